#!/usr/bin/env python3
"""Static ISA summary of the product library's gfx950 kernels (no GPU needed): instructions, code bytes and the instruction mix per
kernel, from llvm-objdump of the device code object.

  python tools/isa_summary.py [out.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gyeeta_amd", "csrc", "gys_engine.hip")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def klass(m):
    if m.startswith(("v_mfma", "v_smfma")):
        return "mfma"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    if m.startswith("s_barrier"):
        return "barrier"
    if m.startswith("v_"):
        return "valu"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as t:
        obj = os.path.join(t, "dev.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
                               "--no-gpu-bundle-output", "-w", "-c", SRC, "-o", obj])
        asm = subprocess.check_output([OBJDUMP, "-d", obj], text=True)
    kernels, cur = [], None
    for line in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            cur = {"name": m.group(2), "mix": {}, "n": 0, "first": None, "last": None}
            kernels.append(cur)
            continue
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-F]+):", line)
        if cur is None or not m:
            continue
        mn, addr = m.group(1), int(m.group(2), 16)
        if mn == "s_nop" or mn.startswith("s_code_end"):
            continue
        c = klass(mn)
        cur["mix"][c] = cur["mix"].get(c, 0) + 1
        cur["n"] += 1
        cur["first"] = addr if cur["first"] is None else cur["first"]
        cur["last"] = addr
    names = subprocess.check_output(["c++filt"] + [k["name"] for k in kernels], text=True).splitlines()
    cols = ["valu", "salu", "vmem", "smem", "lds", "branch", "waitcnt", "barrier"]
    lines = ["# static ISA summary, gfx950 (tools/isa_summary.py; hipcc -O3 of gyeeta_amd/csrc/gys_engine.hip); counts are STATIC instructions",
             "%-32s %7s %8s " % ("kernel", "instrs", "bytes") + " ".join("%7s" % c for c in cols)]
    for k, n in sorted(zip(kernels, names), key=lambda kn: -kn[0]["n"]):
        if not k["n"]:
            continue
        n = re.sub(r"\(.*", "", n).replace("void ", "").replace("gys::", "")
        lines.append("%-32s %7d %8d " % (n[:32], k["n"], k["last"] - k["first"] + 8) + " ".join("%7d" % k["mix"].get(c, 0) for c in cols))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
