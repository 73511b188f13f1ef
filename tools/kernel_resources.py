#!/usr/bin/env python3
"""Static resource usage of every gfx950 kernel in the product library, from the AMDGPU metadata notes of the device code object
(no GPU needed): VGPRs / SGPRs / spills / static LDS / scratch per kernel and the wave occupancy those numbers allow.

  python tools/kernel_resources.py [out.txt]

gfx950 (CDNA4) budget per SIMD: 512 VGPRs per lane (unified VGPR + AGPR file), at most 8 waves; per CU: 4 SIMDs, 160 KiB of LDS
(MI355X_MICROARCH.md).  waves/SIMD by registers = min(8, 512 // ceil(vgprs, 8)); workgroups/CU by LDS = 160 KiB // LDS per workgroup
(dynamic LDS is a launch parameter and is listed separately for the kernels that use it)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gyeeta_amd", "csrc", "gys_engine.hip")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"
# dynamic LDS of the launches bench.py's default window makes (gys_engine.hip run_resp_batch)
DYN_LDS = {"k_resp_host": "+ dynamic LDS, resp_host_lds_bytes(): 151 KiB for the <16,...> instances at the bench's 1000-listener hosts (32 KiB quarter-full table + 24 B x 1000 keys + 6 B x 16384-event tile image): one 1024-thread workgroup per CU"}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    if out and not out.endswith(".txt"):
        sys.exit("kernel_resources.py: the argument is the OUTPUT file (*.txt); the library is always compiled from the sources")
    with tempfile.TemporaryDirectory() as t:
        obj = os.path.join(t, "dev.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
                               "--no-gpu-bundle-output", "-w", "-c", SRC, "-o", obj])
        notes = subprocess.check_output([READELF, "--notes", obj], text=True)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and line.lstrip().startswith("- "):
            cur = {}
            kernels.append(cur)
        if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                                     "private_segment_fixed_size", "max_flat_workgroup_size", "name"):
            cur[k] = v
    names = subprocess.check_output([FILT] + [k["name"] for k in kernels], text=True).splitlines()
    rows = []
    for k, n in zip(kernels, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "").replace("gys::", "")
        vg, ag = int(k["vgpr_count"]), int(k["agpr_count"])
        regs = -(-(vg + ag) // 8) * 8
        wg = int(k["max_flat_workgroup_size"])
        lds = int(k["group_segment_fixed_size"])
        waves_reg = min(8, 512 // max(regs, 8))
        waves_wg = -(-wg // 64)
        note = ""
        for key, txt in DYN_LDS.items():
            if n.startswith(key):
                note = txt
        wg_lds = (160 * 1024) // lds if lds else None
        rows.append((n, wg, vg, ag, int(k["sgpr_count"]), int(k["vgpr_spill_count"]) + int(k["sgpr_spill_count"]), int(k["private_segment_fixed_size"]),
                     lds, waves_reg, wg_lds, waves_wg, note))
    rows.sort(key=lambda r: r[0])
    lines = ["# static kernel resources, gfx950 (tools/kernel_resources.py; hipcc -O3 of gyeeta_amd/csrc/gys_engine.hip)",
             "%-44s %5s %5s %5s %5s %6s %8s %8s %10s %12s" % ("kernel", "wg", "vgpr", "agpr", "sgpr", "spills", "scratchB", "LDS B", "waves/SIMD", "wg/CU by LDS")]
    for n, wg, vg, ag, sg, sp, scr, lds, wr, wl, ww, note in rows:
        lines.append("%-44s %5d %5d %5d %5d %6d %8d %8d %10d %12s  %s" % (n[:44], wg, vg, ag, sg, sp, scr, lds, wr, "-" if wl is None else str(wl), note))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
