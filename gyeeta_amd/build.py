"""Build recipe for libgysketch.so (hipcc cross-compiles gfx950 without a GPU).  Used by __graft_entry__.build()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgysketch.so")
SOURCES = ["gys_engine.hip"]
DEPS = ["gys_engine.hip", "gys_kernels.hpp", "gys_rollup.hpp", "gys_huge.hpp", "gys_device.hpp", "gys_json.hpp", "gys_svcquery.hpp", "gys_svcquery_host.hpp", "gys_regex.hpp", "gys_mconn_shim.hpp", "../../include/gysketch.h", "../../include/gys_tdigest_tbl.h"]


def hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


COMMIT_PATH = os.path.join(LIB_DIR, "build_commit.txt")


def stamp_commit():
    """which source tree the library was built from (the GPU box gets the built .so but not .git): `git rev-parse HEAD` + "+dirty" when
    tracked files differ from it; read back by bench.py (build_commit) and tools/pmc_traffic.py (source_commit of the counter passes)"""
    try:
        root = os.path.dirname(HERE)
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.call(["git", "-C", root, "diff", "--quiet", "HEAD", "--", "gyeeta_amd", "include", "bench.py"], stderr=subprocess.DEVNULL) != 0
        open(COMMIT_PATH, "w").write(head + ("+dirty" if dirty else "") + " " + _hash_sources() + " " + (_hash_device_code(LIB_PATH) or "-") + "\n")
    except Exception:
        pass


def build_commit():
    try:
        return open(COMMIT_PATH).read().strip().split()[0]
    except Exception:
        return None


def sources_sha():
    """sha256 (first 16 hex digits) over the library's source files (DEPS): the same for two builds of the same kernels whatever commits
    lie between them (a documentation commit moves HEAD, not the kernels) -- written next to the commit at build time, compared by
    bench.py between the library it runs (kernel_sources) and the tree profiles/pmc_traffic.json was taken on (source_kernels)"""
    try:
        return open(COMMIT_PATH).read().strip().split()[1]
    except Exception:
        return None


def device_code_sha():
    """sha256 (first 16 hex digits) over the .rodata (kernel descriptors, constant tables) and .text (the kernels' ISA) sections of the
    gfx950 code object inside the library: the identity of the KERNELS themselves.  Two builds of the same sources give the same value
    (the rest of the code object -- notes, symbol tables -- carries per-build names and does not), and a change that only touches host
    code of gys_engine.hip leaves it alone where sources_sha() moves.  bench.py compares it between the library it runs (device_code) and
    the library profiles/pmc_traffic.json was taken with."""
    try:
        f = open(COMMIT_PATH).read().strip().split()
        if len(f) > 2 and f[2] != "-":
            return f[2]
    except Exception:
        pass
    return _hash_device_code(LIB_PATH)


def _hash_device_code(lib_path):
    import hashlib
    import shutil
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    tmp = tempfile.mkdtemp(prefix="gys_devcode_")
    try:
        fat, elf = os.path.join(tmp, "fatbin"), os.path.join(tmp, "gfx950.elf")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path], stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + elf], stderr=subprocess.DEVNULL)
        h = hashlib.sha256()
        for sec in (".rodata", ".text"):
            out = os.path.join(tmp, "sec")
            subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=" + sec, elf, out], stderr=subprocess.DEVNULL)
            data = open(out, "rb").read()
            if not data:
                return None
            h.update(data)
        return h.hexdigest()[:16]
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _hash_sources():
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        h.update(open(os.path.join(CSRC, d), "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-Wno-unused-value", "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES] + [
               "-ldl", "-Wl,-rpath,/opt/rocm/lib"]  # RCCL (the window exchange inside the library) is bound at run time: dlopen("librccl.so") through this RUNPATH, or $GYS_RCCL_LIB
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    stamp_commit()
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
