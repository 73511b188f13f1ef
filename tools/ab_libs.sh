#!/bin/bash
# A/B of kernel variants inside ONE gpurun call (GPU minutes are charged per call: ~50 s of fixed cost each).
#   here (no GPU):   tools/ab_libs.sh build wg6 -DGYS_BINS_WGS=6        -> gyeeta_amd/lib/libgysketch_wg6.so (travels with the snapshot)
#   on the GPU box:  tools/ab_libs.sh bench OUTDIR [bench.py args ...]  -> one lean bench line per library found (the default one first)
# A quarter-size run keeps the per-key rates of the default line (53.7 values per key and window) and takes a few seconds per library:
#   tools/ab_libs.sh bench OUT --hosts 2500 --events 134217728 --steps 6 --warmup 2
# Only compile-time switches of the HIP side can be compared this way: constants shared with the oracle / capi.py (e.g. the t-digest
# buffer size) need their own tree.  capi.py loads $GYS_LIB when set.
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
build)
	tag=$2; shift 2
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w "$@" -o $R/gyeeta_amd/lib/libgysketch_$tag.so \
		$R/gyeeta_amd/csrc/gys_engine.hip -ldl -Wl,-rpath,/opt/rocm/lib && echo "built libgysketch_$tag.so ($*)"
	;;
bench)
	out=$2; shift 2; mkdir -p $out; cd $R
	for lib in gyeeta_amd/lib/libgysketch.so $(ls gyeeta_amd/lib/libgysketch_*.so 2>/dev/null); do
		tag=$(basename $lib .so)
		GYS_LIB=$R/$lib timeout 120 python bench.py --no-cpu-baseline --no-host-fed --no-quantile-check --detail-out $out/$tag.json "$@" > $out/$tag.line 2> $out/$tag.err
		python - $out/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))  # (the full result: bench.py's stdout line is the compact one)
    print("%-28s %.2f G ev/s %.3f ms" % (sys.argv[2], d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
	done
	;;
*) echo "usage: $0 build TAG [-D...] | bench OUTDIR [bench.py args]"; exit 2 ;;
esac
