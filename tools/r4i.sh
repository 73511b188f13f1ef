#!/bin/bash
# round 4, GPU call 9: C2 with the conn ingest's fourteen-unit staging at twelve waves per CU (default library) against the nine-piece
# eight-wave kernel of r4h (libgysketch_conn9p) and the new staging at eight waves (libgysketch_conn14t512); then the conn / listener parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_conn9p libgysketch_conn14t512 libgysketch libgysketch_conn9p libgysketch_conn14t512; do
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $O/conn_$lib.$RANDOM.json 2> $O/conn_$lib.err
done
python - $O <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/conn_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f G rec/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01}, d.get("checks"))
    except Exception as e:
        print(f, "no result", e)
PY
(time timeout 900 python -m pytest tests -m gpu -x -q -k "conn or lstate or c2 or listener or round3 or configs" 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
