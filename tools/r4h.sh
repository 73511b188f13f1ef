#!/bin/bash
# round 4, GPU call 8: A/B of the conn ingest's piece prefetch (the default library against -DGYS_CONN_PREFETCH=0) on C2, then all GPU
# parity tests + smoke() (the multi-host summary query, the JSON entry points' guards and the bench's share-device flow came after r4g)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
for lib in libgysketch libgysketch_connpf0 libgysketch libgysketch_connpf0; do
	GYS_LIB=$R/gyeeta_amd/lib/$lib.so timeout 200 python bench.py --workload conn --no-cpu-baseline --steps 30 --warmup 5 > $O/conn_$lib.$RANDOM.json 2> $O/conn_$lib.err
done
python - $O <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/conn_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f G rec/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.01})
    except Exception as e:
        print(f, "no result", e)
PY
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log 2>&1
grep -v amdgpu $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.log
