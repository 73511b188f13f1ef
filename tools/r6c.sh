#!/bin/bash
# round 6, call c: A/B of the event-phase switches once more (new default = diet without the scalar-branch store and the joint probe loop),
# and what a SECOND resident workgroup per CU is worth today: 20 000 hosts x 500 listeners in the two-workgroup tile form (engine's choice)
# against the same batch pinned to the 1024 x 16 form (GYS_TPT=16)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c; mkdir -p $O
tools/ab_libs.sh bench $O/ab --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab.txt
tools/ab_libs.sh bench $O/ab2 --configs none --steps 20 --warmup 5 2>&1 | tee $O/ab2.txt
LEAN="--no-cpu-baseline --no-host-fed --no-quantile-check --configs none --steps 10 --warmup 3"
for v in "" 16; do
  GYS_TPT=$v timeout 300 python bench.py --hosts 20000 --svcs 500 $LEAN --detail-out $O/svcs500_tpt$v.json > $O/svcs500_tpt$v.line 2> $O/svcs500_tpt$v.err
  python - $O/svcs500_tpt$v.json "svcs500 GYS_TPT=$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-22s %7.2f G %s  %8.3f ms " % (sys.argv[2], d["value"] / 1e9, d["unit"], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items() if v["ms"] > 0.05})
PY
done 2>&1 | tee $O/two_wg.txt
