#!/bin/bash
# GPU call ac: counters of the roll-up kernels (per dispatch): HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ set, on the bench's own state at 10^7 services
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6ac; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-host-fed --configs none --steps 3 --warmup 1 --detail-out none"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/pr_$i
  timeout 500 rocprofv3 --pmc $set --kernel-trace -d /tmp/pr_$i -o p --output-format csv -- python $R/bench.py $ARGS > $O/pmc_$i.log 2>&1
  python - /tmp/pr_$i "$set" >> $O/rollup_counters.txt <<'PY'
import collections, csv, os, sys
root, name = sys.argv[1], sys.argv[2]
print("## pmc set:", name)
for d, _, fs in os.walk(root):
    for f in fs:
        if not f.endswith("counter_collection.csv"):
            continue
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(os.path.join(d, f))):
            if "rollup" not in r["Kernel_Name"] and "k_digest_bins<true" not in r["Kernel_Name"]:
                continue
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][:40], r.get("Grid_Size", ""))
            agg.setdefault(key, {})[r["Counter_Name"]] = agg.setdefault(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for k in sorted(agg):
            print("dispatch %-7d %-40s grid %-9s " % k + "  ".join("%s=%.5g" % kv for kv in sorted(agg[k].items())))
PY
done
cat $O/rollup_counters.txt
