#!/bin/bash
# SURVEY 8f N3, measured: K3 with and without the probability-guided re-layout (same forest, weights = reach
# probabilities), bench time + LDS / texture counters of the walk kernel.  Output: gpurun_out/n3_relayout.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/n3_relayout.txt
: > $out
for mode in plain relayout; do
  flag=""; [ $mode = relayout ] && flag="--relayout"
  echo "== $mode: bench.py --steps 10 --warmup 3 $flag" >> $out
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-host --no-k4 --cpu-rows 20000 $flag 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('ms_per_step %.4f  walk_ms %.4f  quantise_ms %.4f  cpu_parity %s  swaps %s' % (l['ms_per_step'], r['walk_kernel_ms_avg'], r['prepass_kernel_ms_avg'], l['cpu_baseline']['gpu_matches_cpu_bitwise_on_sample'], l['config'].get('relayout_swaps')))" >> $out
  for p in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" "TA_TA_BUSY TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES GRBM_GUI_ACTIVE TD_TD_BUSY"; do
    rm -rf $R/gpurun_out/n3_pmc
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $R/gpurun_out/n3_pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-host --no-k4 $flag > /dev/null 2>&1
    python3 - >> $out <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/n3_pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "qring_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  walk kernel:", {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
  done
done
cat $out
