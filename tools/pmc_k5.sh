#!/bin/bash
# superseded by tools/pmc_script.sh k5 tools/pmc_target.py K5 4; kept for the round-2 profile it produced
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
rm -rf gpurun_out/k5pmc_a gpurun_out/k5pmc_b gpurun_out/k5pmc_c
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d gpurun_out/k5pmc_a -- python3 tools/k5_time.py > gpurun_out/k5pmc_a.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/k5pmc_b -- python3 tools/k5_time.py > gpurun_out/k5pmc_b.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE TD_TD_BUSY TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/k5pmc_c -- python3 tools/k5_time.py > gpurun_out/k5pmc_c.log 2>&1
python3 - <<PY
import csv, glob, collections
kern = collections.defaultdict(lambda: collections.defaultdict(list))
for n in "abc":
    for f in glob.glob("gpurun_out/k5pmc_%s/*/*_counter_collection.csv" % n):
        for r in csv.DictReader(open(f)):
            if "sparse_q" in r["Kernel_Name"] or "quantize" in r["Kernel_Name"]:
                kern[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in kern.items():
    print(k, {c: "%.4g" % (sum(v)/len(v)) for c, v in sorted(cs.items())})
PY
