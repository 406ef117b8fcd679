#!/bin/bash
# PMC passes over bench.py (each pass = its own rocprofv3 run; no tracing domains mixed in), then one JSON summary per
# kernel (averages per launch, counters summed over the chip) stamped with the hash of the kernel sources it was taken
# with: gpurun_out/pmc_<tag>.json.  Copy the ones to be judged into profiles/.
# usage: tools/pmc.sh <tag> [bench args...]      env PASSES="a b c d e" selects passes
export TMPDIR=/tmp
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
tag=$1; shift
BENCH_ARGS="$*"
PASSES=${PASSES:-"a b c d e"}
pass() { n=$1; shift; case " $PASSES " in *" $n "*) ;; *) return 0;; esac
  rm -rf $R/gpurun_out/pmc_${tag}_$n
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-host --no-k4 --no-configs $BENCH_ARGS > $R/gpurun_out/pmc_${tag}_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${tag}_$n.log; exit 1; }; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU && \
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY && \
pass c TA_TA_BUSY TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE TD_TD_BUSY && \
pass d FETCH_SIZE && \
pass e WRITE_SIZE && \
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
import bench, torch
prop = torch.cuda.get_device_properties(0)
kern = collections.defaultdict(lambda: collections.defaultdict(list))
for n in "abcde":
    for f in glob.glob("$R/gpurun_out/pmc_${tag}_%s/*/*_counter_collection.csv" % n):
        for r in csv.DictReader(open(f)):
            if "tahoe" in r["Kernel_Name"]:
                kern[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc passes a-e (tools/pmc.sh; each pass its own run, no trace domains) over bench.py --steps 2 --warmup 1 "
               "--no-cpu --no-host --no-k4 --no-configs $BENCH_ARGS; averages per launch, counters summed over the chip; FETCH_SIZE / WRITE_SIZE in KB",
       "script": "bench.py --steps 2 --warmup 1 (predicts per run:) 3",
       "src_hash": bench.kernel_source_hash(), "num_cus": prop.multi_processor_count, "clock_ghz": getattr(prop, "clock_rate", 2400000) / 1e6,
       "kernels": {k: dict({c: sum(v) / len(v) for c, v in sorted(cs.items())}, launches=len(next(iter(cs.values()))))
                   for k, cs in sorted(kern.items())}}
json.dump(out, open("$R/gpurun_out/pmc_${tag}.json", "w"), indent=1)
for k, cs in out["kernels"].items():
    print(k[-70:], {c: "%.4g" % v for c, v in cs.items()})
PY
