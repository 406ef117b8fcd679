#!/bin/bash
# Everything under profiles/r04 that depends on the final kernels, in two halves (a GPU call is limited to 20 minutes):
#   tools/profile_all.sh a   counter profiles: K3 over bench.py (tools/pmc.sh), K1 K2 K4 K5 KR3 over tools/pmc_target.py
#                            (KR3 = K3's shape from the histogram-style generator: u8 codes, quantize_multi_kernel)
#   tools/profile_all.sh b   kernel-trace stats, the five configurations with their rooflines, the default bench line
# Results land in gpurun_out/; copy them to profiles/r04/ (see profiles/README.md for the names).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
if [ "$1" = a ]; then   # optional second argument: the configurations of this call, e.g. "3 1 2" then "4 5 R3" (a call is limited to 20 minutes)
  for c in ${2:-3 1 2 4 5 R3}; do
    if [ $c = 3 ]; then tools/pmc.sh k3 > gpurun_out/pmc_k3.txt 2>&1; echo "pmc K3 done"; continue; fi
    t=$(echo k$c | tr A-Z a-z); tools/pmc_script.sh $t tools/pmc_target.py K$c 4 > gpurun_out/pmc_$t.txt 2>&1; echo "pmc K$c done"
  done
else
  rm -rf gpurun_out/stats_k3 gpurun_out/stats_k2 gpurun_out/stats_k4 gpurun_out/stats_k5
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_k3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-host --no-k4 --no-configs > gpurun_out/stats_k3.log 2>&1
  for c in 2 4 5 R3; do t=$(echo k$c | tr A-Z a-z); rm -rf gpurun_out/stats_$t; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$t -- python3 tools/pmc_target.py K$c 12 > gpurun_out/stats_$t.log 2>&1; done
  echo "stats done"
  python3 tools/run_configs.py > gpurun_out/configs.log 2>&1; echo "configs done"
  python3 bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.log
fi
