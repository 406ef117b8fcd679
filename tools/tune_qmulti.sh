#!/bin/bash
# Features per workgroup of the many-features quantise kernel (16 = QMULTI 4, 32 = QMULTI 8) on KR3 (254 thresholds per feature)
# and K2 (3072 features x ~40 thresholds, quantised form forced).  A failed build stops the script.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for q in ${CONFIGS:-4 8}; do
  rm -f tahoe_amd/csrc/quantize.o
  make -C tahoe_amd/csrc -s QMULTI=$q
  echo "== QMULTI=$q ($((4 * q)) features per workgroup)"
  timeout -k 10 200 python3 tools/kr3_time.py 1000000 2>&1 | grep '"code8"' | cut -c1-200
  timeout -k 10 200 python3 tools/k2_time.py 2>&1 | grep "qring" | tail -1
done
rm -f tahoe_amd/csrc/quantize.o; make -C tahoe_amd/csrc -s
