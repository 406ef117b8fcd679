#!/bin/bash
# Walkers : ring of QRING's 192-row tile (u16 codes) on K3 (tools/k3_time.py).  A failed build stops the script.
# usage (GPU box): CONFIGS="14:10 15:5" tools/tune_r3.sh > gpurun_out/tune_r3.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for cfg in ${CONFIGS:-14:10 15:5 15:4 13:15}; do
  set -- $(echo $cfg | tr : " ")
  rm -f tahoe_amd/csrc/qring.o tahoe_amd/csrc/sparse.o
  make -C tahoe_amd/csrc -s R3W=$1 R3R=$2
  echo "== walkers $1 ring $2: $(timeout -k 10 200 python3 tools/k3_time.py 2>&1 | tail -1)"
done
rm -f tahoe_amd/csrc/qring.o tahoe_amd/csrc/sparse.o; make -C tahoe_amd/csrc -s
