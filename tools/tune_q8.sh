#!/bin/bash
# Walkers : ring : consumer batch of the 384-row tile of QRING's u8 form, on KR3 (tools/kr3_time.py).  A failed build stops the script.
# usage (GPU box): CONFIGS="13:7:3 14:5:2 12:10:4" tools/tune_q8.sh > gpurun_out/tune_q8.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for cfg in ${CONFIGS:-13:7:3 14:5:2 12:10:4 12:10:5}; do
  set -- $(echo $cfg | tr : " ")
  rm -f tahoe_amd/csrc/qring.o
  make -C tahoe_amd/csrc -s Q8W=$1 Q8R=$2 Q8B=$3
  echo "== walkers $1 ring $2 batch $3"
  timeout -k 10 200 python3 tools/kr3_time.py 1000000 2>&1 | grep '"code8": true'
done
rm -f tahoe_amd/csrc/qring.o; make -C tahoe_amd/csrc -s
