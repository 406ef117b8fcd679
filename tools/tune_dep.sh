#!/bin/bash
# Top walk reading both children beside the feature (5 VALU + 2 LDS per chain-level, one LDS round trip) against reading only the
# chosen child after the compare (4 + 2, two round trips): the 384-row u8 tile on KR3 (make Q8DEP=0/1) and the 192-row u16 tile on
# K3 (make R3DEP=0/1).  A failed build stops the script.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for d in 0 1; do
  rm -f tahoe_amd/csrc/qring.o
  make -C tahoe_amd/csrc -s Q8DEP=$d R3DEP=$d
  echo "== chosen child only: $d"
  timeout -k 10 200 python3 tools/kr3_time.py 1000000 2>&1 | grep '"code8": true' | cut -c1-200
  timeout -k 10 200 python3 tools/k3_time.py 2>&1 | tail -1
done
rm -f tahoe_amd/csrc/qring.o; make -C tahoe_amd/csrc -s
