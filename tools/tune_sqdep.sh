#!/bin/bash
# Sparse quantised walk (K5): both children beside the feature read (SQDEP=0) against only the chosen child after the compare (1).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for d in 0 1; do
  rm -f tahoe_amd/csrc/sparse.o
  make -C tahoe_amd/csrc -s SQDEP=$d
  echo "== SQDEP=$d: $(timeout -k 10 200 python3 tools/k5_time.py 2>&1 | tail -2 | tr '\n' ' ')"
done
rm -f tahoe_amd/csrc/sparse.o; make -C tahoe_amd/csrc -s
