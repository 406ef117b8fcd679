#!/bin/bash
# Timing-only ablation builds of the row-streaming wide form on K2 (results are wrong on purpose): what each part costs.
# usage (on the GPU box): tools/ablate_wstream.sh > gpurun_out/ablate_wstream.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R/tahoe_amd/csrc"
for a in ${ABLATIONS:-0 1 2 3}; do
  rm -f widef.o wkey.o
  if [ $a = 0 ]; then make -s; else make -s ABLATE=$a; fi
  echo "== ablation $a (0 full, 1 no bottom-block gathers, 2 walkers only pass the rows on, 3 loaders load nothing)"
  (cd "$R" && timeout -k 10 120 python3 tools/k2_time.py 2>&1 | grep tilering | tail -1)
done
rm -f widef.o wkey.o; make -s
