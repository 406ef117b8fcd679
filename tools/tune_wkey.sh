#!/bin/bash
# Wave partition of the row-streaming wide form on K2: loaders:walkers:chains per walker (+ 1 summer wave; 16 waves at most)
# usage (GPU box): CONFIGS="5:10:2 4:11:2" tools/tune_wkey.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}/tahoe_amd/csrc
for cfg in ${CONFIGS:-5:10:2 4:11:2 3:12:2 5:10:1}; do
  set -- $(echo $cfg | tr : " "); rm -f wkey.o
  make -s WKL=$1 WKW=$2 WKCHAINS=$3 || exit 1
  echo "loaders $1 walkers $2 chains $3: $(cd ../.. && python3 tools/k2_time.py 2>&1 | grep tilering | tail -1)"
done
rm -f wkey.o; make -s
