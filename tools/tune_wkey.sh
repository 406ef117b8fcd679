#!/bin/bash
# Wave partition of the row-streaming wide form on K2: loaders:walkers:chains per walker:consumers[:consumer priority]
# usage (GPU box): CONFIGS="4:10:2:2:2 3:9:2:4:1:3" tools/tune_wkey.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}/tahoe_amd/csrc
for cfg in ${CONFIGS:-4:8:2:4 3:9:2:4 2:10:2:4 3:10:2:3 4:9:2:3:1}; do
  set -- $(echo $cfg | tr : " "); rm -f wkey.o
  make -s WKL=$1 WKW=$2 WKCHAINS=$3 WKC=$4 WKP=$5 || exit 1
  echo "loaders $1 walkers $2 chains $3 consumers $4 prio ${5:-0}: $(cd ../.. && python3 tools/k2_time.py 2>&1 | grep tilering | tail -1)"
done
rm -f wkey.o; make -s
