#!/bin/bash
# s_sleep arguments of the 384-row u8 tile's two spin loops (walker waiting for ring space : consumer polling) on KR3.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for cfg in ${CONFIGS:-16:1 8:1 32:1 16:2 4:1}; do
  set -- $(echo $cfg | tr : " ")
  rm -f tahoe_amd/csrc/qring.o
  make -C tahoe_amd/csrc -s WS8=$1 CS8=$2
  echo "== walker sleep $1 consumer sleep $2: $(timeout -k 10 200 python3 tools/kr3_time.py 1000000 2>&1 | grep '"code8": true' | cut -c1-200)"
done
rm -f tahoe_amd/csrc/qring.o; make -C tahoe_amd/csrc -s
