#!/bin/bash
# Non-temporal hints on K2's two read-once / write-once streams: the loaders' row loads (bit 0) and the
# leaf-value workspace stores + the summer wave's loads (bit 1) compete with the bottom blocks for L2.  For each build
# (make WKNT=0..3): the TILERING time of tools/k2_time.py, then TD / TA busy, L2 -> L1 requests and L2 hit / miss from two
# rocprofv3 --pmc passes over tools/pmc_target.py K2 4.  A failed build stops the script (no number from a stale library).
# usage (GPU box): tools/k2_nt.sh > gpurun_out/k2_nt.txt     -> profiles/r04/experiments.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for nt in ${VARIANTS:-0 1 2 3}; do
  rm -f tahoe_amd/csrc/wkey.o
  make -C tahoe_amd/csrc -s WKNT=$nt
  echo "== WKNT=$nt (bit 0: row loads nt, bit 1: leaf stores + summer loads nt)"
  timeout -k 10 120 python3 tools/k2_time.py 2>&1 | grep "tilering" | tail -1
  PASSES="c f" tools/pmc_script.sh k2nt$nt tools/pmc_target.py K2 4 > gpurun_out/pmc_k2nt$nt.txt 2>&1
  python3 - <<PY
import json
p = json.load(open("gpurun_out/pmc_k2nt$nt.json"))
for k, v in p["kernels"].items():
    if "wkey_kernel" in k:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        cus = p["num_cus"]
        print("   wkey_kernel: %.4f ms profiled, TD busy %.3f, TA busy %.3f, L2->L1 requests %.4g, TCC hit %.4g miss %.4g (hit rate %.3f)" % (
            cyc / (p["clock_ghz"] * 1e6), v["TD_TD_BUSY"] / (cus * cyc), v["TA_TA_BUSY"] / (cus * cyc), v["TCP_TCC_READ_REQ"],
            v["TCC_HIT_sum"], v["TCC_MISS_sum"], v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1.0)))
PY
done
rm -f tahoe_amd/csrc/wkey.o; make -C tahoe_amd/csrc -s
