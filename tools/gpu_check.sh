#!/bin/bash
# One GPU-box pass: smoke, parity tests, short bench.  Logs under gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1 || { tail -30 gpurun_out/pytest.log; exit 1; }
tail -2 gpurun_out/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 "$@" > gpurun_out/bench.log 2>&1 || { tail -5 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
