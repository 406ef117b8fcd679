# timing-only ablation builds of the walk kernel (built on the GPU box; the in-tree .so is restored at the end)
set -e
cp tahoe_amd/libtahoe_amd.so /tmp/libtahoe_amd.so.keep
for a in 0 1 2 3; do
  touch tahoe_amd/csrc/qring.hip
  make -C tahoe_amd/csrc ABLATE=-DTAHOE_ABLATE=$a > /tmp/build_$a.log 2>&1
  echo "ablate $a: $(python bench.py --steps 10 --warmup 3 --no-cpu --no-host --no-tree-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])")"
done
cp /tmp/libtahoe_amd.so.keep tahoe_amd/libtahoe_amd.so
