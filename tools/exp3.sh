run() { echo "== $*"; env $1 python bench.py --steps 5 --warmup 2 --no-cpu ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'])"; }
run X=1 --depth 2
run X=1 --depth 4
run X=1 --depth 7
run X=1 --depth 10
run X=1 --depth 12
run TAHOE_TILE_ROWS=128 --depth 12
run TAHOE_TILE_ROWS=128 --depth 7
run TAHOE_TILE_ROWS=128 --depth 2
run X=1 --depth 12 --cols 64
run X=1 --depth 2 --cols 64
