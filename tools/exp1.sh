set -e
run() { echo "== $*"; env $1 python bench.py --steps 5 --warmup 2 --no-cpu ${@:2} | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'])"; }
run X=1 --depth 8
run X=1 --depth 10
run X=1 --depth 12
run TAHOE_LDS_LEVELS=6 --depth 12
run TAHOE_LDS_LEVELS=4 --depth 12
run X=1 --depth 12 --strategy 1 --rows 200000
run X=1 --depth 12 --trees 250
