run() { echo "== $*"; env $1 python bench.py --steps 5 --warmup 2 --no-cpu ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'])"; }
run TAHOE_QRING_WALKERS=15
run TAHOE_QRING_WALKERS=12
run TAHOE_QRING_WALKERS=8
run TAHOE_QRING_WALKERS=4
run X=1 --depth 2
run X=1 --depth 7
run X=1 --depth 10
