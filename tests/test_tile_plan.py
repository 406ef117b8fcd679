"""The launch planner of QRING's region form (qreg_plan, tahoe_amd/csrc/qring_internal.h), checked on the host against a brute-force
restatement: a batch is walked as n whole waves of large tiles (192 rows on u16 codes, 384 on u8 codes / narrow forests) followed
by a remainder in whichever tile size is cheaper, n minimising cost_large * waves + remainder.  CPU test: the probe is compiled
with hipcc (which cross-compiles here) and only its host code runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "tile_plan_probe")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "tahoe_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "probes", "tile_plan_probe.cpp")],
                   check=True, capture_output=True)
    return exe


def waves(rows, tile, cus):
    return ((rows + tile - 1) // tile + cus - 1) // cus


def rem_slices(rem, cus, slice_trees):
    if rem == 0 or slice_trees < 120:
        return 1
    tiles = (rem + 127) // 128
    return max(1, min(min(cus // tiles, 8), slice_trees // 60))


def brute(rows, cus, cost3, big, slice_trees=0):
    best = None
    for n in range(waves(rows, big, cus) + 1):
        r3 = min(rows, n * cus * big // 384 * 384)
        rem = rows - r3
        c2, c3 = 100 * waves(rem, 128, cus), cost3 * waves(rem, big, cus)
        if r3 > 0 and rem_slices(rem, cus, slice_trees) > 1:
            c2 = min(c2, 100 // rem_slices(rem, cus, slice_trees) + 20)
        cost = cost3 * waves(r3, big, cus) + min(c2, c3)
        if best is None or cost < best[0]:
            best = (cost, r3 if rem else 0, (2 if c2 <= c3 else 3) if rem else 3)
    return best


def test_plan_is_the_cheapest_cut(probe):
    cases = []
    for cus in (256, 304, 64, 1):
        for cost3, big in ((133, 192), (161, 192), (218, 384)):
            for rows in (1, 63, 128, 129, 384, 385, 10_000, 49_152, 98_304, 98_305, 125_000, 250_000, 500_000, 983_040, 1_000_000, 1_234_567):
                for trees in (0, 1000):
                    cases.append((rows, cus, 0, cost3, big, trees))
    out = subprocess.run([probe], input="".join("%d %d %d %d %d %d\n" % c for c in cases), capture_output=True, text=True, check=True).stdout
    got = [tuple(int(v) for v in ln.split()) for ln in out.splitlines()]
    assert len(got) == len(cases)
    for (rows, cus, _, cost3, big, trees), (rows3, chains) in zip(cases, got):
        assert rows3 % 384 == 0 and rows3 <= rows and chains in (2, 3), (rows, cus, rows3, chains)
        rem = rows - rows3
        s = rem_slices(rem, cus, trees) if rows3 > 0 else 1
        c2 = min(100 * waves(rem, 128, cus), 100 // s + 20) if s > 1 else 100 * waves(rem, 128, cus)
        cost = cost3 * waves(rows3, big, cus) + (cost3 * waves(rem, big, cus) if chains == 3 else c2)
        assert cost == brute(rows, cus, cost3, big, trees)[0], (rows, cus, cost3, big, trees, rows3, chains)


def test_known_plans_and_forced_forms(probe):
    # K3 on 256 CUs: 20 waves of 192-row tiles + 16,960 rows in 128-row tiles; one of 8 GPUs' share: 2 waves + 26,696 rows;
    # KR3 on u8 codes: 10 waves of 384-row tiles + the same remainder; forced forms leave the cut to the caller
    # with tree slices for small remainders (1000 trees): 250 k rows = 5 waves of 192-row tiles + 34 tiles of 128 in 7 slices each
    # (without: 4 waves + 418 tiles of 128), 500 k rows = 10 waves + 67 tiles in 3 slices
    inp = ("1000000 256 0 133 192 0\n125000 256 0 133 192 0\n1000000 256 0 218 384 0\n125000 256 0 218 384 0\n1000000 256 2 133 192 0\n"
           "1000000 256 3 133 192 0\n250000 256 0 133 192 0\n250000 256 0 133 192 1000\n500000 256 0 133 192 1000\n1000000 256 0 133 192 1000\n")
    out = subprocess.run([probe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["983040", "2", "98304", "2", "983040", "2", "98304", "2", "0", "2", "0", "3", "196608", "2", "245760", "2", "491520", "2", "983040", "2"]
