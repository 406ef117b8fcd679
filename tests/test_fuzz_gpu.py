"""A short run of the randomised differential test (tests/fuzz_gpu.py): random dense and sparse forests and batches, every
available strategy against the CPU oracle, leaf indices and float32 sums bit for bit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_shapes_every_strategy_bit_exact(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "10", "3"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz ok" in out.stdout


@pytest.mark.gpu
def test_random_big_batches_every_strategy_bit_exact(built):
    """The same in `big` mode for 45 s: batches of several waves of tiles (the mixed 192- / 128-row plans, tree slices),
    hundreds of trees -- the launches whose hand-over protocol rests on "a wave's LDS operations are performed in issue order"."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "45", "5", "big"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz ok" in out.stdout


@pytest.mark.gpu
def test_ring_soak_200_predicts(built):
    """tools/soak.py --quick: 200 back-to-back K3 predicts (every 20th compared bit for bit with the first, error flag read
    each time), then the sparse and the wide-row ring kernels the same way: the ring protocols under sustained load."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--quick"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "soak ok" in out.stdout and "rows that ever differed: 0" in out.stdout
