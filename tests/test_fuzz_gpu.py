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
