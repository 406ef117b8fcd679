"""The drop-in ./Tahoe CLI (tahoe_amd/host: BaseTahoeTest look-alike over the C ABI) on a GPU box: same stdout
protocol as the reference's main.cu / BaseTahoeTest.h:71-115, and every strategy agrees with the CPU check."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", ["susy_like_c18", "k3_like_c256", "depth0", "depth1_ties"])
def test_cli_protocol(built, case):
    exe = os.path.join(ROOT, "tahoe_amd", "host", "Tahoe")
    g = os.path.join(ROOT, "tests", "golden", case)
    r = subprocess.run([exe, g + ".model.txt", g + ".data.txt"], capture_output=True, text=True, timeout=120)
    out = r.stdout
    assert r.returncode == 0, out + r.stderr
    for line in ("Loading model...", "Loading data...", "Predict on CPU to get standard results...", "Test on GPU...",
                 "Exec.Time/Sample on FIL (baseline) is", "Using strategy 1", "Exec.Time/Sample on strategy 1 is",
                 "Performance model choose #", "Tahoe brings"):
        assert line in out, f"missing line {line!r} in:\n{out}"
    assert "Results are incorrect" not in out and "FAIL:" not in out
    assert out.count("Results are correct") == 1 + len(re.findall(r"^Using strategy \d", out, flags=re.M))
    assert "Using strategy 5" in out or "Strategy 5 is not suitable for this case." in out


def test_cli_unreadable_file_exits_1(built):
    exe = os.path.join(ROOT, "tahoe_amd", "host", "Tahoe")
    r = subprocess.run([exe, "/nonexistent/model", "/nonexistent/data"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "fail to read" in r.stderr
