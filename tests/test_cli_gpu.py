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


def test_cli_on_a_histogram_style_model(built, tmp_path):
    """The CLI end to end on a forest in the style of histogram-trained models, written in the reference's text formats
    (BaseTahoeTest.h:267-402): 28 features, <= 254 thresholds each, 30 k rows with missing values -- strategy 5 walks it on 8-bit
    rank codes in 384-row tiles; every strategy must pass the CPU check of the harness."""
    import tahoe_amd as ta

    T, D, C, R = 150, 6, 28, 30_000
    nodes = ta.synth_forest_hist(T, D, C, seed=77, feature_seed=5, max_bins=254, scale_decades=2.0)
    data = ta.synth_data_hist(R, C, seed=78, feature_seed=5, scale_decades=2.0, missing_prob=0.01, missing=-999.0)
    ta.write_model(str(tmp_path / "m.txt"), nodes, T, D)
    ta.write_data(str(tmp_path / "d.txt"), data, -999.0)
    f = ta.Forest(nodes, T, D, C, missing=-999.0)
    assert f.kernel_form(R) == "qring_region8"
    f.close()
    exe = os.path.join(ROOT, "tahoe_amd", "host", "Tahoe")
    r = subprocess.run([exe, str(tmp_path / "m.txt"), str(tmp_path / "d.txt")], capture_output=True, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 0, out + r.stderr
    assert "Results are incorrect" not in out and "FAIL:" not in out and "Using strategy 5" in out
    assert out.count("Results are correct") == 1 + len(re.findall(r"^Using strategy \d", out, flags=re.M))


def test_cli_result_json_and_leaf_dump(built, tmp_path):
    """TAHOE_RESULT_JSON / TAHOE_LEAF_DUMP: the machine-readable summary, and per-(row, tree) leaf indices that equal
    the oracle's on the golden case."""
    import json

    import numpy as np

    from oracle import oracle

    exe = os.path.join(ROOT, "tahoe_amd", "host", "Tahoe")
    g = os.path.join(ROOT, "tests", "golden", "susy_like_c18")
    js, leaf = tmp_path / "r.json", tmp_path / "leaf.bin"
    env = dict(os.environ, TAHOE_RESULT_JSON=str(js), TAHOE_LEAF_DUMP=str(leaf))
    r = subprocess.run([exe, g + ".model.txt", g + ".data.txt"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(js.read_text())
    assert res["num_cols"] == 18 and len(res["strategy_us_per_sample"]) == 5 and 1 <= res["best_strategy"] <= 5
    assert res["speedup"] > 0 and res["baseline_us_per_sample"] > 0
    nodes, T, D = oracle.load_model(g + ".model.txt")
    data, missing = oracle.load_data(g + ".data.txt")
    _, want_leaf = oracle.predict(nodes, T, D, data, missing, want_leaf=True)
    got = np.fromfile(str(leaf), dtype=np.uint32).reshape(data.shape[0], T)
    assert np.array_equal(got, want_leaf)


@pytest.mark.parametrize("args", [["1"], ["1", "--mode", "allreduce32"], ["1", "--mode", "chain", "--chunk", "100"],
                                  ["--emulate", "3"], ["--emulate", "7", "--mode", "allreduce32"],
                                  ["--emulate", "3", "--mode", "chain", "--chunk", "64"],
                                  ["1", "--mode", "rows"], ["--emulate", "3", "--mode", "rows"], ["--emulate", "7", "--mode", "rows"]])
def test_sharded_host(built, args):
    """./TahoeSharded: the C++ tree-sharding host (one process, a forest shard per device).  A 1-GPU box can check the
    RCCL path with a single shard (bit-exact) and the partition + combination logic with shards emulated on device 0:
    all-reduce modes against the float64 sum with the stated bound, chain mode bit for bit."""
    exe = os.path.join(ROOT, "tahoe_amd", "host", "TahoeSharded")
    g = os.path.join(ROOT, "tests", "golden", "susy_like_c18")
    r = subprocess.run([exe, g + ".model.txt", g + ".data.txt"] + args, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Results are correct" in r.stdout and "Exec.Time/Sample with" in r.stdout


@pytest.mark.parametrize("mode", ["allreduce64", "allreduce32", "chain", "rows"])
def test_sharded_host_1200_trees(built, tmp_path, mode):
    """The same on a forest where float32 association matters (1200 trees): 8 emulated shards.  The gate inside
    TahoeSharded is the float64 bound (all-reduce modes) / bit equality with the sequential float32 sum (chain)."""
    import tahoe_amd as ta

    T, D, C, R = 1200, 6, 16, 3000
    nodes = ta.synth_forest(T, D, C, seed=11, leaf_prob=0.05)
    data = ta.synth_data(R, C, seed=12, missing_prob=0.02, missing=-999.0)
    ta.capi.write_model(str(tmp_path / "m.txt"), nodes, T, D)
    ta.capi.write_data(str(tmp_path / "d.txt"), data, -999.0)
    exe = os.path.join(ROOT, "tahoe_amd", "host", "TahoeSharded")
    r = subprocess.run([exe, str(tmp_path / "m.txt"), str(tmp_path / "d.txt"), "--emulate", "8", "--mode", mode, "--chunk", "1000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Results are correct" in r.stdout
    if mode in ("chain", "rows"):
        assert "max |ours - CPU float32 sum| = 0" in r.stdout


def test_cli_unreadable_file_exits_1(built):
    exe = os.path.join(ROOT, "tahoe_amd", "host", "Tahoe")
    r = subprocess.run([exe, "/nonexistent/model", "/nonexistent/data"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "fail to read" in r.stderr
