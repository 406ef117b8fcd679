"""The bench.py output contract, checked on the line recorded from the last default run on the GPU box
(profiles/r02/bench_default_line.json).  CPU test: guards the keys the driver and the judge read."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_keys():
    line = json.load(open(os.path.join(ROOT, "profiles", "r02", "bench_default_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "samples/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["rows_per_step"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    # the roofline covers the whole step (pre-pass + walk), VERDICT r1 item 3
    assert abs(r["kernel_ms_avg"] - (r["walk_kernel_ms_avg"] + r["prepass_kernel_ms_avg"])) < 1e-3
    assert r["hbm_frac_measured"] is None or 0.0 < r["hbm_frac_measured"] <= 1.0
    k4 = line["k4"]
    assert k4["row_sharded"]["bit_identical_to_cpu_f32"] is True
    assert k4["tree_sharded_allreduce64"]["within_stated_bound"] is True
    assert k4["tree_sharded_allreduce64"]["max_abs_err_vs_f64"] <= k4["tree_sharded_allreduce64"]["cpu_f32_max_abs_err_vs_f64"]
    assert k4["tree_sharded_chain_accuracy"]["bit_identical_to_cpu_f32"] is True
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["gpu_matches_cpu_bitwise_on_sample"] is True


def test_bench_source_keeps_the_oracle_out_of_the_timed_regions():
    """bench.py reaches the oracle through one accessor, used by the cpu_baseline leg and by the checks of the K4 legs'
    results; neither sits inside a timed region (`timed(`, or between the fences of the primary loop)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("from oracle import oracle") == 1
    leg = src.index("# ---- CPU baseline + parity spot check")
    assert src.index("from oracle import oracle") > leg
    primary = src[src.index("    for _ in range(args.warmup):"):src.index("    forest.check()  # a raised error flag")]
    assert "oracle" not in primary
    for chunk in src.split("timed(lambda:")[1:]:
        assert "oracle" not in chunk.split("\n")[0]


def test_committed_pmc_profile_belongs_to_these_kernel_sources():
    """roofline.traffic / roofline.physical come from profiles/r03/pmc_k3.json only while its stamp matches the device
    sources in the tree (bench.kernel_source_hash); a kernel edit without a fresh tools/pmc.sh run turns this red."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    prof = json.load(open(os.path.join(ROOT, "profiles", "r03", "pmc_k3.json")))
    assert prof["src_hash"] == bench.kernel_source_hash()
    phys = bench.physical_ceilings(prof, 3.7, 0.77, 1_000_000, 256)
    for kern in ("walk", "quantise"):
        for key in ("vector_issue_busy", "valu_busy", "lds_busy", "texture_addr_busy", "hbm_frac_counters_raw"):
            assert 0.0 < phys[kern][key] <= 1.0, (kern, key, phys[kern][key])
    assert phys["quantise"]["hbm_frac_compulsory"] < 0.5  # the kernel furthest from its HBM roofline, said so
