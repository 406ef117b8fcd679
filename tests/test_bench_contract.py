"""The bench.py output contract, checked on the line recorded from the last default run on the GPU box
(profiles/r04/bench_default_line.json).  CPU test: guards the keys the driver and the judge read."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def recorded_line():
    return json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_default_line.json")))


def test_recorded_bench_line_has_the_contract_keys():
    line = recorded_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert key in line, key
    assert line["unit"] == "samples/s" and line["higher_is_better"] is True and line["scaling"] == "strong"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["config"]["rows_per_step"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_le_1", "kernel_form"):
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


def test_recorded_line_carries_a_fraction_below_one():
    """next to the byte-model figure (> 1) the line has the one honest fraction this kernel has -- its node
    visits per second against the measured ceiling of its own LDS-resident inner loop (tools/ubench_qwalk.hip)."""
    r = recorded_line()["roofline"]
    c = r["physical"]["walk"]["lds_walk_ceiling"]
    for key in ("kernel", "ubench", "frac", "ubench_source", "node_visits_per_s", "frac_of_instruction_floor"):
        assert key in c, key
    assert 0.0 < c["frac"] <= 1.0 and 0.0 < c["frac_of_instruction_floor"] <= 1.0
    assert abs(c["frac"] - c["ubench"] / c["kernel"]) < 1e-3 and r["frac_le_1"] == c["frac"]
    assert "live" in c["ubench_source"]
    assert abs(c["node_visits_per_s"] - 1_000_000 * 1000 * 12 / (r["walk_kernel_ms_avg"] * 1e-3)) / c["node_visits_per_s"] < 1e-3


def test_recorded_line_has_every_baseline_configuration():
    """K1, K2, K5 are driver-run legs of the same line (K3 = the primary figures, K4 = `k4`): time from the
    in-library hipEvents, the kernel form that ran, byte-model and compulsory fractions, stamped counters, a bitwise check."""
    cfg = recorded_line()["configs"]
    assert set(cfg) >= {"K1", "K2", "K3", "K4", "K5", "KR3", "seconds"} and cfg["seconds"] <= 15.0
    want_form = {"K1": "qring_split", "K2": "tilering_wide_stream", "K5": "sparse_qring", "KR3": "qring_region8"}
    for k in ("K1", "K2", "K5", "KR3"):
        leg = cfg[k]
        assert "error" not in leg, leg
        for key in ("workload", "strategy", "kernel_form", "ms", "samples_per_s", "stream_slots", "ring_rows", "rows_checked",
                    "bitwise_equal_to_cpu_oracle", "roofline"):
            assert key in leg, (k, key)
        assert leg["kernel_form"] == want_form[k] and leg["rows_checked"] >= 2048 and leg["bitwise_equal_to_cpu_oracle"] is True
        assert leg["ms"] > 0 and abs(leg["samples_per_s"] - int(leg["workload"].split(" rows")[0].split()[-1]) / (leg["ms"] * 1e-3)) / leg["samples_per_s"] < 1e-2
        rl = leg["roofline"]
        assert 0.0 < rl["compulsory_frac"] <= 1.0 and rl["frac"] > 0.0
        assert rl["counters"] is None or (rl["counters"]["src_hash"] and 0.0 < max(rl["counters"]["busy"].values()) <= 1.0)
    assert cfg["K2"]["stream_slots"] >= 4 and cfg["K3"]["kernel_form"].startswith("qring_region")


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
    """roofline.traffic / roofline.physical come from profiles/r04/pmc_k3.json only while its stamp matches the device
    sources in the tree (bench.kernel_source_hash); a kernel edit without a fresh tools/pmc.sh run turns this red."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    prof = json.load(open(os.path.join(ROOT, "profiles", "r04", "pmc_k3.json")))
    assert prof["src_hash"] == bench.kernel_source_hash()
    phys = bench.physical_ceilings(prof, 3.7, 0.77, 1_000_000, 256)
    for kern in ("walk", "quantise"):
        for key in ("vector_issue_busy", "valu_busy", "lds_busy", "texture_addr_busy", "hbm_frac_counters_raw"):
            assert 0.0 < phys[kern][key] <= 1.0, (kern, key, phys[kern][key])
    assert phys["quantise"]["hbm_frac_compulsory"] < 0.5  # the kernel furthest from its HBM roofline, said so


def test_walk_ceiling_and_config_roofline_arithmetic():
    """bench.py's host-side arithmetic without a GPU: the <= 1 fraction (here the live micro-benchmark cannot run, so the committed
    figure serves and says so), the per-configuration byte model on a hand example, and the workloads of the `configs` legs."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    # 1 M rows x 1000 trees x 12 levels on 256 CUs in 3.4 ms: 1.2e10 visits = 732,421.875 wave-levels per CU
    c = bench.lds_walk_ceiling(3.4, 1_000_000, 1000, 12, 256, 2.4)
    assert abs(c["kernel"] - 3.4e6 / 732421.875) < 1e-3 and abs(c["node_visits_per_s"] - 1.2e10 / 3.4e-3) / c["node_visits_per_s"] < 1e-9
    assert abs(c["frac"] - c["ubench"] / c["kernel"]) < 1e-3 and 0.0 < c["frac"] <= 1.0
    assert abs(c["frac_of_instruction_floor"] - 6.0 / (c["kernel"] * 2.4)) < 1e-3
    assert "live" in c["ubench_source"]  # either measured live, or the committed figure with "live run unavailable"
    # byte model: rows x (len x (node + 4) + trees x leaf + cols x 4 + 4); compulsory: rows x cols x 4 + nodes x node + rows x 4
    r = bench.config_roofline(1.0, rows=10, cols=4, trees=3, len_sum_per_row=6.0, node_bytes=6, n_nodes=21, cfg="none")
    assert r["algorithmic_bytes"] == 10 * (6 * 10 + 3 * 6 + 16 + 4) and r["compulsory_bytes"] == 10 * 16 + 21 * 6 + 40
    assert r["counters"] is None and abs(r["frac"] - r["algorithmic_bytes"] / 1e-3 / 1e9 / 8000.0) < 1e-4
    assert bench.dense_path_len_sum(__import__("numpy").array([[0, 2, 6], [1, 3, 14]])) == (0 + 1 + 2 + 1 + 2 + 3) / 2
    import tahoe_amd as ta

    kind, (nodes, T, D, C), data = bench.baseline_workload(ta, "K1", through_text_files=True)
    assert kind == "dense" and (T, D, C) == (500, 8, 18) and data.shape == (10_000, 18) and nodes.size == T * 511
    kind, (sn, tr, C5), _ = bench.baseline_workload(ta, "K5")
    assert kind == "sparse" and tr.size == 2000 and C5 == 256 and 9_000_000 < sn.size < 10_500_000


def test_every_stamped_profile_of_the_round_belongs_to_these_sources():
    """Counter profiles and selector runs of profiles/r04 carry src_hash = bench.kernel_source_hash() of the tree they were
    measured on; all of them were re-taken on the final sources (a stale file would be a claim about other kernels)."""
    import sys

    sys.path.insert(0, ROOT)
    import bench

    want = bench.kernel_source_hash()
    names = ["pmc_k1.json", "pmc_k2.json", "pmc_k3.json", "pmc_k4.json", "pmc_k5.json", "pmc_kr3.json", "selector_vs_enumeration.json",
             "selector_holdout.json", "selector_realistic.json", "selector_wide.json"]
    for name in names:
        assert json.load(open(os.path.join(ROOT, "profiles", "r04", name)))["src_hash"] == want, name
    line = recorded_line()
    assert line["roofline"]["kernel_source_hash"] == want
    for k in ("K1", "K2", "K5", "KR3"):
        assert line["configs"][k]["roofline"]["counters"]["src_hash"] == want, k
