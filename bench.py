#!/usr/bin/env python3
"""bench.py -- samples/sec of the batched tree-ensemble traversal on MI355X.

One "step" = one tahoe_forest_predict over one resident batch (BASELINE.json config 3, "K3": synthetic
forest of 1000 trees of depth 12 over 256 features, 1M rows, float32).  Inputs and the forest are
resident in HBM before the timed region (as in the reference, BaseTahoeTest.h:563-573).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1 (default, --scaling strong): the metric's 1M rows are split by rows, rank r predicts rows [R*r/N, R*(r+1)/N) with
    the whole forest (rows are independent -> no data-path collective, every sum bit-identical to 1 GPU);
    value = R / time; "scaling": "strong".   --scaling weak: 1M rows per rank, value = N * R / time.
--shard trees: the forest's trees are split across ranks instead, every rank sees the same 1M rows, the per-rank sums
    are combined as --tree-mode says (tahoe_amd/sharding.py: allreduce64 | allreduce32 | chain).

At N > 1 the line proves what ran: "collective" carries the backend, the ranks the process group reports, an all-reduce of
ones that must sum to that number, the device of every rank, and the fastest / slowest rank's step time.

Secondary legs in the same JSON line (never part of `value`; at N > 1 only with --k4, and then every collective step is
preceded by an all-reduced "everyone is fine" flag so that a failure on one rank cannot strand the others in a collective
and take the primary line with it): BASELINE config 4 ("K4", 8000 trees) both ways --
row shards (bit-exact) and tree shards (all-reduce of float64 partials, and the bit-exact chain) -- each with its
error against a float64 CPU sum on a row sample; at N = 1 they are the one-GPU proxies of the 8-GPU run ("K4 on
R/8 rows" against "one 1000-tree shard on all R rows").
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MISSING = -999.0
PROFILE_ROUND = "r04"
PROFILE_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND)


def algorithmic_bytes_per_sample(T, depth, cols, bits_bytes):
    """the reference's own traversal byte model (main.cu:38-46) made exact for perfect
    depth-`depth` trees: per (row, tree) `depth` internal visits of one node record (4 B threshold +
    b B packed bits) and one feature (4 B), plus the leaf record; plus the row once and the prediction."""
    return T * (depth * (4 + bits_bytes + 4) + (4 + bits_bytes)) + cols * 4 + 4


def kernel_source_hash():
    """Identifies the kernels a profile was taken with: sha256 over the device sources, comments and white space
    removed (an edit of a comment does not make a profile stale)."""
    import re

    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "tahoe_amd", "csrc", "*.hip")) +
                       glob.glob(os.path.join(ROOT, "tahoe_amd", "csrc", "*.h"))):
        with open(path, "r", errors="replace") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"\s+", " ", text)
        h.update(os.path.basename(path).encode() + b"\0" + text.encode())
    return h.hexdigest()[:16]


def load_profile(name):
    """A committed rocprofv3 summary (tools/pmc.sh) -- only if it was taken with the kernels of this tree."""
    try:
        with open(os.path.join(PROFILE_DIR, name)) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        return None
    return prof if prof.get("src_hash") == kernel_source_hash() else None


def physical_ceilings(prof, walk_ms, quant_ms, rows, cols):
    """Per-kernel fractions of the ceilings that physically bind, from the PMC counters of the
    profiled run (ratios inside that run) next to the live kernel times of this one."""
    out = {"source": f"profiles/{PROFILE_ROUND}/pmc_k3.json", "src_hash": prof["src_hash"]}
    for key, kname in (("walk", "qring"), ("quantise", "quantize")):
        parts = [v for n, v in prof["kernels"].items() if kname in n and "bucket_index" not in n]
        if not parts:
            continue
        # a step may walk in two launches (waves of 192-row tiles + a 128-row remainder): one launch after the other, so
        # counters and cycles add up
        k = {c: sum(part.get(c, 0.0) for part in parts) for c in parts[0]}
        out.setdefault("kernels_per_step", {})[key] = len(parts)
        cyc = k["GRBM_GUI_ACTIVE"] / 8.0  # shader cycles of one launch (the counter sums the 8 XCDs)
        cus, simds = prof.get("num_cus", 256), prof.get("num_cus", 256) * 4
        vmem = max(k.get("SQ_ACTIVE_INST_VMEM", 0.0), k.get("SQ_INSTS_VMEM_RD", 0.0) + k.get("SQ_INSTS_VMEM_WR", 0.0))
        issue = (k["SQ_ACTIVE_INST_VALU"] + k["SQ_ACTIVE_INST_LDS"] + vmem) * 4.0 / (simds * cyc)
        fetch, write = k["FETCH_SIZE"] * 1024.0, k["WRITE_SIZE"] * 1024.0
        ms = walk_ms if key == "walk" else quant_ms
        ent = {
            "kernel_ms_live": round(ms, 4),
            "kernel_ms_profiled": round(cyc / (prof.get("clock_ghz", 2.4) * 1e6), 4),
            # one vector instruction (VALU, LDS or VMEM) per SIMD per 4 cycles: SQ_ACTIVE_INST_* count quad-cycles
            "vector_issue_busy": round(issue, 3),
            "valu_busy": round(k["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cyc), 3),
            "lds_busy": round(k["SQ_LDS_IDX_ACTIVE"] / (cus * cyc), 3),
            "texture_addr_busy": round(k["TA_TA_BUSY"] / (cus * cyc), 3),
            "texture_data_busy": round(k["TD_TD_BUSY"] / (cus * cyc), 3),
            "hbm_bytes_counters_raw": int(fetch + write),
            "hbm_GBps_counters_raw": round((fetch + write) / (ms * 1e-3) / 1e9, 1),
            "hbm_frac_counters_raw": round((fetch + write) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        }
        if key == "walk":
            # FETCH_SIZE counts half of a wide coalesced stream (MI355X_MICROARCH.md, HBM): the quantised tiles are one
            tile_stream = rows * cols * 2
            ent["hbm_bytes_counters_corrected"] = int(fetch + write + tile_stream / 2)
            ent["binds"] = ("texture data path: per (tree, tile) a 4-KiB top staged from L2 (4 wave-loads) + one divergent 64-lane gather per chain "
                            "and 16-byte piece of a bottom block, ~45 cycles each whatever the width; vector issue (VALU + LDS + VMEM per SIMD) behind it")
        else:
            moved = rows * cols * 4 + rows * cols * 2  # rows read once, codes written once
            ent["hbm_GBps_compulsory"] = round(moved / (ms * 1e-3) / 1e9, 1)
            ent["hbm_frac_compulsory"] = round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            ent["binds"] = "texture path (8 bytes used of every 64-byte line a lane pulls) above an HBM stream"
        out[key] = ent
    return out


# ---- the one honest fraction <= 1 the walk has: against the ceiling of the LDS-resident walk itself ----
def lds_walk_ceiling(walk_ms, rows, trees, depth, num_cus, clock_ghz):
    """node visits per second of the walk kernel against what tools/ubench_qwalk.hip measures for the LDS-resident part of
    the same walk alone (14 walker waves x 3 chains, 64-row regions, no global memory; the faster of its two level forms:
    both children by ds_read_b64 + select = 5 VALU + 2 LDS, or the chosen child after the compare = 4 + 2, the form the
    192- and 384-row tiles run): 6 vector instructions per 64-row level at best, one vector instruction per CU and cycle.  The unit is a "wave-level":
    one level of the walk for the 64 rows of one wave-chain.  Both times are wall times of whole kernels divided by the
    wave-levels per CU they contain, so the device clock cancels in `frac`."""
    wave_levels_per_cu = (rows / 64.0) * trees * depth / max(num_cus, 1)
    ns_kernel = walk_ms * 1e6 / wave_levels_per_cu
    out = {"unit": "ns per wave-level (64 rows x one tree level) per CU", "kernel": round(ns_kernel, 4),
           "kernel_clk_at_device_clock": round(ns_kernel * clock_ghz, 3),
           "node_visits_per_s": round(rows * trees * depth / (walk_ms * 1e-3), 1),
           "instruction_floor_clk": 6.0, "frac_of_instruction_floor": round(6.0 / (ns_kernel * clock_ghz), 4),
           "instruction_floor_means": "4 VALU + 2 LDS per chain-level, one vector instruction per CU and cycle at the device clock"}
    exe = os.path.join(ROOT, "tools", "ubench_qwalk")
    ub, source = None, None
    if os.path.exists(exe):  # live, on this device, as a child process (its own HIP context)
        import subprocess
        try:
            r = subprocess.run([exe, "--kernel-shape"], capture_output=True, text=True, timeout=60)
            for ln in r.stdout.splitlines():
                if ln.startswith("{"):
                    ub = json.loads(ln)
                    source = "tools/ubench_qwalk --kernel-shape, run live on this device"
        except (OSError, ValueError, subprocess.SubprocessError):
            ub = None
    if ub is None:  # the committed run of the same binary (2.4 GHz assumed in its clk figure)
        ub = {"ns_per_wave_level_per_cu": 7.55 / 2.4, "nw": 14, "k": 3}
        source = "profiles/r04/bench_default_line.json (tools/ubench_qwalk --kernel-shape: 7.55 clk at 2.4 GHz); live run unavailable"
    out.update({"ubench": round(ub["ns_per_wave_level_per_cu"], 4), "ubench_shape": f"{ub.get('nw')} walker waves x {ub.get('k')} chains",
                "ubench_source": source, "frac": round(ub["ns_per_wave_level_per_cu"] / ns_kernel, 4),
                "frac_means": "time the LDS-resident walk alone needs per wave-level (micro-benchmark, no tops staged, no bottom blocks, "
                              "no ring) / time the walk kernel takes per wave-level: <= 1 by construction, the share of the kernel's time "
                              "that its own inner loop accounts for at that loop's measured ceiling"})
    return out


# ---- BASELINE.json's other configurations (K1, K2, K5): workloads as tools/run_configs.py has always built them ----
BASELINE_SHAPES = {"K1": (500, 8, 18, 10_000, 11, 12, 0.05, 0.02), "K2": (500, 8, 3072, 100_000, 21, 22, 0.0, 0.0),
                   "K3": (1000, 12, 256, 1_000_000, 42, 43, 0.0, 0.0), "K4": (8000, 12, 256, 1_000_000, 42, 43, 0.0, 0.0)}
K5_SHAPE = {"trees": 2000, "cols": 256, "rows": 200_000, "min_depth": 4, "max_depth": 24, "leaf_prob": 0.32, "max_tree_nodes": 65535,
            "forest_seed": 44, "data_seed": 43}


def baseline_workload(ta, cfg, through_text_files=False):
    """(kind, forest description, host rows) of a BASELINE configuration.  kind "dense": (nodes, T, D, C); "sparse": (nodes, trees, C)."""
    if cfg == "K5":
        k = K5_SHAPE
        sn, tr = ta.capi.synth_sparse_forest(k["trees"], k["cols"], k["min_depth"], k["max_depth"], k["leaf_prob"], k["max_tree_nodes"], k["forest_seed"])
        return "sparse", (sn, tr, k["cols"]), ta.synth_data(k["rows"], k["cols"], seed=k["data_seed"])
    if cfg == "KR3":  # K3's shape from the histogram-style generator (254 thresholds per feature = LightGBM's max_bin 255): what
        T, D, C, R = 1000, 12, 256, 1_000_000  # trained models look like
        nodes = ta.synth_forest_hist(T, D, C, seed=42, feature_seed=7, max_bins=254, zipf_s=1.0, leaf_prob=0.02, scale_decades=3.0)
        return "dense", (nodes, T, D, C), ta.synth_data_hist(R, C, seed=43, feature_seed=7, scale_decades=3.0)
    T, D, C, R, fs, ds, lp, mp = BASELINE_SHAPES[cfg]
    nodes = ta.synth_forest(T, D, C, seed=fs, leaf_prob=lp)
    data = ta.synth_data(R, C, seed=ds, missing_prob=mp, missing=MISSING)
    if through_text_files:  # K1: the reference's own file formats (BaseTahoeTest.h:267-402), written and parsed back
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            ta.write_model(os.path.join(d, "m.txt"), nodes, T, D)
            ta.write_data(os.path.join(d, "d.txt"), data, MISSING)
            n2, T2, D2 = ta.load_model(os.path.join(d, "m.txt"))
            x2, miss = ta.load_data(os.path.join(d, "d.txt"))
        if n2.tobytes() != nodes.tobytes() or x2.tobytes() != data.tobytes() or (T2, D2) != (T, D):
            raise RuntimeError("text formats did not round-trip")
        nodes, data = n2, x2
    return "dense", (nodes, T, D, C), data


def config_counters(cfg, ms):
    """Busy fractions and HBM-side bytes of configuration `cfg` from its stamped counter profile (tools/pmc_script.sh over
    tools/pmc_target.py), per predict; None when the profile is missing or was taken with other kernel sources."""
    try:
        with open(os.path.join(PROFILE_DIR, "pmc_%s.json" % cfg.lower())) as fh:
            prof = json.load(fh)
        if prof.get("src_hash") != kernel_source_hash():
            return None
        n = int(prof["script"].split()[-1])
        ks = [v for k, v in prof["kernels"].items() if "bucket_index" not in k and "threshold_keys" not in k]  # (not the create-time kernels)
        tot = lambda c: sum(v.get(c, 0.0) * v["launches"] for v in ks) / n  # per predict
        cyc = tot("GRBM_GUI_ACTIVE") / 8.0
        cus = prof.get("num_cus", 256)
        vmem = max(tot("SQ_ACTIVE_INST_VMEM"), tot("SQ_INSTS_VMEM_RD"))
        units = {"vector_issue": (tot("SQ_ACTIVE_INST_VALU") + tot("SQ_ACTIVE_INST_LDS") + vmem) * 4.0 / (cus * 4 * cyc),
                 "valu": tot("SQ_ACTIVE_INST_VALU") * 4.0 / (cus * 4 * cyc), "lds_array": tot("SQ_LDS_IDX_ACTIVE") / (cus * cyc),
                 "texture_addr": tot("TA_TA_BUSY") / (cus * cyc), "texture_data": tot("TD_TD_BUSY") / (cus * cyc)}
        hbm = (tot("FETCH_SIZE") + tot("WRITE_SIZE")) * 1024.0
        return {"profile": f"profiles/{PROFILE_ROUND}/pmc_{cfg.lower()}.json", "src_hash": prof["src_hash"],
                "scope": "all kernels of one predict (pre-pass + walk), i.e. step-wide",
                "kernels_per_predict": round(sum(v["launches"] for v in ks) / n, 2),
                "kernel_ms_profiled": round(cyc / (prof.get("clock_ghz", 2.4) * 1e6), 4),
                "hbm_bytes_raw": int(hbm), "hbm_frac_raw": round(hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "hbm_note": "FETCH_SIZE + WRITE_SIZE, raw (FETCH_SIZE counts half of wide coalesced reads on gfx950)",
                "l2_to_l1_bytes": int(tot("TCP_TCC_READ_REQ") * 128), "busy": {k: round(v, 3) for k, v in units.items()},
                "busiest_unit": max(units, key=units.get)}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return None


def config_roofline(ms, rows, cols, trees, len_sum_per_row, node_bytes, n_nodes, cfg, leaf_bytes=None):
    """SURVEY.md 8(d) for one configuration: the reference's traversal byte model made exact -- per (row, tree) `len` internal
    visits of one node record (`node_bytes` = 4 + b for the dense SoA format, 12 for sparse_node_t) and one 4-byte feature, plus
    the leaf record; plus the row once and the prediction -- over the measured time and the HBM peak; the compulsory bytes (rows +
    forest + predictions once) the same way; the stamped counters of the configuration beside them."""
    leaf_bytes = node_bytes if leaf_bytes is None else leaf_bytes
    alg = rows * (len_sum_per_row * (node_bytes + 4) + trees * leaf_bytes + cols * 4 + 4)
    comp = rows * cols * 4 + n_nodes * node_bytes + rows * 4
    t = ms * 1e-3
    return {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "algorithmic_bytes": int(alg), "achieved": round(alg / t / 1e9, 1),
            "frac": round(alg / t / 1e9 / HBM_PEAK_GBPS, 4), "compulsory_bytes": int(comp),
            "compulsory_GBps": round(comp / t / 1e9, 1), "compulsory_frac": round(comp / t / 1e9 / HBM_PEAK_GBPS, 5),
            "mean_path_len": round(len_sum_per_row / trees, 3), "counters": config_counters(cfg, ms)}


def dense_path_len_sum(leaf_idx):
    """Internal nodes visited per (row, tree) = level of the leaf the walk ended on = floor(log2(heap index + 1)); mean row sum."""
    return float(np.floor(np.log2(leaf_idx.astype(np.float64) + 1.0)).sum(axis=1).mean())


def sparse_path_len_sum(sn, tr, want_leaf):
    """The same for sparse forests: level of every node of every tree (children lie behind their parent), then of each leaf reached."""
    sizes = np.diff(np.append(tr, sn.size))
    level = np.zeros(sn.size, dtype=np.int32)
    frontier = tr.astype(np.int64)
    root_of = np.repeat(tr.astype(np.int64), sizes)
    lvl = 0
    while frontier.size:
        level[frontier] = lvl
        inner = frontier[sn["bits"][frontier] >= 0]  # is_leaf is the sign bit
        kids = root_of[inner] + sn["left_idx"][inner]
        frontier = np.concatenate([kids, kids + 1])
        lvl += 1
    return float(level[tr.astype(np.int64)[None, :] + want_leaf.astype(np.int64)].sum(axis=1).mean())


def config_legs(ta, torch, which=("K1", "K2", "K5", "KR3"), warmup=5, steps=20, check_rows=2048):
    """BASELINE.json's configurations other than the metric's (K3) and K4 -- plus KR3, K3's shape from the histogram-style generator
    (254 thresholds per feature, skewed feature usage, early leaves: what trained models look like; QRING walks it on u8 codes) --
    timed on this device after the primary region (never part of `value`): 5 warm-ups + 20 timed predicts with the in-library hipEvents on resident inputs, which kernel form ran,
    the byte-model and compulsory-byte fractions, the configuration's stamped counters, and a bitwise check of the timed
    launches' output on >= 2048 rows against the CPU oracle (the checker, called after the timing)."""
    oracle = _oracle()
    legs = {}
    for cfg in which:
        t_leg = time.perf_counter()
        try:
            kind, desc, data = baseline_workload(ta, cfg, through_text_files=(cfg == "K1"))
            rows, cols = data.shape
            if kind == "dense":
                nodes, T, D, C = desc
                forest = ta.Forest(nodes, T, D, C, missing=MISSING)
            else:
                sn, tr, C = desc
                T = int(tr.size)
                forest = ta.capi.SparseForest(sn, tr, C, missing=MISSING)
            forest.reserve(rows)
            x = torch.from_numpy(data).cuda()
            out = torch.empty(rows, dtype=torch.float32, device="cuda")
            for _ in range(warmup):
                forest.predict_raw(x, out)
            forest.set_profiling(steps)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for _ in range(steps):
                forest.predict_raw(x, out)
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - tw) / steps * 1e3
            forest.check()
            walk, pre = forest.kernel_times_ms(), forest.prepass_times_ms()
            forest.set_profiling(0)
            ms = float(np.mean(walk) + (np.mean(pre) if len(pre) else 0.0))
            info = forest.info()
            strategy = ta.STRATEGY_NAMES.get(forest.get_strategy(rows), "?")
            leg = {"workload": (f"{cfg}: {T} trees depth {D}, {cols} features, {rows} rows" if kind == "dense" else
                                f"{cfg}: sparse forest {T} trees depth {K5_SHAPE['min_depth']}-{K5_SHAPE['max_depth']}, {int(sn.size)} nodes, "
                                f"{cols} features, {rows} rows") + (", through the text file formats" if cfg == "K1" else "") +
                               (", histogram-style generator (<= 254 thresholds per feature)" if cfg == "KR3" else ""),
                   "strategy": ("sparse_" if kind == "sparse" else "") + strategy, "kernel_form": forest.kernel_form(rows),
                   "ms": round(ms, 4), "ms_source": f"hipEvents inside the library, mean of {len(walk)} predicts (pre-pass + walk)",
                   "prepass_ms": round(float(np.mean(pre)) if len(pre) else 0.0, 4), "ms_min": round(float(np.min(np.asarray(walk) + np.asarray(pre))), 4),
                   "wall_ms_per_predict": round(wall_ms, 4), "samples_per_s": round(rows / (ms * 1e-3), 1),
                   "stream_slots": int(info.stream_slots), "ring_rows": int(info.ring_rows), "qring_tile_rows": int(info.qring_tile_rows)}
            # ---- the checker: the timed launches' own output on a strided row sample, bit for bit ----
            sample = np.unique(np.linspace(0, rows - 1, num=min(check_rows, rows)).astype(np.int64))
            got = out.cpu().numpy()[sample]
            if kind == "dense":
                want, want_leaf = oracle.predict(nodes, T, D, data[sample], MISSING, want_leaf=True, threads=min(os.cpu_count() or 1, 16))
                len_sum, node_bytes, n_nodes = dense_path_len_sum(want_leaf), 4 + info.bits_bytes, T * ta.capi.tree_num_nodes(D)
            else:
                want, want_leaf = oracle.sparse_predict(sn, tr, data[sample], MISSING, want_leaf=True, threads=min(os.cpu_count() or 1, 16))
                len_sum, node_bytes, n_nodes = sparse_path_len_sum(sn, tr, want_leaf), 12, int(sn.size)
            leg["rows_checked"] = int(sample.size)
            leg["bitwise_equal_to_cpu_oracle"] = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
            leg["roofline"] = config_roofline(ms, rows, cols, T, len_sum, node_bytes, n_nodes, cfg)
            if not leg["bitwise_equal_to_cpu_oracle"]:
                leg["error"] = f"{cfg}: GPU sums differ from the CPU oracle on the sampled rows"
            forest.close()
            del x, out
        except Exception as err:  # the primary line must survive a failure of a secondary leg
            leg = {"error": f"{type(err).__name__}: {err}"}
        leg["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
        legs[cfg] = leg
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--cols", type=int, default=256)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--shard", choices=["rows", "trees"], default="rows")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 with row shards: strong = the metric's rows split over the ranks (default), weak = that many rows per rank")
    ap.add_argument("--tree-mode", choices=["allreduce64", "allreduce32", "chain"], default="allreduce64")
    ap.add_argument("--chunk-rows", type=int, default=32768, help="rows per hand-over of the chained tree shards")
    ap.add_argument("--strategy", type=int, default=0, help="0 auto, 1 direct, 2 rowtile, 3 tileblock, 4 tilering, 5 qring")
    ap.add_argument("--cpu-rows", type=int, default=100_000, help="rows of the batch timed on the CPU oracle")
    ap.add_argument("--relayout", action="store_true",
                    help="probability-guided re-layout (SURVEY 8f N3): weights = reach probabilities, TAHOE_CREATE_PROB_RELAYOUT")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-resident (PCIe-inclusive) leg")
    ap.add_argument("--no-k4", "--no-tree-leg", dest="no_k4", action="store_true", help="skip the secondary K4 legs")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary legs of BASELINE configurations K1, K2, K5")
    ap.add_argument("--cpu-rows-multi", type=int, default=20_000, help="N > 1: rows of rank 0's shard timed on the CPU oracle")
    ap.add_argument("--k4", action="store_true", help="N > 1: run the secondary K4 legs too (they contain collectives; off by default there)")
    ap.add_argument("--k4-trees", type=int, default=8000)
    ap.add_argument("--k4-chain", action="store_true",
                    help="N > 1: also time the chained (bit-exact, point-to-point) tree shards of K4; off by default because a stuck "
                         "send/recv would take the process group -- and the primary line -- down with it")
    ap.add_argument("--k4-sample", type=int, default=2048, help="rows of the K4 legs checked against the float64 CPU sum")
    args = ap.parse_args()

    import torch

    import tahoe_amd as ta
    from tahoe_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # TAHOE_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through host memory); the numbers of such a run mean nothing, the code path is the same.
    backend = os.environ.get("TAHOE_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dist = None
    # TAHOE_BENCH_FORCE_DIST=1 with one rank: the process group is created anyway (world_size 1 on the real backend) and every
    # collective of the N > 1 path -- the ones all-reduce, barrier, MAX / MIN all-reduces -- runs through it, so that this
    # file's torch.distributed surface meets RCCL on a one-GPU box before the driver's 8-GPU run does.
    force_dist = world == 1 and os.environ.get("TAHOE_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29531")
        kw = {"world_size": 1, "rank": 0} if force_dist else {}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=datetime.timedelta(seconds=300), **kw)
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300), **kw)

    class HostHop:
        """torch.distributed through host memory (the gloo rehearsal): same calls, CUDA tensors staged on the CPU."""

        ReduceOp = dist.ReduceOp if dist else None

        @staticmethod
        def _via_host(fn, t, **kw):
            host = t.cpu()
            fn(host, **kw)
            t.copy_(host)

        def all_reduce(self, t, **kw):
            self._via_host(dist.all_reduce, t, **kw)

        def recv(self, t, **kw):
            self._via_host(dist.recv, t, **kw)

        def send(self, t, **kw):
            dist.send(t.cpu(), **kw)

        def broadcast(self, t, **kw):
            self._via_host(dist.broadcast, t, **kw)

        def get_rank(self, group=None):
            return dist.get_rank(group)

        def get_world_size(self, group=None):
            return dist.get_world_size(group)

    coll = dist if (backend == "nccl" or dist is None) else HostHop()

    def all_reduce(t, op=None):
        kw = {} if op is None else {"op": op}
        coll.all_reduce(t, **kw)  # nccl = RCCL over xGMI

    # ---- N > 1: the run proves what it ran on -- ranks the process group reports, an all-reduce of ones over them, the
    # device of every rank (n_gpus in the line is this count, not an environment variable) ----
    collective = None
    if use_dist:
        ranks_seen = int(dist.get_world_size())
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        all_reduce(ones)
        devs = torch.zeros(ranks_seen, dtype=torch.float64, device="cuda")
        devs[dist.get_rank()] = float(device_index)
        all_reduce(devs)
        collective = {"backend": "nccl (RCCL)" if backend == "nccl" else backend, "world_size_env": world, "ranks_seen": ranks_seen,
                      "allreduce_ones_sum": float(ones.item()), "allreduce_ones_ok": bool(ones.item() == float(ranks_seen)),
                      "device_of_rank": [int(v) for v in devs.cpu().tolist()],
                      "device_name": torch.cuda.get_device_name(device_index)}
        if ranks_seen != args.gpus or not collective["allreduce_ones_ok"]:
            raise SystemExit(f"bench: --gpus {args.gpus} but the process group has {ranks_seen} ranks / all-reduce of ones = {ones.item()}")

    T, D, C, R = args.trees, args.depth, args.cols, args.rows
    nodes = ta.synth_forest(T, D, C, seed=42)
    if args.relayout:
        ta.capi.set_probability_weights(nodes, T, D)
    n_per_tree = ta.capi.tree_num_nodes(D)
    tree_sharded = world > 1 and args.shard == "trees"
    weak = world > 1 and args.shard == "rows" and args.scaling == "weak"
    if tree_sharded:
        my_rows, first_row = R, 0
    elif weak:
        my_rows, first_row = R, rank * R
    else:  # strong scaling over rows (N = 1: everything)
        lo_r, hi_r = sharding.shard_bounds(R, rank, world)
        my_rows, first_row = hi_r - lo_r, lo_r
    data = ta.synth_data(my_rows, C, seed=43, first_row=first_row)
    x = torch.from_numpy(data).cuda()
    preds = torch.empty(my_rows, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    if tree_sharded:
        tsf = sharding.TreeShardedForest(
            nodes, T, D,
            make_local=lambda my_nodes, my_trees: ta.Forest(my_nodes, my_trees, D, C, missing=MISSING, strategy=args.strategy),
            finish=lambda sums: ta.capi.transform_preds(sums, 0, T, 0.0, 0.0, stream=stream),
            mode=args.tree_mode, chunk_rows=args.chunk_rows)
        tsf.dist = coll
        forest = tsf.local
        forest.set_strategy(args.strategy)
        my_T = forest.num_trees
    else:
        forest = ta.Forest(nodes, T, D, C, missing=MISSING, relayout=args.relayout)
        forest.set_strategy(args.strategy)
        my_T = T
    forest.reserve(min(my_rows, args.chunk_rows) if (tree_sharded and args.tree_mode == "chain") else my_rows)
    info = forest.info()

    def step():
        if tree_sharded:
            tsf.predict(x, preds)
        else:
            forest.predict(x, preds, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
        all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def agree(ok):
        """Every rank learns whether ALL ranks are fine (one all-reduce of a flag): a failure on one rank makes every rank
        leave the leg together instead of stranding the others in the next collective."""
        if not use_dist:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda")
        all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    for _ in range(args.warmup):
        step()
    forest.set_profiling(args.steps if not (tree_sharded and args.tree_mode == "chain") else 0)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local)
    if use_dist:  # the spread over the ranks, for the record (value uses the slowest)
        t_min = torch.tensor([dt_local], dtype=torch.float64, device="cuda")
        all_reduce(t_min, op=dist.ReduceOp.MIN)
        collective["ms_per_step_fastest_rank"] = round(float(t_min.item()) / args.steps * 1e3, 4)
        collective["ms_per_step_slowest_rank"] = round(dt / args.steps * 1e3, 4)
    forest.check()  # a raised error flag (bounded ring wait) voids the run
    kernel_ms = forest.kernel_times_ms()
    prepass_ms = forest.prepass_times_ms()
    forest.set_profiling(0)

    ms_per_step = dt / args.steps * 1e3
    total_rows = R * world if weak else R
    value = total_rows / (dt / args.steps)

    # ---- roofline of the step's kernels (this rank's launch) ----
    # `achieved` follows SURVEY.md 8(d) and the task's contract: the reference's ALGORITHMIC traversal bytes of the rows
    # and trees this launch processes, divided by the measured kernel time of the WHOLE step (quantise pre-pass + walk).
    # Those bytes are mostly served from LDS (2-byte rank codes, 4-byte nodes), which is the point of the layout, so the
    # figure exceeds the HBM peak: it says how the step compares with the reference's byte model run at HBM speed, not
    # that HBM moves that much.  What physically binds is under "physical" (PMC counters of profiles/<round>) and
    # "hbm_frac_measured" (counter bytes over the same time, <= 1).
    strategy_name = ta.STRATEGY_NAMES.get(forest.get_strategy(my_rows), "?")
    b_alg = algorithmic_bytes_per_sample(my_T, D, C, info.bits_bytes) * my_rows
    have_times = len(kernel_ms) > 0
    # (chained tree shards issue one launch per row chunk: kernel events are off there, the step's wall time serves)
    walk_ms = float(np.mean(kernel_ms)) if have_times else ms_per_step
    quant_ms = float(np.mean(prepass_ms)) if len(prepass_ms) else 0.0
    step_kernel_ms = walk_ms + quant_ms
    achieved = b_alg / (step_kernel_ms * 1e-3) / 1e9
    k3_shape = (T, D, C, R) == (1000, 12, 256, 1_000_000) and world == 1 and strategy_name == "qring"
    prof = load_profile("pmc_k3.json") if k3_shape else None
    traffic = None
    physical = None
    if prof:
        traffic = int(sum((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 for n, k in prof["kernels"].items() if "bucket_index" not in n))
        physical = physical_ceilings(prof, walk_ms, quant_ms, my_rows, C)
    if rank == 0 and have_times and forest.kernel_form(my_rows).startswith("qring_region"):
        # the fraction <= 1 of this kernel: its node visits per second against the measured ceiling of the LDS-resident walk
        prop = torch.cuda.get_device_properties(device_index)
        physical = physical or {"source": None, "note": "no counter profile stamped with these kernel sources"}
        physical.setdefault("walk", {})["lds_walk_ceiling"] = lds_walk_ceiling(
            walk_ms, my_rows, my_T, D, prop.multi_processor_count, getattr(prop, "clock_rate", 2_400_000) / 1e6)
    compulsory = my_rows * C * 4 + my_T * n_per_tree * (4 + info.bits_bytes) + my_rows * 4
    roofline = {
        "bound": "hbm", "kernel": f"{strategy_name}: quantise pre-pass + walk" if quant_ms else f"{strategy_name}_kernel",
        "bound_note": "the contract's two bounds are hbm | mfma and this path has no MFMA; what physically binds the walk is vector-"
                      "instruction issue + per-wave latency (LDS-resident), the pre-pass the texture path -- see physical",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4),
        "model_bytes_ratio": round(achieved / HBM_PEAK_GBPS, 4),
        "frac_means": "algorithmic (reference byte model, SURVEY 8d) bytes / (pre-pass + walk kernel time) / HBM peak; > 1 because "
                      "the visits are served from LDS -- see hbm_frac_measured and physical for the ceilings that bind",
        "traffic": traffic,
        "traffic_unit": "HBM-side bytes per step, FETCH_SIZE + WRITE_SIZE of both kernels, raw counters of the rocprofv3 --pmc "
                        f"passes in profiles/{PROFILE_ROUND}/pmc_k3.json (null when that profile was not taken with these kernel sources)",
        "hbm_frac_measured": round(traffic / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
        "kernel_ms_avg": round(step_kernel_ms, 4),
        "kernel_ms_source": "hipEvents on the launch stream, inside the library" if have_times else "wall time of the step",
        "walk_kernel_ms_avg": round(walk_ms, 4), "walk_kernel_ms_min": round(float(np.min(kernel_ms)), 4) if have_times else None,
        "prepass_kernel_ms_avg": round(quant_ms, 4),
        "frac_walk_only": round(b_alg / (walk_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "algorithmic_bytes_per_launch": b_alg,
        "compulsory_bytes_per_launch": compulsory,
        "compulsory_frac": round(compulsory / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
        "physical": physical,
        "frac_le_1": (physical or {}).get("walk", {}).get("lds_walk_ceiling", {}).get("frac"),
        "frac_le_1_means": "physical.walk.lds_walk_ceiling.frac: the walk kernel against the measured ceiling of its own LDS-resident "
                           "inner loop (tools/ubench_qwalk.hip); `frac` above is the contract's byte-model figure and exceeds 1",
        "kernel_form": forest.kernel_form(my_rows),
        "kernel_source_hash": kernel_source_hash(),
    }

    # ---- secondary legs: BASELINE config 4 ("K4": 8000 trees).  Which split serves it?  Measured both ways, each with
    # its error against a float64 CPU sum on a row sample.  N = 1: the one-GPU proxies of the 8-GPU run. ----
    k4 = None
    want_k4 = args.k4 or (world == 1 and not args.no_k4 and (T, D, C, R) == (1000, 12, 256, 1_000_000))
    if want_k4 and not tree_sharded:
        try:
            k4 = k4_legs(args, ta, sharding, torch, coll, dist, world, rank, fence, max_over_ranks, stream, agree)
        except Exception as err:  # the primary line must survive a failure of a secondary leg
            k4 = {"error": f"{type(err).__name__}: {err}"}

    # ---- secondary legs: BASELINE configurations K1, K2, K5 on this device (N = 1 only, after the primary region) ----
    configs = None
    if rank == 0 and world == 1 and not args.no_configs and (T, D, C, R) == (1000, 12, 256, 1_000_000):
        t_cfg = time.perf_counter()
        configs = config_legs(ta, torch)
        configs["K3"] = {"see": "the primary figures of this line", "ms": round(step_kernel_ms, 4), "kernel_form": roofline["kernel_form"]}
        configs["K4"] = {"see": "k4 in this line (row shards / tree shards on one GPU)"}
        configs["seconds"] = round(time.perf_counter() - t_cfg, 2)

    # ---- host-resident batch (rank 0, N = 1 only): the PCIe-inclusive rate, reported beside `value`, never as it ----
    host_leg = None
    if rank == 0 and world == 1 and not args.no_host:
        fence()
        ref = preds.cpu().numpy()
        host_leg = {"unit": "samples/s", "note": "tahoe_forest_predict_host: rows start in host memory, chunked upload "
                    "overlapped with the traversal, predictions back in host memory; PCIe-inclusive, not `value`"}
        pin = ta.PinnedArray(R, C)
        pin.array[:] = data
        out_h = np.empty(R, dtype=np.float32)
        for name, src in (("pinned", pin.array), ("pageable", data)):
            forest.predict_host(src, out_h)  # creates the buffers / warms up
            reps = 3
            th = time.perf_counter()
            for _ in range(reps):
                forest.predict_host(src, out_h)
            th = (time.perf_counter() - th) / reps
            host_leg[name] = {"value": round(R / th, 1), "ms_per_batch": round(th * 1e3, 3),
                              "GBps_over_link": round(R * C * 4 / th / 1e9, 2),
                              "bitwise_equal_to_resident": bool(np.array_equal(out_h.view(np.uint32), ref.view(np.uint32)))}
            if not host_leg[name]["bitwise_equal_to_resident"]:
                raise SystemExit("bench: host-pipeline predictions differ from the resident-batch predictions")
        pin.close()

    # ---- CPU baseline + parity spot check (rank 0; N > 1: a shorter leg on the first rows of rank 0's own shard, so that
    # every line of a scaling run carries its CPU figure from the same run; the other ranks wait at the final barrier) ----
    cpu = None
    if rank == 0 and not args.no_cpu:
        oracle = _oracle()
        n_cpu = min(args.cpu_rows if world == 1 else args.cpu_rows_multi, my_rows)
        tc = time.perf_counter()
        want, _ = oracle.predict(nodes, T, D, data[:n_cpu], MISSING, threads=1)
        cpu_s = time.perf_counter() - tc
        got = preds[:n_cpu].cpu().numpy()
        # (tree shards: rank 0's buffer holds the all-reduced sums, which are not the sequential float32 sum -- DESIGN.md 7 -- or,
        # chained, another rank holds the result: timed, not compared)
        exact = None if tree_sharded else bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        cpu = {
            "value": round(n_cpu / cpu_s, 1), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"first {n_cpu} rows of " + ("the batch" if world == 1 else f"rank 0's shard (rows {first_row}..)") +
                      f", all {T} trees, single thread (the reference's predict_on_cpu is single-threaded)",
            "seconds": round(cpu_s, 2),
            "gpu_matches_cpu_bitwise_on_sample": exact,
        }
        if world == 1:
            ncores = os.cpu_count() or 1
            n_all = min(R, n_cpu * min(ncores, 16))
            tc = time.perf_counter()
            oracle.predict(nodes, T, D, data[:n_all], MISSING, threads=ncores)
            cpu_all_s = time.perf_counter() - tc
            cpu["all_cores"] = {"value": round(n_all / cpu_all_s, 1), "cores": ncores, "rows": n_all, "seconds": round(cpu_all_s, 2),
                                "note": "the same scalar port with the rows cut into one block per hardware thread -- every thread "
                                        "walks the whole 98 MB AoS forest, so this scales far below the core count; a stated baseline "
                                        "of this port, not what the host could do with a tuned CPU traversal"}
        if exact is False:
            raise SystemExit("bench: GPU sums differ from the CPU oracle on the sampled rows")

    if rank == 0:
        sharding_name = "none" if world == 1 else (f"trees/{args.tree_mode}" if tree_sharded else "rows")
        out = {
            "metric": "samples/sec, 1000-tree depth-12 forest @1M rows (batched tree-ensemble traversal)",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": collective["ranks_seen"] if collective else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "us_per_batch": round(ms_per_step * 1e3, 1),
            "higher_is_better": True,
            # the default split at N > 1 is strong (the metric's rows over the ranks); N = 1 carries the same label so that the
            # driver's N = 1 row is comparable with its N > 1 rows
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"K3: synthetic forest {T} trees depth {D}, {C} features, {R} rows"
                                   + (" per GPU" if weak else (f" split over {world} GPUs" if world > 1 else "")),
                       "trees": T, "depth": D, "cols": C, "rows_per_step": total_rows, "rows_per_gpu": my_rows,
                       "sharding": sharding_name, "strategy": strategy_name,
                       **({"relayout_swaps": int(info.relayout_swaps)} if args.relayout else {})},
            "collective": collective,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "host_pipeline": host_leg,
            "k4": k4,
            "configs": configs,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()  # rank 0's CPU leg ends here: nobody tears the group down under it
        dist.destroy_process_group()


def k4_legs(args, ta, sharding, torch, coll, dist, world, rank, fence, max_over_ranks, stream, agree):
    """BASELINE config 4 both ways.  Every rank builds the 8000-tree forest description (counter-based generator: tree t
    is the same everywhere) and keeps what its role needs.  The CPU oracle appears here only as the checker of the sums
    the timed launches produced (never inside a timed region)."""
    oracle = _oracle()
    T4, D, C, R = args.k4_trees, args.depth, args.cols, args.rows
    G = 8  # the configuration's GPU count: what the N = 1 proxies stand for
    per_tree = ta.capi.tree_num_nodes(D)
    nodes4 = ta.synth_forest(T4, D, C, seed=45)
    k4_steps, k4_warm = max(3, min(args.steps, 10)), 2
    sample = min(args.k4_sample, R // max(world, 1) if world > 1 else R // G)

    def timed(fn):
        for _ in range(k4_warm):
            fn()
        fence()
        t = time.perf_counter()
        for _ in range(k4_steps):
            fn()
        fence()
        return max_over_ranks(time.perf_counter() - t) / k4_steps

    def errors(got32, rows_data, ncpu):
        """Errors of float32 sums `got32` for rows `rows_data` against the float64 CPU sum, beside the CPU float32 sum's own."""
        exact = oracle.predict_f64_mt(nodes4, T4, D, rows_data, MISSING, threads=ncpu)
        cpu32, _ = oracle.predict(nodes4, T4, D, rows_data, MISSING, threads=ncpu)
        a = oracle.abs_leaf_sum(nodes4, T4, D, rows_data, MISSING, threads=ncpu)
        e_got = np.abs(got32.astype(np.float64) - exact)
        e_cpu = np.abs(cpu32.astype(np.float64) - exact)
        rel_vs_cpu = np.abs(got32.astype(np.float64) - cpu32.astype(np.float64)) / np.maximum(np.abs(cpu32.astype(np.float64)), 1e-30)
        return {"rows_checked": int(len(exact)), "max_abs_err_vs_f64": float(e_got.max()),
                "cpu_f32_max_abs_err_vs_f64": float(e_cpu.max()),
                "bit_identical_to_cpu_f32": bool(np.array_equal(got32.view(np.uint32), cpu32.view(np.uint32))),
                "max_rel_diff_vs_cpu_f32": float(rel_vs_cpu.max())}, exact, a

    ncpu = min(os.cpu_count() or 1, 32)
    legs = {"workload": f"K4: synthetic forest {T4} trees depth {D}, {C} features, {R} rows",
            "note": "secondary figures, never part of `value`"}

    # (a) row shards: the whole forest on every rank, R/N rows each (N = 1: R/8 rows = one of eight ranks' share)
    n_ranks_modelled = world if world > 1 else G
    lo, hi = sharding.shard_bounds(R, rank if world > 1 else 0, n_ranks_modelled)
    fa = xa = pa = None
    try:  # local part: no collective inside
        xa = torch.from_numpy(ta.synth_data(hi - lo, C, seed=43, first_row=lo)).cuda()
        pa = torch.empty(hi - lo, dtype=torch.float32, device="cuda")
        fa = ta.Forest(nodes4, T4, D, C, missing=MISSING)
        fa.reserve(hi - lo)
        fa.predict(xa, pa, stream=stream)
        fa.check()
        ok, why = True, None
    except Exception as err:
        ok, why = False, f"{type(err).__name__}: {err}"
    if agree(ok):
        ta_s = timed(lambda: fa.predict(xa, pa, stream=stream))
        leg = {"sharding": "rows", "collective": "none", "rows_per_gpu": hi - lo, "trees_per_gpu": T4,
               "tree_groups_per_gpu": fa.info().qring_groups, "ms_per_step": round(ta_s * 1e3, 4),
               "value": round(R / ta_s, 1), "unit": "samples/s",
               "value_means": f"{R} rows / time of one rank's {hi - lo} rows" + ("" if world > 1 else f" (projection to {G} GPUs)")}
        if rank == 0:
            try:
                fa.check()
                e, _, _ = errors(pa[:sample].cpu().numpy(), xa[:sample].cpu().numpy(), ncpu)
                leg.update(e)
                if not e["bit_identical_to_cpu_f32"]:  # reported, not raised
                    leg["error"] = "K4 row shard: sums differ from the CPU oracle"
            except Exception as err:
                leg["error"] = f"{type(err).__name__}: {err}"
    else:
        leg = {"sharding": "rows", "error": why or "another rank failed to set this leg up"}
    legs["row_sharded"] = leg
    if fa is not None:
        fa.close()
    del fa, pa, xa

    # (b) tree shards: T4/N trees per rank, all R rows; N = 1: one of eight shards (the all-reduce is not in the time)
    n_shards = world if world > 1 else G
    s_lo, s_hi = sharding.shard_bounds(T4, rank if world > 1 else 0, n_shards)
    fb = x4 = p4 = None
    try:
        x4 = torch.from_numpy(ta.synth_data(R, C, seed=43, first_row=0)).cuda()
        p4 = torch.empty(R, dtype=torch.float32, device="cuda")
        fb = ta.Forest(nodes4[s_lo * per_tree: s_hi * per_tree], s_hi - s_lo, D, C, missing=MISSING)
        fb.reserve(R)
        fb.predict_raw(x4, p4, stream=stream)
        fb.check()
        ok, why = True, None
    except Exception as err:
        ok, why = False, f"{type(err).__name__}: {err}"
    if not agree(ok):
        legs["tree_sharded_allreduce64"] = {"sharding": "trees", "error": why or "another rank failed to set this leg up"}
        if fb is not None:
            fb.close()
        return legs
    for mode in (("allreduce64", "chain") if (world > 1 and args.k4_chain) else ("allreduce64",)):
        if world > 1:
            tsf = sharding.TreeShardedForest.__new__(sharding.TreeShardedForest)
            tsf.dist, tsf.group, tsf.rank, tsf.world, tsf.num_trees = coll, None, rank, world, T4
            tsf.mode, tsf.chunk_rows, tsf.local, tsf._wide = mode, args.chunk_rows, fb, None
            tsf.tree_range = (s_lo, s_hi)
            tsf.finish = lambda sums: sums
            tb_s = timed(lambda: tsf.predict_sums(x4, p4))
        else:
            tb_s = timed(lambda: fb.predict_raw(x4, p4, stream=stream))
        leg = {"sharding": "trees", "combine": mode, "rows_per_gpu": R, "trees_per_gpu": s_hi - s_lo,
               "ms_per_step": round(tb_s * 1e3, 4), "value": round(R / tb_s, 1), "unit": "samples/s",
               "collective": ("one all-reduce of 8 B/row (float64 partials)" if mode == "allreduce64" else
                              f"point-to-point hand-over of 4 B/row in chunks of {args.chunk_rows} rows") if world > 1 else
                             "not in the time (1 GPU: one of 8 shards); 8 MB all-reduce over xGMI expected << 1 ms"}
        try:
            fb.check()
            ok, why = True, None
        except Exception as err:
            ok, why = False, f"{type(err).__name__}: {err}"
        if not agree(ok):
            leg["error"] = why or "another rank's forest raised its error flag"
            legs["tree_sharded_" + mode] = leg
            break
        if world > 1:
            got = p4[:sample].clone()
            if mode == "chain":  # the last rank of the chain holds the sums: hand the checked rows to rank 0, which reports
                coll.broadcast(got, src=world - 1)
            if rank == 0:
                try:
                    e, exact, a = errors(got.cpu().numpy(), x4[:sample].cpu().numpy(), ncpu)
                    if mode == "chain" and not e["bit_identical_to_cpu_f32"]:
                        leg["error"] = "K4 chained tree shards: sums differ from the CPU oracle"
                    if mode == "allreduce64":
                        bound = sharding.sum_error_bound(a, exact, trees_per_shard=(T4 + world - 1) // world)
                        e["within_stated_bound"] = bool(np.all(np.abs(got.cpu().numpy().astype(np.float64) - exact) <= bound))
                    leg.update(e)
                except Exception as err:
                    leg["error"] = f"{type(err).__name__}: {err}"
        legs["tree_sharded_" + mode] = leg
    if world == 1:
        # accuracy of the 8-shard all-reduce, emulated on this GPU on the row sample: eight shard forests, float64 combine
        xs = x4[:sample].contiguous()
        acc = torch.zeros(sample, dtype=torch.float64, device="cuda")
        chain = torch.zeros(sample, dtype=torch.float32, device="cuda")
        for k in range(G):
            k_lo, k_hi = sharding.shard_bounds(T4, k, G)
            fk = fb if k == 0 else ta.Forest(nodes4[k_lo * per_tree: k_hi * per_tree], k_hi - k_lo, D, C, missing=MISSING)
            acc += fk.predict_raw(xs).double()
            fk.predict_accumulate(xs, chain)
            fk.check()
            if k:
                fk.close()
        e, exact, a = errors(acc.float().cpu().numpy(), xs.cpu().numpy(), ncpu)
        bound = sharding.sum_error_bound(a, exact, trees_per_shard=(T4 + G - 1) // G)
        e["within_stated_bound"] = bool(np.all(np.abs(acc.float().cpu().numpy().astype(np.float64) - exact) <= bound))
        e["emulated"] = f"{G} shard forests on this GPU, partials added in float64, rounded once"
        legs["tree_sharded_allreduce64"].update(e)
        ec, _, _ = errors(chain.cpu().numpy(), xs.cpu().numpy(), ncpu)
        ec["emulated"] = f"{G} shard forests on this GPU, running float32 sums handed from shard to shard"
        legs["tree_sharded_chain_accuracy"] = ec
        if not ec["bit_identical_to_cpu_f32"]:
            ec["error"] = "K4 chained tree shards (emulated): sums differ from the CPU oracle"
        t_rows, t_trees = legs["row_sharded"]["ms_per_step"], legs["tree_sharded_allreduce64"]["ms_per_step"]
        legs["selector"] = {
            "choose_sharding": sharding.choose_sharding(T4, D),
            "row_shard_ms_over_tree_shard_ms": round(t_rows / t_trees, 3),
            "why": "row shards are bit-exact and need no collective; the all-reduce of tree-shard totals is not within 1e-6 "
                   "relative of the CPU's sequential float32 sum (max_rel_diff_vs_cpu_f32), the bit-exact chain pays (N - 1) chunk "
                   "times of pipeline fill; the price of rows is row_shard_ms_over_tree_shard_ms in kernel time per rank (one rank's R/8 "
                   "rows of the whole forest: four tree groups, each with its own quantise pass -- against all R rows of a 1/8 forest, "
                   "all-reduce not counted)"}
    fb.close()
    return legs


def _oracle():
    """The CPU restatement of the reference's predictor (oracle/, test infrastructure): bench.py uses it as the
    cpu_baseline leg and as the checker of GPU results, after the timed regions -- never as the thing measured."""
    from oracle import oracle

    return oracle


if __name__ == "__main__":
    main()
