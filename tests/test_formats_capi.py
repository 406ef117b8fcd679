"""Host-side pieces of the product library, CPU only: the C ABI loads and exports every symbol the header
declares, the text-format loaders agree bit for bit with the oracle's restatement of the reference loaders
(BaseTahoeTest.h:267-402), the writers round-trip, and compute entry points fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ta(built):
    import tahoe_amd

    return tahoe_amd


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tahoe_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tahoe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ta):
    lib = ctypes.CDLL(ta.capi.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/tahoe_amd.h but not exported: {missing}"
    # and the ctypes binding covers the header exactly
    assert sorted(ta.capi.EXPORTED_SYMBOLS) == syms
    assert ta.lib.tahoe_abi_version() == 2


def test_library_exports_nothing_but_the_abi(ta):
    """The boundary promises plain C entry points and nothing else (include/tahoe_amd.h:1-32): no C++ internals, no
    kernel stubs, no unprefixed helpers -- `nm -D --defined-only` lists exactly the header's declarations."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", ta.capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    # (the version node of the linker script shows as an absolute symbol: `A TAHOE_AMD_2`)
    exported = sorted(n.split("@")[0] for n in names if not n.startswith("TAHOE_AMD_"))
    assert exported == header_symbols(), sorted(set(exported) ^ set(header_symbols()))


def test_kernel_form_names_cover_the_enum(ta):
    text = open(os.path.join(ROOT, "include", "tahoe_amd.h")).read()
    forms = dict((int(v), k) for k, v in re.findall(r"TAHOE_FORM_([A-Z0-9_]+) = (\d+)", text))
    assert sorted(forms) == list(range(len(forms)))
    for value, name in forms.items():
        assert ta.lib.tahoe_kernel_form_name(value).decode() == name.lower()
    assert ta.lib.tahoe_kernel_form_name(len(forms)).decode() == "?" and ta.lib.tahoe_kernel_form_name(-1).decode() == "?"
    assert ta.lib.tahoe_forest_get_kernel_form(None, 1) == -1


def test_node_encoding_matches_reference_masks(ta):
    # encode_node / dense_node_decode, Struct.h:103-117
    n = np.zeros(1, dtype=ta.NODE_DTYPE)
    ta.lib.tahoe_encode_node(n.ctypes.data, 12345, 0.5, 1, 0.25, 1)
    assert n["bits"][0] == np.int32(np.uint32(12345 | (1 << 30) | (1 << 31)))
    assert n["val"][0] == np.float32(0.5) and n["weight"][0] == np.float32(0.25)
    ta.lib.tahoe_encode_node(n.ctypes.data, (1 << 30) + 7, 1.0, 0, 0.0, 0)  # fid is masked to 30 bits
    assert n["bits"][0] == 7
    o = np.zeros(1, dtype=oracle.NODE_DTYPE)
    oracle.lib.oracle_encode_node(o.ctypes.data, (1 << 30) + 7, 1.0, 0, 0.0, 0)
    assert o.tobytes() == n.tobytes()
    assert ta.capi.tree_num_nodes(12) == 8191 == oracle.tree_num_nodes(12)


def both_loaders_agree(ta, model, data):
    n1, t1, d1 = ta.load_model(model)
    n2, t2, d2 = oracle.load_model(model)
    assert (t1, d1) == (t2, d2) and n1.tobytes() == n2.tobytes()
    x1, m1 = ta.load_data(data)
    x2, m2 = oracle.load_data(data)
    assert x1.shape == x2.shape and x1.tobytes() == x2.tobytes()
    assert np.float32(m1).tobytes() == np.float32(m2).tobytes()
    return n1, t1, d1, x1, m1


def test_loaders_on_golden_files(ta):
    import glob

    for model in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.model.txt"))):
        stem = model[: -len(".model.txt")]
        nodes, T, D, data, missing = both_loaders_agree(ta, model, stem + ".data.txt")
        exp = np.load(stem + ".expected.npz")
        assert np.array_equal(nodes.view(np.uint32).reshape(-1, 3), exp["nodes_bits"])
        assert np.array_equal(data.view(np.uint32), exp["data_bits"])


def test_writer_round_trip_is_bit_exact(ta, tmp_path):
    T, D, C, R = 7, 4, 11, 33
    nodes = ta.synth_forest(T, D, C, seed=9, leaf_prob=0.2)
    data = ta.synth_data(R, C, seed=10, missing_prob=0.1, missing=-999.0, nan_prob=0.1)
    data[0, 0], data[0, 1], data[0, 2] = np.float32(1e-45), np.float32(3.4028235e38), -0.0  # denormal, max, -0
    m, d = str(tmp_path / "m.txt"), str(tmp_path / "d.txt")
    ta.write_model(m, nodes, T, D)
    ta.write_data(d, data, -999.0)
    n2, T2, D2, x2, miss = both_loaders_agree(ta, m, d)
    assert (T2, D2) == (T, D) and n2.tobytes() == nodes.tobytes()
    assert x2.tobytes() == data.tobytes() and miss == -999.0


def test_loader_edge_cases_follow_the_reference(ta, tmp_path):
    # (1) a file that ends early repeats its last line (unchecked fgets, BaseTahoeTest.h:298-307, :384)
    m = tmp_path / "short_model.txt"
    m.write_text("1\n2\n3\n0.5\n1\n0.25\n0\n7\n")  # root complete, node 1 starts with fid=7 and then EOF
    d = tmp_path / "short_data.txt"
    d.write_text("2\n3\n-999\n0.5\n1.5\n")  # 2x3 values wanted, 2 present
    nodes, T, D, data, missing = both_loaders_agree(ta, str(m), str(d))
    assert (T, D) == (1, 1) and nodes.size == 3
    fid1 = int(nodes["bits"][1]) & ((1 << 30) - 1)
    assert fid1 == 7 and nodes["val"][1] == np.float32(7.0) and (int(nodes["bits"][1]) & 0xFFFFFFFF) >> 31 == 1
    assert data.tolist() == [[0.5, 1.5, 1.5], [1.5, 1.5, 1.5]]
    # (2) atoi/atof semantics: leading spaces, trailing junk, exponents, CRLF
    m2 = tmp_path / "junk_model.txt"
    m2.write_text("1 trees\r\n1 levels\r\n  3xyz\r\n1e-1\r\n 2 \r\n0.5f\r\n1\r\n")
    d2 = tmp_path / "junk_data.txt"
    d2.write_text(" 1\r\n2 cols\r\nnan\r\n0x10\r\n-inf\r\n")
    nodes, T, D, data, missing = both_loaders_agree(ta, str(m2), str(d2))
    assert (T, D) == (1, 0) and (int(nodes["bits"][0]) & ((1 << 30) - 1)) == 3
    assert nodes["val"][0] == np.float32(0.1) and np.isnan(missing)
    assert data[0, 0] == 16.0 and data[0, 1] == -np.inf  # strtod parses hex floats, as atof does
    # (3) unreadable file -> TAHOE_ERR_IO (the reference perror()s and exit(1)s)
    with pytest.raises(ta.TahoeError) as e:
        ta.load_model(str(tmp_path / "nope.txt"))
    assert e.value.status == 2 and "fail to read" in str(e.value)
    # (4) a missing header line keeps the caller's default (the reference keeps its ctor defaults)
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    x, miss = ta.load_data(str(empty), num_rows=2, num_cols=1, missing=0.5)
    assert x.shape == (2, 1) and miss == 0.5
    x2, miss2 = oracle.load_data(str(empty), num_rows=2, num_cols=1, missing=0.5)
    assert x2.shape == (2, 1) and miss2 == 0.5


def test_parallel_loader_equals_the_sequential_reader(ta, tmp_path, monkeypatch):
    """Files big enough to be cut into several byte ranges (one per loader thread): the multi-threaded reader, the
    single-threaded one (TAHOE_LOADER_THREADS=1) and the oracle's fgets loop give the same bytes -- on a complete
    file, on one truncated in the middle of a node (the tail repeats the last line), with blank lines, without a
    final newline, and with a line longer than one fgets unit (falls back to the sequential reader)."""
    T, D, C, R = 300, 7, 40, 6000
    nodes = ta.synth_forest(T, D, C, seed=19, leaf_prob=0.1)
    data = ta.synth_data(R, C, seed=20, missing_prob=0.05, missing=-999.0)
    m, d = tmp_path / "m.txt", tmp_path / "d.txt"
    ta.write_model(str(m), nodes, T, D)
    ta.write_data(str(d), data, -999.0)
    mtext, dtext = m.read_text(), d.read_text()
    assert len(mtext) > 1_000_000 and len(dtext) > 1_000_000
    mlines, dlines = mtext.split("\n"), dtext.split("\n")
    variants = {
        "complete": (mtext, dtext),
        "truncated": ("\n".join(mlines[: len(mlines) * 2 // 3 + 2]) + "\n", "\n".join(dlines[: len(dlines) // 2]) + "\n"),
        "no_final_newline": (mtext.rstrip("\n"), "\n".join(dlines[: len(dlines) // 3])),
        "blank_lines": ("\n".join(mlines[:5000]) + "\n\n\n" + "\n".join(mlines[5000:]),
                        "\n".join(dlines[:7000]) + "\n\n" + "\n".join(dlines[7000:])),
        "long_line": ("\n".join(mlines[:9000]) + "\n" + "1" * 3000 + "\n" + "\n".join(mlines[9000:]),
                      "\n".join(dlines[:9000]) + "\n" + " " * 1023 + "0.5\n" + "\n".join(dlines[9000:])),
    }
    for name, (mt, dt) in variants.items():
        mp, dp = tmp_path / f"{name}.m.txt", tmp_path / f"{name}.d.txt"
        mp.write_text(mt)
        dp.write_text(dt)
        monkeypatch.setenv("TAHOE_LOADER_THREADS", "8")
        n8, T8, D8, x8, miss8 = both_loaders_agree(ta, str(mp), str(dp))
        monkeypatch.setenv("TAHOE_LOADER_THREADS", "3")
        n3, _, _ = ta.load_model(str(mp))
        x3, _ = ta.load_data(str(dp))
        monkeypatch.setenv("TAHOE_LOADER_THREADS", "1")
        n1, T1, D1 = ta.load_model(str(mp))
        x1, miss1 = ta.load_data(str(dp))
        assert (T8, D8) == (T1, D1) == (T, D), name
        assert n8.tobytes() == n1.tobytes() == n3.tobytes(), name
        assert x8.tobytes() == x1.tobytes() == x3.tobytes(), name
        if name == "complete":
            assert n8.tobytes() == nodes.tobytes() and x8.tobytes() == data.tobytes()


def test_binary_files_and_text_cache(ta, tmp_path):
    T, D, C, R = 9, 5, 13, 57
    nodes = ta.synth_forest(T, D, C, seed=29, leaf_prob=0.2)
    data = ta.synth_data(R, C, seed=30, missing_prob=0.1, missing=-999.0, nan_prob=0.1)
    mb, db = str(tmp_path / "m.tbin"), str(tmp_path / "d.tbin")
    ta.save_model_bin(mb, nodes, T, D)
    ta.save_data_bin(db, data, -999.0)
    n2, T2, D2 = ta.load_model_bin(mb)
    x2, miss = ta.load_data_bin(db)
    assert (T2, D2, miss) == (T, D, -999.0) and n2.tobytes() == nodes.tobytes() and x2.tobytes() == data.tobytes()
    # wrong kind, foreign file, truncation, one flipped payload bit -> TAHOE_ERR_IO
    bad = tmp_path / "bad.tbin"
    raw = open(db, "rb").read()
    cases = {"kind": None, "foreign": b"12\n3\n", "short": raw[:-5], "long": raw + b"x",
             "bitflip": raw[:200] + bytes([raw[200] ^ 4]) + raw[201:]}
    for name, content in cases.items():
        with pytest.raises(ta.TahoeError) as e:
            if content is None:
                ta.load_model_bin(db)
            else:
                bad.write_bytes(content)
                ta.load_data_bin(str(bad))
        assert e.value.status == 2, name
    # empty payloads
    ta.save_data_bin(db, np.empty((0, 4), dtype=np.float32), 1.5)
    x0, m0 = ta.load_data_bin(db)
    assert x0.shape == (0, 4) and m0 == 1.5
    # text + cache: first load parses and writes "<path>.tbin", second reads it, a changed text file invalidates it
    m, d = str(tmp_path / "m.txt"), str(tmp_path / "d.txt")
    ta.write_model(m, nodes, T, D)
    ta.write_data(d, data, -999.0)
    for _ in range(2):
        n3, T3, D3 = ta.load_model(m, cached=True)
        x3, miss3 = ta.load_data(d, cached=True)
        assert (T3, D3, miss3) == (T, D, -999.0) and n3.tobytes() == nodes.tobytes() and x3.tobytes() == data.tobytes()
        assert os.path.exists(m + ".tbin") and os.path.exists(d + ".tbin")
    flag = ctypes.c_int(-1)
    nt, dd, ptr = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_void_p()
    assert ta.lib.tahoe_load_model_cached(m.encode(), ctypes.byref(nt), ctypes.byref(dd), ctypes.byref(ptr), ctypes.byref(flag)) == 0
    ta.lib.tahoe_free_host(ptr)
    assert flag.value == 1
    data2 = data.copy()
    data2[3, 2] = 42.0
    ta.write_data(d, data2, -999.0)
    os.utime(d, ns=(1, 1))  # same size is possible; the mtime differs
    x4, _ = ta.load_data(d, cached=True)
    assert x4.tobytes() == data2.tobytes()


def test_synthetic_generators_are_deterministic_and_shardable(ta):
    a = ta.synth_data(100, 7, seed=5, missing_prob=0.1, missing=-1.0, nan_prob=0.1)
    b = np.concatenate([ta.synth_data(40, 7, seed=5, missing_prob=0.1, missing=-1.0, nan_prob=0.1),
                        ta.synth_data(60, 7, seed=5, missing_prob=0.1, missing=-1.0, nan_prob=0.1, first_row=40)])
    assert a.tobytes() == b.tobytes()
    assert (a == -1.0).any() and np.isnan(a).any()
    clean = ta.synth_data(50, 3, seed=6)
    assert clean.min() >= -1.0 and clean.max() < 1.0
    f = ta.synth_forest(3, 4, 9, seed=1, leaf_prob=0.3)
    assert f.tobytes() == ta.synth_forest(3, 4, 9, seed=1, leaf_prob=0.3).tobytes()
    bits = f["bits"].view(np.uint32).reshape(3, 31)
    assert ((bits[:, 15:] >> 31) == 1).all(), "bottom level must be leaves"
    assert ((bits & ((1 << 30) - 1)) < 9).all()


def test_no_gpu_means_loud_failure(ta):
    """The product has no CPU path: without a device, create() must fail (never silently compute)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ta.TahoeError) as e:
        ta.Forest(ta.synth_forest(2, 2, 3, seed=1), 2, 2, 3)
    assert e.value.status == 4  # TAHOE_ERR_NO_DEVICE
    n = ctypes.c_int(-1)
    assert ta.lib.tahoe_device_count(ctypes.byref(n)) == 0 and n.value == 0
    p = ctypes.c_void_p()
    assert ta.lib.tahoe_device_alloc(ctypes.byref(p), 16, 1) == 4


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "tahoe_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle" not in text.lower(), f"{os.path.join(dirpath, fn)} mentions the oracle"
    syms = os.popen(f"nm -D --undefined-only {os.path.join(pkg, 'libtahoe_amd.so')} 2>/dev/null").read()
    assert "oracle_" not in syms


def test_histogram_style_generator_properties(ta):
    """tahoe_synth_forest_hist / tahoe_synth_data_hist (the realistic-forest generator): deterministic, valid forests, at most
    max_bins distinct thresholds per feature, Zipf-skewed feature usage, early leaves, no dead branch on data drawn from the
    same feature distributions (checked with the oracle on the CPU)."""
    T, D, C, R = 60, 7, 40, 3000
    a = ta.synth_forest_hist(T, D, C, seed=5, feature_seed=9, max_bins=63, zipf_s=1.1, leaf_prob=0.03, scale_decades=4.0)
    b = ta.synth_forest_hist(T, D, C, seed=5, feature_seed=9, max_bins=63, zipf_s=1.1, leaf_prob=0.03, scale_decades=4.0)
    assert a.tobytes() == b.tobytes()
    assert a.tobytes() != ta.synth_forest_hist(T, D, C, seed=6, feature_seed=9, max_bins=63, zipf_s=1.1, leaf_prob=0.03, scale_decades=4.0).tobytes()
    x = ta.synth_data_hist(R, C, seed=3, feature_seed=9, scale_decades=4.0, missing_prob=0.01)
    assert x.tobytes() == ta.synth_data_hist(R, C, seed=3, feature_seed=9, scale_decades=4.0, missing_prob=0.01).tobytes()
    # ranks generate disjoint row ranges of the same stream
    assert ta.synth_data_hist(100, C, seed=3, feature_seed=9, scale_decades=4.0, missing_prob=0.01, first_row=500).tobytes() == x[500:600].tobytes()
    inner = a["bits"] >= 0
    fid = a["bits"] & ((1 << 30) - 1)
    per_tree = ta.capi.tree_num_nodes(D)
    assert not inner.reshape(T, per_tree)[:, per_tree // 2:].any()  # the bottom level is all leaves
    assert (fid[inner] < C).all()
    distinct = [np.unique(a["val"][inner & (fid == f)]).size for f in range(C)]
    assert 1 <= max(distinct) <= 63
    use = np.sort(np.bincount(fid[inner], minlength=C))[::-1]
    assert use[0] > 4 * np.median(use)  # skewed usage
    want, leaf = oracle.predict(a, T, D, x, -999.0, want_leaf=True)
    level = np.floor(np.log2(leaf.astype(np.float64) + 1.0))
    assert level.min() < D and level.mean() > 2.0  # early leaves exist, and trees are not trivial
    # the scales differ by decades across features
    spread = np.nanstd(np.where(x == np.float32(-999.0), np.nan, x), axis=0)
    assert spread.max() / spread.min() > 100.0
    # no dead branch: both children of well-visited internal nodes are reached (root level of every tree with an internal root)
    root_inner = inner.reshape(T, per_tree)[:, 0]
    went_right = (leaf >= 2)  # heap index 2 = the right child's subtree starts ... use the level-1 ancestor
    anc = leaf.astype(np.int64) + 1
    while (anc > 3).any():
        anc = np.where(anc > 3, anc >> 1, anc)
    share_right = (anc == 3).mean(axis=0)
    assert ((share_right[root_inner] > 0.0) & (share_right[root_inner] < 1.0)).mean() > 0.9
    with pytest.raises(ta.TahoeError):
        ta.synth_forest_hist(1, 3, 0)
