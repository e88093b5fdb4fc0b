"""Host-side mirror of the reference interface: parameter objects, recall, error behaviour.  No GPU needed."""
import os

import numpy as np
import pytest

import scalablevectorsearch_amd as svs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_search_buffer_config_mirror():
    """tests/svs/index/vamana/search_buffer.cpp:314-331."""
    c = svs.SearchBufferConfig()
    assert (c.get_search_window_size(), c.get_total_capacity()) == (0, 0)
    c = svs.SearchBufferConfig(10)
    assert (c.get_search_window_size(), c.get_total_capacity()) == (10, 10)
    c = svs.SearchBufferConfig(10, 20)
    assert (c.get_search_window_size(), c.get_total_capacity()) == (10, 20)
    with pytest.raises(svs.SvsxError):
        svs.SearchBufferConfig(10, 9)
    assert svs.SearchBufferConfig(3, 4) == svs.SearchBufferConfig(3, 4)


def test_search_parameters_defaults():
    """index/vamana/search_params.h:28-61 defaults."""
    p = svs.VamanaSearchParameters()
    assert p.search_buffer_visited_set is False and p.prefetch_lookahead == 4 and p.prefetch_step == 1
    c = p._c()
    assert (c.search_window_size, c.search_buffer_capacity, c.visited_set) == (0, 0, 0)


def test_k_recall_at_n_mirror(oracle):
    """tests/svs/core/recall.cpp:44-86 incl. the argument errors and the n-recall@n default."""
    gt = np.array([[1, 2, 3, 4]])
    res = np.array([[1, 0, 5, 6, 7, 2, 3, 4]])
    want = {(1, 1): 1.0, (2, 2): 0.5, (3, 3): 1 / 3, (4, 4): 0.25, (4, 5): 0.25, (4, 6): 0.5, (4, 7): 0.75, (4, 8): 1.0}
    for (k, n), v in want.items():
        assert svs.k_recall_at_n(gt, res, k, n) == pytest.approx(v)
    res2 = np.array([[0, 2]])
    assert svs.k_recall_at_n(gt, res2, 1, 2) == 0
    assert svs.k_recall_at_n(gt, res2, 2, 2) == 0.5
    assert svs.k_recall_at_n(gt, res2) == 0.5
    with pytest.raises(svs.SvsxError):
        svs.k_recall_at_n(gt, res2, 2, 1)
    with pytest.raises(svs.SvsxError):
        svs.k_recall_at_n(gt, res2, 2, 5)
    with pytest.raises(svs.SvsxError):
        svs.k_recall_at_n(gt, np.zeros((1, 10), int))
    with pytest.raises(svs.SvsxError):
        svs.k_recall_at_n(gt, np.zeros((1, 10), int), 5, 5)
    # bulk: mean of per-row recalls, and agreement with the oracle's restatement
    rng = np.random.default_rng(0)
    g = rng.integers(0, 100, (16, 4))
    r = rng.integers(0, 100, (16, 8))
    g = np.array([rng.permutation(100)[:4] for _ in range(16)])
    r = np.array([rng.permutation(100)[:8] for _ in range(16)])
    rows = np.mean([svs.k_recall_at_n(g[i : i + 1], r[i : i + 1], 3, 5) for i in range(16)])
    assert svs.k_recall_at_n(g, r, 3, 5) == pytest.approx(rows, rel=1e-5)
    assert svs.k_recall_at_n(g, r, 3, 5) == pytest.approx(oracle.recall(g, r, 3, 5))


def test_distance_functor_surface():
    """concepts/distance.h / core/distance/*.h: compare direction, names, enum <-> functor dispatch."""
    assert svs.DistanceL2().greater is False and svs.DistanceIP().greater and svs.DistanceCosineSimilarity().greater
    assert svs.DistanceL2.name == "squared_l2" and svs.DistanceIP.name == "inner_product"
    from scalablevectorsearch_amd.distance import as_functor

    assert isinstance(as_functor(svs.DistanceType.MIP), svs.DistanceIP)
    assert isinstance(as_functor(svs.DistanceCosineSimilarity), svs.DistanceCosineSimilarity)
    with pytest.raises(svs.SvsxError):
        svs.compute(svs.DistanceL2(), np.zeros((1, 4), np.float64), np.zeros((1, 4), np.float32))
    with pytest.raises(svs.SvsxError):
        svs.compute(svs.DistanceL2(), np.zeros((1, 4), np.float32), np.zeros((1, 5), np.float32))


def test_vecs_and_native_readers(tmp_path):
    """core/io/vecs.h + core/io/native.h v1: the reference's own tiny fixtures data/test_dataset/known_f32.{fvecs,svs}
    (copied as data into tests/golden/) hold the same matrix; round-trip through our writers."""
    import os

    from scalablevectorsearch_amd import io

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    a = io.read_fvecs(os.path.join(gold, "known_f32.fvecs"))
    b = io.read_svs_native(os.path.join(gold, "known_f32.svs"), np.float32)
    assert a.shape == b.shape and a.shape[0] > 0 and (a == b).all()
    io.write_fvecs(tmp_path / "x.fvecs", a)
    io.write_svs_native(tmp_path / "x.svs", a)
    assert (io.read_fvecs(tmp_path / "x.fvecs") == a).all() and (io.read_svs_native(tmp_path / "x.svs", np.float32) == a).all()
    assert open(tmp_path / "x.fvecs", "rb").read() == open(os.path.join(gold, "known_f32.fvecs"), "rb").read()
    g = np.arange(12, dtype=np.uint32).reshape(3, 4)
    io.write_svs_native(tmp_path / "g.svs", g)
    assert (io.read_svs_native(tmp_path / "g.svs", np.uint32) == g).all()
    with pytest.raises(ValueError):
        io.read_svs_native(os.path.join(gold, "known_f32.fvecs"), np.float32)


def test_calibration_binary_search_logic():
    """index/vamana/calibrate.h:249-343 with a synthetic monotone recall model (no GPU)."""
    from scalablevectorsearch_amd.calibrate import CalibrationParameters, optimize_search_buffer

    calls = []

    def recall_model(p):  # recall grows with the window; capacity helps a little
        w, c = p.buffer_config.search_window_size, p.buffer_config.total_capacity
        calls.append((w, c))
        return min(1.0, 0.5 + 0.02 * w + 0.001 * (c - w))

    best, ok = optimize_search_buffer(CalibrationParameters(), svs.VamanaSearchParameters(), 10, 0.9, recall_model)
    assert ok and best.buffer_config == svs.SearchBufferConfig(20, 20)  # 0.5 + 0.02*20 = 0.9
    assert len(calls) <= 12  # binary search over [10, 1000)
    # unreachable target: not converged, returns the largest window tried by lower_bound
    best, ok = optimize_search_buffer(CalibrationParameters(search_window_size_upper=30), svs.VamanaSearchParameters(), 10, 1.5, recall_model)
    assert not ok and best.buffer_config.search_window_size == 29
    # window == k already enough ("maybe oversized") -> split buffer: window below k, capacity k
    best, ok = optimize_search_buffer(CalibrationParameters(), svs.VamanaSearchParameters(), 10, 0.6, recall_model)
    assert ok and best.buffer_config.total_capacity == 10 and best.buffer_config.search_window_size == 5  # 0.5+0.1+0.005>=0.6
    # ROIOnly never splits
    best, ok = optimize_search_buffer(CalibrationParameters(search_buffer_optimization="ROIOnly"), svs.VamanaSearchParameters(), 10, 0.6, recall_model)
    assert best.buffer_config == svs.SearchBufferConfig(10, 10)


def test_tool_scripts_compile():
    """every helper script that a GPU session will run (tools/) at least parses"""
    import ast
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "tools", f) for f in sorted(os.listdir(os.path.join(root, "tools"))) if f.endswith(".py")]
    assert len(files) >= 10
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_filter_sampling_restatement_equals_the_standard_library(tmp_path):
    """sample_filter_hits (svs_runtime_utils.h:445-471) draws ids with std::mt19937(42) through std::uniform_int_distribution<size_t>;
    libsvsx uses those classes, the Python mirror restates them (iterator.py): same draws for every range, and the (checked, hits) pair and
    the first batch size that follow from them"""
    import os
    import subprocess

    from scalablevectorsearch_amd.iterator import (_MT19937, _uniform_below, predict_further_processing, sample_filter_hits,
                                                   sample_size_for_filter_stop)

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "filter_sample")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tests", "cpp", "filter_sample.cpp"), "-o", exe])
    for n in (1, 2, 3, 7, 200, 1500, 65536, 1_000_000, 12_500_000, (1 << 31) - 1, 1 << 32):
        want = [int(v) for v in subprocess.run([exe, str(n), "800"], capture_output=True, text=True, check=True).stdout.split()]
        rng = _MT19937(42)
        assert [_uniform_below(rng, n) for _ in range(800)] == want, n
    n = 1500
    member = np.zeros(n, bool)
    member[::7] = True
    draws = [int(v) for v in subprocess.run([exe, str(n), "200"], capture_output=True, text=True, check=True).stdout.split()]
    assert sample_filter_hits(member, n) == (200, int(member[draws].sum()))
    assert sample_size_for_filter_stop(0.0) == 200 and sample_size_for_filter_stop(0.5) == 200
    assert sample_size_for_filter_stop(0.001) == 999                       # static_cast<size_t>(1.0f / 0.001f): float32 arithmetic, as there
    assert sample_filter_hits(np.ones(50, bool), 50, 200) == (50, 50)     # fewer ids than samples
    assert predict_further_processing(200, 29, 10, 20, n) == 20            # enough hits already: the hint
    assert predict_further_processing(200, 4, 10, 20, n) == 300            # (10 - 4) * 200 / 4


LEGACY_CONFIG = """
__version__ = 'v0.0.2'

[object]
__version__ = 'v0.0.2'
name = 'vamana config parameters'
alpha = 0.949999988079071
max_out_degree = 64   # no prune_to key: defaults to the degree (index.h:128-132)
construction_window_size = 128
max_candidates = 750
entry_point = 77
default_search_window_size = 33
visited_set = true
"""


def test_saved_index_config_readers(tmp_path):
    """svs_config.toml (index/vamana/index.h:88-97,141-170): entry point, build parameters and the default search parameters the reference
    restores on load -- the reference's own data/test_dataset/vamana_config.toml (tests/golden/), a search-parameter variant, and the flat legacy
    layout; the Python reader and the C++ header (include/svsx_io.hpp, compiled here) must agree."""
    import subprocess

    from scalablevectorsearch_amd.io import read_index_config

    golden = os.path.join(ROOT, "tests", "golden", "vamana_config.toml")
    cfg = read_index_config(golden)
    assert cfg["entry_point"] == 9426
    assert cfg["build"] == dict(alpha=1.2000000476837158, graph_max_degree=128, window_size=200, max_candidate_pool_size=1000, prune_to=128, use_full_search_history=True)
    assert cfg["search"] == dict(search_window_size=0, search_buffer_capacity=0, search_buffer_visited_set=False, prefetch_lookahead=0, prefetch_step=0)
    tuned = tmp_path / "svs_config.toml"
    tuned.write_text(open(golden).read().replace("search_buffer_capacity = 0", "search_buffer_capacity = 48").replace("search_window_size = 0", "search_window_size = 24")
                     .replace("search_buffer_visited_set = false", "search_buffer_visited_set = true"))
    assert read_index_config(str(tmp_path))["search"]["search_window_size"] == 24  # a directory: svs_config.toml inside it
    legacy = tmp_path / "legacy.toml"
    legacy.write_text(LEGACY_CONFIG)
    lc = read_index_config(str(legacy))
    assert lc["entry_point"] == 77 and lc["build"]["prune_to"] == 64 and lc["build"]["use_full_search_history"] is True
    assert lc["search"] == dict(search_window_size=33, search_buffer_capacity=33, search_buffer_visited_set=True, prefetch_lookahead=4, prefetch_step=1)
    bad = tmp_path / "bad.toml"
    bad.write_text(open(golden).read().replace("search_window_size = 0", "search_window_size = 9"))
    with pytest.raises(ValueError):
        read_index_config(str(bad))
    # the C++ reader prints the same facts
    src = tmp_path / "cfg.cpp"
    src.write_text('#include "svsx_io.hpp"\n#include <cstdio>\nint main(int c, char** v) { try { auto g = svsx::io::read_index_config(v[1]);\n'
                   'std::printf("%u %d %d %.10g %zu %zu %zu %zu %d %zu %zu %d %zu %zu\\n", g.entry_point, g.has_build, g.has_search, g.alpha, g.graph_max_degree, g.window_size,'
                   ' g.max_candidate_pool_size, g.prune_to, g.use_full_search_history, g.search_window_size, g.search_buffer_capacity, g.search_buffer_visited_set,'
                   ' g.prefetch_lookahead, g.prefetch_step); } catch (const std::exception& e) { std::printf("error %s\\n", e.what()); return 1; } return 0; }\n')
    exe = str(tmp_path / "cfg")
    subprocess.check_call(["g++", "-std=c++20", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    run = lambda p: subprocess.run([exe, str(p)], capture_output=True, text=True)
    assert run(golden).stdout.split() == ["9426", "1", "1", "1.200000048", "128", "200", "1000", "128", "1", "0", "0", "0", "0", "0"]
    assert run(tuned).stdout.split()[9:12] == ["24", "48", "1"]
    assert run(legacy).stdout.split() == ["77", "1", "1", "0.9499999881", "64", "128", "750", "64", "1", "33", "33", "1", "4", "1"]
    assert run(bad).returncode == 1 and "search window larger" in run(bad).stdout
