"""The C ABI library: loads, exports every symbol include/svsx.h declares, validates arguments, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from scalablevectorsearch_amd import _lib

    _lib.build()
    return _lib


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "svsx.h")).read()
    declared = set(re.findall(r"\b(svsx_[a-z0-9_]+)\s*\(", header))
    declared -= set(re.findall(r"\}\s*(svsx_[a-z0-9_]+)\s*;", header))  # struct types (one is the return type of a callback typedef)
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    L = lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.svsx_version()


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include "svsx.h"\nint main(void){ svsx_search_params p = {0}; (void)p; return SVSX_SUCCESS; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++17")):
        subprocess.check_call([cc, std, "-x", "c" if cc == "gcc" else "c++", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                               "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_null_index_reports_not_initialized_and_fills_sentinels(lib):
    """bindings/cpp/src/vamana_index_impl.h:115-123."""
    L = lib.lib()
    q = np.zeros((2, 4), np.float32)
    ids = np.zeros((2, 3), np.uint64)
    d = np.zeros((2, 3), np.float32)
    sp = lib.SearchParams(4, 4, 0, 0, 0)
    rc = L.svsx_search(None, q.ctypes.data, 0, 2, 3, C.byref(sp), ids.ctypes.data, None, d.ctypes.data, None)
    assert rc == lib.NOT_INITIALIZED
    assert b"not initialized" in L.svsx_last_error().lower()
    assert (ids == np.uint64(2**64 - 1)).all() and np.isposinf(d).all()


def test_create_validates_arguments_before_touching_the_device(lib):
    L = lib.lib()
    h = C.c_void_p()
    x = np.zeros((4, 8), np.float32)
    g = np.zeros((4, 3), np.uint32)
    e = np.array([9], np.uint32)  # out of range
    desc = lib.IndexDesc(0, 0, 4, 8, x.ctypes.data, 0, g.ctypes.data, 3, 2, e.ctypes.data, 1, 0, 1.0, 0.0, 0, 0)
    assert L.svsx_index_create(C.byref(desc), C.byref(h)) == lib.INVALID_ARGUMENT
    assert b"entry point" in L.svsx_last_error()
    desc.metric = 7
    assert L.svsx_index_create(C.byref(desc), C.byref(h)) == lib.INVALID_ARGUMENT
    desc.metric, desc.sq_enabled = 0, 1  # SQ needs int8/uint8 codes
    e[0] = 0
    assert L.svsx_index_create(C.byref(desc), C.byref(h)) == lib.INVALID_ARGUMENT
    assert L.svsx_index_create(None, C.byref(h)) == lib.INVALID_ARGUMENT


def test_no_silent_cpu_fallback(lib):
    """Without a HIP device every compute entry point must FAIL (SVSX_RUNTIME_ERROR), never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import scalablevectorsearch_amd as svs

    with pytest.raises(svs.SvsxError) as ei:
        svs.compute(svs.DistanceL2(), np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32))
    assert ei.value.code == lib.RUNTIME_ERROR
    with pytest.raises(svs.SvsxError) as ei:
        svs.VamanaIndex(np.zeros((2, 2), np.uint32), np.zeros((2, 4), np.float32), svs.DistanceL2(), [0])
    assert ei.value.code == lib.RUNTIME_ERROR


def test_round3_entry_points_check_their_arguments_before_any_device_work(lib):
    """svsx_sharded_*, svsx_vamana_build_device / svsx_find_medoid_device, svsx_search_filtered_ex: null handles and inconsistent shapes are
    refused with the runtime's error codes (api_defs.h:110-117) on a box without a GPU too"""
    L = lib.lib()
    h = C.c_void_p()
    one = (C.c_void_p * 1)(None)
    assert L.svsx_sharded_create(None, 0, None, 0, 1, C.byref(h)) == lib.INVALID_ARGUMENT             # no shards
    assert L.svsx_sharded_create(one, 1, None, 0, 1, C.byref(h)) == lib.NOT_INITIALIZED               # a null shard
    assert b"Index not initialized" in L.svsx_last_error()
    assert L.svsx_sharded_create(one, 1, None, 3, 2, C.byref(h)) == lib.INVALID_ARGUMENT              # rank outside the world
    assert L.svsx_sharded_create(one, 1, None, 0, 1, None) == lib.INVALID_ARGUMENT
    assert L.svsx_sharded_search_device(None, None, 0, 1, 1, None, None, None, None) == lib.NOT_INITIALIZED
    assert L.svsx_sharded_unique_id(None) == lib.INVALID_ARGUMENT
    L.svsx_sharded_destroy(None)                                                                      # a no-op, like free(NULL)
    bp = lib.BuildParams(1.2, 64, 128, 750, 60, 1)
    assert L.svsx_vamana_build_device(None, C.byref(bp), None) == lib.NOT_INITIALIZED
    m = C.c_uint32(0)
    assert L.svsx_find_medoid_device(None, C.byref(m)) == lib.NOT_INITIALIZED
    q = np.zeros((1, 4), np.float32)
    ids = np.zeros((1, 1), np.uint64)
    d = np.zeros((1, 1), np.float32)
    assert L.svsx_search_filtered_ex(None, q.ctypes.data, 0, 1, 1, None, None, 0.0, 1, ids.ctypes.data, d.ctypes.data) == lib.NOT_INITIALIZED
    dist = C.c_double(0.0)
    assert L.svsx_get_distance(None, 0, q.ctypes.data, 0, C.byref(dist)) == lib.NOT_INITIALIZED
    alloc = lib.RESULTS_ALLOCATOR(lambda user, counts, n, out: None)
    assert L.svsx_range_search(None, q.ctypes.data, 0, 1, 1.0, None, None, alloc, None) == lib.NOT_INITIALIZED
    assert L.svsx_index_refresh_row_norms(None, None) == lib.NOT_INITIALIZED and L.svsx_index_drop_row_norms(None) == lib.NOT_INITIALIZED
    assert L.svsx_version().startswith(b"svsx 0.5") and L.svsx_abi_version() == 3


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under scalablevectorsearch_amd/ may include, import, link or call it."""
    pkg = os.path.join(ROOT, "scalablevectorsearch_amd")
    banned = ("svs_oracle", "import oracle", "from oracle", "oracle.oracle", "oracle/", "libsvs_ref", "svsref_", "svso_")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for b in banned:
                    assert b not in text, (dirpath, f, b)


def _build_facade(tmp_path):
    import subprocess

    exe = str(tmp_path / "facade_smoke")
    pkg = os.path.join(ROOT, "scalablevectorsearch_amd")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_smoke.cpp"),
                           "-o", exe, "-L", pkg, "-lsvsx", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_facade_compiles_and_validates(lib, tmp_path):
    """include/svsx.hpp (VamanaIndex::search-shaped facade) builds against libsvsx.so; argument checks run on the host."""
    import subprocess

    exe = _build_facade(tmp_path)
    assert subprocess.call([exe, "compile-only"]) == 0


@pytest.mark.gpu
def test_cxx_facade_search_on_gpu(lib, tmp_path):
    import subprocess

    exe = _build_facade(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].split() == ["5", "6", "4"]
    assert lines[1].split() == ["filtered:", "40", "42", "38", "44"]                        # IDFilter overload, inline-record index
    assert lines[2].split()[0] == "range:" and int(lines[2].split()[1]) >= 3                # runtime range_search + get_distance through the C ABI (exit codes 15-19 check the contents)
    assert lines[3].split()[0] == "calibrated:" and lines[3].split()[-1] == "1.00"          # VamanaIndex::calibrate reaches recall 1 on the ring
    assert [ln.split()[2] for ln in lines[4:]] == ["5", "6", "4", "7", "3", "2", "1", "0"]  # BatchIterator, batches of 3


def test_cxx_io_recall_calibration_header(tmp_path):
    """include/svsx_io.hpp -- the C++ host side of SURVEY 8(f) ranks 2 and 3 (file readers, recall, search-buffer calibration) -- against the
    reference's own data files and the Python mirror (io.py, calibrate.py); no GPU involved"""
    import subprocess

    import numpy as np

    from scalablevectorsearch_amd import io

    exe = str(tmp_path / "io_smoke")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "io_smoke.cpp"), "-o", exe])
    fv, sv = os.path.join(ROOT, "tests", "golden", "known_f32.fvecs"), os.path.join(ROOT, "tests", "golden", "known_f32.svs")
    out = subprocess.run([exe, fv, sv, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict((ln.split()[0], ln.split()[1:]) for ln in out.stdout.splitlines())
    a, b = io.read_fvecs(fv), io.read_svs_native(sv, np.float32)
    assert lines["fvecs"][:2] == [str(a.shape[0]), str(a.shape[1])] and abs(float(lines["fvecs"][2]) - float(a.astype(np.float64).sum())) < 1e-6
    assert lines["svs"][:2] == [str(b.shape[0]), str(b.shape[1])] and abs(float(lines["svs"][2]) - float(b.astype(np.float64).sum())) < 1e-6
    assert lines["roundtrip"] == ["1"] and lines["ivecs"] == ["2", "3", "6"] and lines["entry"] == ["4711"]
    assert abs(float(lines["recall"][0]) - 5.0 / 6.0) < 1e-6
    # the same expectations as tests/test_host_logic.py holds for the Python optimize_search_buffer
    assert lines["cal1"] == ["20", "20", "1", "1"] and lines["cal2"] == ["29", "0"] and lines["cal3"] == ["5", "10", "1"] and lines["cal4"] == ["10", "10"]
    assert (io.read_svs_native(str(tmp_path / "rt.svs"), np.float32) == a).all()   # what the C++ writer wrote, read by the Python reader


def test_cxx_search_index_cli_builds_and_checks_its_arguments(lib, tmp_path):
    """tools/search_index.cpp (utils/search_index.cpp:89-178 over the C ABI) links against libsvsx.so; wrong argument counts and unknown
    distance / type names are refused before anything touches a GPU"""
    import subprocess

    exe = str(tmp_path / "search_index")
    pkg = os.path.join(ROOT, "scalablevectorsearch_amd")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "search_index.cpp"),
                           "-o", exe, "-L", pkg, "-lsvsx", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"])
    bad = subprocess.run([exe, "float"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Expected 11 arguments" in bad.stdout
    args = ["float", "float", "q.fvecs", "32", "10", "4", "cfg", "graph", "data", "res"]
    bad = subprocess.run([exe] + args + ["Manhattan"], capture_output=True, text=True)
    assert bad.returncode == 2 and "Unsupported distance type" in bad.stderr
    bad = subprocess.run([exe, "float16"] + args[1:] + ["L2"], capture_output=True, text=True)
    assert bad.returncode == 2 and "Unsupported Query and Data type pair" in bad.stderr
