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
    assert [ln.split()[2] for ln in lines[2:]] == ["5", "6", "4", "7", "3", "2", "1", "0"]  # BatchIterator, batches of 3
