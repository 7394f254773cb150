"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol of include/graphpope_hip.h."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "graphpope_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:pope|sage)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from graphpope_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == declared            # the ctypes table and the header agree


def test_size_queries_need_no_gpu():
    from graphpope_amd import _lib
    lib = _lib.load()
    assert lib.pope_version().startswith(b"graphpope_hip")
    assert [lib.pope_words(k) for k in (1, 64, 65, 128, 129, 130, 256, 257, 1024)] == [1, 1, 2, 2, 4, 4, 4, 8, 16]
    assert lib.pope_plane_bytes(89250, 256) == 89250 * 4 * 8
    assert lib.pope_bfs_scratch_bytes(89250, 899756, 256) >= 2 * 89250 * 4 * 8
    assert lib.pope_csr_scratch_bytes(89250, 899756) >= (89250 + 1) * 4
    assert lib.pope_last_error() == b""


def test_product_path_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from graphpope_amd import engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.require_gpu()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "graphpope_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("SURVEY", ""), f


def test_argument_validation_needs_no_gpu():
    """Bad sizes / null pointers are rejected before any HIP call, with a message behind pope_last_error()."""
    from graphpope_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.pope_csr_build(null, 5, -1, null, null, null, null, null, 0, 0, null) == _lib.ERR_INVALID
    assert b"pope_csr_build" in lib.pope_last_error()
    assert lib.pope_geodesic_finalize(null, 0, 10, 4, null, 0, null, 4, 0, null) == _lib.ERR_INVALID
    assert lib.pope_pairwise_minmax(null, 10, 4, null, 2, 7, null, 2, 0, null, 0, null) == _lib.ERR_INVALID
    assert lib.sage_conv_forward(null, null, 5, 9, 0, null, 4, null, null, null, 4, null, null, null, 0, null, null) == _lib.ERR_INVALID
    assert lib.pope_geodesic_run_workspace_bytes(10, 10, 0, 8) == 0 and lib.pope_geodesic_run_workspace_bytes(89250, 899756, 256, 8) > 0
    assert lib.sage_bn_relu_dropout_forward(null, 8, 4, null, null, null, null, null, 0.1, 1e-5, 1, 0.5, 0, null, null, null, null, 0, null, null, null) == _lib.ERR_INVALID
    assert lib.sage_bn_relu_dropout_backward(null, null, 8, 4, null, null, null, null, 1, 0.5, 0, null, null, null, null, 0, null, null, null) == _lib.ERR_INVALID
    assert lib.sage_adam_step(2, null, null, null, null, null, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, null, null) == _lib.ERR_INVALID
    assert lib.sage_adam_step(0, null, null, null, null, null, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, null, null) == _lib.ERR_INVALID      # step is 1-based
    assert lib.pope_geodesic_column_stats(null, 3, 10, 4, null, null, null, 0, null) == _lib.ERR_INVALID
    assert lib.sage_sample_batch_device(null, null, 10, null, 4, null, 2, 0, null, null, null, null, null, null, 0, null) == _lib.ERR_INVALID
    assert lib.sage_advance_counters(null, null, 3, null) == _lib.ERR_INVALID and lib.sage_copy_segments(2, null, null, null, null) == _lib.ERR_INVALID
    assert lib.pope_assemble_host_result(null, 0, 0, null, 0, 0, null, 0, 4, 1, 0, null) == _lib.ERR_INVALID
    assert lib.pope_host_pin(null, 0) == _lib.ERR_INVALID
    assert lib.sage_bn_scratch_bytes(256) > 0 and lib.pope_column_stats_scratch_bytes(256) > 0
    import pytest
    with pytest.raises(_lib.PopeError, match="null pointer"):
        _lib.check(lib.pope_concat(null, 4, 4, null, 8, null))


def test_result_tensor_pool_reuses_the_pages_of_a_freed_result(monkeypatch):
    """engine.host_result_tensor: pageable, contiguous, handed back when the LAST view dies, reused only for the same size,
    dropped for another size and when GRAPHPOPE_RESULT_POOL_MB forbids keeping it."""
    import torch
    from graphpope_amd import engine
    engine._RESULT_POOL.clear()
    t = engine.host_result_tensor(1000, 700)
    assert t.shape == (1000, 700) and t.dtype == torch.float32 and t.is_contiguous() and not t.is_pinned() and not t.is_shared()
    p = t.data_ptr()
    t.fill_(3.0)
    v = t[:, :4]
    del t
    assert not engine._RESULT_POOL                       # a view still uses the pages
    del v
    assert list(engine._RESULT_POOL) == [2800000]
    t2 = engine.host_result_tensor(1000, 700)
    assert t2.data_ptr() == p and not engine._RESULT_POOL
    t3 = engine.host_result_tensor(1000, 700)            # the pool is empty: fresh pages
    assert t3.data_ptr() != p
    del t2, t3
    assert list(engine._RESULT_POOL) == [2800000]        # one entry, the first to come back
    t4 = engine.host_result_tensor(2000, 700)            # another size evicts it
    assert not engine._RESULT_POOL
    monkeypatch.setenv("GRAPHPOPE_RESULT_POOL_MB", "1")
    del t4
    assert not engine._RESULT_POOL                       # 5.6 MB > 1 MB: unmapped, not kept
    monkeypatch.setenv("GRAPHPOPE_RESULT_POOL_MB", "0")
    t5 = engine.host_result_tensor(1000, 700)            # pool off: torch's own allocation
    del t5
    assert not engine._RESULT_POOL
    assert engine.host_result_tensor(3, 5).shape == (3, 5)


def test_kernel_name_queries_follow_the_shapes():
    """pope_level_kernel_name / pope_finalize_kernel_name are host logic (which instantiation a shape gets: bench.py and the profiles label
    their roofline entries with them): the shapes of BASELINE configs[1], [3], [4] and of their per-rank shards, no GPU needed."""
    import ctypes
    from graphpope_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(64)

    def level(n, k):
        _lib.check(lib.pope_level_kernel_name(n, k, buf, 64))
        return buf.value.decode()

    def fin(n, k, f, shards):
        _lib.check(lib.pope_finalize_kernel_name(n, k, f, 1 if f else 0, shards, buf, 64))
        return buf.value.decode()

    flickr, rmat = 89250, 1 << 22
    assert level(flickr, 256) == "k_bfs_level<4, 1, 0>" and level(flickr, 128) == "k_bfs_level<2, 1, 0>"
    assert level(flickr, 1024) == "k_bfs_level<8, 1, 2>" and level(flickr, 768) == "k_bfs_level<4, 1, 2>"
    assert level(rmat, 512) == "k_bfs_level<8, 3, 0>" and level(rmat, 64) == "k_bfs_level<1, 3, 0>"
    assert level(rmat, 1024) == "k_bfs_level<8, 3, 1>" and level(rmat, 768) == "k_bfs_level<4, 3, 1>"
    assert fin(flickr, 256, 500, 1) == "k_finalize_pipe<2, 1>" and fin(flickr, 1024, 500, 1) == "k_finalize_wide<2>"
    assert fin(rmat, 512, 0, 1) == "k_finalize_lut" and fin(flickr, 256, 500, 8) == "k_finalize_wide<2>" and fin(rmat, 64, 0, 8) == "k_finalize_lut"
    assert lib.pope_level_kernel_name(0, 256, buf, 64) != 0 and lib.pope_finalize_kernel_name(flickr, 0, 0, 0, 1, buf, 64) != 0
