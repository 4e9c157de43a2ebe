"""CPU-side checks of the C-ABI library and the host logic (no GPU compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from onmf_ontf_ndl_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "onmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(onmf_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `make -C onmf_ontf_ndl_amd/csrc` (or __graft_entry__.build())"
    lib = C.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), "include/onmf_hip.h declares %s but the library does not export it" % name
    # the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_matches_header():
    lib = _lib.load()
    text = open(os.path.join(ROOT, "include", "onmf_hip.h")).read()
    want = int(re.search(r"#define ONMF_ABI_VERSION (\d+)", text).group(1))
    assert lib.onmf_abi_version() == want


@pytest.mark.parametrize("n", [1, 2, 5, 16, 25, 64, 100, 256])
def test_host_lambda_max_matches_lapack(n):
    lib = _lib.load()
    rs = np.random.RandomState(n)
    for maker in (lambda: rs.rand(n, 3 * n + 1), lambda: rs.rand(n, 3 * n + 1) - 0.5,
                  lambda: np.outer(rs.rand(n), rs.rand(3 * n))):       # rank one
        M = maker().astype(np.float32)
        S = np.ascontiguousarray((M.astype(np.float64) @ M.T.astype(np.float64)).astype(np.float32))
        got = lib.onmf_host_lambda_max(S.ctypes.data_as(C.c_void_p), n)
        want = np.linalg.eigvalsh(S.astype(np.float64))[-1]
        assert abs(got - want) <= 1e-12 * abs(want)
    Z = np.zeros((n, n), dtype=np.float32)
    assert lib.onmf_host_lambda_max(Z.ctypes.data_as(C.c_void_p), n) == 0.0


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from onmf_ontf_ndl_amd.engine import get_engine
    with pytest.raises(_lib.OnmfLibraryError):
        get_engine()
    from onmf_ontf_ndl_amd import Online_NMF
    nmf = Online_NMF(np.zeros((4, 8)), n_components=2, iterations=3, batch_size=8)
    assert nmf.code.shape == (2, 8)                       # construction needs no GPU (onmf.py:49)
    with pytest.raises(_lib.OnmfLibraryError):
        nmf.train_dict()


def test_weight_matches_reference_expression():
    from onmf_ontf_ndl_amd.onmf import _weight
    assert _weight(4, None) == 0.25
    assert _weight(np.int64(4), 0.5) == 0.5
    assert _weight(1.0, 2) == 1.0
    assert np.isinf(_weight(0, None))                     # t = 0 divides by zero like t ** (-1)


def test_legacy_facade_signature():
    from onmf_ontf_ndl_amd.legacy.onmf import Online_NMF
    X = np.zeros((4, 8))
    nmf = Online_NMF(X, n_components=2, iterations=2, batch_size=8, ini_dict=None, ini_A=None,
                     ini_B=None, history=0, alpha=None)   # image_reconstruction.py:289-297
    assert nmf.initial_agg is None
    with pytest.raises(TypeError):
        Online_NMF(X, n_components=2, ini_dict=np.zeros((4, 2)), ini_A=np.zeros((2, 2)))
