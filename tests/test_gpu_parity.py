"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.

Tolerances (fp32 device arithmetic vs the fp64 reference), stated once:
  codes H, aggregates A/B/C ........ relative Frobenius error <= 2e-5
  dictionary W ...................... relative Frobenius error <= 1e-4 (the bar of the task)
  per-sweep dist .................... relative 1e-4;  sweep counts equal
"""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu

TOL_H = 2e-5
TOL_AGG = 2e-5
TOL_W = 1e-4


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from onmf_ontf_ndl_amd.engine import get_engine
    e = get_engine()
    yield e
    e.set_option(2, 0)
    e.set_option(3, 0)


@pytest.fixture(scope="module")
def orc():
    from oracle import onmf_oracle
    return onmf_oracle


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


# ------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(16, 16, 100), (64, 8, 32), (256, 300, 1024), (25, 1000, 441),
                                   (130, 257, 67), (1, 1, 5)])
def test_matrix_products_all_layouts(eng, shape):
    """The four operand layouts of the MFMA product kernel against float64 NumPy (asymmetric data)."""
    import ctypes as C
    M, N, K = shape
    rs = np.random.RandomState(M + N + K)
    # use the public entry points that exercise each layout:
    #   aggregate: H H^T and H X^T (both operands contraction-fast), X X^T
    r, b, d = M, K, N
    H = rs.rand(r, b).astype(np.float32)
    X = (rs.rand(d, b) - 0.3).astype(np.float32)
    A0 = rs.rand(r, r).astype(np.float32)
    B0 = rs.rand(r, d).astype(np.float32)
    C0 = rs.rand(d, d).astype(np.float32)
    w = 0.3
    A1, B1, C1 = eng.aggregate(eng.dev(X), eng.dev(H), w, eng.dev(A0), eng.dev(B0), eng.dev(C0), want_c=True)
    H64, X64 = H.astype(np.float64), X.astype(np.float64)
    assert relerr(host(A1), (1 - w) * A0 + w * H64 @ H64.T) < 2e-6
    assert relerr(host(B1), (1 - w) * B0 + w * H64 @ X64.T) < 2e-6
    assert relerr(host(C1), (1 - w) * C0 + w * X64 @ X64.T) < 2e-6
    # null inputs mean zero aggregates
    A2, B2, _ = eng.aggregate(eng.dev(X), eng.dev(H), 1.0, None, None)
    assert relerr(host(A2), H64 @ H64.T) < 2e-6
    assert relerr(host(B2), H64 @ X64.T) < 2e-6


@pytest.mark.parametrize("shape", [(49, 100, 1000), (64, 16, 129), (33, 7, 4100), (147, 100, 333), (64, 64, 8192),
                                   (12, 250, 640), (200, 60, 70)])
def test_products_with_narrow_outputs_and_operands(eng, shape):
    """Outputs of at most 64 rows take the product kernel's 64-row tile, operands narrower than a tile its
    range-checked loads: W H (rows of W fast, rows of H slow), H H^T | H X^T with few atoms, against float64."""
    d, r, b = shape
    rs = np.random.RandomState(7 * d + r + b)
    W = (rs.rand(d, r) - 0.4).astype(np.float32)
    H = rs.rand(r, b).astype(np.float32)
    X = (rs.rand(d, b) - 0.2).astype(np.float32)
    R = eng.patch_product(eng.dev(W), eng.dev(H))
    assert relerr(host(R), W.astype(np.float64) @ H.astype(np.float64)) < 2e-6
    A1, B1, _ = eng.aggregate(eng.dev(X), eng.dev(H), 1.0, None, None)
    H64, X64 = H.astype(np.float64), X.astype(np.float64)
    assert relerr(host(A1), H64 @ H64.T) < 2e-6
    assert relerr(host(B1), H64 @ X64.T) < 2e-6
    assert np.array_equal(host(A1), host(A1).T)                      # exactly symmetric whatever the tile


def test_fill_uniform_statistics(eng):
    u = host(eng.rand(256, 4099, seed=7))
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 2e-3
    u2 = host(eng.rand(256, 4099, seed=7))
    np.testing.assert_array_equal(u, u2)
    assert np.abs(host(eng.rand(256, 4099, seed=8)) - u).max() > 0.5


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("joint", [False, True])
def test_unfold3(eng, orc, mode, joint):
    T = np.random.RandomState(3).rand(16, 3, 20).astype(np.float32)
    want = orc.unfold3(T, mode)
    want = want.T if joint else want
    got = host(eng.unfold3(eng.dev(T), mode, transpose=joint))
    np.testing.assert_array_equal(got, want.astype(np.float64))


def test_gather_and_code_accumulate_duplicates(eng):
    rs = np.random.RandomState(5)
    X = rs.rand(7, 20).astype(np.float32)
    idx = rs.randint(20, size=33)
    got = host(eng.gather_columns(eng.dev(X), idx))
    np.testing.assert_array_equal(got, X[:, idx].astype(np.float64))
    code = rs.rand(4, 20).astype(np.float32)
    H = rs.rand(4, 33).astype(np.float32)
    want = code.copy()
    want[:, idx] += H                                    # NumPy semantics: last duplicate wins
    out = host(eng.code_accumulate(eng.dev(code), idx, eng.dev(H)))
    np.testing.assert_allclose(out, want.astype(np.float64), rtol=1e-6)
    dense = host(eng.code_accumulate(eng.dev(code), None, eng.dev(rs.rand(4, 20).astype(np.float32))))
    assert dense.shape == (4, 20)


# ------------------------------------------------------------------------------------------
# golden vectors
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", [1, 0])
def test_g1_sparse_code(eng, kernel):
    g = load_golden("g1_code")
    eng.set_option(2, 1)            # evaluate every stopping test in full so dist can be compared
    eng.set_option(3, kernel)
    X, W = eng.dev(g["X"]), eng.dev(g["W"])
    for alpha in (0, 1):
        for key, tol in (("01", 0.01), ("0", 0.0)):
            H, sweeps, dist = eng.sparse_code(X, W, H0=eng.dev(g["H0"]), alpha=alpha, sub_iter=10,
                                              stopping_diff=tol)
            name = "a%d_t%s" % (alpha, key)
            assert sweeps == len(g["dist_" + name])
            assert relerr(host(H), g["H_" + name]) < TOL_H
            np.testing.assert_allclose(dist, g["dist_" + name], rtol=1e-4)
    # early stop (the reference stopped after fewer than 10 sweeps here)
    H, sweeps, dist = eng.sparse_code(X, W, H0=eng.dev(g["H0_early"]), alpha=0, sub_iter=10,
                                      stopping_diff=0.01)
    assert sweeps == len(g["dist_early"]) < 10
    assert relerr(host(H), g["H_early"]) < TOL_H
    # one column (the per-patch call of image_reconstruction.py:384)
    H, sweeps, _ = eng.sparse_code(eng.dev(g["X"][:, :1]), W, H0=eng.dev(g["H0"][:, :1]), alpha=1,
                                   sub_iter=10, stopping_diff=0.01)
    assert sweeps == len(g["dist_single"])
    assert relerr(host(H), g["H_single"]) < TOL_H
    eng.set_option(2, 0)
    eng.set_option(3, 0)


def test_g1_radius_branch(eng, orc):
    """r given (src/onmf.py:260-263): the reference's own output, then a larger case against the oracle with a
    radius that shortens the first row's move, one that does not, and an early stop."""
    from onmf_ontf_ndl_amd import update_code_within_radius
    g = load_golden("g1_code")
    X, W, H0 = g["X"][:, :6], g["W"], g["H0"][:, :6]
    H = update_code_within_radius(X.copy(), W, H0=H0.copy(), r=0.5, alpha=0, sub_iter=3, stopping_diff=0.0)
    assert H.dtype == np.float64 and relerr(H, g["H_radius"]) < TOL_H
    rs = np.random.RandomState(77)
    d, r, b = 90, 40, 333
    X, W, H0 = rs.rand(d, b), rs.rand(d, r), rs.rand(r, b)
    free = orc.code_sweeps(X, W, H0=H0.copy(), alpha=0.5, sub_iter=4, stopping_diff=0.0, norm="gram")
    eng.set_option(2, 1)            # every stopping test in full, so that dist can be compared
    for radius, tol in ((0.25, 0.0), (3.0, 0.0), (1e6, 0.0), (0.25, 0.05)):
        trace = []
        want = orc.code_sweeps(X, W, H0=H0.copy(), r=radius, alpha=0.5, sub_iter=4, stopping_diff=tol, norm="gram",
                               trace=trace)
        H, sweeps, dist = update_code_within_radius(X, W, H0=H0.copy(), r=radius, alpha=0.5, sub_iter=4,
                                                    stopping_diff=tol, return_info=True)
        assert sweeps == len(trace)
        assert relerr(H, want) < TOL_H
        np.testing.assert_allclose(dist, [t[1] for t in trace], rtol=1e-3)
        # (the radius binds in the first two cases - the result differs from the free solve - and not in the third)
        assert (relerr(want, free) > 1e-4) == (radius < 1e6) or tol > 0
    eng.set_option(2, 0)
    for kernel in (1,):                     # the fallback kernel for the later sweeps as well
        eng.set_option(3, kernel)
        try:
            H = update_code_within_radius(X, W, H0=H0.copy(), r=0.25, alpha=0.5, sub_iter=4, stopping_diff=0.0)
        finally:
            eng.set_option(3, 0)
        assert relerr(H, orc.code_sweeps(X, W, H0=H0.copy(), r=0.25, alpha=0.5, sub_iter=4, stopping_diff=0.0,
                                         norm="gram")) < TOL_H


def test_radius_entry_point_rejects_what_the_reference_cannot_run(eng):
    """r = 0 divides by zero in the reference as soon as W has a second atom (src/onmf.py:262); the C-ABI
    answers a radius that is not positive and finite with ONMF_ERR_ARG and a message, the engine passes that on,
    and the radius does not combine with the exact coordinate-descent coder."""
    import torch
    from onmf_ontf_ndl_amd import _lib
    X = torch.rand(20, 30, device=eng.device)
    W = torch.rand(20, 4, device=eng.device)
    for bad in (0.0, -1.0, float("inf"), float("nan")):
        with pytest.raises(_lib.OnmfLibraryError, match="radius"):
            eng.sparse_code(X, W, H0=torch.rand(4, 30, device=eng.device), radius=bad)
    with pytest.raises(ValueError):
        eng.sparse_code(X, W, radius=1.0, mode=_lib.CODER_EXACT_CD)
    H, sweeps, _ = eng.sparse_code(X, W, H0=torch.rand(4, 30, device=eng.device), radius=1.0, sub_iter=3,
                                   stopping_diff=0.0)
    assert sweeps == 3 and float(H.min()) >= 0


def test_g1_bounds_decide_like_the_full_test(eng):
    g = load_golden("g1_code")
    X, W = eng.dev(g["X"]), eng.dev(g["W"])
    for h0key, distkey in (("H0", "dist_a0_t01"), ("H0_early", "dist_early")):
        H, sweeps, _ = eng.sparse_code(X, W, H0=eng.dev(g[h0key]), alpha=0, sub_iter=10,
                                       stopping_diff=0.01)
        assert sweeps == len(g[distkey])


def test_g2_update_dict(eng):
    g = load_golden("g2_dict")
    W1 = host(eng.update_dict(eng.dev(g["W"]), eng.dev(g["A"]), eng.dev(g["B"])))
    assert relerr(W1, g["W1"]) < TOL_W
    assert np.all(W1[:, 3] == 0)
    r, d = g["B"].shape
    Wz = host(eng.update_dict(eng.dev(g["W"]), eng.zeros(r, r), eng.zeros(r, d)))
    assert relerr(Wz, g["W1_zero_agg"]) < 1e-6


def test_g3_steps_through_class_surface(eng):
    from onmf_ontf_ndl_amd import Online_NMF
    g = load_golden("g3_steps")
    d, r = g["W0"].shape
    np.random.seed(13)
    X = np.random.rand(*g["X"].shape)
    W = np.random.rand(d, r)
    np.testing.assert_array_equal(X, g["X"])
    agg = [np.zeros((r, r)), np.zeros((r, d))]
    nmf = Online_NMF(X, n_components=r)
    for s in range(3):
        H, agg, W = nmf.step(X, agg, W, np.int64(s + 1))     # H0 from NumPy's global stream
        assert isinstance(H, np.ndarray) and H.dtype == np.float64
        assert relerr(H, g["H_%d" % s]) < TOL_H
        assert relerr(agg[0], g["A_%d" % s]) < TOL_AGG
        assert relerr(agg[1], g["B_%d" % s]) < TOL_AGG
        assert relerr(W, g["W_%d" % s]) < TOL_W
        assert nmf.history == g["hist_%d" % s]
        assert nmf.last_sweeps == len(g["dist_%d" % s])


def test_g4_train_dict_and_warm_restart(eng):
    from onmf_ontf_ndl_amd import Online_NMF
    g = load_golden("g4_train_warm")
    r = g["W0"].shape[1]
    np.random.seed(14)
    X = np.random.rand(*g["X"].shape)
    n = X.shape[1]
    nmf = Online_NMF(X, n_components=r, iterations=5, batch_size=n, subsample=False)
    W, agg, code = nmf.train_dict()
    assert relerr(W, g["W"]) < TOL_W and relerr(agg[0], g["A"]) < TOL_AGG
    assert relerr(agg[1], g["B"]) < TOL_AGG and relerr(code, g["code"]) < TOL_H
    assert nmf.history == 5.0
    nmf2 = Online_NMF(X, n_components=r, iterations=5, batch_size=n, ini_dict=W, ini_agg=agg,
                      history=nmf.history, subsample=False)
    W2, agg2, code2 = nmf2.train_dict()
    assert relerr(W2, g["W_b"]) < TOL_W and relerr(agg2[0], g["A_b"]) < TOL_AGG
    assert relerr(agg2[1], g["B_b"]) < TOL_AGG and relerr(code2, g["code_b"]) < TOL_H
    assert nmf2.history == 10.0


def test_g5_subsample_duplicate_indices(eng):
    from onmf_ontf_ndl_amd import Online_NMF
    g = load_golden("g5_subsample")
    r = g["W0"].shape[1]
    np.random.seed(15)
    X = np.random.rand(*g["X"].shape)
    nmf = Online_NMF(X, n_components=r, iterations=4, batch_size=32, subsample=True)
    W, agg, code = nmf.train_dict()
    assert relerr(W, g["W"]) < TOL_W and relerr(agg[0], g["A"]) < TOL_AGG
    assert relerr(code, g["code"]) < TOL_H
    assert nmf.history == g["hist"]


def test_g6_full_code_third_aggregate(eng, orc):
    from onmf_ontf_ndl_amd import Online_NMF
    g = load_golden("g6_full_code")
    r = g["W0"].shape[1]
    np.random.seed(16)
    X = np.random.rand(*g["X"].shape)
    nmf = Online_NMF(X, n_components=r, iterations=4, batch_size=X.shape[1])
    W, agg, _ = nmf.train_dict(full_code=True)
    assert len(agg) == 3
    for got, key in zip([W, *agg], ["W", "A", "B", "C"]):
        assert relerr(got, g[key]) < (TOL_W if key == "W" else TOL_AGG)
    loss = orc.surrogate_loss(W, *agg)
    assert abs(loss - g["loss"]) < 1e-3 * max(1.0, abs(g["loss"]))
    # the device monitor on the same state (ising_reconstruction.py:133)
    from onmf_ontf_ndl_amd.onmf import surrogate_loss
    dev_loss = surrogate_loss(W, *agg)
    # the three traces are each ~|C|-sized and cancel: tolerance relative to the largest of them
    scale = max(1.0, abs(np.trace(agg[2])))
    assert abs(dev_loss - loss) < 1e-5 * scale
    assert abs(surrogate_loss(W, agg[0], agg[1]) - (loss - np.trace(agg[2]))) < 1e-5 * scale


@pytest.mark.parametrize("d,r", [(1024, 256), (441, 25), (7, 3)])
def test_surrogate_loss_matches_float64(eng, orc, d, r):
    rs = np.random.RandomState(d + r)
    W = rs.rand(d, r).astype(np.float32)
    H = rs.rand(r, 50)
    X = rs.rand(d, 50)
    A, B, Cm = (H @ H.T / 50).astype(np.float32), (H @ X.T / 50).astype(np.float32), (X @ X.T / 50).astype(np.float32)
    got = eng.surrogate_loss(eng.dev(W), eng.dev(A), eng.dev(B), eng.dev(Cm))
    ref = orc.surrogate_loss(W.astype(np.float64), A.astype(np.float64), B.astype(np.float64), Cm.astype(np.float64))
    scale = abs(np.trace(W.astype(np.float64) @ A @ W.T.astype(np.float64)))
    assert abs(got - ref) < 2e-6 * scale


def test_g7_beta_and_alpha(eng):
    from onmf_ontf_ndl_amd import Online_NMF
    g = load_golden("g7_beta_alpha")
    r = g["W0"].shape[1]
    np.random.seed(17)
    X = np.random.rand(*g["X"].shape)
    kw = dict(n_components=r, iterations=3, batch_size=X.shape[1], alpha=0.3, beta=0.5)
    nmf = Online_NMF(X, **kw)
    W, agg, _ = nmf.train_dict()
    assert relerr(W, g["W"]) < TOL_W and relerr(agg[1], g["B"]) < TOL_AGG
    nmf2 = Online_NMF(X, ini_dict=W, ini_agg=agg, history=nmf.history, **kw)
    W2, agg2, _ = nmf2.train_dict()
    assert relerr(W2, g["W_b"]) < TOL_W and relerr(agg2[0], g["A_b"]) < TOL_AGG
    assert nmf2.history == g["hist_b"]


def test_g8_stream_of_warm_restarts(eng):
    """12 driver iterations (iterations=2 each) with carried W, aggregates, history."""
    g = load_golden("g8_stream")
    r = g["W0"].shape[1]
    d = g["W0"].shape[0]
    W, A, B = eng.dev(g["W0"]), eng.zeros(r, r), eng.zeros(r, d)
    hist = 0
    sweeps = []
    for it in range(g["Xs"].shape[0]):
        t = hist + 1
        H, A, B, _, W, s, _ = eng.step(eng.dev(g["Xs"][it]), A, B, None, W, 1.0 / t,
                                       H0=eng.dev(g["H0s"][it]))
        sweeps.append(s)
        hist = t + 1
    assert sweeps == list(g["sweeps"])
    assert relerr(host(W), g["W"]) < TOL_W
    assert relerr(host(A), g["A"]) < 5e-5 and relerr(host(B), g["B"]) < 5e-5
    H, _, _ = eng.sparse_code(eng.dev(g["Xs"][-1]), W, H0=eng.dev(g["H_last0"]), alpha=0)
    Hh, Wh = host(H), host(W)
    resid = np.linalg.norm(g["Xs"][-1] - Wh @ Hh) / np.linalg.norm(g["Xs"][-1])
    assert abs(resid - g["resid"]) < 1e-5


def test_g11_renoir_stream_through_class_surface(eng):
    """BASELINE configs[0] on real data: 10 x 10 grey patches of the reference's Data/renoir/0.jpg, r = 16,
    1000 patches per driver iteration, 16 iterations with (W, aggregates, history) carried through the
    constructor as the reference's drivers do.  Everything random comes from NumPy's global stream in the
    reference's order; the step with index 2 must stop after 4 sweeps like the reference's."""
    from onmf_ontf_ndl_amd import Online_NMF
    g = load_golden("g11_renoir")
    k, r, nb = 10, 16, 1000
    np.random.seed(0)
    W = agg = None
    hist = 0
    sweeps = []
    for it in range(g["patches"].shape[0]):
        np.random.randint(0, g["image_shape"][0] - k, size=nb)      # the sampler's corner draws
        np.random.randint(0, g["image_shape"][1] - k, size=nb)
        X = g["patches"][it] / 255
        nmf = Online_NMF(X, n_components=r, iterations=2, batch_size=nb, ini_dict=W, ini_agg=agg,
                         history=hist, subsample=False)
        W, agg, _ = nmf.train_dict()
        hist = nmf.history
        sweeps.append(nmf.last_sweeps)
    assert sweeps == list(g["sweeps"]) and sweeps[2] == 4
    assert relerr(W, g["W"]) < TOL_W
    assert relerr(agg[0], g["A"]) < 5e-5 and relerr(agg[1], g["B"]) < 5e-5
    assert hist == g["hist"]
    H = Online_NMF(X, n_components=r).sparse_code(X, W)
    assert abs(np.linalg.norm(X - W @ H) / np.linalg.norm(X) - g["resid"]) < 1e-5


# ------------------------------------------------------------------------------------------
# against the oracle on seeded inputs (sizes the oracle finishes in seconds)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,r,b", [(100, 16, 1000), (441, 25, 1000), (300, 100, 517), (256, 64, 2048),
                                   (64, 256, 300), (33, 7, 1), (90, 300, 700)])
@pytest.mark.parametrize("kernel", [1, 0])
def test_step_against_oracle(eng, orc, d, r, b, kernel):
    rs = np.random.RandomState(d + r + b)
    X = rs.rand(d, b)
    if d == 441:
        X = (X < 0.1).astype(np.float64)                  # 0/1 adjacency-like patches
    W = rs.rand(d, r)
    W /= np.maximum(1, np.linalg.norm(W, axis=0))
    Hw = rs.rand(r, b)
    A = Hw @ Hw.T / 3
    B = Hw @ X.T / 3
    H0 = rs.rand(r, b)
    tr = []
    H_o, agg_o, W_o, _ = orc.minibatch_step(X, [A, B], W, 4, H0=H0, trace=tr, norm="gram")
    eng.set_option(3, kernel)
    H, A1, B1, _, W1, sweeps, _ = eng.step(eng.dev(X), eng.dev(A), eng.dev(B), None, eng.dev(W), 0.25,
                                           H0=eng.dev(H0))
    eng.set_option(3, 0)
    assert sweeps == len(tr)
    assert relerr(host(H), H_o) < TOL_H
    assert relerr(host(A1), agg_o[0]) < TOL_AGG and relerr(host(B1), agg_o[1]) < TOL_AGG
    assert relerr(host(W1), W_o) < TOL_W


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("shape", [(1500, 48), (64, 16), (1024, 256), (2500, 40), (700, 23), (300, 100), (33, 7),
                                   (4096, 64), (5000, 17), (200, 300), (130, 513)])
def test_update_dict_both_kernels(eng, orc, kernel, shape):
    """The dictionary pass with cooperating one-wave workgroups (1; d = 1500 -> 24 of them) and with the
    panel kernel (2; one workgroup per 1024 rows, which exchange the atoms' norms when there are
    several), against the oracle."""
    if kernel == 2 and shape[0] > 16384:
        with pytest.raises(Exception):
            eng.set_option(6, 2)
            try:
                eng.update_dict(eng.zeros(shape[0], shape[1]), eng.zeros(shape[1], shape[1]),
                                eng.zeros(shape[1], shape[0]))
            finally:
                eng.set_option(6, 0)
        return
    d, r = shape
    rs = np.random.RandomState(77 + d)
    W = rs.rand(d, r)
    Hw = rs.rand(r, 400)
    A = Hw @ Hw.T / 400          # aggregates are running means of H H^T and H X^T
    B = Hw @ rs.rand(d, 400).T / 400
    eng.set_option(6, kernel)
    try:
        W1 = host(eng.update_dict(eng.dev(W), eng.dev(A), eng.dev(B)))
    finally:
        eng.set_option(6, 0)
    assert relerr(W1, orc.dictionary_pass(W, A, B)) < TOL_W
    assert np.linalg.norm(W1, axis=0).max() <= 1 + 1e-5


def test_update_dict_small_aggregates_keep_norms_below_one(eng, orc):
    """Aggregates scaled so that most atoms end inside the unit ball (no projection)."""
    rs = np.random.RandomState(5)
    d, r = 700, 33
    W = rs.rand(d, r) * 0.02
    Hw = rs.rand(r, 50)
    A = Hw @ Hw.T
    B = Hw @ (0.02 * rs.rand(d, 50)).T
    for kernel in (1, 2):
        eng.set_option(6, kernel)
        try:
            W1 = host(eng.update_dict(eng.dev(W), eng.dev(A), eng.dev(B)))
        finally:
            eng.set_option(6, 0)
        assert relerr(W1, orc.dictionary_pass(W, A, B)) < TOL_W


def test_torch_in_torch_out_and_device_rng(eng):
    import torch
    from onmf_ontf_ndl_amd import Online_NMF, update_code_within_radius
    X = torch.rand(50, 300, device=eng.device)
    nmf = Online_NMF(X, n_components=8, iterations=4, batch_size=128, subsample=True, rng="device",
                     seed=3)
    W, agg, code = nmf.train_dict()
    assert isinstance(W, torch.Tensor) and W.is_cuda and W.dtype == torch.float32
    assert tuple(W.shape) == (50, 8) and tuple(agg[0].shape) == (8, 8) and tuple(agg[1].shape) == (8, 50)
    assert tuple(code.shape) == (8, 300)
    assert float(W.min()) >= 0 and float(torch.linalg.norm(W, dim=0).max()) <= 1 + 1e-5
    assert torch.isfinite(agg[0]).all() and torch.isfinite(agg[1]).all()
    H = update_code_within_radius(X, W, H0=None, r=None, alpha=1, sub_iter=10, stopping_diff=0.01)
    assert isinstance(H, torch.Tensor) and tuple(H.shape) == (8, 300) and float(H.min()) >= 0
    with pytest.raises(ValueError):
        update_code_within_radius(X, W, r=0)


def test_size_independent_properties_at_scale(eng):
    """b = 8192, d = 1024, r = 64 (the single-GPU bench shape): properties instead of an oracle run."""
    import torch
    d, r, b = 1024, 64, 8192
    g = torch.Generator(device=eng.device).manual_seed(1234)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    H, sweeps, _ = eng.sparse_code(X, W, H0=H0, alpha=0.0, sub_iter=10, stopping_diff=0.0)
    assert sweeps == 10 and float(H.min()) >= 0
    # columns are independent: coding a slice alone gives the same columns
    Hs, _, _ = eng.sparse_code(X[:, 1000:1300].contiguous(), W, H0=H0[:, 1000:1300].contiguous(),
                               alpha=0.0, sub_iter=10, stopping_diff=0.0)
    assert relerr(host(Hs), host(H[:, 1000:1300])) < 1e-5
    # each sweep is a descent step for 0.5*||X - W H||^2: the residual after 10 sweeps beats H0's
    r0 = float(torch.linalg.norm(X - W @ H0))
    r1 = float(torch.linalg.norm(X - W @ H))
    assert r1 < r0
    # aggregates are linear in w and symmetric
    A1, B1, _ = eng.aggregate(X, H, 1.0, None, None)
    A2, B2, _ = eng.aggregate(X, H, 0.25, A1, B1)
    assert relerr(host(A2), host(A1)) < 1e-6 and relerr(host(B2), host(B1)) < 1e-6
    assert float((A1 - A1.t()).abs().max()) == 0.0
    ref = (H.double() @ X.double().t())
    assert relerr(host(B1), ref.cpu().numpy()) < 1e-5
    # dictionary pass keeps the constraint set: nonnegative, column norms <= 1
    W1 = eng.update_dict(W, A1, B1)
    assert float(W1.min()) >= 0 and float(torch.linalg.norm(W1, dim=0).max()) <= 1 + 1e-5


def test_step_with_a_tall_dictionary_against_oracle(eng, orc):
    """d > 1024: the dictionary pass runs on several workgroups that exchange the atoms' norms, and in
    two parts around the code solve."""
    d, r, b = 2100, 40, 300
    rs = np.random.RandomState(d + r + b)
    X = rs.rand(d, b)
    W = rs.rand(d, r)
    W /= np.maximum(1, np.linalg.norm(W, axis=0))
    Hw = rs.rand(r, b)
    A = Hw @ Hw.T / 300
    B = Hw @ X.T / 300
    H0 = rs.rand(r, b)
    tr = []
    H_o, agg_o, W_o, _ = orc.minibatch_step(X, [A, B], W, 4, H0=H0, trace=tr, norm="gram")
    H, A1, B1, _, W1, sweeps, _ = eng.step(eng.dev(X), eng.dev(A), eng.dev(B), None, eng.dev(W), 0.25, H0=eng.dev(H0))
    assert sweeps == len(tr)
    assert relerr(host(H), H_o) < TOL_H
    assert relerr(host(A1), agg_o[0]) < TOL_AGG and relerr(host(B1), agg_o[1]) < TOL_AGG
    assert relerr(host(W1), W_o) < TOL_W


@pytest.mark.parametrize("d,r,b", [(64, 130, 33001), (32, 256, 32768), (48, 100, 70000)])
def test_last_columns_of_the_big_product_come_from_the_side_stream(eng, d, r, b):
    """When W^T X is a whole number of chip rounds, its last columns are computed by a separate launch
    on a second stream (contraction split, other summation order).  One sweep from the same start
    must give the same code as the single-launch path (second stream off) to rounding - an indexing
    slip in the column ranges would show as O(1) differences in the last columns."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(b + r)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    Hs, _, _ = eng.sparse_code(X, W, H0=H0, sub_iter=1, stopping_diff=0.0)
    eng.set_option(4, 0)                 # no second stream: one launch for all columns
    try:
        H1, _, _ = eng.sparse_code(X, W, H0=H0, sub_iter=1, stopping_diff=0.0)
    finally:
        eng.set_option(4, 1)
    assert float((Hs - H1).abs().max()) <= 1e-5 * max(1.0, float(H1.abs().max()))
    # and against float64 on the last 5000 columns
    Xh, Wh, H0h = host(X[:, -5000:]), host(W), host(H0[:, -5000:])
    from oracle import onmf_oracle as orc
    Href = orc.code_sweeps(Xh, Wh, H0=H0h, sub_iter=1, stopping_diff=0.0)
    assert relerr(host(Hs[:, -5000:]), Href) <= TOL_H


def test_sharded_step_without_a_process_group_is_the_plain_step(eng):
    """dist.sharded_step (code, partial sums, collective, recursion, dictionary pass) with no process
    group must reproduce eng.step at a fixed sweep count - only the recursion's arithmetic differs
    (separate multiply-adds instead of the product's epilogue)."""
    import torch
    from onmf_ontf_ndl_amd import dist as odist
    g = torch.Generator(device=eng.device).manual_seed(77)
    d, r, b = 96, 24, 1500
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    Hh = torch.rand(r, 300, device=eng.device, generator=g)
    A = Hh @ Hh.T
    B = Hh @ torch.rand(d, 300, device=eng.device, generator=g).T
    H1, A1, B1, _, W1, s1, _ = eng.step(X, A, B, None, W, 0.25, H0=H0, sub_iter=10, stopping_diff=0.0)
    H2, A2, B2, W2, s2 = odist.sharded_step(eng, X, A, B, W, 0.25, H0=H0, sub_iter=10, stopping_diff=0.0)
    assert s1 == s2 == 10
    assert torch.equal(H1, H2) and torch.equal(W1, W2)
    assert relerr(host(A2), host(A1)) <= 1e-6 and relerr(host(B2), host(B1)) <= 1e-6


def test_full_size_step_equals_its_parts(eng):
    """d = 1024, r = 256, b = 65536 (the headline shape): one step call - all sweeps in one launch,
    dictionary pass in two parts beside the products - gives bit for bit what the three entry points give
    one after the other, and the results have the properties the domain demands."""
    import torch
    d, r, b = 1024, 256, 65536
    g = torch.Generator(device=eng.device).manual_seed(99)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    Hw = torch.rand(r, 512, device=eng.device, generator=g)
    A = Hw @ Hw.t() / 512
    B = Hw @ torch.rand(d, 512, device=eng.device, generator=g).t() / 512
    H1, A1, B1, _, W1, sweeps, _ = eng.step(X, A, B, None, W, 0.2, H0=H0)
    H2, sweeps2, _ = eng.sparse_code(X, W, H0=H0, alpha=0.0, sub_iter=10, stopping_diff=0.01)
    A2, B2, _ = eng.aggregate(X, H2, 0.2, A, B)
    W2 = eng.update_dict(W, A, B)                      # the OLD aggregates (src/onmf.py:161)
    assert sweeps == sweeps2
    assert torch.equal(H1, H2) and torch.equal(A1, A2) and torch.equal(B1, B2) and torch.equal(W1, W2)
    assert float(H1.min()) >= 0
    assert float((A1 - A1.t()).abs().max()) == 0.0
    assert float(W1.min()) >= 0 and float(torch.linalg.norm(W1, dim=0).max()) <= 1 + 1e-5
    # columns are independent: a slice coded alone agrees to rounding (its W^T X is summed in another
    # order - a few-tile product is split over k - and ten sweeps at r = 256 amplify those last bits)
    Hs, _, _ = eng.sparse_code(X[:, 4096:4352].contiguous(), W, H0=H0[:, 4096:4352].contiguous(), alpha=0.0,
                               sub_iter=sweeps, stopping_diff=-1.0)
    assert relerr(host(Hs), host(H1[:, 4096:4352])) < 1e-4
    # each sweep descends on 0.5 * ||X - W H||^2
    assert float(torch.linalg.norm(X - W @ H1)) < float(torch.linalg.norm(X - W @ H0))


def test_sixty_step_stream_stays_within_the_dictionary_tolerance(eng, orc):
    """The bar of the task is on the LEARNED dictionary: 60 chained steps on fresh random batches
    (d = 100, r = 16, b = 1000, state carried on the GPU in fp32) against the fp64 oracle on the same
    stream - same sweep counts at every step, W within 1e-4 at the end (measured: 4e-6)."""
    d, r, b = 100, 16, 1000
    rs = np.random.RandomState(5)
    W = rs.rand(d, r)
    A, B = np.zeros((r, r)), np.zeros((r, d))
    Wd, Ad, Bd = eng.dev(W), eng.dev(A), eng.dev(B)
    for t in range(1, 61):
        X = rs.rand(d, b)
        H0 = rs.rand(r, b)
        tr = []
        _, (A, B), W, _ = orc.minibatch_step(X, [A, B], W, t, H0=H0, trace=tr, norm="gram")
        _, Ad, Bd, _, Wd, sweeps, _ = eng.step(eng.dev(X), Ad, Bd, None, Wd, 1.0 / t, H0=eng.dev(H0))
        assert sweeps == len(tr), t
    assert relerr(host(Wd), W) < TOL_W
    assert relerr(host(Ad), A) < TOL_AGG and relerr(host(Bd), B) < TOL_AGG
