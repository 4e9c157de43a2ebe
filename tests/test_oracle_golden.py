"""The CPU oracle must reproduce the vectors generated from the reference (tests/golden)."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import onmf_oracle as orc

TOL = 1e-11          # fp64 restatement vs fp64 reference: BLAS summation order only


def test_g1_code_sweeps_all_variants():
    g = load_golden("g1_code")
    for alpha in (0, 1):
        for key, tol in (("01", 0.01), ("0", 0.0)):
            tr = []
            H = orc.code_sweeps(g["X"], g["W"], H0=g["H0"].copy(), alpha=alpha, sub_iter=10,
                                stopping_diff=tol, trace=tr)
            name = "a%d_t%s" % (alpha, key)
            assert relerr(H, g["H_" + name]) < TOL
            d = np.array([x[1] for x in tr])
            assert len(d) == len(g["dist_" + name])
            np.testing.assert_allclose(d, g["dist_" + name], rtol=1e-9)


def test_g1_early_stop_and_single_column():
    g = load_golden("g1_code")
    tr = []
    H = orc.code_sweeps(g["X"], g["W"], H0=g["H0_early"].copy(), alpha=0, sub_iter=10,
                        stopping_diff=0.01, trace=tr)
    assert len(tr) == len(g["dist_early"]) < 10          # the reference stopped early here
    assert relerr(H, g["H_early"]) < TOL
    tr = []
    Hs = orc.code_sweeps(g["X"][:, :1], g["W"], H0=g["H0"][:, :1].copy(), alpha=1, sub_iter=10,
                         stopping_diff=0.01, trace=tr)
    assert len(tr) == len(g["dist_single"])
    assert relerr(Hs, g["H_single"]) < TOL


def test_g1_gram_norm_equals_svd_norm():
    g = load_golden("g1_code")
    tr = []
    orc.code_sweeps(g["X"], g["W"], H0=g["H0"].copy(), alpha=0, sub_iter=10, stopping_diff=0.01,
                    norm="gram", trace=tr)
    np.testing.assert_allclose([x[1] for x in tr], g["dist_a0_t01"], rtol=1e-7)


def test_g1_radius_branch():
    g = load_golden("g1_code")
    H = orc.code_sweeps(g["X"][:, :6], g["W"], H0=g["H0"][:, :6].copy(), r=0.5, alpha=0,
                        sub_iter=3, stopping_diff=0.0)
    assert relerr(H, g["H_radius"]) < TOL


def test_g2_dictionary_pass():
    g = load_golden("g2_dict")
    assert relerr(orc.dictionary_pass(g["W"], g["A"], g["B"]), g["W1"]) < TOL
    assert np.all(g["W1"][:, 3] == 0)                    # the clipped-to-zero atom is in the data
    r, d = g["B"].shape
    Wz = orc.dictionary_pass(g["W"], np.zeros((r, r)), np.zeros((r, d)))
    assert relerr(Wz, g["W1_zero_agg"]) < TOL


def test_g3_steps_chain():
    g = load_golden("g3_steps")
    r, d = g["W0"].shape[1], g["W0"].shape[0]
    W = g["W0"]
    agg = [np.zeros((r, r)), np.zeros((r, d))]
    for s in range(3):
        tr = []
        H, agg, W, hist = orc.minibatch_step(g["X"], agg, W, s + 1, H0=g["H0_%d" % s], trace=tr)
        assert relerr(H, g["H_%d" % s]) < TOL
        assert relerr(agg[0], g["A_%d" % s]) < TOL
        assert relerr(agg[1], g["B_%d" % s]) < TOL
        assert relerr(W, g["W_%d" % s]) < TOL
        assert hist == g["hist_%d" % s]
        assert len(tr) == len(g["dist_%d" % s])
    # the first step sees zero aggregates: the dictionary is only clipped and normalised
    W1 = g["W_0"]
    np.testing.assert_allclose(W1, g["W0"] / np.maximum(1, np.linalg.norm(g["W0"], axis=0)),
                               rtol=1e-12)


def _run_train(g, seed, cls_kwargs, full_code=False, draw_x=None):
    np.random.seed(seed)
    X = np.random.rand(*g["X"].shape)
    np.testing.assert_array_equal(X, g["X"])
    nmf = orc.OracleOnlineNMF(X, **cls_kwargs)
    return nmf, nmf.train_dict(full_code=full_code)


def test_g4_train_then_warm_restart_global_rng_order():
    g = load_golden("g4_train_warm")
    r = g["W0"].shape[1]
    n = g["X"].shape[1]
    nmf, (W, agg, code) = _run_train(g, 14, dict(n_components=r, iterations=5, batch_size=n))
    assert relerr(W, g["W"]) < TOL and relerr(agg[0], g["A"]) < TOL and relerr(agg[1], g["B"]) < TOL
    assert relerr(code, g["code"]) < TOL
    assert nmf.history == g["hist"] == 5.0
    nmf2 = orc.OracleOnlineNMF(g["X"], n_components=r, iterations=5, batch_size=n, ini_dict=W,
                               ini_agg=agg, history=nmf.history)
    W2, agg2, code2 = nmf2.train_dict()
    assert relerr(W2, g["W_b"]) < TOL and relerr(agg2[0], g["A_b"]) < TOL
    assert relerr(agg2[1], g["B_b"]) < TOL and relerr(code2, g["code_b"]) < TOL
    assert nmf2.history == g["hist_b"] == 10.0           # t skipped the value 5


def test_g5_subsample_duplicates():
    g = load_golden("g5_subsample")
    r = g["W0"].shape[1]
    nmf, (W, agg, code) = _run_train(g, 15, dict(n_components=r, iterations=4, batch_size=32,
                                                 subsample=True))
    assert relerr(W, g["W"]) < TOL and relerr(agg[0], g["A"]) < TOL
    assert relerr(code, g["code"]) < TOL
    assert nmf.history == g["hist"]


def test_g6_full_code_and_surrogate_loss():
    g = load_golden("g6_full_code")
    r = g["W0"].shape[1]
    n = g["X"].shape[1]
    nmf, (W, agg, code) = _run_train(g, 16, dict(n_components=r, iterations=4, batch_size=n),
                                     full_code=True)
    assert len(agg) == 3
    for got, key in zip([W, *agg], ["W", "A", "B", "C"]):
        assert relerr(got, g[key]) < TOL
    assert abs(orc.surrogate_loss(W, *agg) - g["loss"]) < 1e-9 * max(1.0, abs(g["loss"]))


def test_g7_beta_alpha_with_restart():
    g = load_golden("g7_beta_alpha")
    r = g["W0"].shape[1]
    n = g["X"].shape[1]
    kw = dict(n_components=r, iterations=3, batch_size=n, alpha=0.3, beta=0.5)
    nmf, (W, agg, _) = _run_train(g, 17, kw)
    assert relerr(W, g["W"]) < TOL and relerr(agg[1], g["B"]) < TOL
    nmf2 = orc.OracleOnlineNMF(g["X"], ini_dict=W, ini_agg=agg, history=nmf.history, **kw)
    W2, agg2, _ = nmf2.train_dict()
    assert relerr(W2, g["W_b"]) < TOL and relerr(agg2[0], g["A_b"]) < TOL
    assert nmf2.history == g["hist_b"]


def test_g8_stream_with_injected_draws():
    g = load_golden("g8_stream")
    r = g["W0"].shape[1]
    W, agg, hist = None, None, 0
    sweeps = []
    for it in range(g["Xs"].shape[0]):
        X = g["Xs"][it]
        if W is None:
            W = g["W0"]
            agg = [np.zeros((r, r)), np.zeros((r, X.shape[0]))]
        tr = []
        # iterations=2 -> one step at t = hist + 1 with the call's initial aggregates
        _, agg, W, hist = orc.minibatch_step(X, agg, W, hist + 1, H0=g["H0s"][it], trace=tr)
        sweeps.append(len(tr))
    assert sweeps == list(g["sweeps"])
    assert relerr(W, g["W"]) < 1e-10 and relerr(agg[0], g["A"]) < 1e-10
    assert relerr(agg[1], g["B"]) < 1e-10
    assert hist == g["hist"]
    H = orc.code_sweeps(g["Xs"][-1], W, H0=g["H_last0"], alpha=0, sub_iter=10, stopping_diff=0.01)
    assert relerr(H, g["H_last"]) < 1e-10
    resid = np.linalg.norm(g["Xs"][-1] - W @ H) / np.linalg.norm(g["Xs"][-1])
    assert abs(resid - g["resid"]) < 1e-10


def test_g11_renoir_stream_through_the_global_rng():
    """The C1 configuration on real image patches (10 x 10 grey windows of the reference's
    Data/renoir/0.jpg, stored as uint8), 16 driver iterations with carried state; the step with index 2
    stops after 4 sweeps.  Replayed through NumPy's global generator in the reference's order: two corner
    draws per iteration, the initial dictionary once, H0 per step."""
    g = load_golden("g11_renoir")
    k, r, nb = 10, 16, 1000
    np.random.seed(0)
    W = agg = None
    hist = 0
    sweeps = []
    for it in range(g["patches"].shape[0]):
        np.random.randint(0, g["image_shape"][0] - k, size=nb)      # the corner draws of the sampler
        np.random.randint(0, g["image_shape"][1] - k, size=nb)
        X = g["patches"][it] / 255
        nmf = orc.OracleOnlineNMF(X, n_components=r, iterations=2, batch_size=nb, ini_dict=W, ini_agg=agg,
                                  history=hist, subsample=False)
        W, agg, _ = nmf.train_dict()
        hist = nmf.history
        sweeps.append(len(nmf.sweep_log[-1]))
        np.testing.assert_allclose([v for _, v in nmf.sweep_log[-1]], g["dists"][it][:sweeps[-1]], rtol=1e-9)
    assert sweeps == list(g["sweeps"]) and sweeps[2] == 4
    assert relerr(W, g["W"]) < 1e-10 and relerr(agg[0], g["A"]) < 1e-10 and relerr(agg[1], g["B"]) < 1e-10
    assert hist == g["hist"]
    H = orc.OracleOnlineNMF(X, n_components=r).sparse_code(X, W)
    assert relerr(H, g["H_last"]) < 1e-10
    assert abs(np.linalg.norm(X - W @ H) / np.linalg.norm(X) - g["resid"]) < 1e-10


@pytest.mark.parametrize("name,alpha", [("default2", None), ("a05", 0.5), ("a005", 0.05)])
def test_g9_lasso_matches_sklearn_vectors(name, alpha):
    g = load_golden("g9_lasso")
    want = g["H_" + name]
    got = orc.lasso_nonneg_code(g["X"], g["W"], alpha, solver="sklearn")
    np.testing.assert_allclose(got, want, atol=1e-12)
    cd = orc.lasso_nonneg_code(g["X"], g["W"], alpha, solver="cd")
    a = 2 if alpha is None else alpha
    # Same convex problem.  scikit-learn's positive LARS path stops slightly short of the optimum
    # (objective up to 0.4 % above the coordinate-descent fixed point at small alpha), so the
    # comparison is: CD objective never worse, same support, codes within 5e-2 absolute.
    f_cd = orc.lasso_objective(g["X"], g["W"], cd.T, a)
    f_sk = orc.lasso_objective(g["X"], g["W"], want.T, a)
    assert np.all(f_cd <= f_sk * (1 + 1e-9) + 1e-12)
    assert np.all(f_sk <= f_cd * (1 + 5e-3))
    assert np.array_equal(cd > 0, want > 0)
    np.testing.assert_allclose(cd, want, atol=5e-2)


def test_unfold3_shapes_and_layout():
    T = np.arange(4 * 3 * 5, dtype=np.float64).reshape(4, 3, 5)
    assert orc.unfold3(T, 0).shape == (4, 15)
    assert orc.unfold3(T, 1).shape == (3, 20)
    assert orc.unfold3(T, 2).shape == (5, 12)
    np.testing.assert_array_equal(orc.unfold3(T, 0), T.reshape(4, 15))
    np.testing.assert_array_equal(orc.unfold3(T, 2)[1], T[:, :, 1].reshape(-1))
    np.testing.assert_array_equal(orc.unfold3(T, 1)[2], T[:, 2, :].reshape(-1))


# ----------------------------------------------------------------------------------------
# scikit-learn's positive LARS path (the coder of src/ontf.py:79-86), restated
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,alpha", [("default2", None), ("a05", 0.5), ("a005", 0.05)])
def test_g9_lars_restatement_matches_sklearn_vectors(name, alpha):
    g = load_golden("g9_lasso")
    info = []
    H = orc.lars_pos_code(g["X"], g["W"], alpha, info=info)
    assert relerr(H, g["H_" + name]) < TOL
    assert all(i["end"] == "alpha_min" for i in info)
    # ... and it is NOT the optimum of the convex problem where the path ran out of positive correlations
    if alpha == 0.05:
        cd = orc.lasso_nonneg_code(g["X"], g["W"], alpha, solver="cd")
        assert np.abs(cd - H).max() > 1e-3


def _unfold_joint(T):
    return np.reshape(np.moveaxis(T, 2, 0), (T.shape[2], -1)).T


@pytest.mark.parametrize("case,r,alpha,beta,batch,seed", [("g12_ntf_a", 8, 0.5, None, 40, 27),
                                                          ("g12_ntf_b", 12, None, 0.7, 64, 28)])
def test_g12_ntf_streams_coded_by_sklearn(case, r, alpha, beta, batch, seed):
    """OracleOnlineNTF with the restated path (solver="lars") reproduces, from NumPy's global stream alone,
    the streams that scikit-learn's coder and the reference's update_dict produced."""
    g = load_golden(case)
    steps = int(g["steps"])
    np.random.seed(seed)
    T = np.random.rand(*g["T"].shape)
    assert np.array_equal(T, g["T"])
    ntf = orc.OracleOnlineNTF(T, n_components=r, iterations=steps + 1, batch_size=batch, mode=2,
                              learn_joint_dict=True, alpha=alpha, beta=beta, subsample=True, solver="lars")
    W, A, B, _ = ntf.train_dict_single()
    assert relerr(W, g["W_%d" % steps]) < 1e-10
    assert relerr(A, g["A_%d" % steps]) < 1e-10 and relerr(B, g["B_%d" % steps]) < 1e-10
    assert ntf.history == steps + 1
    # one step alone on stored inputs, codes included
    Xu = _unfold_joint(T)
    i = steps
    H, A1, B1, W1 = ntf.step(Xu[:, g["idx_%d" % i]], g["A_%d" % (i - 1)], g["B_%d" % (i - 1)], g["W_%d" % (i - 1)],
                             np.float64(i))
    assert relerr(H, g["H_%d" % i]) < 1e-10 and relerr(W1, g["W_%d" % i]) < TOL


def test_g12_cd_optimum_is_not_the_references_dictionary():
    """What following the path buys: the same stream coded to the optimum of the convex problem ends with a
    dictionary 1e-3 away from the one the reference's coder leads to."""
    g = load_golden("g12_ntf_a")
    np.random.seed(27)
    T = np.random.rand(*g["T"].shape)
    ntf = orc.OracleOnlineNTF(T, n_components=8, iterations=7, batch_size=40, mode=2, learn_joint_dict=True,
                              alpha=0.5, subsample=True, solver="cd")
    W, _, _, _ = ntf.train_dict_single()
    assert relerr(W, g["W_6"]) > 1e-3


def test_g12_lars_restatement_at_c3_shape():
    g = load_golden("g12_ntf_c3")
    rs = np.random.RandomState(29)
    d, r, b = 1200, 100, 100
    X = rs.rand(d, b).astype(np.float32).astype(np.float64)
    W = rs.rand(d, r).astype(np.float32).astype(np.float64) / 32
    cols = slice(0, 24)                      # pure-Python path: a slice keeps the CPU suite short
    for name, alpha in (("a2", 2), ("a01", 0.1)):
        H = orc.lars_pos_code(X[:, cols], W, alpha)
        assert relerr(H, g["H_" + name][cols]) < 1e-10
