"""GPU parity of the Online_NTF surface (src/ontf.py of the reference).

The reference's coder is scikit-learn LassoLars (positive).  The default GPU coder (coder="lars") follows
that routine's path event by event and is held to scikit-learn's own vectors: g9 (codes), g12 (streams of
Online_NTF steps: codes, aggregates and the learned dictionary after 6 and 7 steps; the coder at d = 1200,
r = 100).  Tolerances: codes and aggregates 1e-4 relative (measured 1e-6 ... 1e-5), dictionary 1e-4.
coder="cd" solves the convex problem itself by coordinate descent; scikit-learn's path ends short of that
optimum, so its codes are compared with the oracle's coordinate-descent solution (<= 2e-4 absolute) and with
scikit-learn's vectors at objective level (never worse than 1e-5 relative, support equal, codes within 5e-2)."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import onmf_oracle
    return onmf_oracle


@pytest.mark.parametrize("name,alpha", [("default2", None), ("a05", 0.5), ("a005", 0.05)])
def test_g9_lars_codes_match_sklearn(name, alpha):
    """The default coder against scikit-learn's own output for the call of src/ontf.py:79-86."""
    from onmf_ontf_ndl_amd import Online_NTF
    g = load_golden("g9_lasso")
    ntf = Online_NTF(np.zeros((2, 3, 4)), n_components=g["W"].shape[1], alpha=alpha)
    H = ntf.joint_sparse_code_tensor(g["X"], g["W"])       # (b x r), like the reference
    want = g["H_" + name]
    assert H.shape == want.shape and H.dtype == np.float64
    print("g9 %s: lars codes rel err %.2e, max abs %.2e" % (name, relerr(H, want), np.abs(H - want).max()))
    assert relerr(H, want) < 1e-4
    assert np.array_equal(H > 1e-9, want > 1e-9)       # (an atom that left the active set keeps a residue of 1e-17)


def _unfold_joint(T):
    return np.reshape(np.moveaxis(T, 2, 0), (T.shape[2], -1)).T


@pytest.mark.parametrize("case,r,alpha,beta,batch,seed", [("g12_ntf_a", 8, 0.5, None, 40, 27),
                                                          ("g12_ntf_b", 12, None, 0.7, 64, 28)])
def test_g12_ntf_stream_dictionary(case, r, alpha, beta, batch, seed):
    """Streams of Online_NTF steps whose codes came from scikit-learn (tests/golden/make_golden.py g12):
    step by step through Online_NTF.step on the stored minibatches, then end to end through
    train_dict_single on NumPy's global stream.  The learned dictionary must be within 1e-4."""
    from onmf_ontf_ndl_amd import Online_NTF
    g = load_golden(case)
    steps = int(g["steps"])
    T = g["T"]
    Xu = _unfold_joint(T)
    ntf = Online_NTF(T, n_components=r, mode=2, learn_joint_dict=True, alpha=alpha, beta=beta)
    W, A, B = g["W0"], np.zeros((r, r)), np.zeros((r, Xu.shape[0]))
    worst = dict(H=0.0, A=0.0, B=0.0, W=0.0)
    for i in range(1, steps + 1):
        X = Xu[:, g["idx_%d" % i]]
        H, A, B, W = ntf.step(X, A, B, W, np.int64(i))
        for key, got in (("H", H), ("A", A), ("B", B), ("W", W)):
            worst[key] = max(worst[key], relerr(got, g["%s_%d" % (key, i)]))
        # carry the reference's own state, so that every step is compared on identical inputs
        A, B, W = g["A_%d" % i], g["B_%d" % i], g["W_%d" % i]
    print("%s step by step, worst rel err: %s" % (case, worst))
    assert worst["H"] < 1e-4 and worst["A"] < 1e-4 and worst["B"] < 1e-4 and worst["W"] < 1e-4
    # end to end: the state carried on the GPU, minibatches drawn from NumPy's global stream (src/ontf.py:213, :230)
    np.random.seed(seed)
    T2 = np.random.rand(*T.shape)
    assert np.array_equal(T2, T)
    ntf = Online_NTF(T2, n_components=r, iterations=steps + 1, batch_size=batch, mode=2, learn_joint_dict=True,
                     alpha=alpha, beta=beta, subsample=True)
    W, A, B, _ = ntf.train_dict_single()
    errs = dict(W=relerr(W, g["W_%d" % steps]), A=relerr(A, g["A_%d" % steps]), B=relerr(B, g["B_%d" % steps]))
    print("%s train_dict_single after %d steps, rel err vs the scikit-learn-coded stream: %s" % (case, steps, errs))
    assert errs["W"] < 1e-4 and errs["A"] < 1e-4 and errs["B"] < 1e-4
    assert ntf.history == steps + 1


def test_g12_lars_coder_at_c3_shape():
    """d = 1200, r = 100, b = 100 (the batch the reference's main uses): scikit-learn's codes for alpha = 2 and
    alpha = 0.1 on inputs redrawn from RandomState(29) as the generating script drew them."""
    from onmf_ontf_ndl_amd import Online_NTF
    from onmf_ontf_ndl_amd.engine import get_engine
    g = load_golden("g12_ntf_c3")
    rs = np.random.RandomState(29)
    d, r, b = 1200, 100, 100
    X = rs.rand(d, b).astype(np.float32)
    W = rs.rand(d, r).astype(np.float32) / np.float32(32)
    for name, alpha in (("a2", 2), ("a01", 0.1)):
        ntf = Online_NTF(np.zeros((2, 3, 4)), n_components=r, alpha=alpha)
        H = ntf.joint_sparse_code_tensor(X, W)
        want = g["H_" + name]
        print("g12 c3 %s: rel err %.2e, max abs %.2e, nonzeros per column %.1f" % (
            name, relerr(H, want), np.abs(H - want).max(), (want > 0).sum(axis=1).mean()))
        assert relerr(H, want) < 1e-4


@pytest.mark.parametrize("name,alpha", [("default2", None), ("a05", 0.5), ("a005", 0.05)])
def test_g9_lasso_codes(orc, name, alpha):
    from onmf_ontf_ndl_amd import Online_NTF
    g = load_golden("g9_lasso")
    X, W = g["X"], g["W"]
    ntf = Online_NTF(np.zeros((2, 3, 4)), n_components=W.shape[1], alpha=alpha, coder="cd")
    H = ntf.joint_sparse_code_tensor(X, W)                  # (b x r), like the reference
    assert H.shape == g["H_" + name].shape and H.dtype == np.float64
    a = 2 if alpha is None else alpha
    cd = orc.lasso_nonneg_code(X, W, alpha, solver="cd")
    np.testing.assert_allclose(H, cd, atol=2e-4)
    f_gpu = orc.lasso_objective(X, W, H.T, a)
    f_sk = orc.lasso_objective(X, W, g["H_" + name].T, a)
    assert np.all(f_gpu <= f_sk * (1 + 1e-5) + 1e-7)
    assert np.array_equal(H > 1e-6, g["H_" + name] > 1e-6)
    np.testing.assert_allclose(H, g["H_" + name], atol=5e-2)


def test_ntf_step_against_oracle(orc):
    from onmf_ontf_ndl_amd import Online_NTF
    rs = np.random.RandomState(3)
    d, r, b = 60, 9, 40
    X = rs.rand(d, b)
    W = rs.rand(d, r)
    W /= np.linalg.norm(W, axis=0)
    Hw = rs.rand(r, 50)
    A, B = Hw @ Hw.T / 5, Hw @ rs.rand(d, 50).T / 5
    ref = orc.OracleOnlineNTF(np.zeros((2, 3, 4)), n_components=r, alpha=0.1, solver="cd")
    H_o, A_o, B_o, W_o = ref.step(X, A, B, W, 5)
    ntf = Online_NTF(np.zeros((2, 3, 4)), n_components=r, alpha=0.1, coder="cd")
    H, A1, B1, W1 = ntf.step(X, A, B, W, np.int64(5))
    assert H.shape == (b, r)
    np.testing.assert_allclose(H, H_o, atol=2e-4)
    assert relerr(A1, A_o) < 1e-4 and relerr(B1, B_o) < 1e-4
    assert relerr(W1, W_o) < 1e-4
    assert ntf.history == 6.0


@pytest.mark.parametrize("mode,joint", [(0, False), (1, False), (2, False), (2, True), (0, True)])
def test_train_dict_single_shapes_and_constraints(orc, mode, joint):
    from onmf_ontf_ndl_amd import Online_NTF
    np.random.seed(4)
    T = np.random.rand(16, 3, 20)
    r = 5
    ntf = Online_NTF(T, n_components=r, iterations=3, batch_size=7, mode=mode,
                     learn_joint_dict=joint, alpha=0.05, subsample=True, coder="cd")
    W, A, B, code = ntf.train_dict_single()
    Xu = orc.unfold3(T, mode)
    Xu = Xu.T if joint else Xu
    d = Xu.shape[0]
    assert W.shape == (d, r) and A.shape == (r, r) and B.shape == (r, d)
    assert code.shape == (T.shape[1], r)                   # the reference never fills it (ontf.py:237)
    assert W.min() >= 0 and np.linalg.norm(W, axis=0).max() <= 1 + 1e-5
    assert ntf.history == 3.0
    # same random stream as the oracle class: W0 then per step (idx) - compare end to end
    np.random.seed(4)
    T2 = np.random.rand(16, 3, 20)
    ref = orc.OracleOnlineNTF(T2, n_components=r, iterations=3, batch_size=7, mode=mode,
                              learn_joint_dict=joint, alpha=0.05, subsample=True, solver="cd")
    W_o, A_o, B_o, _ = ref.train_dict_single()
    assert relerr(W, W_o) < 2e-4 and relerr(A, A_o) < 2e-4 and relerr(B, B_o) < 2e-4


@pytest.mark.parametrize("r,d,b,alpha", [(5, 12, 70, 0.05), (30, 40, 33, 0.2), (64, 90, 20, 0.3), (65, 90, 20, 0.3),
                                         (130, 150, 12, 0.5), (200, 260, 9, 1.0), (201, 260, 9, 1.0),
                                         (256, 300, 8, 0.02)])
def test_lars_coder_every_register_and_storage_form(orc, r, d, b, alpha):
    """One to four atoms per lane; the inverse of the active Gram matrix in LDS (r <= 200) and in the global
    workspace (r > 200); small penalties so that many atoms enter and some leave again.  Against the oracle's
    restatement of scikit-learn's routine (itself held to scikit-learn's vectors)."""
    from onmf_ontf_ndl_amd.engine import get_engine
    eng = get_engine()
    rs = np.random.RandomState(1000 + r)
    W = rs.rand(d, r).astype(np.float32)
    W /= np.linalg.norm(W, axis=0).astype(np.float32)
    X = rs.rand(d, b).astype(np.float32)
    if r == 5:
        X[:, 3] = 0.0                                        # a column with no correlation at all: code 0
    H, info = eng.sparse_code_lars(eng.dev(X), eng.dev(W), alpha=alpha, want_info=True)
    H = H.cpu().numpy().astype(np.float64).T
    info = info.cpu().numpy()
    rec = []
    want = orc.lars_pos_code(X.astype(np.float64), W.astype(np.float64), alpha, info=rec)
    events = np.array([i["n_iter"] for i in rec])
    drops = np.array([i["drops"] for i in rec])
    print("r=%d: rel err %.2e, events %d..%d (gpu %d..%d), atoms dropped %d (gpu %d)" % (
        r, relerr(H, want), events.min(), events.max(), info[:, 0].min(), info[:, 0].max(), drops.sum(),
        info[:, 3].sum()))
    assert relerr(H, want) < 1e-5
    assert np.array_equal(info[:, 0], events) and np.array_equal(info[:, 3], drops)
    assert np.all(info[:, 1] == 1)
    if r == 5:
        assert np.all(H[3] == 0) and info[3, 0] == 0


def test_lars_coder_second_launch_for_columns_that_outgrow_their_slots(orc):
    """A call sizes its first launch by the active-set sizes of the call before; columns that need more slots
    are coded again by a second launch with room for every atom.  Forced here (16 slots for columns that hold
    up to ~40 atoms), and through the history rule: a sparse call (large penalty) followed by a dense one."""
    from onmf_ontf_ndl_amd import _lib
    from onmf_ontf_ndl_amd.engine import get_engine
    eng = get_engine()
    rs = np.random.RandomState(77)
    d, r, b = 120, 90, 150
    W = rs.rand(d, r).astype(np.float32)
    W /= np.linalg.norm(W, axis=0).astype(np.float32)
    X = rs.rand(d, b).astype(np.float32)
    want = orc.lars_pos_code(X.astype(np.float64), W.astype(np.float64), 0.3)
    active = (want > 1e-12).sum(axis=1)
    assert active.max() > 20 and active.min() < 64
    Xd, Wd = eng.dev(X), eng.dev(W)
    host = lambda t: t.cpu().numpy().astype(np.float64).T
    import torch
    before = eng.counters()["lars_recoded"]
    try:
        eng.set_option(_lib.OPT_LARS_SLOTS, 16)
        H16, info = eng.sparse_code_lars(Xd, Wd, alpha=0.3, want_info=True)
        torch.cuda.synchronize()
        recoded = eng.counters()["lars_recoded"] - before
        eng.set_option(_lib.OPT_LARS_SLOTS, 64)
        H64 = eng.sparse_code_lars(Xd, Wd, alpha=0.3)
    finally:
        eng.set_option(_lib.OPT_LARS_SLOTS, 0)
    print("forced: %d of %d columns recoded; rel err %.2e" % (recoded, b, relerr(host(H16), want)))
    assert 0 < recoded <= b
    assert relerr(host(H16), want) < 1e-5 and torch.equal(H16, H64)
    assert np.all(info.cpu().numpy()[:, 1] == 1)
    # history rule: after a sparse call the first launch is small, and the dense call still comes out right
    Hs = eng.sparse_code_lars(Xd, Wd, alpha=5.0)
    torch.cuda.synchronize()
    print("slots used by the sparse call:", eng.counters()["lars_slots_used"])
    assert eng.counters()["lars_slots_used"] <= 9          # -> 16 slots in the next call's first launch
    before = eng.counters()["lars_recoded"]
    Hd = eng.sparse_code_lars(Xd, Wd, alpha=0.3)
    torch.cuda.synchronize()
    print("after a sparse call: %d columns recoded, slots used %d" % (eng.counters()["lars_recoded"] - before,
                                                                     eng.counters()["lars_slots_used"]))
    assert eng.counters()["lars_recoded"] > before
    assert torch.equal(Hd, H16)
    assert relerr(host(Hs), orc.lars_pos_code(X.astype(np.float64), W.astype(np.float64), 5.0)) < 1e-5
