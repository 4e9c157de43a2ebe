#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference implementation.

Run in the build container only (needs /root/reference, which never travels):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

It imports the reference's ``src/onmf.py`` unmodified, runs it under fixed legacy-NumPy seeds
and stores inputs + outputs as small ``.npz`` files.  Nothing of the reference's text is stored;
the files hold arrays only.  ``np.linalg.norm`` is wrapped (not the reference) so the spectral
norms the reference computes at src/onmf.py:265 are recorded as the per-sweep ``dist`` trace.

The sklearn LassoLars vectors (case g9) come from scikit-learn itself, called with the argument
values of src/ontf.py:79-86; src/ontf.py is not imported because its ``tensorly`` and
``progressbar`` imports are not installed here.
"""
import os
import sys

import numpy as np

REF = os.environ.get("ONMF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

from src.onmf import Online_NMF, update_code_within_radius  # noqa: E402


class NormTap:
    """Record every 2-D ``norm(M, 2)`` call as the reference makes them (num, den, num, den...)."""

    def __init__(self):
        self.vals = []
        self._orig = np.linalg.norm

    def __enter__(self):
        tap = self

        def wrapped(x, ord=None, *a, **k):
            v = tap._orig(x, ord, *a, **k)
            if ord == 2 and np.ndim(x) == 2:
                tap.vals.append(float(v))
            return v
        np.linalg.norm = wrapped
        return self

    def __exit__(self, *exc):
        np.linalg.norm = self._orig

    def dists(self):
        v = np.array(self.vals, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            return v[0::2] / v[1::2]


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def g1_code():
    """update_code_within_radius alone, H0 given; alpha in {0,1}; tol in {0.01, 0}."""
    rs = np.random.RandomState(11)
    d, r, b = 64, 8, 32
    X = rs.rand(d, b)
    W = rs.rand(d, r)
    H0 = rs.rand(r, b)
    out = dict(X=X, W=W, H0=H0)
    for alpha in (0, 1):
        for tol in (0.01, 0.0):
            with NormTap() as tap:
                H = update_code_within_radius(X, W, H0=H0.copy(), r=None, alpha=alpha,
                                              sub_iter=10, stopping_diff=tol)
            key = "a%d_t%s" % (alpha, "01" if tol else "0")
            out["H_" + key] = H
            out["dist_" + key] = tap.dists()
    # a case that stops early: H0 already near the fixed point (many sweeps from the same start)
    Hc = update_code_within_radius(X, W, H0=H0.copy(), r=None, alpha=0, sub_iter=400,
                                   stopping_diff=0.0)
    with NormTap() as tap:
        He = update_code_within_radius(X, W, H0=Hc.copy(), r=None, alpha=0, sub_iter=10,
                                       stopping_diff=0.01)
    out["H0_early"] = Hc
    out["H_early"] = He
    out["dist_early"] = tap.dists()
    # single column (the per-patch call of image_reconstruction.py:384), alpha=1
    x1 = X[:, :1].copy()
    h1 = H0[:, :1].copy()
    with NormTap() as tap:
        Hs = update_code_within_radius(x1, W, H0=h1.copy(), r=None, alpha=1, sub_iter=10,
                                       stopping_diff=0.01)
    out["H_single"] = Hs
    out["dist_single"] = tap.dists()
    # radius branch (never taken by any caller; kept for the oracle only)
    Hr = update_code_within_radius(X[:, :6].copy(), W, H0=H0[:, :6].copy(), r=0.5, alpha=0,
                                   sub_iter=3, stopping_diff=0.0)
    out["H_radius"] = Hr
    save("g1_code", **out)


def g2_dict():
    """update_dict alone with a random PSD A and a B; includes atoms that clip to zero."""
    rs = np.random.RandomState(12)
    d, r = 48, 12
    W = rs.rand(d, r)
    Hh = rs.rand(r, 200)
    Xx = rs.rand(d, 200)
    A = Hh @ Hh.T
    B = Hh @ Xx.T
    B[3, :] = -1.0           # drives atom 3 to the all-zero column (norm 0 -> divide by max(1,0))
    nmf = Online_NMF(Xx, n_components=r)
    W1 = nmf.update_dict(W, A, B)
    Wz = nmf.update_dict(W, np.zeros((r, r)), np.zeros((r, d)))
    save("g2_dict", W=W, A=A, B=B, W1=W1, W1_zero_agg=Wz)


def g3_steps():
    """step() three times, aggregates carried by the caller, zero initial aggregates."""
    np.random.seed(13)
    d, r, b = 40, 6, 96
    X = np.random.rand(d, b)
    W = np.random.rand(d, r)
    agg = [np.zeros((r, r)), np.zeros((r, d))]
    nmf = Online_NMF(X, n_components=r)
    out = dict(X=X, W0=W)
    for s in range(3):
        st = np.random.get_state()
        with NormTap() as tap:
            H, agg, W = nmf.step(X, agg, W, np.int64(s + 1))
        after = np.random.get_state()
        np.random.set_state(st)
        H0 = np.random.rand(r, b)
        np.random.set_state(after)
        out.update({"H0_%d" % s: H0, "H_%d" % s: H, "A_%d" % s: agg[0], "B_%d" % s: agg[1],
                    "W_%d" % s: W, "dist_%d" % s: tap.dists(),
                    "hist_%d" % s: np.float64(nmf.history)})
    save("g3_steps", **out)


def _replay_train(seed, d, r, n, iterations, batch, subsample, draw_w):
    """Re-draw what train_dict draws, in its order (W, then per step [idx], H0)."""
    np.random.seed(seed)
    X = np.random.rand(d, n)
    st = np.random.get_state()
    W0 = np.random.rand(d, r) if draw_w else None
    idxs, h0s = [], []
    for _ in range(1, iterations):
        if subsample:
            idx = np.random.randint(n, size=batch)
        else:
            idx = np.arange(n)
        idxs.append(idx)
        h0s.append(np.random.rand(r, len(idx)))
    end = np.random.get_state()
    np.random.set_state(st)
    return X, W0, idxs, h0s, end


def g4_train_warm():
    """train_dict(iterations=5) then a warm restart (ini_dict/ini_agg/history)."""
    d, r, n, its = 30, 5, 64, 5
    X, W0, idxs, h0s, _ = _replay_train(14, d, r, n, its, n, False, True)
    nmf = Online_NMF(X, n_components=r, iterations=its, batch_size=n, subsample=False)
    with NormTap() as tap:
        W, agg, code = nmf.train_dict()
    out = dict(X=X, W0=W0, H0s=np.stack(h0s), W=W, A=agg[0], B=agg[1], code=code.copy(),
               hist=np.float64(nmf.history), dist=tap.dists())
    # warm restart: draws only the H0s
    st = np.random.get_state()
    h0b = [np.random.rand(r, n) for _ in range(1, its)]
    np.random.set_state(st)
    nmf2 = Online_NMF(X, n_components=r, iterations=its, batch_size=n, ini_dict=W, ini_agg=agg,
                      history=nmf.history, subsample=False)
    W2, agg2, code2 = nmf2.train_dict()
    out.update(H0s_b=np.stack(h0b), W_b=W2, A_b=agg2[0], B_b=agg2[1], code_b=code2.copy(),
               hist_b=np.float64(nmf2.history))
    save("g4_train_warm", **out)


def g5_subsample():
    """subsample=True with batch_size < n and duplicate indices."""
    d, r, n, its, batch = 24, 4, 20, 4, 32     # batch > n guarantees duplicates
    X, W0, idxs, h0s, _ = _replay_train(15, d, r, n, its, batch, True, True)
    nmf = Online_NMF(X, n_components=r, iterations=its, batch_size=batch, subsample=True)
    W, agg, code = nmf.train_dict()
    save("g5_subsample", X=X, W0=W0, idx=np.stack(idxs), H0s=np.stack(h0s), W=W, A=agg[0],
         B=agg[1], code=code.copy(), hist=np.float64(nmf.history))


def g6_full_code():
    """full_code=True carries the third aggregate C; surrogate loss from (W,A,B,C)."""
    d, r, n, its = 20, 4, 48, 4
    X, W0, idxs, h0s, _ = _replay_train(16, d, r, n, its, n, False, True)
    nmf = Online_NMF(X, n_components=r, iterations=its, batch_size=n, subsample=False)
    W, agg, code = nmf.train_dict(full_code=True)
    loss = np.trace(W @ agg[0] @ W.T) - 2 * np.trace(W @ agg[1]) + np.trace(agg[2])
    save("g6_full_code", X=X, W0=W0, H0s=np.stack(h0s), W=W, A=agg[0], B=agg[1], C=agg[2],
         code=code.copy(), loss=np.float64(loss))


def g7_beta_alpha():
    """beta=0.5 forgetting and alpha=0.3 L1 shift through train_dict, with a warm restart."""
    d, r, n, its = 20, 4, 48, 3
    X, W0, idxs, h0s, _ = _replay_train(17, d, r, n, its, n, False, True)
    nmf = Online_NMF(X, n_components=r, iterations=its, batch_size=n, subsample=False,
                     alpha=0.3, beta=0.5)
    W, agg, _ = nmf.train_dict()
    st = np.random.get_state()
    h0b = [np.random.rand(r, n) for _ in range(1, its)]
    np.random.set_state(st)
    nmf2 = Online_NMF(X, n_components=r, iterations=its, batch_size=n, ini_dict=W, ini_agg=agg,
                      history=nmf.history, alpha=0.3, beta=0.5, subsample=False)
    W2, agg2, _ = nmf2.train_dict()
    save("g7_beta_alpha", X=X, W0=W0, H0s=np.stack(h0s), W=W, A=agg[0], B=agg[1],
         H0s_b=np.stack(h0b), W_b=W2, A_b=agg2[0], B_b=agg2[1], hist_b=np.float64(nmf2.history))


def g8_stream():
    """Downsized C1-style stream: 12 driver iterations of iterations=2 on image-like patches.

    The patch data are synthetic smooth 8x8 patches (no reference image is stored).
    """
    np.random.seed(18)
    k, r, b, outer = 8, 6, 80, 12
    yy, xx = np.mgrid[0:64, 0:64] / 64.0
    img = 0.5 + 0.25 * np.sin(6.0 * xx) * np.cos(4.0 * yy) + 0.25 * xx * yy
    W = agg = None
    hist = 0
    Xs, H0s, sweeps = [], [], []
    for _ in range(outer):
        a = np.random.randint(0, 64 - k, size=b)
        c = np.random.randint(0, 64 - k, size=b)
        X = np.stack([img[i:i + k, j:j + k].reshape(-1) for i, j in zip(a, c)], axis=1)
        st = np.random.get_state()
        if W is None:
            W0 = np.random.rand(k * k, r)
        H0 = np.random.rand(r, b)
        np.random.set_state(st)
        nmf = Online_NMF(X, n_components=r, iterations=2, batch_size=b, ini_dict=W, ini_agg=agg,
                         history=hist, subsample=False)
        with NormTap() as tap:
            W, agg, _ = nmf.train_dict()
        hist = nmf.history
        Xs.append(X)
        H0s.append(H0)
        sweeps.append(len(tap.dists()))
    nmf = Online_NMF(Xs[-1], n_components=r)
    st = np.random.get_state()
    Hl0 = np.random.rand(r, b)
    np.random.set_state(st)
    Hl = nmf.sparse_code(Xs[-1], W)
    resid = np.linalg.norm(Xs[-1] - W @ Hl) / np.linalg.norm(Xs[-1])
    save("g8_stream", Xs=np.stack(Xs), W0=W0, H0s=np.stack(H0s), sweeps=np.array(sweeps),
         W=W, A=agg[0], B=agg[1], hist=np.float64(hist), H_last0=Hl0, H_last=Hl,
         resid=np.float64(resid))


def g9_lasso():
    """sklearn SparseCoder(lasso_lars, positive) with the arguments of src/ontf.py:79-86."""
    from sklearn.decomposition import SparseCoder
    rs = np.random.RandomState(19)
    d, r, b = 48, 10, 24
    W = rs.rand(d, r)
    W /= np.linalg.norm(W, axis=0)
    X = rs.rand(d, b)
    out = dict(X=X, W=W)
    for name, alpha in (("default2", 2), ("a05", 0.5), ("a005", 0.05)):
        coder = SparseCoder(dictionary=W.T, transform_n_nonzero_coefs=None, transform_alpha=alpha,
                            transform_algorithm="lasso_lars", positive_code=True)
        out["H_" + name] = coder.transform(X.T)
    save("g9_lasso", **out)


def g10_image():
    """Patch matrix and overlap average from scikit-learn (the calls of image_reconstruction.py:163-166
    and :354), and one reference coder call per window as image_reconstruction.py:384 makes them
    (alpha=1, sub_iter=10, stopping_diff=0.01, H0=None drawn from the global generator)."""
    from sklearn.feature_extraction.image import extract_patches_2d, reconstruct_from_patches_2d
    rs = np.random.RandomState(23)
    k = 4
    grey = rs.rand(11, 9)
    patches = extract_patches_2d(grey, (k, k))
    X_grey = patches.reshape(len(patches), -1).T
    vals = rs.rand(*patches.shape)
    fold_grey = reconstruct_from_patches_2d(vals, grey.shape)
    color = rs.rand(9, 10, 3)
    W = rs.rand(k * k * 3, 6)
    W /= np.linalg.norm(W, axis=0)
    np.random.seed(77)
    codes, starts, sweeps, codes_loose, sweeps_loose = [], [], [], [], []
    for i in range(0, color.shape[0] - k, 2):
        for j in range(0, color.shape[1] - k, 2):
            patch = color[i:i + k, j:j + k, :].reshape((-1, 1))
            state = np.random.get_state()
            starts.append(np.random.rand(W.shape[1], 1))     # the draw the call below repeats
            np.random.set_state(state)
            with NormTap() as tap:
                codes.append(update_code_within_radius(patch, W, H0=None, r=None, alpha=1, sub_iter=10,
                                                       stopping_diff=0.01))
            sweeps.append(len(tap.dists()))
            # the same start under a looser test, so that the windows stop after different sweep counts
            with NormTap() as tap:
                codes_loose.append(update_code_within_radius(patch, W, H0=starts[-1].copy(), r=None, alpha=1,
                                                             sub_iter=10, stopping_diff=0.03))
            sweeps_loose.append(len(tap.dists()))
    save("g10_image", k=np.int64(k), grey=grey, X_grey=X_grey, vals=vals.reshape(len(vals), -1).T,
         fold_grey=fold_grey, color=color, W=W, stride=np.int64(2), H0=np.hstack(starts),
         code=np.hstack(codes), sweeps=np.array(sweeps), tol_loose=np.float64(0.03),
         code_loose=np.hstack(codes_loose), sweeps_loose=np.array(sweeps_loose))


def g11_renoir():
    """The C1 configuration on real data: 10 x 10 grey patches of Data/renoir/0.jpg, r = 16, 1000 patches
    per driver iteration (iterations=2 each: one step), state carried - the first 16 iterations of the
    100-iteration stream whose step index 2 stops after 4 sweeps.  The patches are stored as the image's
    own uint8 values (X = patches / 255, as image_reconstruction.py:88 scales the image); the corner draws
    use NumPy's global stream, which the replaying test repeats to stay in step with the H0 draws."""
    from PIL import Image
    np.random.seed(0)
    img8 = np.asarray(Image.open(os.path.join(REF, "Data", "renoir", "0.jpg")).convert("L"))
    data = img8 / 255
    k, r, nb, outer = 10, 16, 1000, 16
    W = agg = None
    hist = 0
    patches, sweeps, dists = [], [], []
    for _ in range(outer):
        a = np.random.randint(0, data.shape[0] - k, size=nb)
        c = np.random.randint(0, data.shape[1] - k, size=nb)
        P8 = np.stack([img8[i:i + k, j:j + k].reshape(-1) for i, j in zip(a, c)], axis=1)
        X = np.stack([data[i:i + k, j:j + k].reshape(-1) for i, j in zip(a, c)], axis=1)
        assert np.array_equal(P8 / 255, X)
        nmf = Online_NMF(X, n_components=r, iterations=2, batch_size=nb, ini_dict=W, ini_agg=agg,
                         history=hist, subsample=False)
        with NormTap() as tap:
            W, agg, _ = nmf.train_dict()
        hist = nmf.history
        patches.append(P8)
        d_ = tap.dists()
        sweeps.append(len(d_))
        dists.append(np.pad(d_, (0, 10 - len(d_)), constant_values=np.nan))
    nmf = Online_NMF(X, n_components=r)
    Hl = nmf.sparse_code(X, W)
    resid = np.linalg.norm(X - W @ Hl) / np.linalg.norm(X)
    save("g11_renoir", patches=np.stack(patches).astype(np.uint8), image_shape=np.array(data.shape),
         sweeps=np.array(sweeps), dists=np.stack(dists), W=W, A=agg[0], B=agg[1], hist=np.float64(hist),
         H_last=Hl, resid=np.float64(resid))
    print("g11: sweeps", sweeps, "resid", resid)


def _ntf_stream(shape, r, alpha, beta, batch, steps, seed):
    """The loop of Online_NTF.train_dict_single (src/ontf.py:203-236) for mode 2 / learn_joint_dict, with the
    reference's pieces that can be run here: scikit-learn's coder called with the arguments of src/ontf.py:79-86,
    the aggregate lines of src/ontf.py:147-148 and the reference's own update_dict (src/onmf.py:92-116, the body
    src/ontf.py:91-115 repeats).  src/ontf.py itself does not import here (tensorly, progressbar)."""
    from sklearn.decomposition import SparseCoder
    np.random.seed(seed)
    T = np.random.rand(*shape)
    Xu = np.reshape(np.moveaxis(T, 2, 0), (T.shape[2], -1)).T           # tl_unfold(T, 2).T, src/ontf.py:207
    d, n = Xu.shape
    W = np.random.rand(d, r)                                            # src/ontf.py:213
    A, B = np.zeros((r, r)), np.zeros((r, d))
    helper = Online_NMF(Xu, n_components=r)
    out = dict(T=T, W0=W.copy())
    for i in range(1, steps + 1):
        idx = np.random.randint(n, size=batch)                          # src/ontf.py:230
        X = Xu[:, idx]
        coder = SparseCoder(dictionary=W.T, transform_n_nonzero_coefs=None,
                            transform_alpha=2 if alpha is None else alpha,
                            transform_algorithm="lasso_lars", positive_code=True)
        H = coder.transform(X.T)
        t = np.float64(i)
        wt = t ** (-(1 if beta is None else beta))
        A1 = (1 - wt) * A + wt * np.dot(H.T, H)
        B1 = (1 - wt) * B + wt * np.dot(H.T, X.T)
        W = helper.update_dict(W, A, B)                                 # old aggregates, src/ontf.py:151
        A, B = A1, B1
        out.update({"idx_%d" % i: idx, "H_%d" % i: H, "A_%d" % i: A, "B_%d" % i: B, "W_%d" % i: W})
    return out


def g12_ntf_lars():
    """Online_NTF streams coded by scikit-learn's positive LassoLars (src/ontf.py:79-86, :136-151), and the coder
    alone at the C3 shape (d = 1200, r = 100, b = 100; inputs are float32 values drawn from RandomState(29),
    redrawn by the tests, only the codes are stored)."""
    from sklearn.decomposition import SparseCoder
    a = _ntf_stream((16, 3, 200), 8, 0.5, None, 40, 6, 27)
    save("g12_ntf_a", alpha=np.float64(0.5), steps=np.int64(6), **a)
    bq = _ntf_stream((25, 3, 300), 12, None, 0.7, 64, 7, 28)
    save("g12_ntf_b", steps=np.int64(7), beta=np.float64(0.7), **bq)
    rs = np.random.RandomState(29)
    d, r, b = 1200, 100, 100
    X = rs.rand(d, b).astype(np.float32).astype(np.float64)
    W = rs.rand(d, r).astype(np.float32).astype(np.float64) / 32       # exact scaling; column norms ~ 0.62
    out = {}
    for name, alpha in (("a2", 2), ("a01", 0.1)):
        coder = SparseCoder(dictionary=W.T, transform_n_nonzero_coefs=None, transform_alpha=alpha,
                            transform_algorithm="lasso_lars", positive_code=True)
        out["H_" + name] = coder.transform(X.T)
    save("g12_ntf_c3", **out)


if __name__ == "__main__":
    cases = (g1_code, g2_dict, g3_steps, g4_train_warm, g5_subsample, g6_full_code,
             g7_beta_alpha, g8_stream, g9_lasso, g10_image, g11_renoir, g12_ntf_lars)
    want = set(sys.argv[1:])
    for fn in cases:
        if not want or fn.__name__.split("_")[0] in want:
            fn()
