"""CPU oracle for the online-NMF minibatch step (TEST INFRASTRUCTURE, not product code).

This module is a NumPy restatement of the hot path of HanbaekLyu/ONMF_ONTF_NDL:

* ``code_sweeps``            <- ``update_code_within_radius``  (src/onmf.py:233-271)
* ``dictionary_pass``        <- ``Online_NMF.update_dict``      (src/onmf.py:92-116,
                                                                 identical body at src/ontf.py:91-115)
* ``aggregate_update``       <- the (A,B[,C]) recursion in ``step`` (src/onmf.py:150-158,
                                                                 transposed form src/ontf.py:142-148)
* ``minibatch_step``         <- ``Online_NMF.step``             (src/onmf.py:119-167)
* ``OracleOnlineNMF``        <- ``Online_NMF`` incl. ``train_dict`` (src/onmf.py:20-226)
* ``unfold3`` / ``OracleOnlineNTF`` <- ``Online_NTF``           (src/ontf.py:19-244)
* ``lasso_nonneg_code``      <- ``joint_sparse_code_tensor``    (src/ontf.py:59-89) which hands
                               the solve to scikit-learn's ``SparseCoder`` (LassoLars, positive)
* ``lars_pos_path``          <- the routine behind that call: ``_lars_path_solver`` of scikit-learn
                               1.7.2 (a dependency the reference neither vendors nor pins), restated
                               branch by branch and pinned by vectors generated from the library

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The shipped GPU path never
routes through this file.

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference's own
``src/onmf.py`` under fixed seeds (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py``
holds this restatement to those vectors at ~1e-12.  The sklearn LassoLars coder of ``src/ontf.py``
is pinned by vectors generated from scikit-learn 1.7.2 with the call arguments of ontf.py:79-86;
``tensorly.unfold`` (ontf.py:204,207) is absent from this environment and is restated from its
documented definition - that single reshape is "parity unpinned".

Reference behaviours that are kept on purpose (they are observable):
  * the dictionary pass inside a step reads the aggregates from BEFORE the step (onmf.py:161);
  * ``train_dict`` hands the *initial* aggregates to every one of its steps (onmf.py:217);
  * ``history`` ends at t+1, so a warm restart skips one value of t (onmf.py:162, :204-220);
  * the stopping test uses spectral norms (``np.linalg.norm(M, 2)``, onmf.py:265);
  * ``iterations=N`` runs N-1 steps (onmf.py:206);
  * aggregates are sums over the batch, not means (onmf.py:155-156);
  * ``code[:, idx] += H`` does not accumulate duplicate indices (onmf.py:221).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "spectral_norm", "code_sweeps", "dictionary_pass", "aggregate_update", "minibatch_step",
    "OracleOnlineNMF", "unfold3", "lasso_nonneg_code", "lasso_objective", "OracleOnlineNTF",
    "surrogate_loss", "lars_pos_path", "lars_pos_code",
]


# ----------------------------------------------------------------------------------------
# sparse-code solve
# ----------------------------------------------------------------------------------------
def spectral_norm(M, method="svd"):
    """Largest singular value of a 2-D array.

    ``method="svd"`` is what ``np.linalg.norm(M, 2)`` does (src/onmf.py:265).
    ``method="gram"`` gets the same number from the small Gram matrix (rows <= columns assumed
    cheap side), which is how a GPU has to do it; used as the fair "optimised CPU" variant.
    """
    M = np.asarray(M)
    if method == "svd":
        return np.linalg.norm(M, 2)
    small = M @ M.T if M.shape[0] <= M.shape[1] else M.T @ M
    lam = np.linalg.eigvalsh(small.astype(np.float64))[-1]
    return np.sqrt(max(lam, 0.0))


def code_sweeps(X, W, H0=None, r=None, alpha=0, sub_iter=10, stopping_diff=0.1, *,
                norm="svd", dtype=np.float64, trace=None):
    """Row-wise projected Gauss-Seidel solve for H >= 0 in X ~ W H  (src/onmf.py:233-271).

    Per sweep i (0-based) and row k, in order, with Gm = W^T W and Cm = W^T X:
        grad  = Gm[k, :] @ H - Cm[k, :] + alpha
        H[k]  = max(0, H[k] - grad / (sqrt(i + 10) * (Gm[k, k] + 1)))
    The loop runs while i < sub_iter and dist > stopping_diff, dist being the ratio of the
    spectral norms of (H - H_before_sweep) and H_before_sweep (onmf.py:252, :265).

    ``H0=None`` draws U[0,1) from NumPy's global legacy generator exactly where the reference
    does (onmf.py:245-246).  ``r`` is the radius argument (onmf.py:260-263): with ``r`` given the
    iterate is pulled back toward H0 after every row; note the reference then rebinds H0 to the
    iterate itself (line 263), which this restatement keeps.
    ``trace`` (a list) receives (sweep_index, dist) pairs.
    """
    X = np.asarray(X, dtype=dtype)
    W = np.asarray(W, dtype=dtype)
    Gm = W.T @ W
    Cm = W.T @ X
    n_atoms, n_cols = W.shape[1], X.shape[1]
    if H0 is None:
        H0 = np.random.rand(n_atoms, n_cols)
    H = np.array(H0, dtype=dtype, copy=True)
    anchor = np.asarray(H0, dtype=dtype)
    shift = np.asarray(alpha, dtype=dtype)

    sweep, dist = 0, 1
    while sweep < sub_iter and dist > stopping_diff:
        before = H.copy()
        damp = (sweep + 10) ** 0.5
        for k in range(n_atoms):
            g = Gm[k] @ H - Cm[k] + shift
            H[k] = np.maximum(H[k] - g * (1 / (damp * (Gm[k, k] + 1))), 0)
            if r is not None:
                gap = np.linalg.norm(H - anchor, 2)
                H = anchor + (r / max(r, gap)) * (H - anchor)
            anchor = H            # onmf.py:263 (aliasing kept: with r=None this is a no-op)
        with np.errstate(divide="ignore", invalid="ignore"):
            dist = spectral_norm(H - before, norm) / spectral_norm(before, norm)
        if trace is not None:
            trace.append((sweep, float(dist)))
        sweep += 1
    return H


# ----------------------------------------------------------------------------------------
# dictionary pass and aggregates
# ----------------------------------------------------------------------------------------
def dictionary_pass(W, A, B, dtype=np.float64):
    """One block-coordinate pass over the atoms (src/onmf.py:92-116).

    For j = 0..r-1 in order:  u = W[:, j] - (W @ A[:, j] - B[j, :]) / (A[j, j] + 1);
    u = max(u, 0);  W[:, j] = u / max(1, ||u||_2).   Later atoms see earlier updates.
    """
    W1 = np.array(W, dtype=dtype, copy=True)
    A = np.asarray(A, dtype=dtype)
    B = np.asarray(B, dtype=dtype)
    for j in range(W1.shape[1]):
        resid = W1 @ A[:, j] - B[j, :]
        u = np.maximum(W1[:, j] - resid * (1 / (A[j, j] + 1)), 0)
        W1[:, j] = u * (1 / max(1, np.linalg.norm(u)))
    return W1


def aggregate_update(aggregates, H, X, t, beta=None, code_is_transposed=False):
    """(1-w) * old + w * new with w = t**(-beta), beta=1 if None (src/onmf.py:150-158).

    ``code_is_transposed`` selects the ontf form where H is (b x r) (src/ontf.py:147-148).
    A third aggregate C tracks X X^T (onmf.py:157-158).
    """
    t = np.float64(t)
    w = t ** (-(1 if beta is None else beta))
    Hm = H.T if code_is_transposed else H
    out = [(1 - w) * aggregates[0] + w * (Hm @ Hm.T),
           (1 - w) * aggregates[1] + w * (Hm @ X.T)]
    if len(aggregates) == 3:
        out.append((1 - w) * aggregates[2] + w * (X @ X.T))
    return out


def minibatch_step(X, aggregates, W, t, alpha=None, beta=None, H0=None, *,
                   sub_iter=10, stopping_diff=0.01, norm="svd", dtype=np.float64, trace=None):
    """One online-NMF step (src/onmf.py:119-167): code, aggregates, dictionary (old aggregates).

    Returns (H1, aggregates1, W1, history) with history = t + 1 (onmf.py:162).
    """
    a = 0 if alpha is None else alpha            # onmf.py:82-84
    H1 = code_sweeps(X, W, H0=H0, r=None, alpha=a, sub_iter=sub_iter,
                     stopping_diff=stopping_diff, norm=norm, dtype=dtype, trace=trace)
    agg1 = aggregate_update(aggregates, H1, np.asarray(X, dtype=dtype), t, beta)
    W1 = dictionary_pass(W, aggregates[0], aggregates[1], dtype=dtype)   # onmf.py:161
    return H1, agg1, W1, np.float64(t) + 1


def surrogate_loss(W, A, B, C):
    """tr(W A W^T) - 2 tr(W B) + tr(C)  (ising_reconstruction.py:133)."""
    return float(np.trace(W @ A @ W.T) - 2 * np.trace(W @ B) + np.trace(C))


class OracleOnlineNMF:
    """Same constructor and return conventions as ``Online_NMF`` (src/onmf.py:20-226)."""

    def __init__(self, X, n_components=100, iterations=500, batch_size=20, ini_dict=None,
                 ini_agg=None, history=0, alpha=None, beta=None, subsample=False,
                 norm="svd", dtype=np.float64):
        self.X = X
        self.n_components = n_components
        self.batch_size = batch_size
        self.iterations = iterations
        self.subsample = subsample
        self.initial_dict = ini_dict
        self.initial_agg = ini_agg
        self.history = history
        self.alpha = alpha
        self.beta = beta
        self.code = np.zeros((n_components, X.shape[1]))
        self.norm = norm
        self.dtype = dtype
        self.sweep_log = []          # per step: list of (sweep, dist)

    def sparse_code(self, X, W, H0=None):
        a = 0 if self.alpha is None else self.alpha
        tr = []
        H = code_sweeps(X, W, H0=H0, r=None, alpha=a, sub_iter=10, stopping_diff=0.01,
                        norm=self.norm, dtype=self.dtype, trace=tr)
        self.sweep_log.append(tr)
        return H

    def update_dict(self, W, A, B):
        return dictionary_pass(W, A, B, dtype=self.dtype)

    def step(self, X, aggregates, W, t, H0=None):
        H1 = self.sparse_code(X, W, H0=H0)
        agg1 = aggregate_update(aggregates, H1, np.asarray(X, dtype=self.dtype), t, self.beta)
        W1 = self.update_dict(W, aggregates[0], aggregates[1])
        self.history = np.float64(t) + 1
        return H1, agg1, W1

    def train_dict(self, full_code=False):
        d, n = self.X.shape
        r = self.n_components
        if self.initial_dict is None:
            W = np.random.rand(d, r)                       # onmf.py:190
            held = [np.zeros((r, r)), np.zeros((r, d))]
            if full_code:
                held.append(np.zeros((d, d)))
        else:
            W = self.initial_dict
            held = [self.initial_agg[0], self.initial_agg[1]]
            if full_code:
                held.append(self.initial_agg[2])
        t0 = self.history
        returned = self.initial_agg if self.initial_dict is not None else held
        for i in np.arange(1, self.iterations):
            idx = np.arange(n)
            if self.subsample:
                idx = np.random.randint(n, size=self.batch_size)    # onmf.py:212
            # every step is handed the aggregates the call started with (onmf.py:217-219)
            H, returned, W = self.step(self.X[:, idx], list(held), W, t0 + i)
            self.code[:, idx] += H                                  # onmf.py:221
        return W, returned, self.code


# ----------------------------------------------------------------------------------------
# tensor (ontf) variant
# ----------------------------------------------------------------------------------------
def unfold3(T, mode):
    """Mode-``mode`` unfolding, rows = T.shape[mode], remaining axes flattened in C order.

    Restates ``tensorly.unfold`` (used at src/ontf.py:204,207) from its documented definition
    ``reshape(moveaxis(T, mode, 0), (T.shape[mode], -1))``; tensorly is not installed here, so
    this one reshape is parity-unpinned.
    """
    T = np.asarray(T)
    return np.reshape(np.moveaxis(T, mode, 0), (T.shape[mode], -1))


def lasso_objective(X, W, H, alpha):
    """Column-wise 0.5*||x - W h||^2 + alpha*||h||_1 for H given as (r x b)."""
    R = X - W @ H
    return 0.5 * np.sum(R * R, axis=0) + alpha * np.sum(np.abs(H), axis=0)


def lasso_nonneg_code(X, W, alpha=None, solver="sklearn", sweeps=20000, tol=1e-12):
    """argmin_{h>=0} 0.5||x - W h||^2 + alpha ||h||_1 per column; returns H as (b x r).

    ``solver="sklearn"`` makes the call of src/ontf.py:79-86 verbatim in meaning
    (SparseCoder, lasso_lars, positive_code, transform_alpha = alpha or 2).
    ``solver="lars"`` is this module's own restatement of the path scikit-learn follows
    (``lars_pos_code``): the same numbers as ``"sklearn"`` to rounding, without the library.
    ``solver="cd"`` is exact cyclic coordinate descent on the same convex problem, run to a fixed
    point - the optimum, which scikit-learn's positive path does NOT always reach (it misjudges the
    last segment once no atom with a positive correlation is left outside the active set).
    """
    a = 2 if alpha is None else alpha
    if solver == "lars":
        return lars_pos_code(X, W, a)
    if solver == "sklearn":
        from sklearn.decomposition import SparseCoder
        coder = SparseCoder(dictionary=np.asarray(W).T, transform_n_nonzero_coefs=None,
                            transform_alpha=a, transform_algorithm="lasso_lars",
                            positive_code=True)
        return coder.transform(np.asarray(X).T)
    W = np.asarray(W, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    Gm, Cm = W.T @ W, W.T @ X
    H = np.zeros((W.shape[1], X.shape[1]))
    for _ in range(sweeps):
        biggest = 0.0
        for k in range(H.shape[0]):
            if Gm[k, k] <= 0:
                continue
            new = np.maximum(H[k] - (Gm[k] @ H - Cm[k] + a) / Gm[k, k], 0)
            biggest = max(biggest, float(np.max(np.abs(new - H[k]), initial=0.0)))
            H[k] = new
        if biggest <= tol:
            break
    return H.T


def _givens_delete(L, n, out):
    """Remove row/column ``out`` from the n x n lower Cholesky factor held in L[:n, :n], in place.

    Restates ``sklearn.utils.arrayfuncs.cholesky_delete`` (scikit-learn 1.7.2, arrayfuncs.pyx:86-117):
    rows below ``out`` move up by one, then one Givens rotation per remaining row restores the
    triangle (BLAS ``rotg`` sign convention: r takes the sign of the larger entry; a negative
    diagonal is flipped together with c and s)."""
    for i in range(out, n - 1):
        L[i, :i + 2] = L[i + 1, :i + 2]
    for i in range(out, n - 1):
        a, b = L[i, i], L[i, i + 1]
        roe = a if abs(a) > abs(b) else b
        scale = abs(a) + abs(b)
        if scale == 0.0:
            c, s, rr = 1.0, 0.0, 0.0
        else:
            rr = scale * np.sqrt((a / scale) ** 2 + (b / scale) ** 2)
            if roe < 0:
                rr = -rr
            c, s = a / rr, b / rr
        if rr < 0:
            rr, c, s = -rr, -c, -s
        L[i, i], L[i, i + 1] = rr, 0.0
        lo = L[i + 1:n - 1, i].copy()
        hi = L[i + 1:n - 1, i + 1].copy()
        L[i + 1:n - 1, i] = c * lo + s * hi
        L[i + 1:n - 1, i + 1] = c * hi - s * lo


def lars_pos_path(Gram, Xy, n_samples, alpha_min, max_iter=1000, eps=np.finfo(float).eps, info=None):
    """Positive LARS-lasso path on a precomputed Gram matrix, ended at ``alpha_min``.

    Restates ``sklearn.linear_model._least_angle._lars_path_solver`` (scikit-learn 1.7.2,
    _least_angle.py:411-913) for the one argument set that src/ontf.py:79-86 reaches through
    ``SparseCoder`` -> ``_sparse_encode_precomputed`` (_dict_learning.py:118-135) -> ``LassoLars.fit``
    (_least_angle.py:1132-1149): method="lasso", positive=True, Gram and Xy given, return_path=False.
    Every branch of that routine is kept, including the ones that end the path before alpha_min
    (regressor limit, "alpha is increasing" bail-out) and the degenerate-pivot skip, because
    they decide which coefficients the reference hands to its aggregate update.
    ``info`` (a dict) receives n_iter, drops and the branch that ended the path."""
    Gram = np.asarray(Gram, dtype=np.float64)
    n_feat = Xy.shape[0]
    Cov = np.array(Xy, dtype=np.float64)         # correlations of the inactive variables, in `rest` order
    rest = list(range(n_feat))                   # inactive variables, aligned with Cov
    active = []
    coef = np.zeros(n_feat)
    prev_coef = np.zeros(n_feat)
    alpha = prev_alpha = 0.0
    L = np.zeros((n_feat, n_feat))
    n_iter = n_active = n_drops = 0
    drop = False
    tiny32 = np.finfo(np.float32).tiny
    eq_tol = np.finfo(np.float32).eps
    end = "?"
    while True:
        if Cov.size:
            c_idx = int(np.argmax(Cov))
            C = Cov[c_idx]
        else:
            C = 0.0
        alpha = C / n_samples
        if alpha <= alpha_min + eq_tol:                                  # :656-667
            if abs(alpha - alpha_min) > eq_tol:
                if n_iter > 0:
                    ss = (prev_alpha - alpha_min) / (prev_alpha - alpha)
                    coef = prev_coef + ss * (coef - prev_coef)
                alpha = alpha_min
            end = "alpha_min"
            break
        if n_iter >= max_iter or n_active >= n_feat:                     # :669-670
            end = "full"
            break
        if not drop:                                                     # :671-741
            new = rest[c_idx]
            # the chosen variable trades places with the first inactive one (:687-690)
            Cov[c_idx], Cov[0] = Cov[0], Cov[c_idx]
            rest[c_idx], rest[0] = rest[0], rest[c_idx]
            cov_full, rest_full = Cov, rest
            Cov, rest = Cov[1:], rest[1:]
            c = Gram[new, new]
            L[n_active, :n_active] = Gram[new, active]
            if n_active:
                # forward substitution with the current factor (:705-713)
                for i in range(n_active):
                    L[n_active, i] = (L[n_active, i] - L[i, :i] @ L[n_active, :i]) / L[i, i]
            v = L[n_active, :n_active] @ L[n_active, :n_active]
            diag = max(np.sqrt(abs(c - v)), eps)
            L[n_active, n_active] = diag
            if diag < 1e-7:                                              # :719-738
                Cov, rest = cov_full, rest_full
                Cov[0] = 0
                Cov[c_idx], Cov[0] = Cov[0], Cov[c_idx]
                # the routine swaps the correlations back but leaves its index and Gram arrays swapped
                # (`rest` here): position 0 keeps the old first variable's correlation under the
                # skipped variable's name, position c_idx holds 0 under the old first one's
                continue
            active.append(new)
            n_active += 1
        if n_iter > 0 and prev_alpha < alpha:                            # :748-761
            end = "alpha_up"
            break
        # direction: solve (L L^T) w = 1, scale to unit correlation decrease (:764-788)
        ones = np.ones(n_active)
        y = np.zeros(n_active)
        for i in range(n_active):
            y[i] = (ones[i] - L[i, :i] @ y[:i]) / L[i, i]
        w = np.zeros(n_active)
        for i in range(n_active - 1, -1, -1):
            w[i] = (y[i] - L[i + 1:n_active, i] @ w[i + 1:n_active]) / L[i, i]
        if n_active == 1 and w[0] == 0:
            w[0] = 1.0
            AA = 1.0
        else:
            AA = 1.0 / np.sqrt(np.sum(w))
            if not np.isfinite(AA):
                k = 0
                while not np.isfinite(AA):
                    Lr = L[:n_active, :n_active].copy()
                    Lr.flat[::n_active + 1] += (2 ** k) * eps
                    w = np.linalg.solve(Lr @ Lr.T, ones)
                    AA = 1.0 / np.sqrt(max(np.sum(w), eps))
                    k += 1
            w = w * AA
        corr = Gram[np.ix_(rest, active)] @ w if rest else np.zeros(0)   # :800
        corr = np.around(corr, decimals=15)                              # :804
        with np.errstate(divide="ignore", invalid="ignore"):
            ratios = (C - Cov) / (AA - corr + tiny32)
            pos = ratios[ratios > 0]
            g1 = pos.min() if pos.size else np.finfo(np.float64).max
            gamma = min(g1, C / AA)                                      # :806-808
            z = -coef[active] / (w + tiny32)                             # :815
        zp = z[z > 0]
        z_pos = zp.min() if zp.size else np.finfo(np.float64).max
        drop = False
        if z_pos < gamma:
            out = np.where(z == z_pos)[0][::-1]
            gamma = z_pos
            drop = True
        n_iter += 1
        prev_coef, prev_alpha = coef, alpha
        coef = np.zeros(n_feat)
        coef[active] = prev_coef[active] + gamma * w                     # :847
        Cov = Cov - gamma * corr                                         # :850
        if drop:                                                         # :853-893
            for ii in out:
                _givens_delete(L, n_active, int(ii))
            n_active -= 1
            gone = [active.pop(int(ii)) for ii in out]
            temp = np.asarray(Xy, dtype=np.float64)[gone] - Gram[gone] @ coef
            Cov = np.r_[temp, Cov]
            rest = gone + rest
            n_drops += len(gone)
    if info is not None:
        info.update(n_iter=n_iter, drops=n_drops, end=end, n_active=n_active, alpha=alpha)
    return coef


def lars_pos_code(X, W, alpha=None, info=None):
    """Column-wise restatement of the coder call of src/ontf.py:79-86; returns H as (b x r).

    scikit-learn scales the penalty by the number of features (_dict_learning.py:119) and the
    path's alphas by the number of samples (_least_angle.py:655), both d here, so the path ends
    where the largest correlation of an inactive atom reaches ``alpha`` itself."""
    a = 2 if alpha is None else alpha
    W = np.asarray(W, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    d = W.shape[0]
    Gm, Cm = W.T @ W, W.T @ X
    H = np.zeros((X.shape[1], W.shape[1]))
    for c in range(X.shape[1]):
        rec = {}
        H[c] = lars_pos_path(Gm, Cm[:, c], d, float(a) / d, info=rec)
        if info is not None:
            info.append(rec)
    return H


class OracleOnlineNTF:
    """Same surface as ``Online_NTF`` (src/ontf.py:19-244); aggregates ARE carried (ontf.py:236)."""

    def __init__(self, X, n_components=100, iterations=500, sub_iterations=10, batch_size=20,
                 ini_dict=None, ini_A=None, ini_B=None, history=0, mode=0,
                 learn_joint_dict=False, alpha=None, beta=None, subsample=True, solver="sklearn",
                 cd_sweeps=20000, cd_tol=1e-12):
        self.X = X
        self.n_components = n_components
        self.batch_size = batch_size
        self.iterations = iterations
        self.sub_iterations = sub_iterations
        self.initial_dict = ini_dict
        self.initial_A = ini_A
        self.initial_B = ini_B
        self.history = history
        self.alpha = alpha
        self.beta = beta
        self.mode = mode
        self.learn_joint_dict = learn_joint_dict
        self.code = np.zeros((X.shape[1], n_components))
        self.subsample = subsample
        self.solver = solver
        self.cd_sweeps = cd_sweeps          # solver="cd": sweep cap and tolerance (a negative tolerance
        self.cd_tol = cd_tol                # runs exactly cd_sweeps sweeps)

    def joint_sparse_code_tensor(self, X, W):
        return lasso_nonneg_code(X, W, self.alpha, solver=self.solver, sweeps=self.cd_sweeps, tol=self.cd_tol)

    def update_dict(self, W, A, B):
        return dictionary_pass(W, A, B)

    def step(self, X, A, B, W, t):
        H1 = self.joint_sparse_code_tensor(X, W)                       # (b x r)
        A1, B1 = aggregate_update([A, B], H1, X, t, self.beta, code_is_transposed=True)
        W1 = self.update_dict(W, A, B)                                 # ontf.py:151
        self.history = np.float64(t) + 1
        return H1, A1, B1, W1

    def unfolded(self):
        M = unfold3(self.X, self.mode)
        return M.T if self.learn_joint_dict else M                    # ontf.py:203-208

    def train_dict_single(self):
        Xu = self.unfolded()
        d, n = Xu.shape
        r = self.n_components
        if self.initial_dict is None:
            W = np.random.rand(d, r)
            A, B = np.zeros((r, r)), np.zeros((r, d))
        else:
            W, A, B = self.initial_dict, self.initial_A, self.initial_B
        t0 = self.history
        for i in np.arange(1, self.iterations):
            Xb = Xu
            if self.subsample:
                idx = np.random.randint(n, size=self.batch_size)
                Xb = Xu[:, idx]
            _, A, B, W = self.step(Xb, A, B, W, t0 + i)
        return W, A, B, self.code
