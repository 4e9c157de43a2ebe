"""Drop-in for the reference's ``src/ontf.py``: ``Online_NTF`` (src/ontf.py:19-244).

The tensor is unfolded along ``mode`` (optionally transposed: ``learn_joint_dict``), then the same
online step runs as in ``onmf``; unlike ``Online_NMF.train_dict`` the aggregates ARE carried between
the steps of one call (src/ontf.py:236).  The code matrix of this class is (b x r), the transpose of
onmf's (src/ontf.py:86, :147-148).

The reference hands the sparse-code solve to scikit-learn (``SparseCoder``, lasso_lars, positive
codes, alpha = ``self.alpha`` or 2; src/ontf.py:79-86), i.e. per column the positive LARS-lasso path of
    min_{h >= 0}  0.5 * ||x - W h||^2 + alpha * ||h||_1 .
``coder="lars"`` (the default) follows that path on the GPU event by event, one column per wavefront
(``onmf_sparse_code_lars``), including the way scikit-learn's routine ends it - short of the optimum
once no atom with a positive correlation is left outside the active set - because the reference's
aggregates, and so its dictionary, are built from exactly those codes.
``coder="cd"`` solves the convex problem itself by exact cyclic coordinate descent from h = 0: every
column iterates until ITS relative change per sweep, ||h_new - h_old|| / ||h_old||, falls under
``cd_tol`` (at most ``cd_max_sweeps`` sweeps), all in one kernel launch without host round trips.  It
reaches the optimum; it does not reproduce the reference's codes.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .engine import get_engine
from .onmf import _is_torch, _out, _weight

__all__ = ["Online_NTF"]

DEBUG = False


class Online_NTF():

    def __init__(self,
                 X, n_components=100,
                 iterations=500,
                 sub_iterations=10,
                 batch_size=20,
                 ini_dict=None,
                 ini_A=None,
                 ini_B=None,
                 history=0,
                 mode=0,
                 learn_joint_dict=False,
                 alpha=None,
                 beta=None,
                 subsample=True,
                 *,
                 rng="numpy",
                 seed=0,
                 device=None,
                 coder="lars",
                 lars_max_iter=1000,
                 cd_tol=1e-6,
                 cd_max_sweeps=4000):
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
        self.code = np.zeros(shape=(X.shape[1], n_components))
        self.subsample = subsample
        self.rng = rng
        self.seed = seed
        self._device = device
        self._torch_io = _is_torch(X)
        if coder not in ("lars", "cd"):
            raise ValueError("coder must be 'lars' or 'cd'")
        self.coder = coder
        self.lars_max_iter = lars_max_iter          # SparseCoder's transform_max_iter (its default)
        self.cd_tol = cd_tol
        self.cd_max_sweeps = cd_max_sweeps
        self._sweeps_each = None

    def _engine(self):
        return get_engine(self._device)

    def _code_dev(self, eng, Xd, Wd):
        """Device solve; returns H as (r x b)."""
        a = 2 if self.alpha is None else self.alpha          # src/ontf.py:79-84
        if self.coder == "lars":
            if Wd.shape[1] > 256:
                raise ValueError("coder='lars' supports n_components <= 256; use coder='cd'")
            return eng.sparse_code_lars(Xd, Wd, alpha=a, max_iter=self.lars_max_iter)
        H0 = eng.zeros(Wd.shape[1], Xd.shape[1])
        H, sweeps = eng.sparse_code_each(Xd, Wd, H0=H0, alpha=a, sub_iter=self.cd_max_sweeps,
                                         stopping_diff=self.cd_tol, mode=_lib.CODER_EXACT_CD, want_sweeps=True)
        self._sweeps_each = sweeps                           # device int32[b]; read on demand
        return H

    @property
    def last_sweeps(self):
        """Sweeps run by the slowest column of the most recent solve (None before the first)."""
        s = getattr(self, "_sweeps_each", None)
        return None if s is None else int(s.max())

    def joint_sparse_code_tensor(self, X, W):
        """Code H (samples x topics) with W H^T ~ X (src/ontf.py:59-89)."""
        eng = self._engine()
        H = self._code_dev(eng, eng.dev(X), eng.dev(W))
        return _out(H.t().contiguous(), _is_torch(X))

    def update_dict(self, W, A, B):
        """Same block-coordinate pass as onmf (src/ontf.py:91-115)."""
        eng = self._engine()
        return _out(eng.update_dict(eng.dev(W), eng.dev(A), eng.dev(B)), _is_torch(W))

    def _step_dev(self, eng, Xd, Ad, Bd, Wd, t):
        w = _weight(t, self.beta)
        if self.coder == "lars" and Wd.shape[1] <= 256:
            a = 2 if self.alpha is None else self.alpha      # src/ontf.py:79-84
            H, A1, B1, W1 = eng.ntf_step(Xd, Ad, Bd, Wd, w, alpha=a, max_iter=self.lars_max_iter)
            self.history = np.float64(t) + 1
            return H, A1, B1, W1
        H = self._code_dev(eng, Xd, Wd)
        A1, B1, _ = eng.aggregate(Xd, H, w, Ad, Bd)          # src/ontf.py:147-148
        W1 = eng.update_dict(Wd, Ad, Bd)                      # old aggregates, src/ontf.py:151
        self.history = np.float64(t) + 1
        return H, A1, B1, W1

    def step(self, X, A, B, W, t):
        """One iteration (src/ontf.py:117-154): returns H1 (b x r), A1, B1, W1."""
        eng = self._engine()
        lt = _is_torch(X)
        H, A1, B1, W1 = self._step_dev(eng, eng.dev(X), eng.dev(A), eng.dev(B), eng.dev(W), t)
        return _out(H.t().contiguous(), lt), _out(A1, lt), _out(B1, lt), _out(W1, lt)

    def unfold(self):
        """tl_unfold(X, mode), transposed when learn_joint_dict (src/ontf.py:203-208); device tensor."""
        eng = self._engine()
        T = eng.dev(self.X)
        if T.dim() != 3:
            raise ValueError("Online_NTF expects a 3-way tensor, got shape %s" % (tuple(T.shape),))
        return eng.unfold3(T, self.mode, transpose=bool(self.learn_joint_dict))

    def train_dict_single(self):
        """Learn W along one mode (src/ontf.py:156-244): returns W, A, B, code."""
        eng = self._engine()
        lt = self._torch_io
        Xu = self.unfold()
        d, n = Xu.shape
        r = self.n_components
        if self.initial_dict is None:
            if self.rng == "numpy":
                Wd = eng.numpy_rand(d, r)                    # src/ontf.py:213
            else:
                Wd = eng.rand(d, r, eng.next_seed(self.seed))
            if DEBUG:
                print('W.shape', tuple(Wd.shape))
            Ad, Bd = eng.zeros(r, r), eng.zeros(r, d)
            ran = True
        else:
            Wd, Ad, Bd = eng.dev(self.initial_dict), eng.dev(self.initial_A), eng.dev(self.initial_B)
            ran = False
        t0 = self.history
        for i in np.arange(1, self.iterations):
            Xb = Xu
            if self.subsample:
                if self.rng == "numpy":
                    idx = np.random.randint(n, size=self.batch_size)   # src/ontf.py:230
                else:
                    idx = eng.torch.randint(n, (self.batch_size,), device=eng.device)
                Xb = eng.gather_columns(Xu, idx)
            _, Ad, Bd, Wd = self._step_dev(eng, Xb, Ad, Bd, Wd, t0 + i)
            ran = True
        if not ran:
            return self.initial_dict, self.initial_A, self.initial_B, self.code
        return _out(Wd, lt), _out(Ad, lt), _out(Bd, lt), self.code
