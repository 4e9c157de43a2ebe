"""Drop-in for the reference's ``src/onmf.py``: ``Online_NMF`` and ``update_code_within_radius``.

Same constructor, method names, argument order and return conventions as the reference
(src/onmf.py:20-271); the arithmetic runs in libonmf_hip.so on an MI355X.  Inputs may be NumPy
arrays (results come back as float64 NumPy arrays, like the reference's) or ROCm ``torch`` tensors
(results stay on the GPU as float32 tensors).

Extra keyword-only arguments (not in the reference):
  rng     "numpy" (default): draw the initial dictionary, the subsample indices and every H0 from
          NumPy's global generator in the reference's order (src/onmf.py:190, :212, :246), so a
          seeded run follows the same random stream.  "device": draw them on the GPU.
  seed    base seed of the device generator.
  device  torch device of the GPU to use.

Behaviours of the reference that callers can observe are kept: the dictionary pass inside a step
uses the aggregates from before that step (onmf.py:161); ``train_dict`` hands its initial
aggregates to every step and returns those of the last one (onmf.py:217-226); ``history`` ends at
t+1 (onmf.py:162); ``iterations=N`` runs N-1 steps (onmf.py:206).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .engine import get_engine

__all__ = ["Online_NMF", "update_code_within_radius", "sparse_code_update", "dict_update", "surrogate_loss"]

DEBUG = False


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _out(t, like_torch):
    """Device tensor -> what the caller works with."""
    if t is None:
        return None
    if like_torch:
        return t
    return t.detach().cpu().numpy().astype(np.float64)


def _weight(t, beta):
    """t ** (-beta) with beta = 1 when None (src/onmf.py:150-155); accepts any real t."""
    t = np.float64(t)
    b = 1 if beta is None else beta
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(t ** (-b))


class Online_NMF():

    def __init__(self,
                 X,
                 n_components=100,
                 iterations=500,
                 batch_size=20,
                 ini_dict=None,
                 ini_agg=None,
                 history=0,
                 alpha=None,
                 beta=None,
                 subsample=False,
                 *,
                 rng="numpy",
                 seed=0,
                 device=None):
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
        if rng not in ("numpy", "device"):
            raise ValueError("rng must be 'numpy' or 'device'")
        self.rng = rng
        self.seed = seed
        self._device = device
        self._torch_io = _is_torch(X)
        self._code_dev = None            # device copy of the accumulated code (lazy)
        self._code_host = None
        self._X_dev = None
        self.last_sweeps = None          # sweeps run by the most recent sparse-code solve
        self.last_dist = None            # its per-sweep dist values (NaN where bounds decided)

    # -- code: (r x n) zeros until train_dict accumulates into it (src/onmf.py:49, :221) -------
    @property
    def code(self):
        if self._code_dev is not None:
            return _out(self._code_dev, self._torch_io)
        if self._code_host is None:
            n = self.X.shape[1]
            if self._torch_io:
                self._code_dev = self._engine().zeros(self.n_components, n)
                return self._code_dev
            self._code_host = np.zeros(shape=(self.n_components, n))
        return self._code_host

    @code.setter
    def code(self, value):
        self._code_dev = None
        self._code_host = value

    # -- plumbing -------------------------------------------------------------------------
    def _engine(self):
        return get_engine(self._device)

    def _alpha(self):
        return 0 if self.alpha is None else self.alpha      # src/onmf.py:82-84

    def _draw_h0(self, eng, r, b):
        if self.rng == "numpy":
            return eng.numpy_rand(r, b), None               # src/onmf.py:245-246
        return None, eng.next_seed(self.seed)

    # -- reference API ----------------------------------------------------------------------
    def sparse_code(self, X, W):
        """H (r x n) >= 0 with W H ~ X: ten damped Gauss-Seidel sweeps at most, stopping_diff 0.01
        (src/onmf.py:51-90)."""
        if DEBUG:
            print('sparse_code', 'X.shape:', X.shape, 'W.shape:', W.shape)
        eng = self._engine()
        like_torch = _is_torch(X)
        Xd, Wd = eng.dev(X), eng.dev(W)
        H0, seed = self._draw_h0(eng, Wd.shape[1], Xd.shape[1])
        H, sweeps, dist = eng.sparse_code(Xd, Wd, H0=H0, alpha=self._alpha(), sub_iter=10,
                                          stopping_diff=0.01, seed=seed)
        self.last_sweeps, self.last_dist = sweeps, dist
        return _out(H, like_torch)

    def update_dict(self, W, A, B):
        """One block-coordinate pass over the atoms (src/onmf.py:92-116)."""
        eng = self._engine()
        W1 = eng.update_dict(eng.dev(W), eng.dev(A), eng.dev(B))
        return _out(W1, _is_torch(W))

    def step(self, X, aggregates, W, t):
        """One online-NMF iteration (src/onmf.py:119-167): returns H1, [A1, B1(, C1)], W1."""
        eng = self._engine()
        like_torch = _is_torch(X)
        Xd, Wd = eng.dev(X), eng.dev(W)
        Ad, Bd = eng.dev(aggregates[0]), eng.dev(aggregates[1])
        Cd = eng.dev(aggregates[2]) if len(aggregates) == 3 else None
        H1, agg1, W1 = self._step_dev(eng, Xd, Ad, Bd, Cd, Wd, t, want_code=True)
        return _out(H1, like_torch), [_out(a, like_torch) for a in agg1], _out(W1, like_torch)

    def _step_dev(self, eng, Xd, Ad, Bd, Cd, Wd, t, want_code):
        r, b = Wd.shape[1], Xd.shape[1]
        H0, seed = self._draw_h0(eng, r, b)
        w = _weight(t, self.beta)
        H1, A1, B1, C1, W1, sweeps, dist = eng.step(Xd, Ad, Bd, Cd, Wd, w, alpha=self._alpha(),
                                                    H0=H0, seed=seed, want_code=want_code)
        self.last_sweeps, self.last_dist = sweeps, dist
        self.history = np.float64(t) + 1                    # src/onmf.py:162
        agg1 = [A1, B1] + ([C1] if Cd is not None else [])
        return H1, agg1, W1

    def train_dict(self, full_code=False):
        """Learn W from the columns of self.X (src/onmf.py:169-226): returns W, aggregates, code."""
        eng = self._engine()
        like_torch = self._torch_io
        if self._X_dev is None:
            self._X_dev = eng.dev(self.X)
        Xd = self._X_dev
        d, n = Xd.shape
        r = self.n_components
        returned = self.initial_agg

        if self.initial_dict is None:
            if self.rng == "numpy":
                Wd = eng.numpy_rand(d, r)                   # src/onmf.py:190
            else:
                Wd = eng.rand(d, r, eng.next_seed(self.seed))
            Ad, Bd = eng.zeros(r, r), eng.zeros(r, d)
            Cd = eng.zeros(d, d) if full_code else None
            returned = None
        else:
            Wd = eng.dev(self.initial_dict)
            Ad, Bd = eng.dev(self.initial_agg[0]), eng.dev(self.initial_agg[1])
            Cd = eng.dev(self.initial_agg[2]) if full_code else None
        t0 = self.history
        W_out = None if self.initial_dict is None else self.initial_dict
        last = None
        # the device copy of the running code starts from whatever self.code holds
        if self._code_dev is not None:
            code_dev = self._code_dev
        elif self._code_host is None:
            code_dev = eng.zeros(r, n)
        else:
            code_dev = eng.dev(self._code_host)

        for i in np.arange(1, self.iterations):
            idx = None
            Xb = Xd
            if self.subsample:
                if self.rng == "numpy":
                    idx = np.random.randint(n, size=self.batch_size)   # src/onmf.py:212
                else:
                    idx = eng.torch.randint(n, (self.batch_size,), device=eng.device)
                Xb = eng.gather_columns(Xd, idx)
            # every step starts from the aggregates this call was given (src/onmf.py:217-219)
            H, last, Wd = self._step_dev(eng, Xb, Ad, Bd, Cd, Wd, t0 + i, want_code=True)
            eng.code_accumulate(code_dev, idx, H)                       # src/onmf.py:221
            W_out = None

        self._code_dev, self._code_host = code_dev, None
        if last is not None:
            returned = [_out(a, like_torch) for a in last]
        elif returned is None:
            returned = [_out(a, like_torch) for a in ([Ad, Bd] + ([Cd] if full_code else []))]
        W_ret = W_out if W_out is not None else _out(Wd, like_torch)
        return W_ret, returned, self.code


####################################
# custom sparsecoder

def update_code_within_radius(X, W, H0=None, r=None, alpha=0, sub_iter=10, stopping_diff=0.1,
                              *, device=None, return_info=False):
    """argmin_{H>=0} |X - W H| + alpha |H| by row-wise projected Gauss-Seidel sweeps
    (src/onmf.py:233-271).  ``H0=None`` draws U[0,1) from NumPy's global generator like the
    reference.  ``r`` (the radius): the reference pulls the iterate back toward H0 after every row
    and then rebinds H0 to the iterate (src/onmf.py:260-263), which leaves one effect - the first row's
    first move is scaled by r / max(r, its norm); that is what happens here.  ``r`` must be positive
    (the reference divides by zero for r = 0 as soon as W has a second atom).
    ``return_info=True`` also returns (sweeps, dist_trace)."""
    if r is not None and not r > 0:
        raise ValueError("update_code_within_radius: the radius must be positive (or None)")
    eng = get_engine(device)
    like_torch = _is_torch(X)
    Xd, Wd = eng.dev(X), eng.dev(W)
    H0 = eng.numpy_rand(Wd.shape[1], Xd.shape[1]) if H0 is None else eng.dev(H0)       # src/onmf.py:245-246
    H, sweeps, dist = eng.sparse_code(Xd, Wd, H0=H0, alpha=alpha, sub_iter=sub_iter,
                                      stopping_diff=stopping_diff, radius=r)
    out = _out(H, like_torch)
    return (out, sweeps, dist) if return_info else out


# names used in the task statement for the two halves of the step
def sparse_code_update(X, W, H0=None, alpha=0, sub_iter=10, stopping_diff=0.01, **kw):
    return update_code_within_radius(X, W, H0=H0, r=None, alpha=alpha, sub_iter=sub_iter,
                                     stopping_diff=stopping_diff, **kw)


def dict_update(W, A, B, device=None):
    eng = get_engine(device)
    return _out(eng.update_dict(eng.dev(W), eng.dev(A), eng.dev(B)), _is_torch(W))


def surrogate_loss(W, A, B, C=None, device=None):
    """tr(W A W^T) - 2 tr(W B) + tr(C): the convergence monitor that the reference's drivers compute
    from the three aggregates after each round (ising_reconstruction.py:133, :164)."""
    eng = get_engine(device)
    return eng.surrogate_loss(eng.dev(W), eng.dev(A), eng.dev(B), None if C is None else eng.dev(C))
