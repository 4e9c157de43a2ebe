"""Host-side driver of the HIP library on torch-ROCm tensors (device memory + streams only).

One ``Engine`` per GPU owns a library context (scratch buffers).  Every method takes and returns
float32 ``torch`` tensors on that GPU; the facades in ``onmf.py`` / ``ontf.py`` convert to and from
whatever the caller handed in.  All arithmetic happens in libonmf_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings

import numpy as np

from . import _lib

_engines = {}


def get_engine(device=None):
    """Engine for a GPU (default: torch's current device)."""
    import torch
    if not torch.cuda.is_available():
        raise _lib.OnmfLibraryError("onmf_ontf_ndl_amd needs a ROCm GPU: torch.cuda.is_available() "
                                    "is False and there is no CPU path")
    if device is None:
        index = torch.cuda.current_device()
    else:
        dev = torch.device(device)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
    eng = _engines.get(index)
    if eng is None:
        eng = _engines[index] = Engine(index)
    return eng


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    def __init__(self, index):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.index = int(index)
        self.device = torch.device("cuda", self.index)
        ctx = C.c_void_p()
        rc = self.lib.onmf_create(self.index, C.byref(ctx))
        _lib.check(rc, None, "onmf_create")
        self.ctx = ctx
        self._seed_counter = 0
        self._timeouts_seen = 0
        self._mt_table_set = False

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.onmf_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def dev(self, a, dtype=None):
        """numpy array / torch tensor -> contiguous float32 tensor on this GPU (no copy if already so)."""
        torch = self.torch
        dtype = dtype or torch.float32
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        arr = np.ascontiguousarray(a)
        return torch.from_numpy(arr).to(device=self.device, dtype=dtype).contiguous()

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float32, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float32, device=self.device)

    def next_seed(self, base=0):
        self._seed_counter += 1
        return (int(base) * 0x9E3779B97F4A7C15 + self._seed_counter) & 0xFFFFFFFFFFFFFFFF

    def set_option(self, key, value):
        _lib.check(self.lib.onmf_set_option(self.ctx, int(key), int(value)), self.ctx, "onmf_set_option")

    def get_profile(self):
        ms, n = C.c_double(0.0), C.c_int(0)
        _lib.check(self.lib.onmf_get_profile(self.ctx, C.byref(ms), C.byref(n)), self.ctx,
                   "onmf_get_profile")
        return ms.value, n.value

    def counters(self):
        """Recovery events since the context was created: one-launch solves and dictionary passes whose
        workgroups gave up waiting for each other, and solves that ran one launch per sweep since."""
        out = {}
        for name, key in (("sweep_timeouts", _lib.CNT_SWEEP_TIMEOUTS), ("dict_timeouts", _lib.CNT_DICT_TIMEOUTS),
                          ("fallback_solves", _lib.CNT_FALLBACK_SOLVES),
                          ("device_resolves", _lib.CNT_DEVICE_RESOLVES), ("lars_recoded", _lib.CNT_LARS_RECODED), ("lars_slots_used", _lib.CNT_LARS_SLOTS_USED), ("sweeps_total", _lib.CNT_SWEEPS_TOTAL),
                          ("deferred_redone", _lib.CNT_DEFERRED_REDONE), ("lag3_redone", _lib.CNT_LAG3_REDONE)):
            v = C.c_longlong(0)
            rc = self.lib.onmf_get_counter(self.ctx, key, C.byref(v))
            if rc == -1 and os.environ.get("ONMF_HIP_LIB"):
                continue        # a diagnostic build (chosen through ONMF_HIP_LIB) older than this counter
            _lib.check(rc, self.ctx, "onmf_get_counter")
            out[name] = int(v.value)
        return out

    def _warn_on_timeouts(self, now=False):
        """One warning the first time a device-side wait gave up (results stay correct; later solves are
        slower for a while).  Typical cause: another process's kernels on the same GPU."""
        if self._timeouts_seen:
            return
        # (looked at every 16th call: the warning can wait, a short step cannot)
        self._calls = getattr(self, "_calls", 0) + 1
        if (self._calls & 15) and not now:
            return
        c = self.counters()
        n = c["sweep_timeouts"] + c["dict_timeouts"]
        if n:
            self._timeouts_seen = n
            warnings.warn("libonmf_hip: workgroups of a one-launch kernel gave up waiting for each other on GPU %d "
                          "(%s); the work was redone by the slower launch-per-step path and the fast path rests "
                          "for the next solves.  Is another process using this GPU?" % (self.index, c),
                          RuntimeWarning, stacklevel=3)

    def _check_index(self, idx, n, what):
        """NumPy raises IndexError for an index outside [-n, n); so does this (one small reduction)."""
        if idx.numel():
            lo, hi = int(idx.min()), int(idx.max())
            if lo < -n or hi >= n:
                raise IndexError("%s: index %d is out of bounds for axis 1 with size %d"
                                 % (what, lo if lo < -n else hi, n))

    # ------------------------------------------------------------------ the path
    def sparse_code(self, X, W, H0=None, alpha=0.0, sub_iter=10, stopping_diff=0.01, seed=None,
                    mode=_lib.CODER_DAMPED, radius=None):
        """H (r x b), sweeps run, per-sweep dist (NaN where bounds decided) - onmf.py:233-271.
        ``radius``: the reference's r argument (onmf.py:260-263)."""
        d, b = X.shape
        r = W.shape[1]
        if W.shape[0] != d:
            raise ValueError("W has %d rows, X has %d" % (W.shape[0], d))
        if H0 is not None and tuple(H0.shape) != (r, b):
            raise ValueError("H0 must be (%d, %d)" % (r, b))
        H = self.empty(r, b)
        sweeps = C.c_int(0)
        trace = (C.c_float * max(1, int(sub_iter)))()
        if seed is None:
            seed = self.next_seed()
        if radius is not None:
            if mode != _lib.CODER_DAMPED:
                raise ValueError("sparse_code: a radius goes with the damped coder")
            rc = self.lib.onmf_sparse_code_within_radius(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(H0),
                                                         C.c_uint64(seed), float(radius), float(alpha),
                                                         int(sub_iter), float(stopping_diff), _ptr(H),
                                                         C.byref(sweeps), trace, self.stream())
            _lib.check(rc, self.ctx, "onmf_sparse_code_within_radius")
            return H, sweeps.value, np.array(trace[:sweeps.value], dtype=np.float64)
        rc = self.lib.onmf_sparse_code(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(H0),
                                       C.c_uint64(seed), float(alpha), int(sub_iter),
                                       float(stopping_diff), int(mode), _ptr(H), C.byref(sweeps),
                                       trace, self.stream())
        _lib.check(rc, self.ctx, "onmf_sparse_code")
        self._warn_on_timeouts()
        return H, sweeps.value, np.array(trace[:sweeps.value], dtype=np.float64)

    def aggregate(self, X, H, w, A, B, Cagg=None, want_c=False):
        d, b = X.shape
        r = H.shape[0]
        A1, B1 = self.empty(r, r), self.empty(r, d)
        C1 = self.empty(d, d) if want_c else None
        rc = self.lib.onmf_aggregate(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(H), float(w),
                                     _ptr(A), _ptr(B), _ptr(Cagg), _ptr(A1), _ptr(B1), _ptr(C1),
                                     self.stream())
        _lib.check(rc, self.ctx, "onmf_aggregate")
        return A1, B1, C1

    def update_dict(self, W, A, B, out=None):
        """``out``: where to put W1 (``out is W`` updates the dictionary in place)."""
        d, r = W.shape
        if tuple(A.shape) != (r, r) or tuple(B.shape) != (r, d):
            raise ValueError("update_dict: A must be (r,r) and B (r,d)")
        W1 = self.empty(d, r) if out is None else out
        if tuple(W1.shape) != (d, r) or not W1.is_contiguous() or W1.device != self.device or \
                W1.dtype != self.torch.float32:
            raise ValueError("update_dict: out must be a contiguous float32 (d,r) tensor on %s" % self.device)
        rc = self.lib.onmf_update_dict(self.ctx, d, r, _ptr(W), _ptr(A), _ptr(B), _ptr(W1),
                                       self.stream())
        _lib.check(rc, self.ctx, "onmf_update_dict")
        if d > 1024:
            self._warn_on_timeouts()
        return W1

    def update_dict_static(self, W, A, B, out=None):
        """``update_dict`` through the entry point that allocates no device memory (d <= 1024, r <= 256;
        its workspace is a static array of the library).  ``out``: where to put W1 (the library itself
        allocates nothing; this wrapper allocates ``out`` when none is given)."""
        d, r = W.shape
        if tuple(A.shape) != (r, r) or tuple(B.shape) != (r, d):
            raise ValueError("update_dict: A must be (r,r) and B (r,d)")
        W1 = self.empty(d, r) if out is None else out
        if tuple(W1.shape) != (d, r) or not W1.is_contiguous() or W1.device != self.device or \
                W1.dtype != self.torch.float32:
            raise ValueError("update_dict_static: out must be a contiguous float32 (d,r) tensor on %s" % self.device)
        rc = self.lib.onmf_update_dict_static(self.ctx, d, r, _ptr(W), _ptr(A), _ptr(B), _ptr(W1), self.stream())
        _lib.check(rc, self.ctx, "onmf_update_dict_static")
        return W1

    def settle(self):
        """Have the library look at the verdicts of a step that was not waited for (``step(..., defer=True)``)."""
        _lib.check(self.lib.onmf_settle(self.ctx), self.ctx, "onmf_settle")

    def step(self, X, A, B, Cagg, W, w, alpha=0.0, H0=None, seed=None, sub_iter=10,
             stopping_diff=0.01, want_code=True, out=None, defer=False):
        """One minibatch step (onmf.py:119-167). Returns H1, A1, B1, C1, W1, sweeps, dist.
        ``out`` = (A1, B1, W1): tensors to write the new state into (e.g. views of a packed buffer);
        they must not be the input tensors if the step is to keep its three-stream schedule.
        ``defer``: do not wait for the stopping test's verdicts (ONMF_OPT_DEFER_VERDICT must be set): sweeps and
        dist come back as None; inputs must stay untouched and H1, A1, B1, C1 are only to be handed to the next
        call on this engine (or read after ``settle()``)."""
        d, b = X.shape
        r = W.shape[1]
        if out is None:
            A1, B1, W1 = self.empty(r, r), self.empty(r, d), self.empty(d, r)
        else:
            A1, B1, W1 = out
            if tuple(A1.shape) != (r, r) or tuple(B1.shape) != (r, d) or tuple(W1.shape) != (d, r) or \
                    not (A1.is_contiguous() and B1.is_contiguous() and W1.is_contiguous()):
                raise ValueError("step: out must be contiguous (r,r), (r,d), (d,r) tensors")
        C1 = self.empty(d, d) if Cagg is not None else None
        H = self.empty(r, b) if want_code else None
        sweeps = C.c_int(0)
        trace = (C.c_float * max(1, int(sub_iter)))()
        if seed is None:
            seed = self.next_seed()
        if defer:
            # the tensors of this call must outlive it: the library may run the solve again at the next call
            self._deferred_keep = (X, A, B, Cagg, W, H0, H, A1, B1, C1, W1)
            rc = self.lib.onmf_step(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(A), _ptr(B),
                                    _ptr(Cagg), _ptr(H0), C.c_uint64(seed), float(w), float(alpha),
                                    int(sub_iter), float(stopping_diff), _ptr(H), _ptr(A1), _ptr(B1),
                                    _ptr(C1), _ptr(W1), None, None, self.stream())
            _lib.check(rc, self.ctx, "onmf_step")
            return H, A1, B1, C1, W1, None, None
        rc = self.lib.onmf_step(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(A), _ptr(B),
                                _ptr(Cagg), _ptr(H0), C.c_uint64(seed), float(w), float(alpha),
                                int(sub_iter), float(stopping_diff), _ptr(H), _ptr(A1), _ptr(B1),
                                _ptr(C1), _ptr(W1), C.byref(sweeps), trace, self.stream())
        _lib.check(rc, self.ctx, "onmf_step")
        self._warn_on_timeouts()
        return H, A1, B1, C1, W1, sweeps.value, np.array(trace[:sweeps.value], dtype=np.float64)

    def partial_floats(self, d, r):
        return int(self.lib.onmf_partial_floats(int(d), int(r)))

    def step_partial(self, X, A, B, W, P, W_out=None, alpha=0.0, H0=None, seed=None, sub_iter=10,
                     stopping_diff=0.0, want_code=True, state_ready_event=None):
        """The step on this rank's columns of a split minibatch: code, BARE sums [H H^T | H X^T] into the
        packed vector ``P`` (``partial_floats`` long), dictionary pass on (A, B) - see onmf_step_partial in
        include/onmf_hip.h.  ``state_ready_event``: a torch.cuda.Event after which A, B and P may be
        touched.  Returns H, W1, sweeps."""
        d, b = X.shape
        r = W.shape[1]
        if P.numel() < self.partial_floats(d, r) or not P.is_contiguous():
            raise ValueError("step_partial: P must be a contiguous vector of %d floats" % self.partial_floats(d, r))
        W1 = self.empty(d, r) if W_out is None else W_out
        H = self.empty(r, b) if want_code else None
        sweeps = C.c_int(0)
        trace = (C.c_float * max(1, int(sub_iter)))()
        if seed is None:
            seed = self.next_seed()
        ev = C.c_void_p(state_ready_event.cuda_event) if state_ready_event is not None else C.c_void_p(0)
        rc = self.lib.onmf_step_partial(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(A), _ptr(B),
                                        _ptr(H0), C.c_uint64(seed), float(alpha), int(sub_iter),
                                        float(stopping_diff), _ptr(H), _ptr(P), _ptr(W1), C.byref(sweeps), trace,
                                        ev, self.stream())
        _lib.check(rc, self.ctx, "onmf_step_partial")
        self._warn_on_timeouts()
        return H, W1, sweeps.value

    def step_apply(self, P, w, A, B, A_out=None, B_out=None):
        """A1 = (1-w) A + w P[H H^T], B1 = (1-w) B + w P[H X^T] (src/onmf.py:155-156) on the current stream."""
        r, d = B.shape
        A1 = self.empty(r, r) if A_out is None else A_out
        B1 = self.empty(r, d) if B_out is None else B_out
        rc = self.lib.onmf_step_apply(self.ctx, d, r, _ptr(P), float(w), _ptr(A), _ptr(B), _ptr(A1), _ptr(B1),
                                      self.stream())
        _lib.check(rc, self.ctx, "onmf_step_apply")
        return A1, B1

    # ------------------------------------------------------------------ around the path
    def gather_columns(self, X, idx):
        d, n = X.shape
        idx = self.dev(idx, self.torch.int64)
        self._check_index(idx, n, "gather_columns")
        b = idx.numel()
        Xb = self.empty(d, b)
        rc = self.lib.onmf_gather_columns(self.ctx, d, n, _ptr(X), X.stride(0), _ptr(idx), b,
                                          _ptr(Xb), self.stream())
        _lib.check(rc, self.ctx, "onmf_gather_columns")
        return Xb

    def code_accumulate(self, code, idx, H):
        r, n = code.shape
        if idx is None:
            b, ip = n, C.c_void_p(0)
        else:
            idx = self.dev(idx, self.torch.int64)
            self._check_index(idx, n, "code_accumulate")
            b, ip = idx.numel(), _ptr(idx)
        rc = self.lib.onmf_code_accumulate(self.ctx, r, n, _ptr(code), ip, b, _ptr(H), self.stream())
        _lib.check(rc, self.ctx, "onmf_code_accumulate")
        return code

    def unfold3(self, T, mode, transpose=False):
        n0, n1, n2 = T.shape
        nm = T.shape[mode]
        rest = T.numel() // nm
        out = self.empty(rest, nm) if transpose else self.empty(nm, rest)
        rc = self.lib.onmf_unfold3(self.ctx, n0, n1, n2, _ptr(T), int(mode), int(bool(transpose)),
                                   _ptr(out), self.stream())
        _lib.check(rc, self.ctx, "onmf_unfold3")
        return out

    # ------------------------------------------------------------------ images either side of the path
    def sparse_code_each(self, X, W, H0=None, alpha=0.0, sub_iter=10, stopping_diff=0.01, seed=None,
                         mode=_lib.CODER_DAMPED, want_sweeps=False):
        """One independent solve per column, each with its own stopping test
        (image_reconstruction.py:384 calls onmf.py:233-271 once per patch).  No host round trip."""
        d, b = X.shape
        r = W.shape[1]
        if W.shape[0] != d:
            raise ValueError("W has %d rows, X has %d" % (W.shape[0], d))
        if H0 is not None and tuple(H0.shape) != (r, b):
            raise ValueError("H0 must be (%d, %d)" % (r, b))
        H = self.empty(r, b)
        sweeps = self.torch.empty(b, dtype=self.torch.int32, device=self.device) if want_sweeps else None
        if seed is None:
            seed = self.next_seed()
        rc = self.lib.onmf_sparse_code_each(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(H0),
                                            C.c_uint64(seed), float(alpha), int(sub_iter),
                                            float(stopping_diff), int(mode), _ptr(H), _ptr(sweeps),
                                            self.stream())
        _lib.check(rc, self.ctx, "onmf_sparse_code_each")
        return (H, sweeps) if want_sweeps else H

    def sparse_code_lars(self, X, W, alpha=2.0, max_iter=1000, want_info=False):
        """scikit-learn's positive LARS-lasso coder as src/ontf.py:79-86 calls it, one column per wavefront;
        H (r x b).  ``want_info``: also an int32 (b, 4) tensor {events, end, active atoms, atoms dropped}."""
        d, b = X.shape
        r = W.shape[1]
        if W.shape[0] != d:
            raise ValueError("W has %d rows, X has %d" % (W.shape[0], d))
        H = self.empty(r, b)
        info = self.torch.empty(b, 4, dtype=self.torch.int32, device=self.device) if want_info else None
        rc = self.lib.onmf_sparse_code_lars(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), float(alpha),
                                            int(max_iter), _ptr(H), _ptr(info), self.stream())
        _lib.check(rc, self.ctx, "onmf_sparse_code_lars")
        return (H, info) if want_info else H

    def ntf_step(self, X, A, B, W, w, alpha=2.0, max_iter=1000, want_code=True):
        """Online_NTF.step (src/ontf.py:117-154) with the LARS coder: H (r x b), A1, B1, W1; the dictionary pass
        runs beside the coder on the library's second stream."""
        d, b = X.shape
        r = W.shape[1]
        if W.shape[0] != d or tuple(A.shape) != (r, r) or tuple(B.shape) != (r, d):
            raise ValueError("ntf_step: shapes of X (d,b), W (d,r), A (r,r), B (r,d) do not agree")
        H = self.empty(r, b) if want_code else None
        A1, B1, W1 = self.empty(r, r), self.empty(r, d), self.empty(d, r)
        rc = self.lib.onmf_ntf_step(self.ctx, d, r, b, _ptr(X), X.stride(0), _ptr(W), _ptr(A), _ptr(B), float(w),
                                    float(alpha), int(max_iter), _ptr(H), _ptr(A1), _ptr(B1), _ptr(W1), None,
                                    self.stream())
        _lib.check(rc, self.ctx, "onmf_ntf_step")
        if d > 1024:
            self._warn_on_timeouts()
        return H, A1, B1, W1

    def patch_product(self, W, H):
        """W @ H (image_reconstruction.py:352, :385)."""
        d, r = W.shape
        b = H.shape[1]
        R = self.empty(d, b)
        rc = self.lib.onmf_patch_product(self.ctx, d, r, b, _ptr(W), _ptr(H), _ptr(R), b, self.stream())
        _lib.check(rc, self.ctx, "onmf_patch_product")
        return R

    def extract_patches(self, img, k, corners=None, grid=None, stride=1):
        """Patch matrix (k*k*ch x b) of an (h, w[, ch]) image: ``corners`` (b x 2 top-left corners,
        image_reconstruction.py:173-206) or ``grid`` = (rows, cols) windows ``stride`` apart."""
        h, w = img.shape[0], img.shape[1]
        ch = img.shape[2] if img.dim() == 3 else 1
        if corners is not None:
            corners = self.dev(corners, self.torch.int32)
            if corners.dim() != 2 or corners.shape[1] != 2:
                raise ValueError("corners must be (b, 2)")
            if corners.numel() and (int(corners.min()) < 0 or int(corners[:, 0].max()) > h - k or
                                    int(corners[:, 1].max()) > w - k):
                raise ValueError("a patch corner lies outside the image")
            b, gr, gc = corners.shape[0], 0, 0
        else:
            gr, gc = grid
            b = gr * gc
        X = self.empty(k * k * ch, b)
        rc = self.lib.onmf_extract_patches(self.ctx, _ptr(img), h, w, ch, int(k), _ptr(corners), b, gr, gc,
                                           int(stride), _ptr(X), b, self.stream())
        _lib.check(rc, self.ctx, "onmf_extract_patches")
        return X

    def fold_patches(self, R, shape, k, grid, stride=1):
        """Overlap-average of patch values (d x b on ``grid``) into an image of ``shape``
        (image_reconstruction.py:354, :388-392).  Returns (image, overlap count)."""
        h, w = int(shape[0]), int(shape[1])
        ch = int(shape[2]) if len(shape) == 3 else 1
        gr, gc = grid
        if R.shape[0] != k * k * ch or R.shape[1] != gr * gc:
            raise ValueError("R must be (%d, %d)" % (k * k * ch, gr * gc))
        out = self.empty(*shape)
        count = self.empty(h, w)
        rc = self.lib.onmf_fold_patches(self.ctx, _ptr(R), R.stride(0), h, w, ch, int(k), gr, gc, int(stride),
                                        _ptr(out), _ptr(count), self.stream())
        _lib.check(rc, self.ctx, "onmf_fold_patches")
        return out, count

    def surrogate_loss(self, W, A, B, Cagg=None):
        """tr(W A W^T) - 2 tr(W B) + tr(C) (ising_reconstruction.py:133), a Python float."""
        d, r = W.shape
        if tuple(A.shape) != (r, r) or tuple(B.shape) != (r, d) or (Cagg is not None and tuple(Cagg.shape) != (d, d)):
            raise ValueError("surrogate_loss: A must be (r,r), B (r,d), C (d,d)")
        loss = C.c_double(0.0)
        rc = self.lib.onmf_surrogate_loss(self.ctx, d, r, _ptr(W), _ptr(A), _ptr(B), _ptr(Cagg), C.byref(loss),
                                          self.stream())
        _lib.check(rc, self.ctx, "onmf_surrogate_loss")
        return loss.value

    # ------------------------------------------------------------------ network patches
    def ndl_graph(self, row_ptr, nbr):
        """CSR graph (sorted rows) on this GPU, as the chain kernels take it."""
        i32 = self.torch.int32
        return {"n": int(len(row_ptr) - 1), "row_ptr": self.dev(row_ptr, i32), "nbr": self.dev(nbr, i32)}

    def ndl_init_embeddings(self, graph, motif, n_chains, pivots=None, seed=0):
        """tree_sample from one pivot per chain (network_reconstruction_nx.py:108-134, :351-352)."""
        i32 = self.torch.int32
        motif = self.dev(motif, i32)
        k = motif.shape[0]
        piv = None if pivots is None else self.dev(pivots, i32)
        emb = self.torch.zeros(n_chains, k, dtype=i32, device=self.device)
        rc = self.lib.onmf_ndl_init_embeddings(self.ctx, graph["n"], _ptr(graph["row_ptr"]), _ptr(graph["nbr"]), k,
                                               _ptr(motif), int(n_chains), _ptr(piv), C.c_uint64(seed), _ptr(emb),
                                               self.stream())
        _lib.check(rc, self.ctx, "onmf_ndl_init_embeddings")
        return emb

    def ndl_sample_patches(self, graph, motif, emb, steps, glauber=True, seed=0, step_offset=0, want_trace=False):
        """Advance every chain ``steps`` times; returns X (k*k x n_chains*steps) and updates ``emb`` in place
        (network_reconstruction_nx.py:315-329).  ``want_trace``: also the embedding behind every patch."""
        i32 = self.torch.int32
        motif = self.dev(motif, i32)
        k = motif.shape[0]
        if emb.dtype != i32 or emb.dim() != 2 or emb.shape[1] != k or not emb.is_contiguous():
            raise ValueError("emb must be a contiguous int32 (n_chains, %d) tensor" % k)
        n_chains = emb.shape[0]
        X = self.empty(k * k, n_chains * int(steps))
        trace = self.torch.empty(n_chains * int(steps), k, dtype=i32, device=self.device) if want_trace else None
        rc = self.lib.onmf_ndl_sample_patches(self.ctx, graph["n"], _ptr(graph["row_ptr"]), _ptr(graph["nbr"]), k,
                                              _ptr(motif), _lib.NDL_GLAUBER if glauber else _lib.NDL_PIVOT, n_chains,
                                              int(steps), _ptr(emb), C.c_uint64(seed), C.c_uint64(step_offset),
                                              _ptr(X), X.stride(0), _ptr(trace), self.stream())
        _lib.check(rc, self.ctx, "onmf_ndl_sample_patches")
        return (X, trace) if want_trace else X

    def ndl_accumulate_edges(self, n_nodes, trace, R):
        """Per ordered node pair: sum and count of the patch entries that fall on it
        (network_reconstruction_nx.py:477-491).  Dense (n x n) tensors."""
        T, k = trace.shape
        if R.shape[0] != k * k or R.shape[1] != T:
            raise ValueError("R must be (%d, %d)" % (k * k, T))
        total = self.zeros(n_nodes, n_nodes)
        count = self.zeros(n_nodes, n_nodes)
        rc = self.lib.onmf_ndl_accumulate_edges(self.ctx, int(n_nodes), k, T, _ptr(trace), _ptr(R), R.stride(0),
                                                _ptr(total), _ptr(count), self.stream())
        _lib.check(rc, self.ctx, "onmf_ndl_accumulate_edges")
        return total, count

    NUMPY_RAND_ON_DEVICE_FROM = 32768       # smaller draws: NumPy itself and one upload are as quick

    def numpy_rand(self, rows, cols):
        """``np.random.rand(rows, cols)`` as a float32 tensor on this GPU, drawn out of NumPy's global legacy
        generator (the reference's H0 and initial dictionary, src/onmf.py:245-246, :190): the generator's state
        goes to the device, the draw is made there (onmf_numpy_rand), and NumPy gets the advanced state back -
        the host's random stream continues exactly as if NumPy had made the draw.  Bit-identical to
        ``dev(np.random.rand(rows, cols))``."""
        n = int(rows) * int(cols)
        state = np.random.get_state()
        if n < self.NUMPY_RAND_ON_DEVICE_FROM or state[0] != "MT19937":
            return self.dev(np.random.rand(rows, cols))
        if not self._mt_table_set:
            from . import mt_jump
            table = np.ascontiguousarray(mt_jump.load_table(), dtype=np.uint32)
            rc = self.lib.onmf_numpy_set_jump_table(self.ctx, table.ctypes.data_as(C.c_void_p), table.shape[0],
                                                    mt_jump.CHUNK_WORDS)
            _lib.check(rc, self.ctx, "onmf_numpy_set_jump_table")
            self._mt_table_set = True
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        key_out = np.empty(624, dtype=np.uint32)
        pos_out = C.c_int(0)
        out = self.empty(rows, cols)
        rc = self.lib.onmf_numpy_rand(self.ctx, key.ctypes.data_as(C.c_void_p), int(state[2]), n, _ptr(out),
                                      key_out.ctypes.data_as(C.c_void_p), C.byref(pos_out), self.stream())
        _lib.check(rc, self.ctx, "onmf_numpy_rand")
        np.random.set_state(("MT19937", key_out, int(pos_out.value), state[3], state[4]))
        return out

    def rand(self, rows, cols, seed=None):
        out = self.empty(rows, cols)
        if seed is None:
            seed = self.next_seed()
        rc = self.lib.onmf_fill_uniform(self.ctx, _ptr(out), out.numel(), C.c_uint64(seed),
                                        self.stream())
        _lib.check(rc, self.ctx, "onmf_fill_uniform")
        return out
