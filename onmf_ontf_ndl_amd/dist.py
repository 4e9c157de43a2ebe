"""Multi-GPU operation of the ONMF step: one process per GPU, ``torch.distributed`` (RCCL on ROCm).

The reference is single-process; the only quantities that couple minibatch columns are the sums
H H^T and H X^T (src/onmf.py:155-156) and the replicated dictionary.  Two ways to use several GPUs:

* independent streams, periodic averaging (``average_state``): every rank runs its own patch stream
  with its own (W, A, B) and, every few steps, the ranks all-reduce the packed state [A | B | W] and
  divide by the world size.  Averages of nonnegative matrices with column norms <= 1 stay in that
  set, so W needs no re-projection.  One collective of r*r + 2*r*d floats per exchange.
* one large minibatch split by columns (``sharded_step``, built on ``allreduce_partial_aggregates``):
  each rank codes its slice and the partial sums are added before the (1-w, w) recursion - equal to
  the reference run on the concatenated batch up to summation order.

Both helpers work on any tensor device; under the ``gloo`` backend device tensors are staged through
host memory.
"""
from __future__ import annotations

import os


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* if there is one.

    ``ONMF_DIST_BACKEND=gloo`` forces the host backend (e.g. to rehearse several ranks on a machine
    with fewer GPUs than ranks: the local device index then wraps around the devices present)."""
    import torch
    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not td.is_initialized():
        if backend is None:
            backend = os.environ.get("ONMF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend != "nccl" and torch.cuda.is_available():
            local %= max(1, torch.cuda.device_count())
        if backend == "nccl":
            torch.cuda.set_device(local)
            td.init_process_group(backend=backend, device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend=backend)
    return rank, world, local


def all_reduce_sum(flat, group=None):
    """In-place sum over the ranks.  A ``gloo`` group only ever sees host memory: a device tensor is
    copied to the host (synchronously), reduced there and copied back."""
    import torch.distributed as td
    if flat.is_cuda and td.get_backend(group) == "gloo":
        host = flat.cpu()
        td.all_reduce(host, op=td.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        td.all_reduce(flat, op=td.ReduceOp.SUM, group=group)
    return flat


def pack_state(W, A, B):
    """[A | B | W] as one contiguous vector (one collective instead of three)."""
    import torch
    return torch.cat([A.reshape(-1), B.reshape(-1), W.reshape(-1)])


def unpack_state(flat, d, r):
    a, b = r * r, r * d
    A = flat[:a].reshape(r, r)
    B = flat[a:a + b].reshape(r, d)
    W = flat[a + b:a + b + d * r].reshape(d, r)
    return W, A, B


def average_state(W, A, B, group=None):
    """All-reduce(SUM) of the packed state, divided by the world size. Returns new (W, A, B)."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size(group) == 1:
        return W, A, B
    d, r = W.shape
    flat = pack_state(W, A, B)
    all_reduce_sum(flat, group)
    flat /= td.get_world_size(group)
    W1, A1, B1 = unpack_state(flat, d, r)
    # own allocations: slices of the packed vector start at odd offsets for odd r, and the kernels
    # take their 16-byte paths only for 16-byte aligned matrices
    return W1.clone(), A1.clone(), B1.clone()


def allreduce_partial_aggregates(HHt, HXt, group=None):
    """Sum the per-rank partial products H_g H_g^T and H_g X_g^T in place (one collective)."""
    import torch
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size(group) == 1:
        return HHt, HXt
    flat = torch.cat([HHt.reshape(-1), HXt.reshape(-1)])
    all_reduce_sum(flat, group)
    n = HHt.numel()
    HHt.copy_(flat[:n].reshape(HHt.shape))
    HXt.copy_(flat[n:].reshape(HXt.shape))
    return HHt, HXt


def sharded_step(eng, X_local, A, B, W, w, *, alpha=0.0, H0=None, seed=None, sub_iter=10,
                 stopping_diff=0.0, group=None):
    """One minibatch step (src/onmf.py:119-167) on a minibatch whose columns are split over the ranks.

    Every rank holds the same (W, A, B) and its own columns ``X_local``.  It codes them, forms its
    partial sums H H^T and H X^T (src/onmf.py:155-156 are sums over columns), the ranks add them
    with one collective, and every rank applies the (1 - w, w) recursion and the dictionary pass -
    on the OLD aggregates, as the reference does (src/onmf.py:161) - to identical inputs, so the
    state stays replicated without a broadcast.  The result equals the reference's step on the
    concatenated batch up to summation order when ``stopping_diff <= 0`` (fixed sweep count); with a
    positive tolerance each rank applies the stopping test to its own columns, which the reference
    applies to the whole batch.

    ``eng`` is an ``Engine`` (or anything with its ``sparse_code`` / ``aggregate`` / ``update_dict``).
    Returns (H_local, A1, B1, W1, sweeps).
    """
    H, sweeps, _ = eng.sparse_code(X_local, W, H0=H0, alpha=alpha, sub_iter=sub_iter,
                                   stopping_diff=stopping_diff, seed=seed)
    HHt, HXt, _ = eng.aggregate(X_local, H, 1.0, None, None)        # w = 1, no old part: the bare sums
    allreduce_partial_aggregates(HHt, HXt, group)
    A1 = (1.0 - w) * A + w * HHt
    B1 = (1.0 - w) * B + w * HXt
    W1 = eng.update_dict(W, A, B)
    return H, A1, B1, W1, sweeps


def split_columns(n, rank, world):
    """Column range [lo, hi) of rank's slice when n columns are split as evenly as possible."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
