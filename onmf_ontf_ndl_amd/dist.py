"""Multi-GPU operation of the ONMF step: one process per GPU, ``torch.distributed`` (RCCL on ROCm).

The reference is single-process; the only quantities that couple minibatch columns are the sums
H H^T and H X^T (src/onmf.py:155-156) and the replicated dictionary.  Two ways to use several GPUs:

* independent streams, periodic averaging (``PackedState`` + ``average_state_``): every rank runs its
  own patch stream with its own (W, A, B) and, every few steps, the ranks all-reduce the packed state
  [A | B | W] in place and divide by the world size.  Averages of nonnegative matrices with column
  norms <= 1 stay in that set, so W needs no re-projection.  One collective of r*r + 2*r*d floats per
  exchange, on one resident buffer (no packing copies).
* one large minibatch split by columns (``ShardedStepper``; ``sharded_step`` is its plain serial form):
  each rank codes its slice and the partial sums are added before the (1-w, w) recursion - equal to the
  reference run on the concatenated batch up to summation order.  The all-reduce of step t's sums runs
  on a stream of its own underneath step t+1's W^T X and sweeps: only the dictionary pass of step t+1
  needs the aggregates of step t (src/onmf.py:161).

All helpers work on any tensor device; a ``gloo`` group only ever sees host memory (device tensors are
staged through it synchronously).
"""
from __future__ import annotations

import os


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* if there is one.

    Backend: ``backend`` or ``ONMF_DIST_BACKEND``; else ``nccl`` (RCCL) when every rank has a GPU of its
    own, ``gloo`` when ranks must share GPUs (RCCL cannot form a group with two ranks on one device;
    the local device index then wraps around the devices present) or there is no GPU.
    Returns (rank, world, local device index)."""
    import torch
    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not td.is_initialized():
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if backend is None:
            backend = os.environ.get("ONMF_DIST_BACKEND") or ("nccl" if n_dev >= local_world else "gloo")
        if backend == "nccl":
            if local >= n_dev:
                raise RuntimeError("rank %d: local rank %d has no GPU of its own (%d present); RCCL needs one "
                                   "device per rank - set ONMF_DIST_BACKEND=gloo to share GPUs" % (rank, local, n_dev))
            torch.cuda.set_device(local)
            td.init_process_group(backend=backend, device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend=backend)
    if td.get_backend() != "nccl" and torch.cuda.is_available():
        local %= max(1, torch.cuda.device_count())
    return rank, world, local


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def share_of_cores(cores, position, sharers):
    """The ``position``-th of ``sharers`` contiguous, near-equal shares of ``cores`` (never empty)."""
    cores = sorted(cores)
    if sharers <= 1 or len(cores) < sharers:
        return cores
    base, extra = divmod(len(cores), sharers)
    lo = position * base + min(position, extra)
    return cores[lo:lo + base + (1 if position < extra else 0)]


def _gpu_numa_node(index):
    """NUMA node of GPU ``index`` from sysfs, or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def pin_to_local_cores(local_rank, local_world, device_index=None):
    """Keep this rank's process on cores of its own: the cores of its GPU's NUMA node that the process may
    use, shared out among the ranks whose GPUs sit on the same node; an even share of the allowed cores when
    the topology cannot be read.  The small shapes are bound by the host's launch rate, which suffers when
    eight ranks' threads wander over two sockets.  Returns the cores chosen, or None where the platform has
    no ``sched_setaffinity``."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    chosen = None
    node = _gpu_numa_node(device_index) if device_index is not None else None
    if node is not None:
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                local_cores = [c for c in parse_cpulist(f.read()) if c in set(allowed)]
            mates = [k for k in range(local_world) if _gpu_numa_node(k) == node]
            if local_cores and local_rank in mates:
                chosen = share_of_cores(local_cores, mates.index(local_rank), len(mates))
        except (OSError, ValueError):
            chosen = None
    if not chosen:
        chosen = share_of_cores(allowed, local_rank % max(1, local_world), local_world)
    try:
        os.sched_setaffinity(0, chosen)
    except OSError:
        return None
    return chosen


def _active(group=None):
    import torch.distributed as td
    return td.is_available() and td.is_initialized()


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


def _up4(n):
    return (n + 3) & ~3


class PackedState:
    """(A, B, W) as views of ONE contiguous float32 vector [A | B | W], every part starting on a 16-byte
    boundary (the kernels take their vector paths only for 16-byte aligned matrices).  The all-reduce
    of the state is then one in-place collective on ``flat``."""

    def __init__(self, d, r, device, dtype=None):
        import torch
        self.d, self.r = int(d), int(r)
        oa, ob, ow = 0, _up4(r * r), _up4(r * r) + _up4(r * d)
        self.flat = torch.zeros(ow + _up4(d * r), device=device, dtype=dtype or torch.float32)
        self.A = self.flat[oa:oa + r * r].view(r, r)
        self.B = self.flat[ob:ob + r * d].view(r, d)
        self.W = self.flat[ow:ow + d * r].view(d, r)

    def load(self, W, A, B):
        self.W.copy_(W)
        self.A.copy_(A)
        self.B.copy_(B)
        return self

    def views(self):
        return self.A, self.B, self.W


def average_state_(state, group=None, force=False):
    """In place: all-reduce(SUM) of the packed state, divided by the world size.  ``force`` runs the
    collective on a one-rank group too (it is then the identity; used to exercise the backend)."""
    import torch.distributed as td
    if not _active(group):
        return state
    world = td.get_world_size(group)
    if world == 1 and not force:
        return state
    if state.flat.is_cuda and td.get_backend(group) == "nccl":
        # RCCL averages in the collective itself: one operation on the buffer, ordered on the current
        # stream (torch runs it on its communicator's stream between two events; the host does not wait)
        td.all_reduce(state.flat, op=td.ReduceOp.AVG, group=group)
        return state
    all_reduce_sum(state.flat, group)
    if world > 1:
        state.flat.mul_(1.0 / world)
    return state


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
    """All-reduce(SUM) of the state, divided by the world size; returns new (W, A, B).  For loose
    tensors; a stream that keeps its state in a ``PackedState`` uses ``average_state_`` (no copies)."""
    import torch.distributed as td
    if not _active(group) or td.get_world_size(group) == 1:
        return W, A, B
    d, r = W.shape
    st = PackedState(d, r, W.device, W.dtype).load(W, A, B)
    average_state_(st, group)
    return st.W, st.A, st.B


def allreduce_partial_aggregates(HHt, HXt, group=None):
    """Sum the per-rank partial products H_g H_g^T and H_g X_g^T in place (one collective)."""
    import torch
    import torch.distributed as td
    if not _active(group) or td.get_world_size(group) == 1:
        return HHt, HXt
    flat = torch.cat([HHt.reshape(-1), HXt.reshape(-1)])
    all_reduce_sum(flat, group)
    n = HHt.numel()
    HHt.copy_(flat[:n].reshape(HHt.shape))
    HXt.copy_(flat[n:].reshape(HXt.shape))
    return HHt, HXt


def sharded_step(eng, X_local, A, B, W, w, *, alpha=0.0, H0=None, seed=None, sub_iter=10,
                 stopping_diff=0.0, group=None):
    """One minibatch step (src/onmf.py:119-167) on a minibatch whose columns are split over the ranks -
    the plain form: three calls and a blocking collective.  ``ShardedStepper`` is the scheduled form.

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


class _CommStream:
    """Where ``ShardedStepper`` runs a step's exchange: a HIP stream of its own (default priority), ordered
    behind what the main stream holds at that moment.  ``after_main(fn)`` returns an event that fires when
    ``fn``'s work is done."""

    def __init__(self, device):
        import torch
        self.device = device
        self.comm = torch.cuda.Stream(device=device)

    def after_main(self, fn):
        import torch
        main = torch.cuda.current_stream(self.device)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(done)
            fn()
            ready = torch.cuda.Event()
            ready.record(self.comm)
        return ready

    def sync_main(self):
        import torch
        torch.cuda.current_stream(self.device).synchronize()


class _Inline:
    """The same interface with nothing to overlap: the exchange runs at once (host tensors; a ``gloo``
    group, whose collective is staged through the host anyway)."""

    def __init__(self, device=None):
        self.device = device

    def after_main(self, fn):
        self.sync_main()
        fn()
        return None

    def sync_main(self):
        if self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            import torch
            torch.cuda.current_stream(self.device).synchronize()


class ShardedStepper:
    """The column-split step with the library's own schedule and the exchange off the critical path.

    Holds the replicated state (W, A, B) on the GPU.  ``step(X_local, w)``:
      main stream   onmf_step_partial: W^T X, sweeps, bare sums P_t = [H H^T | H X^T], dictionary pass
                    W_{t+1} = update_dict(W_t, A_{t-1}, B_{t-1}) beside them (three-stream schedule)
      comm stream   all-reduce(P_t), then (A_t, B_t) = (1-w)(A_{t-1}, B_{t-1}) + w P_t, then an event
    The next step's code solve starts at once (it needs W_{t+1} only); its dictionary pass and its own
    sums wait for the event inside the library.  Buffers alternate between two sets, so nothing a
    pending exchange reads is overwritten.

    ``sched`` (optional) is where the exchange runs (``after_main(fn) -> event``, ``sync_main()``): by
    default a stream of its own on the engine's GPU, or inline when the collective is host-staged.
    """

    def __init__(self, eng, W, A, B, group=None, sched=None):
        self.eng, self.group = eng, group
        d, r = W.shape
        self.d, self.r = d, r
        self.W = [W.clone().contiguous(), eng.empty(d, r)]
        self.A = [A.clone().contiguous(), eng.empty(r, r)]
        self.B = [B.clone().contiguous(), eng.empty(r, d)]
        self.P = [eng.zeros(eng.partial_floats(d, r)) for _ in range(2)]
        self.cur = 0
        self.ready = None                       # event: (A, B)[cur] are final and P[1 - cur] is free
        if sched is None:
            import torch.distributed as td
            staged = _active(group) and td.get_world_size(group) > 1 and td.get_backend(group) == "gloo"
            on_gpu = getattr(eng.device, "type", "cpu") == "cuda"
            sched = _CommStream(eng.device) if (on_gpu and not staged) else _Inline(eng.device)
        self.sched = sched
        self.sweeps = None

    def state(self):
        """(W, A, B) after the last step (waits for a pending exchange)."""
        if self.ready is not None:
            self.ready.synchronize()
        c = self.cur
        return self.W[c], self.A[c], self.B[c]

    def step(self, X_local, w, *, alpha=0.0, H0=None, seed=None, sub_iter=10, stopping_diff=0.0, want_code=True):
        eng, c, n = self.eng, self.cur, 1 - self.cur
        P = self.P[c]
        H, _, self.sweeps = eng.step_partial(X_local, self.A[c], self.B[c], self.W[c], P, W_out=self.W[n], alpha=alpha,
                                             H0=H0, seed=seed, sub_iter=sub_iter, stopping_diff=stopping_diff,
                                             want_code=want_code, state_ready_event=self.ready)
        A_old, B_old, A_new, B_new, group = self.A[c], self.B[c], self.A[n], self.B[n], self.group

        def exchange():
            if _active(group):
                all_reduce_sum(P, group)          # RCCL: ordered on the stream it is issued on
            eng.step_apply(P, w, A_old, B_old, A_new, B_new)

        self.ready = self.sched.after_main(exchange)
        self.cur = n
        return H


def split_columns(n, rank, world):
    """Column range [lo, hi) of rank's slice when n columns are split as evenly as possible."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
