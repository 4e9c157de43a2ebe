"""Multi-process (world_size 2, gloo, CPU tensors) checks of the multi-GPU helpers."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from onmf_ontf_ndl_amd import dist as odist


def _worker(rank, world, init_file, out_dir):
    import torch.distributed as td
    from oracle import onmf_oracle as orc
    td.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(100 + rank)
        d, r = 12, 4
        W = torch.from_numpy(rs.rand(d, r))
        A = torch.from_numpy(rs.rand(r, r))
        B = torch.from_numpy(rs.rand(r, d))
        W1, A1, B1 = odist.average_state(W, A, B)
        # every rank can rebuild what the other one held
        exp = [np.random.RandomState(100 + k) for k in range(world)]
        Ws, As, Bs = zip(*[(e.rand(d, r), e.rand(r, r), e.rand(r, d)) for e in exp])
        np.testing.assert_allclose(W1.numpy(), np.mean(Ws, axis=0), rtol=1e-12)
        np.testing.assert_allclose(A1.numpy(), np.mean(As, axis=0), rtol=1e-12)
        np.testing.assert_allclose(B1.numpy(), np.mean(Bs, axis=0), rtol=1e-12)
        assert W1.min() >= 0 and np.linalg.norm(W1.numpy(), axis=0).max() <= np.linalg.norm(Ws[0], axis=0).max() + 1
        # the resident packed buffer: one in-place collective, views stay valid
        st = odist.PackedState(d, r, "cpu", torch.float64).load(W, A, B)
        views = st.views()
        odist.average_state_(st)
        np.testing.assert_allclose(st.W.numpy(), np.mean(Ws, axis=0), rtol=1e-12)
        np.testing.assert_allclose(st.A.numpy(), np.mean(As, axis=0), rtol=1e-12)
        assert all(v.data_ptr() == u.data_ptr() for v, u in zip(views, st.views()))

        # one minibatch split by columns: summed partial products == the products of the whole batch
        full = np.random.RandomState(7)
        n = 30
        X = full.rand(d, n)
        Wd = full.rand(d, r)
        H0 = full.rand(r, n)
        lo, hi = odist.split_columns(n, rank, world)
        H_loc = orc.code_sweeps(X[:, lo:hi], Wd, H0=H0[:, lo:hi], alpha=0, sub_iter=10, stopping_diff=-1.0)
        HHt = torch.from_numpy(H_loc @ H_loc.T)
        HXt = torch.from_numpy(H_loc @ X[:, lo:hi].T)
        odist.allreduce_partial_aggregates(HHt, HXt)
        H_all = orc.code_sweeps(X, Wd, H0=H0, alpha=0, sub_iter=10, stopping_diff=-1.0)
        np.testing.assert_allclose(HHt.numpy(), H_all @ H_all.T, rtol=1e-10)
        np.testing.assert_allclose(HXt.numpy(), H_all @ X.T, rtol=1e-10)
        # the whole sharded step, with the oracle standing in for the GPU engine: equals the reference's
        # step on the concatenated batch (fixed sweep count), and the state stays replicated
        class OracleEngine:
            def sparse_code(self, X, W, H0=None, alpha=0.0, sub_iter=10, stopping_diff=0.0, seed=None):
                H = orc.code_sweeps(X.numpy(), W.numpy(), H0=H0.numpy(), alpha=alpha, sub_iter=sub_iter,
                                    stopping_diff=-1.0)
                return torch.from_numpy(H), sub_iter, None

            def aggregate(self, X, H, w, A, B):
                assert w == 1.0 and A is None and B is None
                return H @ H.T, H @ X.T, None

            def update_dict(self, W, A, B):
                return torch.from_numpy(orc.dictionary_pass(W.numpy(), A.numpy(), B.numpy()))

        A0 = torch.from_numpy(full.rand(r, r)); A0 = A0 @ A0.T
        B0 = torch.from_numpy(full.rand(r, d))
        Wt = torch.from_numpy(Wd)
        t = 3
        H_l, A1, B1, W1, sw = odist.sharded_step(OracleEngine(), torch.from_numpy(X[:, lo:hi]), A0, B0, Wt, 1.0 / t,
                                                 H0=torch.from_numpy(H0[:, lo:hi]), sub_iter=10, stopping_diff=0.0)
        Href, (Aref, Bref), Wref, _ = orc.minibatch_step(X, [A0.numpy(), B0.numpy()], Wd, t, H0=H0, sub_iter=10,
                                                         stopping_diff=-1.0)
        np.testing.assert_allclose(H_l.numpy(), Href[:, lo:hi], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(A1.numpy(), Aref, rtol=1e-10)
        np.testing.assert_allclose(B1.numpy(), Bref, rtol=1e-10)
        np.testing.assert_allclose(W1.numpy(), Wref, rtol=1e-10, atol=1e-14)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        td.destroy_process_group()


class OracleStepEngine:
    """Stands in for ``Engine`` under ``ShardedStepper``: the two entry points it drives, on host tensors, with
    the oracle's arithmetic.  ``state_ready_event`` is honoured the way the library does: the code solve and the
    dictionary's own input W are touched at once, A, B and P only after the event."""

    device = torch.device("cpu")

    def __init__(self):
        from oracle import onmf_oracle as orc
        self.orc = orc
        self.log = []

    def empty(self, *shape):
        return torch.full(shape, float("nan"), dtype=torch.float64)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def partial_floats(self, d, r):
        return ((r * r + 3) & ~3) + r * d

    def step_partial(self, X, A, B, W, P, W_out=None, alpha=0.0, H0=None, seed=None, sub_iter=10, stopping_diff=0.0,
                     want_code=True, state_ready_event=None):
        r, d = B.shape
        H = self.orc.code_sweeps(X.numpy(), W.numpy(), H0=H0.numpy(), alpha=alpha, sub_iter=sub_iter,
                                 stopping_diff=-1.0)
        if state_ready_event is not None:
            state_ready_event.synchronize()
        assert not torch.isnan(A).any() and not torch.isnan(B).any(), "aggregates read before they were final"
        P[:r * r] = torch.from_numpy(H @ H.T).reshape(-1)
        P[(r * r + 3) & ~3:] = torch.from_numpy(H @ X.numpy().T).reshape(-1)
        W_out.copy_(torch.from_numpy(self.orc.dictionary_pass(W.numpy(), A.numpy(), B.numpy())))
        return torch.from_numpy(H), W_out, sub_iter

    def step_apply(self, P, w, A, B, A_out=None, B_out=None):
        r, d = B.shape
        A_out.copy_((1.0 - w) * A + w * P[:r * r].view(r, r))
        B_out.copy_((1.0 - w) * B + w * P[(r * r + 3) & ~3:].view(r, d))
        return A_out, B_out


class LazyExchange:
    """An exchange scheduler whose work runs as late as the program allows: only when somebody waits for its
    event - what a slow communication stream looks like to the buffers."""

    class Event:
        def __init__(self, fn):
            self.fn = fn

        def synchronize(self):
            if self.fn is not None:
                fn, self.fn = self.fn, None
                fn()

    def __init__(self):
        self.pending = None

    def after_main(self, fn):
        prev = self.pending                     # one stream: exchanges complete in the order they were issued

        def chained():
            if prev is not None:
                prev.synchronize()
            fn()
        self.pending = LazyExchange.Event(chained)
        return self.pending

    def sync_main(self):
        pass


def _chain_reference(orc, Xs, H0s, W, A, B, t0):
    for k, (X, H0) in enumerate(zip(Xs, H0s)):
        _, (A, B), W, _ = orc.minibatch_step(X, [A, B], W, t0 + k, H0=H0, sub_iter=10, stopping_diff=-1.0)
    return W, A, B


def test_sharded_stepper_buffer_rotation_with_a_late_exchange():
    """Four chained steps through ShardedStepper with every exchange delayed until the library would wait for
    it: the two buffer sets must keep what a pending exchange still reads, and the result must equal the
    reference's chain (single rank: the sums are the whole batch's)."""
    from oracle import onmf_oracle as orc
    rs = np.random.RandomState(5)
    d, r, n, steps = 10, 3, 14, 4
    Xs = [rs.rand(d, n) for _ in range(steps)]
    H0s = [rs.rand(r, n) for _ in range(steps)]
    W = rs.rand(d, r)
    A = rs.rand(r, r); A = A @ A.T
    B = rs.rand(r, d)
    for sched in (LazyExchange(), None):
        st = odist.ShardedStepper(OracleStepEngine(), torch.from_numpy(W), torch.from_numpy(A), torch.from_numpy(B),
                                  sched=sched)
        for k in range(steps):
            st.step(torch.from_numpy(Xs[k]), 1.0 / (3 + k), H0=torch.from_numpy(H0s[k]))
        W1, A1, B1 = st.state()
        Wr, Ar, Br = _chain_reference(orc, Xs, H0s, W, A, B, 3)
        np.testing.assert_allclose(W1.numpy(), Wr, rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(A1.numpy(), Ar, rtol=1e-10)
        np.testing.assert_allclose(B1.numpy(), Br, rtol=1e-10)


def _stepper_worker(rank, world, init_file, out_dir):
    import torch.distributed as td
    from oracle import onmf_oracle as orc
    td.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(11)                       # the same stream on every rank
        d, r, n, steps = 9, 3, 17, 3
        Xs = [rs.rand(d, n) for _ in range(steps)]
        H0s = [rs.rand(r, n) for _ in range(steps)]
        W = rs.rand(d, r)
        A = rs.rand(r, r); A = A @ A.T
        B = rs.rand(r, d)
        lo, hi = odist.split_columns(n, rank, world)
        st = odist.ShardedStepper(OracleStepEngine(), torch.from_numpy(W), torch.from_numpy(A), torch.from_numpy(B))
        for k in range(steps):
            st.step(torch.from_numpy(Xs[k][:, lo:hi].copy()), 1.0 / (3 + k), H0=torch.from_numpy(H0s[k][:, lo:hi].copy()))
        W1, A1, B1 = st.state()
        Wr, Ar, Br = _chain_reference(orc, Xs, H0s, W, A, B, 3)
        np.testing.assert_allclose(W1.numpy(), Wr, rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(A1.numpy(), Ar, rtol=1e-10)
        np.testing.assert_allclose(B1.numpy(), Br, rtol=1e-10)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        td.destroy_process_group()


def test_two_rank_gloo_sharded_stepper_against_the_whole_batch():
    """ShardedStepper over two ranks (columns split, sums all-reduced every step, state replicated) equals the
    reference's chain on the concatenated batches."""
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rdzv")
        mp.spawn(_stepper_worker, args=(2, init_file, tmp), nprocs=2, join=True)
        assert os.path.exists(os.path.join(tmp, "ok0")) and os.path.exists(os.path.join(tmp, "ok1"))


def test_two_rank_gloo_average_and_partial_sums():
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rdzv")
        mp.spawn(_worker, args=(2, init_file, tmp), nprocs=2, join=True)
        assert os.path.exists(os.path.join(tmp, "ok0")) and os.path.exists(os.path.join(tmp, "ok1"))


def test_single_process_helpers():
    W, A, B = torch.rand(5, 3), torch.rand(3, 3), torch.rand(3, 5)
    flat = odist.pack_state(W, A, B)
    W2, A2, B2 = odist.unpack_state(flat, 5, 3)
    assert torch.equal(W, W2) and torch.equal(A, A2) and torch.equal(B, B2)
    assert odist.average_state(W, A, B)[0] is W            # no process group: identity
    st = odist.PackedState(5, 3, "cpu").load(W, A, B)
    assert torch.equal(st.W, W) and torch.equal(st.A, A) and torch.equal(st.B, B)
    assert st.flat.numel() == 12 + 16 + 16 and all(v.data_ptr() % 16 == 0 for v in st.views())
    assert odist.average_state_(st) is st
    spans = [odist.split_columns(10, k, 3) for k in range(3)]
    assert spans == [(0, 4), (4, 7), (7, 10)]
    assert odist.init_from_env() == (0, 1, 0) or os.environ.get("WORLD_SIZE", "1") != "1"
