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
