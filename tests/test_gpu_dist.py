"""The multi-GPU helpers on the one GPU of the test box: a ONE-rank ``nccl`` (RCCL) process group, so that
the collective really goes through RCCL's code path (communicator set-up, the all-reduce kernel on RCCL's
stream, stream ordering with the library's own streams).  The two-rank arithmetic is covered on CPU
tensors by tests/test_dist_cpu.py (gloo)."""
import os
import tempfile

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from onmf_ontf_ndl_amd.engine import get_engine
    return get_engine()


@pytest.fixture(scope="module")
def rccl_group(eng):
    import torch
    import torch.distributed as td
    if td.is_initialized():
        yield None
        return
    tmp = tempfile.mkdtemp()
    td.init_process_group("nccl", init_method="file://" + os.path.join(tmp, "rdzv"), rank=0, world_size=1,
                          device_id=eng.device)
    try:
        yield None
    finally:
        torch.cuda.synchronize()
        td.destroy_process_group()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def make(eng, d, r, b, seed):
    import torch
    g = torch.Generator(device=eng.device).manual_seed(seed)
    Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(3)]
    H0s = [torch.rand(r, b, device=eng.device, generator=g) for _ in range(3)]
    W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
    Hh = torch.rand(r, 300, device=eng.device, generator=g)
    A = Hh @ Hh.T / 300
    B = Hh @ torch.rand(d, 300, device=eng.device, generator=g).T / 300
    return Xs, H0s, W, A, B


def test_packed_state_average_runs_through_rccl(eng, rccl_group):
    import torch
    import torch.distributed as td
    from onmf_ontf_ndl_amd import dist as odist
    assert td.get_backend() == "nccl"
    d, r = 70, 9                                    # odd sizes: the parts still start on 16-byte boundaries
    st = odist.PackedState(d, r, eng.device)
    for v in st.views():
        assert v.data_ptr() % 16 == 0 and v.is_contiguous()
    g = torch.Generator(device=eng.device).manual_seed(1)
    W, A, B = (torch.rand(d, r, device=eng.device, generator=g), torch.rand(r, r, device=eng.device, generator=g),
               torch.rand(r, d, device=eng.device, generator=g))
    st.load(W, A, B)
    before = st.flat.clone()
    odist.average_state_(st, force=True)            # one rank: sum over ranks / 1 = identity, through RCCL
    torch.cuda.synchronize()
    assert torch.equal(st.flat, before)
    assert torch.equal(st.W, W) and torch.equal(st.A, A) and torch.equal(st.B, B)
    # the packed views are what the step reads and writes: a step into the other buffer, then back
    X = torch.rand(d, 200, device=eng.device, generator=g)
    st2 = odist.PackedState(d, r, eng.device)
    eng.step(X, st.A, st.B, None, st.W, 0.5, seed=3, stopping_diff=0.0, want_code=False, out=st2.views())
    _, A1, B1, _, W1, _, _ = eng.step(X, A, B, None, W, 0.5, seed=3, stopping_diff=0.0, want_code=False)
    assert torch.equal(st2.W, W1) and torch.equal(st2.A, A1) and torch.equal(st2.B, B1)


@pytest.mark.parametrize("d,r,b", [(96, 24, 1500), (1100, 40, 700), (256, 256, 4096)])
def test_sharded_stepper_equals_the_plain_step(eng, rccl_group, d, r, b):
    """Three chained steps: onmf_step_partial + all-reduce over RCCL on a stream of its own + onmf_step_apply,
    the exchange of step t overlapping step t+1, against onmf_step on the same batches.  Codes and
    dictionaries bit for bit (same kernels, same order); aggregates to rounding (the recursion is applied by
    another kernel than the product's epilogue)."""
    import torch
    from onmf_ontf_ndl_amd import dist as odist
    Xs, H0s, W, A, B = make(eng, d, r, b, 50 + d)
    stepper = odist.ShardedStepper(eng, W, A, B)
    Wp, Ap, Bp = W, A, B
    for t in range(3):
        w = 1.0 / (t + 2)
        H1, Ap, Bp, _, Wp, s1, _ = eng.step(Xs[t], Ap, Bp, None, Wp, w, H0=H0s[t], stopping_diff=0.0)
        H2 = stepper.step(Xs[t], w, H0=H0s[t], stopping_diff=0.0)
        assert s1 == stepper.sweeps == 10
        if t < 2:                       # same W bit for bit up to here (W_2 already saw the other recursion)
            assert torch.equal(H1, H2), t
        else:
            assert relerr(host(H2), host(H1)) < 1e-5
    W2, A2, B2 = stepper.state()
    assert relerr(host(A2), host(Ap)) < 1e-6 and relerr(host(B2), host(Bp)) < 1e-6
    # the dictionary of step t+1 depends on the aggregates of step t-1 ... t: equal to rounding, and equal
    # bit for bit as long as the aggregates are
    assert relerr(host(W2), host(Wp)) < 1e-5
    # and the plain serial form of the same step
    H3, A3, B3, W3, s3 = odist.sharded_step(eng, Xs[0], A, B, W, 0.5, H0=H0s[0], stopping_diff=0.0)
    H4, A4, B4, _, W4, _, _ = eng.step(Xs[0], A, B, None, W, 0.5, H0=H0s[0], stopping_diff=0.0)
    assert torch.equal(H3, H4) and torch.equal(W3, W4)
    assert relerr(host(A3), host(A4)) < 1e-6


def test_step_partial_and_apply_are_the_step(eng):
    """Without any process group: partial sums + apply == onmf_step, and the packed layout of P."""
    import torch
    d, r, b = 130, 33, 900
    Xs, H0s, W, A, B = make(eng, d, r, b, 7)
    P = eng.zeros(eng.partial_floats(d, r))
    assert eng.partial_floats(d, r) == ((r * r + 3) // 4) * 4 + r * d
    H, W1, sweeps = eng.step_partial(Xs[0], A, B, W, P, H0=H0s[0], stopping_diff=0.0)
    A1, B1 = eng.step_apply(P, 0.25, A, B)
    H2, A2, B2, _, W2, _, _ = eng.step(Xs[0], A, B, None, W, 0.25, H0=H0s[0], stopping_diff=0.0)
    assert torch.equal(H, H2) and torch.equal(W1, W2)
    assert relerr(host(A1), host(A2)) < 1e-6 and relerr(host(B1), host(B2)) < 1e-6
    HHt = P[:r * r].view(r, r)
    HXt = P[((r * r + 3) // 4) * 4:].view(r, d)
    assert relerr(host(HHt), host(H) @ host(H).T) < 2e-6
    assert relerr(host(HXt), host(H) @ host(Xs[0]).T) < 2e-6
