"""The device-side screen of the stopping test must reach the verdicts of the full test.

Sweep counts and codes are compared between ONMF_OPT_EXACT_STOP (every test evaluated in full, the
reference's arithmetic) and the default path (rigorous bounds, full test only where they straddle
stopping_diff), including thresholds placed right next to an actual dist value so that the bounds
cannot decide and the fallback has to."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

EXACT, SCREEN = 2, 5


@pytest.fixture(scope="module")
def eng():
    from onmf_ontf_ndl_amd.engine import get_engine
    e = get_engine()
    yield e
    e.set_option(EXACT, 0)
    e.set_option(SCREEN, 1)


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def both(eng, X, W, H0, tol, alpha=0.0):
    eng.set_option(EXACT, 1)
    He, se, de = eng.sparse_code(X, W, H0=H0, alpha=alpha, stopping_diff=tol)
    eng.set_option(EXACT, 0)
    Hs, ss, ds = eng.sparse_code(X, W, H0=H0, alpha=alpha, stopping_diff=tol)
    return (He, se, de), (Hs, ss, ds)


@pytest.mark.parametrize("d,r,b", [(64, 16, 500), (200, 64, 3000), (128, 100, 1111), (96, 256, 2048)])
def test_screen_agrees_with_full_test(eng, d, r, b):
    import torch
    g = torch.Generator(device=eng.device).manual_seed(d + r)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    (He, se, de), (Hs, ss, ds) = both(eng, X, W, H0, 0.01)
    assert ss == se
    assert torch.equal(He, Hs)                 # same kernels' arithmetic on the code either way
    # every value the screen reports is a valid bound of the exact dist on the continuing side
    for i in range(ss - 1):
        assert ds[i] <= de[i] * (1 + 1e-3)
    # thresholds hugging an actual dist value: just below (loop goes on) and just above (loop ends)
    k = min(4, se - 1)
    for tol in (de[k] * 0.97, de[k] * 1.03):
        (He2, se2, _), (Hs2, ss2, _) = both(eng, X, W, H0, float(tol))
        ends = [i for i, v in enumerate(de) if not v > tol]      # dist is not monotone in the sweep index
        assert se2 == (ends[0] + 1 if ends else len(de))
        assert ss2 == se2
        assert torch.equal(He2, Hs2)


def test_bounds_stay_valid_after_the_host_settles_a_sweep(eng):
    """A long solve whose tolerance sits just under the dist of a middle sweep: the bounds cannot
    decide there, the host does and relaunches.  The sweep that had run ahead of that verdict runs
    again, and every value reported afterwards must still bound the exact dist from below."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(321)
    d, r, b = 96, 48, 2500
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    eng.set_option(EXACT, 1)
    _, _, de = eng.sparse_code(X, W, H0=H0, sub_iter=30, stopping_diff=0.0)
    eng.set_option(EXACT, 0)
    assert len(de) == 30
    for k in (6, 14, 22):
        tol = float(min(de[:k + 1]) * 0.98)
        eng.set_option(EXACT, 1)
        He, se, dex = eng.sparse_code(X, W, H0=H0, sub_iter=30, stopping_diff=tol)
        eng.set_option(EXACT, 0)
        Hs, ss, ds = eng.sparse_code(X, W, H0=H0, sub_iter=30, stopping_diff=tol)
        assert ss == se and torch.equal(He, Hs)
        for i in range(ss - 1):
            assert ds[i] <= dex[i] * (1 + 1e-3), (k, i, ds[i], dex[i])


def test_screen_early_stop_and_zero_code(eng):
    import torch
    g = torch.Generator(device=eng.device).manual_seed(5)
    d, r, b = 64, 16, 400
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.linalg.norm(W, dim=0)
    # start at a (near) fixed point: the loop must end after the first sweep, like the full test
    Hc, _, _ = eng.sparse_code(X, W, H0=torch.rand(r, b, device=eng.device, generator=g),
                               sub_iter=60, stopping_diff=-1.0)
    (He, se, _), (Hs, ss, _) = both(eng, X, W, Hc, 0.01)
    assert se == ss == 1 and torch.equal(He, Hs)
    # all-zero start with X = 0: H stays 0, dist = 0/0 -> the reference's loop ends after one sweep
    Z = torch.zeros(d, b, device=eng.device)
    (He, se, _), (Hs, ss, _) = both(eng, Z, W, torch.zeros(r, b, device=eng.device), 0.01)
    assert se == ss == 1 and float(Hs.abs().max()) == 0.0


def test_host_decisions_match_when_screen_is_off(eng):
    import torch
    g = torch.Generator(device=eng.device).manual_seed(6)
    d, r, b = 100, 32, 700
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    eng.set_option(SCREEN, 0)
    Hh, sh, _ = eng.sparse_code(X, W, H0=H0, stopping_diff=0.02)
    eng.set_option(SCREEN, 1)
    Hs, ss, _ = eng.sparse_code(X, W, H0=H0, stopping_diff=0.02)
    assert sh == ss and relerr(host(Hs), host(Hh)) == 0.0


PERSISTENT = 7


@pytest.mark.parametrize("d,r,b,tol", [(64, 16, 500, 0.01), (200, 64, 3000, 0.05), (96, 256, 2048, 0.01),
                                       (100, 32, 70000, 0.02), (40, 8, 1, 0.01)])
def test_one_launch_for_all_sweeps_equals_one_launch_per_sweep(eng, d, r, b, tol):
    """All sweeps of a solve in one launch (H in registers, verdicts two sweeps back, one sweep run
    ahead at a stop) must return what one launch per sweep returns: same count, same code, same
    reported bounds.  b = 70000 needs 274 workgroups: more than the chip has CUs, so that solve
    falls back to one launch per sweep by itself."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(7 * d + r)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    out = {}
    for mode in (1, 0):
        eng.set_option(PERSISTENT, mode)
        try:
            for rep in range(2):                 # second solve: probes kept from the first
                out[(mode, rep)] = eng.sparse_code(X, W, H0=H0, alpha=0.1, sub_iter=12, stopping_diff=tol)
        finally:
            eng.set_option(PERSISTENT, 1)
    eng.set_option(EXACT, 1)
    He, se, _ = eng.sparse_code(X, W, H0=H0, alpha=0.1, sub_iter=12, stopping_diff=tol)
    eng.set_option(EXACT, 0)
    for key, (H, s, _) in out.items():
        assert s == se, key
        assert torch.equal(H, He), key


def test_a_timed_out_one_launch_solve_is_redone_launch_by_launch(eng):
    """The waits of a one-launch solve are given no time at all: whichever workgroup finds the others'
    sums incomplete at its first look gives up (every kernel still ends normally).  The library then
    runs the same solve with one launch per sweep; the caller sees the ordinary result."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(42)
    d, r, b = 128, 48, 40000                       # 157 workgroups
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    eng.set_option(PERSISTENT, 0)
    ref = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    eng.set_option(PERSISTENT, 1)
    eng.set_option(8, -2)                        # every wait gives up at its first look
    before = eng.counters()
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # the engine warns once about the first timeout
            got = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
            again = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
            eng._warn_on_timeouts(now=True)          # (it looks at the counters every few calls only)
    finally:
        eng.set_option(8, 2000)
        eng.set_option(PERSISTENT, 1)            # (also ends the rest the one-launch mode was sent to)
    for out in (got, again):
        assert out[1] == ref[1]
        assert torch.equal(out[0], ref[0])
    # the event is visible to the caller: counted, and the following solve ran launch by launch
    after = eng.counters()
    assert after["sweep_timeouts"] == before["sweep_timeouts"] + 1
    assert after["fallback_solves"] >= before["fallback_solves"] + 2
    # with the rest over, a one-launch solve works again
    eng.get_profile()
    eng.set_option(1, 1)
    third = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    _, launches = eng.get_profile()
    eng.set_option(1, 0)
    assert torch.equal(third[0], ref[0]) and launches <= 2


@pytest.mark.parametrize("d,r,b", [(64, 16, 500), (200, 64, 3000), (128, 40, 1111), (100, 25, 1000)])
def test_inconclusive_verdicts_are_settled_on_the_device_for_small_r(eng, d, r, b):
    """r <= 64, ONMF_OPT_DEVICE_RESOLVE on: once a solve has needed a full evaluation of the stopping test, the next one carries a round
    of device-side resolution (Gram matrices, eigenvalues and the comparison on the GPU, the sweeps
    continued from there).  Thresholds are placed right next to the exact dist of a middle sweep - just
    below (the loop must go on) and just above (it must end there) - so that the bounds cannot decide.
    Sweep counts and codes must equal those of the path that evaluates every test in full on the host."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(7 * d + r)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    eng.set_option(EXACT, 1)
    _, _, de = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=0.0)
    eng.set_option(EXACT, 0)
    assert len(de) == 12
    eng.set_option(9, 1)                          # device rounds behind every solve that needed a full evaluation
    for k in (3, 6, 9):
        for tol in (float(min(de[:k + 1]) * 0.995), float(de[k] * 1.005)):
            eng.set_option(EXACT, 1)
            He, se, _ = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=tol)
            eng.set_option(EXACT, 0)
            before = eng.counters()["device_resolves"]
            first = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=tol)     # host settles (and arms the device round)
            second = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=tol)
            third = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=tol)
            for out in (first, second, third):
                assert out[1] == se, (k, tol)
                assert torch.equal(out[0], He), (k, tol)
            # the settled sweep's value in the trace is the exact dist, as the host path reports it
            if se - 1 < len(de):
                np.testing.assert_allclose(third[2][se - 1], de[se - 1], rtol=1e-4)
            assert eng.counters()["device_resolves"] >= before + 1, (k, tol)
    # switched off: same results, nothing settled on the device
    eng.set_option(9, 0)
    try:
        before = eng.counters()["device_resolves"]
        tol = float(de[6] * 1.005)
        eng.set_option(EXACT, 1)
        He, se, _ = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=tol)
        eng.set_option(EXACT, 0)
        for _ in range(2):
            Hs, ss, _ = eng.sparse_code(X, W, H0=H0, sub_iter=12, stopping_diff=tol)
            assert ss == se and torch.equal(Hs, He)
        assert eng.counters()["device_resolves"] == before
    finally:
        eng.set_option(9, -1)                     # (the default)


@pytest.mark.parametrize("d,r,b", [(300, 64, 4096), (120, 25, 1000), (200, 128, 2048)])
def test_bounds_stay_valid_and_tight_over_a_stream_of_solves(eng, d, r, b):
    """r <= 128: the sweeps leave a Collatz-Wielandt bound of sigma_max of the new code in their sums, formed
    with a positive vector that takes one power step per solve, and the probes take theirs behind the solve.
    Over a stream of solves on fresh minibatches every reported value must stay a lower bound of the exact
    dist, the sweep counts must agree with the full test - also for a threshold hugging a late sweep's
    value - and, once vector and probes have settled, the bound must be tight (it decides tests within a
    few per cent of the threshold: that is what it is for) on the late sweeps, where the test gets close -
    the probes are tuned to the last move of the solve before."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(7 * d + r)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    ratios = []
    for t in range(6):
        X = torch.rand(d, b, device=eng.device, generator=g)
        H0 = torch.rand(r, b, device=eng.device, generator=g)
        (He, se, de), (Hs, ss, ds) = both(eng, X, W, H0, 0.001)
        assert ss == se == 10 and torch.equal(He, Hs)
        for i in range(ss - 1):
            assert ds[i] <= de[i] * (1 + 1e-3), (t, i, ds[i], de[i])
        ratios.append(min(ds[i] / de[i] for i in range(6, 9)))
        tol = float(de[7] * (0.98 if t % 2 else 1.02))
        (He2, se2, de2), (Hs2, ss2, _) = both(eng, X, W, H0, tol)
        assert ss2 == se2 and torch.equal(He2, Hs2), (t, tol, de)
    assert max(ratios[3:]) > 0.8, ratios


@pytest.mark.parametrize("d,r,b", [(64, 256, 3000), (50, 48, 20000), (40, 200, 17000), (64, 100, 5000)])
def test_workgroup_geometries_agree_bit_for_bit(eng, d, r, b):
    """The probing kernel comes with 64, 128 and 256 columns per workgroup, picked by the batch size so that
    the sweeps spread over the chip; columns are independent, so the codes must not depend on the choice.
    ONMF_OPT_SWEEP_KERNEL = 4 forces the 256-column form."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(r + b)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    Ha, sa, _ = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    eng.set_option(3, 4)
    try:
        Hb, sb, _ = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    finally:
        eng.set_option(3, 0)
    assert sa == sb and torch.equal(Ha, Hb)


DEFER = 11


@pytest.mark.parametrize("d,r,b,tol_kind", [(100, 16, 1000, "plain"), (200, 64, 3000, "plain"), (200, 64, 3000, "hug"),
                                            (96, 256, 2048, "hug"), (128, 100, 4096, "plain")])
def test_steps_that_do_not_wait_for_their_verdicts(eng, d, r, b, tol_kind):
    """ONMF_OPT_DEFER_VERDICT: a chain of onmf_step calls that return no host scalars and never wait for the
    GPU must leave the same state, bit for bit, and run the same number of sweeps as the chain that reads
    every verdict - also when the tolerance hugs an actual dist value, so that verdicts come out
    inconclusive and the library has to run those steps' solves again at the next call."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(7 * d + r)
    Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(6)]
    W0 = torch.rand(d, r, device=eng.device, generator=g)
    W0 = W0 / torch.clamp(torch.linalg.norm(W0, dim=0), min=1.0)
    tol = 0.01
    if tol_kind == "hug":
        eng.set_option(EXACT, 1)
        _, _, de = eng.sparse_code(Xs[1], W0, H0=None, seed=1001, stopping_diff=0.0)
        eng.set_option(EXACT, 0)
        tol = float(de[5] * 0.995)
    results = []
    for defer in (False, True):
        eng.set_option(DEFER, 1 if defer else 0)
        A, B, W = eng.zeros(r, r), eng.zeros(r, d), W0.clone()
        c0 = eng.counters()
        sweeps = []
        try:
            for t, X in enumerate(Xs):
                H, A, B, _, W, s, _ = eng.step(X, A, B, None, W, 1.0 / (t + 1), seed=1000 + t, stopping_diff=tol,
                                               defer=defer)
                sweeps.append(s)
            eng.settle()
        finally:
            eng.set_option(DEFER, 0)
        torch.cuda.synchronize()
        c1 = eng.counters()
        results.append((H.clone(), A.clone(), B.clone(), W.clone(), c1["sweeps_total"] - c0["sweeps_total"],
                        c1["deferred_redone"] - c0["deferred_redone"], sweeps))
    (H1, A1, B1, W1, n1, _, sw1), (H2, A2, B2, W2, n2, redone, sw2) = results
    print("d=%d r=%d b=%d tol=%.4g: sweeps %s, deferred chain: %d solves run again" % (d, r, b, tol, sw1, redone))
    assert all(s is None for s in sw2) and n1 == sum(sw1) == n2
    assert torch.equal(W1, W2) and torch.equal(A1, A2) and torch.equal(B1, B2) and torch.equal(H1, H2)


@pytest.mark.parametrize("d,r,b", [(100, 16, 1000), (200, 64, 3000), (120, 100, 2048), (64, 256, 3000)])
def test_tests_acted_on_three_sweeps_later_change_nothing(eng, d, r, b):
    """The kernels with a control wave (64 columns per workgroup) may act on a sweep's test three sweeps later
    instead of two, so that no wave ever waits for a verdict (ONMF_OPT_VERDICT_LAG3, the default): same sweep
    counts and the same bits as with the option off and as the full test - for tolerances that never stop,
    that stop in the middle (at every sweep in turn: a stop now leaves two sweeps run in vain) and that hug a
    sweep's dist so closely that the bounds cannot decide (the solve is then run again in the plain form)."""
    import torch
    LAG3 = 13
    g = torch.Generator(device=eng.device).manual_seed(11 * d + r)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    eng.set_option(EXACT, 1)
    _, _, trace = eng.sparse_code(X, W, H0=H0, alpha=0.05, sub_iter=12, stopping_diff=0.0)
    eng.set_option(EXACT, 0)
    dists = [float(v) for v in trace[:11]]
    tols = [0.0] + [0.5 * (dists[i] + dists[i + 1]) for i in range(0, 10)] + [dists[4] * (1 + 1e-6), dists[7] * (1 - 1e-6)]
    for tol in tols:
        eng.set_option(EXACT, 1)
        He, se, _ = eng.sparse_code(X, W, H0=H0, alpha=0.05, sub_iter=12, stopping_diff=tol)
        eng.set_option(EXACT, 0)
        for lag3 in (1, 0, 1):
            eng.set_option(LAG3, lag3)
            try:
                H, s, _ = eng.sparse_code(X, W, H0=H0, alpha=0.05, sub_iter=12, stopping_diff=tol)
            finally:
                eng.set_option(LAG3, 1)
            assert s == se, (tol, lag3, s, se)
            assert torch.equal(H, He), (tol, lag3)


@pytest.mark.parametrize("d,r,b,tol", [(441, 25, 1000, 0.01), (100, 16, 700, 0.02), (150, 32, 2000, 0.02)])
def test_early_stopping_streams_settle_their_tests_on_the_device_by_default(eng, d, r, b, tol):
    """Default options, r <= 32, a stream whose solves stop early: after the first solve that needed a full
    evaluation the following ones carry device rounds, each of which settles the inconclusive sweep and the
    one after it.  The chain - waited for, and not waited for (ONMF_OPT_DEFER_VERDICT) - must run the sweep
    counts of the chain that evaluates every test in full on the host and leave the same state bit for bit;
    and the device must indeed have settled tests."""
    import torch
    g = torch.Generator(device=eng.device).manual_seed(13 * d + r)
    Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(8)]
    W0 = torch.rand(d, r, device=eng.device, generator=g)
    eng.set_option(9, -1)                # (the default, whatever an earlier test left)
    eng.set_option(EXACT, 1)             # (tolerances that these streams reach after 3 to 8 sweeps)

    def chain(defer):
        A, B, W = eng.zeros(r, r), eng.zeros(r, d), W0.clone()
        c0 = eng.counters()
        sweeps = []
        for t, X in enumerate(Xs):
            H, A, B, _, W, s, _ = eng.step(X, A, B, None, W, 1.0 / (t + 1), seed=2000 + t, stopping_diff=tol, defer=defer)
            sweeps.append(s)
        eng.settle()
        torch.cuda.synchronize()
        c1 = eng.counters()
        return (H.clone(), A.clone(), B.clone(), W.clone(), c1["sweeps_total"] - c0["sweeps_total"],
                c1["device_resolves"] - c0["device_resolves"], sweeps)

    He, Ae, Be, We, ne, _, swe = chain(False)
    eng.set_option(EXACT, 0)
    assert ne == sum(swe) and sum(s < 10 for s in swe) >= 5, swe          # most of the stream's solves stop early
    Hw, Aw, Bw, Ww, nw, resolved_w, sww = chain(False)
    eng.set_option(DEFER, 1)
    try:
        Hd, Ad, Bd, Wd, nd, resolved_d, _ = chain(True)
    finally:
        eng.set_option(DEFER, 0)
    print("d=%d r=%d b=%d tol=%.4g: sweeps %s, tests settled on the device: %d waited, %d not waited" % (
        d, r, b, tol, swe, resolved_w, resolved_d))
    assert sww == swe and nw == ne == nd
    for got in ((Hw, Aw, Bw, Ww), (Hd, Ad, Bd, Wd)):
        assert all(torch.equal(x, y) for x, y in zip(got, (He, Ae, Be, We)))
    assert resolved_w > 0 and resolved_d > 0
