"""onmf_update_dict_static - the dictionary pass for callers that cannot allocate device memory - must
give what onmf_update_dict gives, bit for bit (same kernels; only the workspace differs: a static device
array of the library instead of buffers the context allocates).  Two calls in a row, the second on a
smaller input, so that anything the first call left in the shared array would show."""
import ctypes as C

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _state(rs, d, r):
    W = rs.rand(d, r).astype(np.float32)
    Hw = rs.rand(r, 300)
    A = (Hw @ Hw.T / 300).astype(np.float32)
    B = (Hw @ rs.rand(d, 300).T / 300).astype(np.float32)
    return W, A, B


def test_static_entry_matches_update_dict_on_two_calls_in_a_row():
    import torch
    from onmf_ontf_ndl_amd import _lib
    from onmf_ontf_ndl_amd.engine import get_engine, _ptr
    from oracle import onmf_oracle as orc
    eng = get_engine()
    lib = eng.lib
    rs = np.random.RandomState(123)
    first = True
    for d, r in ((1024, 256), (300, 100), (1000, 37), (64, 16)):        # each smaller than the first
        W, A, B = _state(rs, d, r)
        Wd, Ad, Bd = eng.dev(W), eng.dev(A), eng.dev(B)
        out = eng.empty(d, r)                       # outputs are the caller's, allocated before the call
        torch.cuda.synchronize()
        mem_before = torch.cuda.memory_allocated(eng.device)
        free_before, _ = torch.cuda.mem_get_info(eng.device)
        rc = lib.onmf_update_dict_static(eng.ctx, d, r, _ptr(Wd), _ptr(Ad), _ptr(Bd), _ptr(out), eng.stream())
        assert rc == 0, lib.onmf_last_error(eng.ctx)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info(eng.device)
        assert torch.cuda.memory_allocated(eng.device) == mem_before
        # the driver handed out no device memory either (the very first launch of a kernel makes the
        # runtime set up that kernel's code on the device: not the entry point's doing, so from call 2 on)
        assert first or free_after >= free_before
        first = False
        want = eng.update_dict(Wd, Ad, Bd)           # the existing entry point (context-owned workspace)
        assert torch.equal(out, want), (d, r)
        ref = orc.dictionary_pass(W.astype(np.float64), A.astype(np.float64), B.astype(np.float64))
        assert relerr(out.cpu().numpy(), ref) < 1e-4


def test_static_entry_rejects_sizes_beyond_its_array():
    from onmf_ontf_ndl_amd import _lib
    from onmf_ontf_ndl_amd.engine import get_engine, _ptr
    eng = get_engine()
    for d, r in ((1025, 16), (64, 257), (4096, 256)):
        W, A, B, out = eng.zeros(d, r), eng.zeros(r, r), eng.zeros(r, d), eng.zeros(d, r)
        rc = eng.lib.onmf_update_dict_static(eng.ctx, d, r, _ptr(W), _ptr(A), _ptr(B), _ptr(out), eng.stream())
        assert rc == -1
        assert b"static scratch" in eng.lib.onmf_last_error(eng.ctx)
        with pytest.raises(_lib.OnmfLibraryError):
            eng.update_dict_static(W, A, B)


def test_static_entry_calls_from_two_streams_are_ordered_on_the_gpu():
    """Two calls on two streams share the GPU's one scratch array: the library orders the second behind the
    first (an event per device), so both give what onmf_update_dict gives."""
    import torch
    from onmf_ontf_ndl_amd.engine import get_engine
    eng = get_engine()
    rs = np.random.RandomState(321)
    cases = [_state(rs, 1024, 256), _state(rs, 900, 200)]
    devs = [tuple(eng.dev(x) for x in c) for c in cases]
    outs = [eng.empty(1024, 256), eng.empty(900, 200)]
    streams = [torch.cuda.Stream(device=eng.device), torch.cuda.Stream(device=eng.device)]
    torch.cuda.synchronize()
    for rep in range(3):
        for (W, A, B), out, st in zip(devs, outs, streams):
            with torch.cuda.stream(st):
                eng.update_dict_static(W, A, B, out=out)
    torch.cuda.synchronize()
    for (W, A, B), out in zip(devs, outs):
        assert torch.equal(out, eng.update_dict(W, A, B))
    with pytest.raises(ValueError):
        eng.update_dict_static(*devs[0], out=eng.empty(256, 1024))
