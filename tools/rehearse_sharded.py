#!/usr/bin/env python3
"""Rehearse dist.sharded_step with several ranks on whatever GPUs there are (ranks may share a card
under the host backend):

    ONMF_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29531 tools/rehearse_sharded.py

Every rank codes its slice of one minibatch; the result (aggregates, dictionary, code) is compared
on every rank with the single-GPU step on the whole minibatch, for three consecutive steps.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as td
    from onmf_ontf_ndl_amd import dist as odist
    from onmf_ontf_ndl_amd.engine import get_engine
    rank, world, local = odist.init_from_env()
    eng = get_engine(torch.device("cuda", local))
    d, r, b = 256, 64, 8192
    g = torch.Generator(device=eng.device).manual_seed(5)          # same data on every rank
    Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(3)]
    H0s = [torch.rand(r, b, device=eng.device, generator=g) for _ in range(3)]
    W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
    A, B = eng.zeros(r, r), eng.zeros(r, d)
    Wf, Af, Bf = W, A, B
    lo, hi = odist.split_columns(b, rank, world)
    worst = 0.0
    for t in range(1, 4):
        X, H0 = Xs[t - 1], H0s[t - 1]
        Hl, A, B, W, _ = odist.sharded_step(eng, X[:, lo:hi].contiguous(), A, B, W, 1.0 / t,
                                            H0=H0[:, lo:hi].contiguous(), sub_iter=10, stopping_diff=0.0)
        Hf, Af, Bf, _, Wf, _, _ = eng.step(X, Af, Bf, None, Wf, 1.0 / t, H0=H0, sub_iter=10, stopping_diff=0.0)
        for name, got, ref in (("H", Hl, Hf[:, lo:hi]), ("A", A, Af), ("B", B, Bf), ("W", W, Wf)):
            err = float(torch.linalg.norm(got - ref) / torch.linalg.norm(ref))
            worst = max(worst, err)
            print("rank %d step %d %s rel err %.2e" % (rank, t, name, err), flush=True)
    assert worst <= 1e-5, worst
    print("rank %d ok (worst %.2e)" % (rank, worst), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
