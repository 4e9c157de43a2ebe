#!/usr/bin/env python3
"""Throughput of the ONMF minibatch step (sparse code + aggregates + dictionary pass) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            (starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one resident synthetic batch (X ~ U[0,1), d x b).
At N > 1 every rank runs its own patch stream (weak scaling) and the ranks average the packed
state [A|B|W] over RCCL every --sync-every steps, inside the timed region (--mode sharded: one
minibatch of N*b columns split over the ranks, partial sums all-reduced every step).
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (d, r, b)
    "headline": (1024, 256, 65536),      # the shape the 50 %-of-roofline target is quoted on
    "c2": (1024, 64, 8192),              # the reference-scale case (r = 64)
    "c4": (4096, 256, 65536),            # configs[3], per GPU
    "c3": (1200, 100, 8192),
    "c1": (100, 16, 1000),
    "c5": (441, 25, 1000),
    # Online_NTF.step (lasso coder, alpha = 2) on mode-2 joint unfoldings of 20 x 20 x 3 patches
    "c3-ntf": (1200, 100, 8192),
    "c3-ntf-100": (1200, 100, 100),      # the batch the reference's main uses
}
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0
SWEEPS = 10
SWEEP_KERNEL_SOURCE = os.path.join(ROOT, "onmf_ontf_ndl_amd", "csrc", "sweep_tile.h")


def step_flops(d, r, b, sweeps):
    """W^T X + H X^T + sweeps + H H^T + W^T W + dictionary pass (flops, fused multiply-add = 2)."""
    return 4.0 * d * r * b + 2.0 * r * r * b * (sweeps + 1) + 4.0 * d * r * r


def host_description():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    pools = []
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: p.get(k) for k in ("user_api", "internal_api", "num_threads", "version")}
                 for p in threadpool_info()]
    except Exception:
        pass
    return model, pools


def cpu_baseline(d, r, budget_cols):
    """Time the NumPy oracle (float64, spectral-norm test by SVD like the reference) on a bounded
    sample of the same workload: same d and r, fewer columns."""
    from oracle import onmf_oracle as orc
    model, pools = host_description()
    threads = max([p.get("num_threads") or 1 for p in pools] or [os.cpu_count() or 1])
    rs = np.random.RandomState(1234)
    b = budget_cols
    X = rs.rand(d, b)
    W = rs.rand(d, r)
    W /= np.maximum(1, np.linalg.norm(W, axis=0))
    A, B = np.zeros((r, r)), np.zeros((r, d))
    # one untimed warm step so (A,B) are non-zero, then timed steps
    _, (A, B), W, _ = orc.minibatch_step(X, [A, B], W, 1, H0=rs.rand(r, b))

    def timed(norm, A, B, W):
        t0 = time.perf_counter()
        n = 0
        while True:
            tr = []
            _, (A, B), W, _ = orc.minibatch_step(X, [A, B], W, n + 2, H0=rs.rand(r, b), norm=norm, trace=tr)
            n += 1
            el = time.perf_counter() - t0
            if el > 10.0 or n >= 5:
                return n, el, len(tr)

    n, el, sweeps = timed("svd", A, B, W)
    # the same steps with the stopping test's norms taken from the r x r Gram matrices instead of two
    # SVDs of r x b: what a CPU implementation tuned for speed would do (same numbers, same sweeps)
    n_g, el_g, _ = timed("gram", A, B, W)
    return {"value": b * n / el, "unit": "patches/s", "cores": int(threads), "kind": "port",
            "sample": "oracle/onmf_oracle.py minibatch_step, float64, SVD stopping test, d=%d r=%d, "
                      "%d steps of %d columns (%d sweeps in the last), %.1f s" % (d, r, n, b, sweeps, el),
            "value_gram_norm_test": b * n_g / el_g, "cpu_model": model, "os_cpu_count": os.cpu_count(),
            "threadpools": pools}


def parity_block(eng, X, A, B, W, t, cols=512):
    """Outside the timed region: one step on a slice of a bench batch at a fixed sweep count, on the GPU
    and by the float64 oracle, from the state the timed steps ended in."""
    from oracle import onmf_oracle as orc
    d, r = W.shape
    Xs = X[:, :cols].contiguous()
    rs = np.random.RandomState(99)
    H0 = rs.rand(r, cols).astype(np.float32)
    host = lambda t_: t_.detach().cpu().numpy().astype(np.float64)
    Xh, Ah, Bh, Wh = host(Xs), host(A), host(B), host(W)
    H, A1, B1, _, W1, sweeps, _ = eng.step(Xs, A, B, None, W, 1.0 / t, alpha=0.0, H0=eng.dev(H0), sub_iter=SWEEPS,
                                           stopping_diff=0.0)
    H_o, (A_o, B_o), W_o, _ = orc.minibatch_step(Xh, [Ah, Bh], Wh, t, H0=H0.astype(np.float64), sub_iter=SWEEPS,
                                                 stopping_diff=-1.0, norm="gram")
    rel = lambda a, ref: float(np.linalg.norm(host(a) - ref) / max(np.linalg.norm(ref), 1e-300))
    res = lambda Wm, Hm: float(np.linalg.norm(Xh - Wm @ Hm) / np.linalg.norm(Xh))
    return {"what": "one step on %d columns of a bench batch from the state after the timed steps, %d sweeps, "
                    "GPU (float32) against oracle/onmf_oracle.py (float64)" % (cols, SWEEPS),
            "sweeps_gpu": int(sweeps), "sweeps_oracle": SWEEPS,
            "rel_err_H": rel(H, H_o), "rel_err_A": rel(A1, A_o), "rel_err_B": rel(B1, B_o), "rel_err_W": rel(W1, W_o),
            "residual_gpu": res(host(W), host(H)), "residual_oracle": res(Wh, H_o)}


def measured_traffic(d, r, b, sweeps_per_launch):
    """HBM bytes per sweep launch from the committed PMC passes (profiles/), when they are for this shape
    AND for the sweep kernel as it is now (the record carries the SHA-256 of the kernel's source file).

    PMC counters cannot be collected from inside this process; the file holds the two rocprofv3
    passes (FETCH_SIZE, WRITE_SIZE) of the same kernel at the same shape, per sweep, and how they
    were corrected.  A launch of n sweeps reads H once, Cm n times and writes H n times.
    """
    prof = os.path.join(ROOT, "profiles")
    try:
        names = sorted(n for n in os.listdir(prof) if n.endswith("_sweep_hbm_traffic.json"))
        with open(os.path.join(prof, names[-1])) as f:
            rec = json.load(f)
        with open(SWEEP_KERNEL_SOURCE, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
    except (OSError, ValueError, IndexError):
        return None
    if rec.get("shape") != {"d": d, "r": r, "b": b} or "hbm_bytes_per_sweep_in_launch" not in rec:
        return None
    if rec.get("kernel_source_sha256") != sha:
        return None                      # the kernel changed since the counters were read
    return rec["hbm_bytes_first_read"] + sweeps_per_launch * rec["hbm_bytes_per_sweep_in_launch"]


def measured_step_hbm(d, r, b, ms_per_step):
    """Whole-step HBM figure: bytes per step over ALL kernels of a step, from the committed FETCH_SIZE / WRITE_SIZE
    passes of this same command (profiles/*_step_hbm_traffic.json, made by tools/step_hbm.py), divided by
    this run's time per step.  None when no record for this shape is there; `sources_unchanged` tells whether
    the kernels' sources are the ones the counters were read from."""
    prof = os.path.join(ROOT, "profiles")
    try:
        names = sorted(n for n in os.listdir(prof) if n.endswith("_step_hbm_traffic.json"))
        with open(os.path.join(prof, names[-1])) as f:
            rec = json.load(f)
        if rec.get("shape") != {"d": d, "r": r, "b": b}:
            return None
        sha = csrc_sha256()
    except (OSError, ValueError, IndexError, KeyError):
        return None
    by = rec["hbm_bytes_per_step"]
    return {"bytes_per_step": by, "achieved": by / (ms_per_step * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": by / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "algorithmic_bytes_per_step": rec.get("algorithmic_bytes_per_step"),
            "source": "profiles/" + names[-1], "sources_unchanged": rec.get("csrc_sha256") == sha}


def csrc_sha256():
    """One hash over the library's sources (names and contents, in name order)."""
    src = os.path.join(ROOT, "onmf_ontf_ndl_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(src)):
        if n.endswith((".hip", ".h", ".cpp")):
            h.update(n.encode())
            with open(os.path.join(src, n), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--stopping-diff", type=float, default=0.01,
                    help="0.01 = the reference's sparse_code setting; 0 = always 10 sweeps")
    ap.add_argument("--sync-every", type=int, default=4, help="steps between (A,B,W) all-reduces at N>1")
    ap.add_argument("--mode", default="streams", choices=["streams", "sharded"],
                    help="N>1: independent streams with periodic averaging, or one minibatch split by columns")
    ap.add_argument("--batches", type=int, default=3, help="distinct resident batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short extra measurement of the r = 64 case that is reported under 'also'")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="library option (ONMF_OPT_* number = value), for experiments")
    ap.add_argument("--cpu-cols", type=int, default=0, help="columns of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-defer", action="store_true",
                    help="read every step's sweep count (one host wait per step) instead of letting the library look "
                         "at a step's stopping-test verdicts at the next call (ONMF_OPT_DEFER_VERDICT)")
    ap.add_argument("--coder", default="lars", choices=["lars", "cd"],
                    help="c3-ntf workloads: scikit-learn's positive LARS path (the reference's coder) or exact "
                         "coordinate descent on the same convex problem")
    return ap.parse_args(argv)


def launch_ranks(n, argv, script=None):
    """Start n copies of this script (or ``script``), one rank each, and wait for them.  Runs before anything
    in this process has touched a GPU (nothing here imports torch).  Returns the first non-zero exit code of
    a rank (the other ranks are then terminated: they would wait for it forever), else 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env))
    rc = 0
    pending = list(procs)
    while pending:
        time.sleep(0.2)
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:            # one rank failed: the others would wait for it forever
                    q.terminate()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as td
    from onmf_ontf_ndl_amd import dist as odist
    from onmf_ontf_ndl_amd import _lib
    from onmf_ontf_ndl_amd.engine import get_engine

    rank, world, local = odist.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but the process group has %d rank(s); refusing to report a line "
                         "for another GPU count" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    torch.cuda.set_device(local)
    pinned = odist.pin_to_local_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local) if world > 1 else None
    eng = get_engine(torch.device("cuda", local))
    for kv in args.opt:
        key, value = kv.split("=")
        eng.set_option(int(key), int(value))
    d, r, b = WORKLOADS[args.workload]
    ntf = args.workload.startswith("c3-ntf")
    backend = td.get_backend() if world > 1 else None

    gen = torch.Generator(device=eng.device).manual_seed(1234 + rank)
    Xs = [torch.rand(d, b, device=eng.device, generator=gen) for _ in range(max(1, args.batches))]
    gw = torch.Generator(device=eng.device).manual_seed(4321)
    W0 = torch.rand(d, r, device=eng.device, generator=gw)
    W0 = eng.update_dict(W0, eng.zeros(r, r), eng.zeros(r, d))   # clip + normalise, as the reference's first step does
    # the state lives in two packed [A|B|W] buffers: a step reads one and writes the other, the
    # all-reduce at N > 1 is one in-place collective on the buffer just written
    state = [odist.PackedState(d, r, eng.device), odist.PackedState(d, r, eng.device)]
    state[0].W.copy_(W0)
    cur = 0
    t = 1
    sweeps_seen = []
    stepper = None
    if world > 1 and args.mode == "sharded":
        stepper = odist.ShardedStepper(eng, state[0].W, state[0].A, state[0].B)
    defer = not args.no_defer and not ntf and stepper is None
    if defer:
        eng.set_option(_lib.OPT_DEFER_VERDICT, 1)
    coder = None
    if ntf:
        from onmf_ontf_ndl_amd import Online_NTF
        coder = Online_NTF(torch.zeros(2, 3, 4), n_components=r, alpha=None, device=eng.device, coder=args.coder)

    def one_step(i):
        nonlocal cur, t
        X = Xs[i % len(Xs)]
        if ntf:
            st = state[cur]
            _, A1, B1, W1 = coder.step(X, st.A, st.B, st.W, t)
            state[1 - cur].load(W1, A1, B1)
            s = 0
        elif stepper is not None:
            stepper.step(X, 1.0 / t, seed=1000 * rank + t, sub_iter=SWEEPS, stopping_diff=args.stopping_diff,
                         want_code=False)
            s = stepper.sweeps
        else:
            st, nx = state[cur], state[1 - cur]
            _, _, _, _, _, s, _ = eng.step(X, st.A, st.B, None, st.W, 1.0 / t, alpha=0.0, H0=None,
                                           seed=1000 * rank + t, sub_iter=SWEEPS,
                                           stopping_diff=args.stopping_diff, want_code=False, out=nx.views(),
                                           defer=defer)
        cur = 1 - cur
        t += 1
        if world > 1 and stepper is None and args.sync_every > 0 and (t % args.sync_every) == 0:
            if defer:
                eng.settle()         # the exchange reads the step's outputs: they have to be final
            odist.average_state_(state[cur])
        return s

    for i in range(max(args.warmup, 0)):
        one_step(i)
    eng.settle()
    eng.set_option(_lib.OPT_PROFILE, 1)
    eng.get_profile()
    sweeps_before = eng.counters()["sweeps_total"]
    if world > 1:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sweeps_seen.append(one_step(args.warmup + i))
    eng.settle()                 # (a step that was not waited for is settled inside the timed region)
    torch.cuda.synchronize()
    if world > 1:
        td.barrier()
    elapsed = time.perf_counter() - t0
    sweep_ms, sweep_launches = eng.get_profile()
    eng.set_option(_lib.OPT_PROFILE, 0)
    if defer:
        # the steps returned no counts: the library's own total over the timed steps
        total = eng.counters()["sweeps_total"] - sweeps_before
        sweeps_seen = [total / max(1, args.steps)] * args.steps
        eng.set_option(_lib.OPT_DEFER_VERDICT, 0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else eng.device)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = 1e3 * elapsed / max(1, args.steps)
        value = world * b * args.steps / elapsed
        mean_sweeps = float(np.mean(sweeps_seen)) if sweeps_seen else 0.0
        if ntf:
            if args.coder == "lars":
                desc = ("%s: Online_NTF.step, d=%d r=%d batch=%d per GPU, lasso coder alpha=2: scikit-learn's positive "
                        "LARS path per column (the reference's coder), (A,B,W) carried" % (args.workload, d, r, b))
                _, info = eng.sparse_code_lars(Xs[0], state[cur].W, alpha=2.0, want_info=True)
                info = info.cpu().numpy()
                cfg = {"workload": desc, "lars_events_per_column_mean": float(info[:, 0].mean()),
                       "lars_events_per_column_max": int(info[:, 0].max()),
                       "lars_active_atoms_mean": float(info[:, 2].mean())}
            else:
                desc = ("%s: Online_NTF.step, d=%d r=%d batch=%d per GPU, lasso coder alpha=2 by coordinate descent "
                        "(cd_tol=%g, at most %d sweeps per column), (A,B,W) carried"
                        % (args.workload, d, r, b, coder.cd_tol, coder.cd_max_sweeps))
                cfg = {"workload": desc, "sweeps_slowest_column_last_step": coder.last_sweeps}
        else:
            flops = step_flops(d, r, b, mean_sweeps)
            cfg = {"workload": "%s: d=%d r=%d batch=%d per GPU, alpha=0 beta=1 sub_iter=%d "
                               "stopping_diff=%g, device-drawn H0, (A,B,W) carried"
                               % (args.workload, d, r, b, SWEEPS, args.stopping_diff),
                   "mean_sweeps": mean_sweeps,
                   "host_waits_per_step": 0 if defer else 1,
                   "step_tflops_per_gpu": flops / (ms * 1e-3) / 1e12,
                   "step_frac_of_fp32_mfma_peak": flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        if world > 1:
            cfg.update({"multi_gpu_mode": args.mode, "backend": backend,
                        "sync_every": args.sync_every if stepper is None else 1,
                        "rank0_cpu_cores": len(pinned) if pinned else None})
        cfg["recovery_counters"] = eng.counters()
        out = {
            "metric": "patches/sec per ONMF minibatch step (d x r dict)",
            "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg,
        }
        if sweep_launches > 0 and not ntf:
            # one launch runs one or several sweeps (all of a solve, when its grid fits the chip)
            sweeps_per_launch = float(np.sum(sweeps_seen)) / sweep_launches
            per_launch = 2.0 * r * r * b * sweeps_per_launch  # algorithmic flops of an average launch
            avg_ms = sweep_ms / sweep_launches
            ach = per_launch / (avg_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "code sweeps (one launch = the sweeps of one solve over the batch)",
                               "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                               "traffic": measured_traffic(d, r, b, sweeps_per_launch),
                               "avg_launch_ms": avg_ms, "launches": sweep_launches,
                               "sweeps_per_launch": sweeps_per_launch}
        if world == 1 and not ntf:
            hb = measured_step_hbm(d, r, b, ms)
            if hb is not None:
                out["step_hbm"] = hb
        if world == 1 and args.workload == "headline" and not args.no_also:
            # the small reference-scale case (d=1024, r=64, batch=8192) beside the headline number: it is
            # bound by launches and the host's part in borderline stopping tests, not by the kernels
            d2, r2, b2 = WORKLOADS["c2"]
            X2 = torch.rand(d2, b2, device=eng.device, generator=gen)
            W2 = eng.update_dict(torch.rand(d2, r2, device=eng.device, generator=gw), eng.zeros(r2, r2), eng.zeros(r2, d2))
            A2, B2 = eng.zeros(r2, r2), eng.zeros(r2, d2)
            t2 = 1
            for i in range(25):
                if i == 5:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                _, A2, B2, _, W2, _, _ = eng.step(X2, A2, B2, None, W2, 1.0 / t2, alpha=0.0, H0=None, seed=t2,
                                                  sub_iter=SWEEPS, stopping_diff=args.stopping_diff, want_code=False)
                t2 += 1
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            out["also"] = {"workload": "c2: d=%d r=%d batch=%d, same settings, 20 steps after 5" % (d2, r2, b2),
                           "ms_per_step": 1e3 * el2 / 20, "value": b2 * 20 / el2, "unit": "patches/s"}
        if world == 1 and not ntf and not args.no_parity:
            st = state[cur]
            out["parity"] = parity_block(eng, Xs[0], st.A.clone(), st.B.clone(), st.W.clone(), t)
        if world == 1 and not ntf and not args.no_cpu_baseline:
            cols = args.cpu_cols or min(b, 4096)
            out["cpu_baseline"] = cpu_baseline(d, r, cols)
        print(json.dumps(out), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
