#!/usr/bin/env python3
"""Throughput of the motif-sampling chains (get_patches_glauber, network_reconstruction_nx.py:315-329)
on a synthetic sparse graph.  Prints one JSON line per configuration.

    python tools/bench_ndl.py [--nodes 20000] [--degree 20] [--k 21]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=20000)
    ap.add_argument("--degree", type=int, default=20)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--cpu-steps", type=int, default=200)
    args = ap.parse_args()
    import torch
    from onmf_ontf_ndl_amd.engine import get_engine
    from oracle import ndl_oracle as nd
    eng = get_engine()
    rs = np.random.RandomState(0)
    n = args.nodes
    m = n * args.degree // 2
    edges = np.concatenate([np.stack([np.arange(n), (np.arange(n) + 1) % n], axis=1), rs.randint(n, size=(m, 2))])
    edges = edges[edges[:, 0] != edges[:, 1]]
    G = nd.Graph(n, edges)
    g = eng.ndl_graph(*G.csr())
    B = nd.path_adj(0, args.k - 1)
    k = args.k
    for glauber in (True, False):
        for chains, steps in ((1, 1000), (64, 1000), (1024, 64), (8192, 8), (65536, 1)):
            emb = eng.ndl_init_embeddings(g, B, chains, seed=1)
            eng.ndl_sample_patches(g, B, emb, min(steps, 8), glauber=glauber, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            X = eng.ndl_sample_patches(g, B, emb, steps, glauber=glauber, seed=1, step_offset=8)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(json.dumps({"chain": "glauber" if glauber else "pivot", "k": k, "chains": chains, "steps": steps,
                              "patches": chains * steps, "ms": el * 1e3, "patches_per_s": chains * steps / el,
                              "write_GBps": k * k * 4.0 * chains * steps / el / 1e9}), flush=True)
        emb = nd.init_embedding(G, B, 1, 0)
        t0 = time.perf_counter()
        nd.sample_patches(G, B, emb, glauber, args.cpu_steps, 1)
        el = time.perf_counter() - t0
        print(json.dumps({"chain": "glauber" if glauber else "pivot", "cpu_port_patches_per_s": args.cpu_steps / el}),
              flush=True)


if __name__ == "__main__":
    main()
