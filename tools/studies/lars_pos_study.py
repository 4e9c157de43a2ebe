#!/usr/bin/env python3
"""CPU study: how scikit-learn's positive LassoLars path ends, branch by branch.

Compares a NumPy restatement of the path (oracle.onmf_oracle.lars_pos_path) with scikit-learn's
own output on random problems of the shapes Online_NTF produces, and logs which branch ended the path."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import onmf_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")


def main():
    from collections import Counter
    for (d, r, b, alpha, seed, normalise) in [(48, 10, 24, 2, 19, True), (48, 10, 24, 0.5, 19, True),
                                              (48, 10, 24, 0.05, 19, True), (75, 8, 200, 0.5, 1, False),
                                              (75, 12, 300, 0.1, 2, False), (1200, 100, 64, 2, 3, False),
                                              (300, 40, 128, 0.2, 4, True)]:
        rs = np.random.RandomState(seed)
        W = rs.rand(d, r)
        if normalise:
            W /= np.linalg.norm(W, axis=0)
        else:
            W = orc.dictionary_pass(W, np.zeros((r, r)), np.zeros((r, d)))
        X = rs.rand(d, b)
        Hs = orc.lasso_nonneg_code(X, W, alpha, solver="sklearn")
        info = []
        Hl = orc.lars_pos_code(X, W, alpha, info=info)
        Hc = orc.lasso_nonneg_code(X, W, alpha, solver="cd", sweeps=20000)
        ends = Counter(i["end"] for i in info)
        print("d=%d r=%d b=%d alpha=%g: |lars-sk|max %.3e  |cd-sk|max %.3e  iters max %d drops %d  ends %s" % (
            d, r, b, alpha, np.abs(Hl - Hs).max(), np.abs(Hc - Hs).max(), max(i["n_iter"] for i in info),
            sum(i["drops"] for i in info), dict(ends)))


if __name__ == "__main__":
    main()
