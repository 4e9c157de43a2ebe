"""CPU oracle for the motif-sampling chains of network dictionary learning (TEST INFRASTRUCTURE,
not product code).

Restatement of the patch source in the reference's ``network_reconstruction_nx.py``:

* ``path_adj``        <- ``Network_Reconstructor.path_adj``            (:86-95)
* ``find_parent``     <- ``find_parent``                               (:100-106)
* ``tree_sample``     <- ``tree_sample``                               (:108-134)
* ``glauber_update``  <- ``glauber_gen_update``                        (:136-173)
* ``rw_update``       <- ``RW_update``                                 (:175-199)
* ``pivot_update``    <- ``Pivot_update``                              (:265-278)
* ``patch``           <- the induced-adjacency loop of ``chd_gen_mx``  (:302-305)
* ``sample_patches``  <- ``get_patches_glauber``                       (:315-329)

The reference draws from NumPy's global Mersenne-Twister stream (``np.random.choice`` /
``np.random.rand``); a device cannot follow that stream, so the product path draws from
Philox4x32-10 keyed by (seed, chain, step, slot) and this restatement takes the SAME draws
(``philox4x32`` below): every embedding and patch is then reproduced bit for bit.  What is restated
is the transition rule; a uniform choice among n items is ``(draw * n) >> 32`` and the Metropolis
test ``rand() > deg_x/deg_y`` is ``draw * deg_y > deg_x << 32``.

PARITY UNPINNED: ``network_reconstruction_nx.py`` cannot be imported in the build container
(``seaborn`` is not installed) and the reference holds no fixtures for it, so the transition rules
are restated from the text and checked through their invariants (valid homomorphisms, stationary
laws on small graphs) rather than against outputs of the reference.

Only ``tests/`` may import this module.
"""
import numpy as np

M32 = 0xFFFFFFFF


def philox4x32(counter, seed):
    """Philox4x32-10: counter = 4 uint32, key = the two halves of a 64-bit seed -> 4 uint32."""
    c = [int(x) & M32 for x in counter]
    k0, k1 = seed & M32, (seed >> 32) & M32
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c[3] ^ k1) & M32, p0 & M32]
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c


def fill_uniform(n, seed):
    """The float32 U[0,1) stream of the library's uniform fill: element 4q+e = word e of counter (q, 0)."""
    out = np.empty(4 * ((n + 3) // 4), dtype=np.float32)
    for q in range((n + 3) // 4):
        w = philox4x32([q & M32, q >> 32, 0, 0], seed)
        for e in range(4):
            out[4 * q + e] = np.float32(w[e] >> 8) * np.float32(1.0 / 16777216.0)
    return out[:n]


INIT_STEP = (1 << 64) - 1


def draw(seed, step, chain, slot):
    return philox4x32([step & M32, (step >> 32) & M32, chain, slot >> 2], seed)[slot & 3]


def scaled(x, n):
    return (x * n) >> 32


class Graph:
    """Undirected simple graph on nodes 0..n-1 with sorted neighbour lists."""

    def __init__(self, n, edges):
        nb = [set() for _ in range(n)]
        for a, b in edges:
            nb[int(a)].add(int(b))
            nb[int(b)].add(int(a))
        self.n = n
        self.nbrs = [sorted(s) for s in nb]
        self.sets = nb

    def csr(self):
        row_ptr = np.zeros(self.n + 1, dtype=np.int32)
        for i, l in enumerate(self.nbrs):
            row_ptr[i + 1] = row_ptr[i] + len(l)
        nbr = np.array([v for l in self.nbrs for v in l], dtype=np.int32)
        return row_ptr, nbr

    def has_edge(self, a, b):
        return b in self.sets[a]


def path_adj(k1, k2):
    """Adjacency of the path motif with k1 nodes left and k2 right of the root (:86-95)."""
    if k1 == 0 or k2 == 0:
        return np.eye(max(k1, k2) + 1, k=1, dtype=int)
    A = np.eye(k1 + k2 + 1, k=1, dtype=int)
    A[k1, k1 + 1] = 0
    A[0, k1 + 1] = 1
    return A


def find_parent(B, i):
    return int(np.min(np.nonzero(B[:, i] == 1)[0]))


def tree_sample(G, B, x, seed, step, chain):
    k = B.shape[0]
    emb = [x]
    if B.sum() == 0:
        return emb + [scaled(draw(seed, step, chain, 4 + i), G.n) for i in range(1, k)]
    for i in range(1, k):
        nbs = G.nbrs[emb[find_parent(B, i)]]
        emb.append(nbs[scaled(draw(seed, step, chain, 4 + i), len(nbs))] if nbs else emb[find_parent(B, i)])
    return emb


def rw_update(G, x, d0, d1):
    nx_ = G.nbrs[x]
    if not nx_:
        return scaled(d0, G.n)
    y = nx_[scaled(d0, len(nx_))]
    if d1 * len(G.nbrs[y]) > (len(nx_) << 32):
        y = x
    return y


def glauber_update(G, B, emb, seed, step, chain):
    k = B.shape[0]
    d0, d1, d2 = (draw(seed, step, chain, s) for s in range(3))
    emb = list(emb)
    if k == 1:
        emb[0] = rw_update(G, emb[0], d0, d1)
        return emb
    j = scaled(d0, k)
    nbh = sorted(set(np.nonzero(B[:, j] == 1)[0].tolist()) | set(np.nonzero(B[j, :] == 1)[0].tolist()))
    if not nbh:
        emb[j] = scaled(d1, G.n)
        return emb
    common = [c for c in G.nbrs[emb[nbh[0]]] if all(G.has_edge(emb[r], c) for r in nbh[1:])]
    emb[j] = common[scaled(d1, len(common))] if common else scaled(d2, G.n)
    return emb


def pivot_update(G, B, emb, seed, step, chain):
    x0 = rw_update(G, emb[0], draw(seed, step, chain, 0), draw(seed, step, chain, 1))
    return tree_sample(G, B, x0, seed, step, chain)


def patch(G, emb):
    k = len(emb)
    return np.array([[1.0 if G.has_edge(emb[q], emb[r]) else 0.0 for r in range(k)] for q in range(k)])


def init_embedding(G, B, seed, chain, pivot=None):
    x = scaled(draw(seed, INIT_STEP, chain, 0), G.n) if pivot is None else int(pivot)
    return tree_sample(G, B, x, seed, INIT_STEP, chain)


def sample_patches(G, B, emb, glauber, steps, seed, chain=0, step_offset=0):
    """``steps`` chain updates from ``emb``; returns (k*k x steps patches, final embedding)."""
    k = B.shape[0]
    X = np.zeros((k * k, steps))
    for s in range(steps):
        step = step_offset + s
        emb = glauber_update(G, B, emb, seed, step, chain) if glauber else pivot_update(G, B, emb, seed, step, chain)
        X[:, s] = patch(G, emb).reshape(-1)
    return X, emb


def reconstruct_network(G, B, W, emb, glauber, recons_iter, seed, chain=0, step_offset=0, coder=None):
    """The loop of ``reconstruct_network`` (network_reconstruction_nx.py:464-491): per chain step, code
    the patch (``coder(patch_column, W) -> code column``; the reference uses positive LassoLars with
    alpha = 0), rebuild it and fold its entries into a running mean per ordered node pair.  Returns
    (mean, count) as dicts keyed by (a, b), the thresholded undirected edge set (:500-508) and the
    final embedding."""
    k = B.shape[0]
    mean, count = {}, {}
    for s in range(recons_iter):
        step = step_offset + s
        emb = glauber_update(G, B, emb, seed, step, chain) if glauber else pivot_update(G, B, emb, seed, step, chain)
        x = patch(G, emb).reshape(-1, 1)
        rec = (W @ coder(x, W)).reshape(k, k)
        for q in range(k):
            for r in range(k):
                key = (emb[q], emb[r])
                j = count.get(key, 0)
                mean[key] = (j * mean.get(key, 0.0) + rec[q, r]) / (j + 1)
                count[key] = j + 1
    edges = {tuple(sorted(key)) for key, v in mean.items() if np.round(v) > 0}
    return mean, count, edges, emb
