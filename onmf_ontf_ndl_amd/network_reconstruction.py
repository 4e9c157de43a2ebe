"""The patch source of the reference's ``network_reconstruction_nx.py`` and its dictionary
training loop: ``Network_Reconstructor`` with its data path only (motif-sampling chains, k x k
adjacency patches, the warm-restarted ``Online_NMF`` chain) - no plotting, no files written.

Same constructor arguments and method names as the reference (network_reconstruction_nx.py:19-390).
The graph may be given as a path to a comma-separated edge list (as the reference reads it, :51-55),
as an (m, 2) integer array of edges, or as a networkx graph.  Nodes keep the reference's order
(first appearance in the edge list = ``nx.Graph(edgelist)`` order); embeddings are arrays of node
labels, as in the reference.

The chains run on the GPU, one chain per wavefront.  Their randomness is a counter-based device
generator keyed by ``seed`` - NumPy's global stream, which the reference draws from, cannot be
followed on a device; the transition rules are the reference's.  Extra keyword-only arguments:
``n_chains`` (default 1 = the reference's single sequential chain; more = independent replicas that
share the work of one ``get_patches_glauber`` call), ``seed``, ``rng`` / ``device`` (passed to
``Online_NMF``).
"""
from __future__ import annotations

import numpy as np

from .engine import get_engine
from .legacy.onmf import Online_NMF
from .onmf import _out

__all__ = ["Network_Reconstructor"]


class Network_Reconstructor():
    def __init__(self, source, n_components=100, MCMC_iterations=500, sub_iterations=100, loc_avg_depth=1,
                 sample_size=1000, batch_size=10, k1=1, k2=2, patches_file='', is_stack=False, alpha=None,
                 is_WAN=False, is_glauber_dict=True, is_glauber_recons=True, *, n_chains=1, seed=0, rng="numpy",
                 device=None):
        self.source = source
        self.n_components = n_components
        self.MCMC_iterations = MCMC_iterations
        self.sub_iterations = sub_iterations
        self.sample_size = sample_size
        self.batch_size = batch_size
        self.loc_avg_depth = loc_avg_depth
        self.k1 = k1
        self.k2 = k2
        self.patches_file = patches_file
        self.is_stack = is_stack
        self.W = np.zeros(shape=(k1 + k2 + 1, n_components))
        self.code = np.zeros(shape=(n_components, sample_size))
        self.alpha = alpha
        self.is_WAN = is_WAN
        self.is_glauber_dict = is_glauber_dict
        self.is_glauber_recons = is_glauber_recons
        self.n_chains = int(n_chains)
        self.seed = int(seed)
        self.rng = rng
        self._device = device
        self._step = 0
        self.nmf = None
        self.nodes, self.row_ptr, self.nbr = self.read_networks_as_graph(source)
        self._index = {int(v): i for i, v in enumerate(self.nodes)}
        self._graph = None

    # -- graph ----------------------------------------------------------------------------
    def read_networks_as_graph(self, path):
        """Edge list -> (node labels in first-appearance order, CSR row pointers, sorted neighbours)
        (network_reconstruction_nx.py:51-55 builds ``nx.Graph(edgelist)`` from the same file)."""
        if isinstance(path, (str, bytes)):
            edges = np.genfromtxt(path, delimiter=',', dtype=int)
        elif hasattr(path, "edges") and hasattr(path, "nodes"):
            nodes = list(path.nodes)
            edges = np.array([(u, v) for u, v in path.edges], dtype=np.int64).reshape(-1, 2)
            return self._csr(np.asarray(nodes, dtype=np.int64), edges)
        else:
            edges = np.asarray(path)
        edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        flat = edges.reshape(-1)
        _, first = np.unique(flat, return_index=True)
        nodes = flat[np.sort(first)]
        return self._csr(nodes, edges)

    @staticmethod
    def _csr(nodes, edges):
        order = np.argsort(nodes, kind="stable")
        pos = np.searchsorted(nodes[order], edges.reshape(-1))
        idx = order[pos].reshape(-1, 2)
        a = np.concatenate([idx[:, 0], idx[:, 1]])
        b = np.concatenate([idx[:, 1], idx[:, 0]])
        pairs = np.unique(np.stack([a, b], axis=1), axis=0)          # undirected, no multi-edges, rows sorted
        n = len(nodes)
        row_ptr = np.zeros(n + 1, dtype=np.int32)
        np.add.at(row_ptr, pairs[:, 0] + 1, 1)
        row_ptr = np.cumsum(row_ptr).astype(np.int32)
        return nodes, row_ptr, pairs[:, 1].astype(np.int32)

    def _engine(self):
        return get_engine(self._device)

    def _g(self):
        if self._graph is None:
            self._graph = self._engine().ndl_graph(self.row_ptr, self.nbr)
        return self._graph

    def _to_index(self, emb):
        emb = np.asarray(emb)
        return np.vectorize(self._index.__getitem__, otypes=[np.int32])(emb).astype(np.int32)

    def _next_steps(self, n):
        s = self._step
        self._step += int(n)
        return s

    # -- motif ----------------------------------------------------------------------------
    def path_adj(self, k1, k2):
        """Adjacency of the path motif: k1 nodes left and k2 right of the root (:86-95)."""
        if k1 == 0 or k2 == 0:
            return np.eye(max(k1, k2) + 1, k=1, dtype=int)
        A = np.eye(k1 + k2 + 1, k=1, dtype=int)
        A[k1, k1 + 1] = 0
        A[0, k1 + 1] = 1
        return A

    def find_parent(self, B, i):
        return int(np.min(np.nonzero(np.asarray(B)[:, i] == 1)[0]))

    def _check_motif(self, B):
        B = np.asarray(B)
        k = B.shape[0]
        if B.ndim != 2 or B.shape[1] != k or not 1 <= k <= 64:
            raise ValueError("the motif must be a k x k matrix with 1 <= k <= 64")
        if B.sum() != 0:
            for i in range(1, k):
                col = np.nonzero(B[:, i] == 1)[0]
                if len(col) == 0 or col.min() >= i:
                    raise ValueError("motif node %d has no parent before it (depth-first order needed)" % i)
        return B.astype(np.int32)

    # -- chains ---------------------------------------------------------------------------
    def tree_sample(self, B, x):
        """Embedding of the motif with its root on node x (:108-134)."""
        eng = self._engine()
        B = self._check_motif(B)
        emb = eng.ndl_init_embeddings(self._g(), B, 1, pivots=self._to_index([x]), seed=self.seed + self._next_steps(1))
        return self.nodes[emb.cpu().numpy()[0]]

    def _advance(self, B, emb, steps, glauber):
        eng = self._engine()
        B = self._check_motif(B)
        emb = np.asarray(emb)
        single = emb.ndim == 1
        e = eng.dev(self._to_index(emb).reshape(-1, B.shape[0]), eng.torch.int32)
        X = eng.ndl_sample_patches(self._g(), B, e, steps, glauber=glauber, seed=self.seed,
                                   step_offset=self._next_steps(steps))
        out = self.nodes[e.cpu().numpy()]
        return X, (out[0] if single else out)

    def glauber_gen_update(self, B, emb):
        return self._advance(B, emb, 1, True)[1]

    def Pivot_update(self, emb):
        return self._advance(self.path_adj(self.k1, self.k2), emb, 1, False)[1]

    def RW_update(self, x):
        return self._advance(np.zeros((1, 1), dtype=int), [x], 1, True)[1][0]

    def chd_gen_mx(self, B, emb, iterations=1000, is_glauber=True):
        """Mean adjacency patch over ``iterations`` chain steps and the final embedding (:280-313)."""
        X, emb = self._advance(B, emb, iterations, is_glauber)
        k = np.asarray(B).shape[0]
        return _out(X.mean(dim=1).reshape(k, k), False), emb

    def get_patches_glauber(self, B, emb):
        """``sample_size`` patches as columns (k*k x sample_size) and the embedding(s) to go on from (:315-329)."""
        X, emb = self._patches_dev(B, emb, self.is_glauber_dict)
        return _out(X, False), emb

    def get_single_patch_glauber(self, B, emb):
        X, emb = self._advance(B, emb, 1, self.is_glauber_recons)
        return _out(X, False), emb

    def _patches_dev(self, B, emb, glauber):
        emb = np.asarray(emb)
        chains = 1 if emb.ndim == 1 else emb.shape[0]
        steps = -(-int(self.sample_size) // chains)
        X, emb = self._advance(B, emb, steps, glauber)
        return X[:, :self.sample_size], emb

    def initial_embedding(self, B):
        """One embedding per chain from uniformly drawn pivots (train_dict's start, :351-352)."""
        eng = self._engine()
        B = self._check_motif(B)
        emb = eng.ndl_init_embeddings(self._g(), B, self.n_chains, pivots=None, seed=self.seed + self._next_steps(1))
        out = self.nodes[emb.cpu().numpy()]
        return out[0] if self.n_chains == 1 else out

    # -- training -------------------------------------------------------------------------
    def train_dict(self, filename=None):
        """``MCMC_iterations`` rounds of: ``sample_size`` fresh patches from the chain -> an ``Online_NMF``
        of ``sub_iterations`` warm-started from the previous round (:342-390).  Nothing is saved;
        ``filename`` is accepted for compatibility."""
        B = self.path_adj(self.k1, self.k2)
        emb = self.initial_embedding(B)
        W = At = Bt = Ct = None
        code = None
        for t in range(self.MCMC_iterations):
            X, emb = self._patches_dev(B, emb, self.is_glauber_dict)
            if t == 0:
                self.nmf = Online_NMF(X.contiguous(), self.n_components, iterations=self.sub_iterations,
                                      batch_size=self.batch_size, alpha=self.alpha, rng=self.rng, device=self._device)
            else:
                self.nmf = Online_NMF(X.contiguous(), self.n_components, iterations=self.sub_iterations,
                                      batch_size=self.batch_size, ini_dict=W, ini_A=At, ini_B=Bt, ini_C=Ct,
                                      alpha=self.alpha, history=self.nmf.history, rng=self.rng, device=self._device)
            W, At, Bt, Ct, H = self.nmf.train_dict()
            if t > 0:
                code = H.clone() if code is None else code + H          # (:378)
        self.W = _out(W, False)
        if code is not None:
            self.code = self.code + _out(code, False)
        self.emb = emb
        return self.W

    def reconstruct_network(self, recons_iter=100, *, return_weights=False, coder="lars", cd_tol=1e-6,
                            cd_max_sweeps=4000):
        """``recons_iter`` chain steps; every patch is coded against the dictionary (the reference's call:
        scikit-learn's positive LassoLars with transform_alpha = 0, network_reconstruction_nx.py:466-473),
        reconstructed and averaged onto the ordered node pairs its embedding covers; pairs whose mean rounds
        to 1 become the edges of the result (network_reconstruction_nx.py:444-511).  All patches are coded in
        one batched solve - ``coder="lars"``: the library's path followed event by event (the reference's
        codes); ``coder="cd"``: the non-negative least-squares problem itself by coordinate descent - and
        accumulated by one kernel into dense n x n sums.  Returns a ``networkx.Graph`` when
        networkx is importable, else an (m, 2) array of node-label pairs; ``return_weights`` adds the
        (n x n) mean and count matrices (node order = ``self.nodes``)."""
        eng = self._engine()
        n = len(self.nodes)
        if n > 46000:
            raise ValueError("reconstruct_network keeps dense %d x %d accumulators; the graph is too large" % (n, n))
        B = self._check_motif(self.path_adj(self.k1, self.k2))
        emb = eng.ndl_init_embeddings(self._g(), B, 1, pivots=None, seed=self.seed + self._next_steps(1))
        X, trace = eng.ndl_sample_patches(self._g(), B, emb, int(recons_iter), glauber=self.is_glauber_recons,
                                          seed=self.seed, step_offset=self._next_steps(recons_iter), want_trace=True)
        Wd = eng.dev(self.W)
        from . import _lib
        if coder == "lars" and Wd.shape[1] <= 256:
            H = eng.sparse_code_lars(X, Wd, alpha=0.0)
        elif coder in ("lars", "cd"):
            H0 = eng.zeros(Wd.shape[1], X.shape[1])
            # every patch is its own least-squares problem: one launch, each column with its own stopping test
            H = eng.sparse_code_each(X, Wd, H0=H0, alpha=0.0, sub_iter=cd_max_sweeps, stopping_diff=cd_tol,
                                     mode=_lib.CODER_EXACT_CD)
        else:
            raise ValueError("coder must be 'lars' or 'cd'")
        total, count = eng.ndl_accumulate_edges(n, trace, eng.patch_product(Wd, H))
        mean = eng.torch.where(count > 0, total / count.clamp(min=1), eng.torch.zeros_like(total))
        keep = eng.torch.round(mean) > 0                       # np.round: exactly 0.5 goes to 0
        pairs = eng.torch.nonzero(keep | keep.t()).cpu().numpy()
        pairs = pairs[pairs[:, 0] <= pairs[:, 1]]
        edges = np.stack([self.nodes[pairs[:, 0]], self.nodes[pairs[:, 1]]], axis=1) if len(pairs) else \
            np.zeros((0, 2), dtype=self.nodes.dtype)
        self.emb = self.nodes[emb.cpu().numpy()[0]]
        try:
            import networkx as nx
            out = nx.Graph()
            out.add_edges_from(map(tuple, edges.tolist()))
        except ImportError:                                   # pragma: no cover
            out = edges
        if return_weights:
            return out, _out(mean, False), _out(count, False)
        return out
