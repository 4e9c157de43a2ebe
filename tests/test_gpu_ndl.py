"""GPU parity of the motif-sampling chains and their adjacency patches against the CPU oracle fed the
same Philox draws: integer work, bit-exact (embeddings and 0/1 patches).  Also the device uniform fill
against the oracle's Philox (pins that both sides draw the same numbers), chain invariants and the
stationary law of the pivot on a small graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from onmf_ontf_ndl_amd.engine import get_engine
    return get_engine()


@pytest.fixture(scope="module")
def nd():
    from oracle import ndl_oracle
    return ndl_oracle


def small_graph(nd, n, extra, seed, isolated=0):
    rs = np.random.RandomState(seed)
    m = n - isolated
    edges = [(i, (i + 1) % m) for i in range(m)] + [tuple(rs.randint(m, size=2)) for _ in range(extra)]
    edges = [(a, b) for a, b in edges if a != b]
    return nd.Graph(n, edges)


def test_uniform_fill_is_the_oracles_philox(eng, nd):
    for n, seed in ((10, 1), (257, 0xDEADBEEFCAFE), (4, 2 ** 63 + 5)):
        got = eng.rand(1, n, seed=seed).cpu().numpy().reshape(-1)
        assert np.array_equal(got, nd.fill_uniform(n, seed))


@pytest.mark.parametrize("k1,k2", [(0, 5), (2, 3), (0, 20), (0, 0), (3, 0), (0, 1)])
@pytest.mark.parametrize("glauber", [True, False])
def test_chains_equal_the_oracle_bit_for_bit(eng, nd, k1, k2, glauber):
    G = small_graph(nd, 60, 90, seed=k1 * 7 + k2, isolated=2)
    row_ptr, nbr = G.csr()
    g = eng.ndl_graph(row_ptr, nbr)
    B = nd.path_adj(k1, k2)
    k = B.shape[0]
    seed, chains, steps = 1234 + k, 5, 37
    emb = eng.ndl_init_embeddings(g, B, chains, seed=seed)
    ref0 = [nd.init_embedding(G, B, seed, c) for c in range(chains)]
    assert np.array_equal(emb.cpu().numpy(), np.array(ref0))
    X = eng.ndl_sample_patches(g, B, emb, steps, glauber=glauber, seed=seed, step_offset=11).cpu().numpy()
    for c in range(chains):
        Xr, er = nd.sample_patches(G, B, ref0[c], glauber, steps, seed, chain=c, step_offset=11)
        assert np.array_equal(emb.cpu().numpy()[c], np.array(er)), (c, emb.cpu().numpy()[c], er)
        assert np.array_equal(X[:, c * steps:(c + 1) * steps], Xr)


def test_chain_from_given_pivots_and_isolated_nodes(eng, nd):
    G = small_graph(nd, 30, 20, seed=3, isolated=3)
    row_ptr, nbr = G.csr()
    g = eng.ndl_graph(row_ptr, nbr)
    B = nd.path_adj(0, 4)
    pivots = [29, 0, 28, 5]                      # 27..29 are isolated: the path contracts onto the pivot
    emb = eng.ndl_init_embeddings(g, B, 4, pivots=np.array(pivots, dtype=np.int32), seed=9)
    ref = [nd.init_embedding(G, B, 9, c, pivot=p) for c, p in enumerate(pivots)]
    assert np.array_equal(emb.cpu().numpy(), np.array(ref))
    assert list(emb.cpu().numpy()[0]) == [29] * 5
    for glauber in (True, False):
        e = emb.clone()
        X = eng.ndl_sample_patches(g, B, e, 25, glauber=glauber, seed=9).cpu().numpy()
        for c in range(4):
            Xr, er = nd.sample_patches(G, B, ref[c], glauber, 25, 9, chain=c)
            assert np.array_equal(e.cpu().numpy()[c], np.array(er))
            assert np.array_equal(X[:, c * 25:(c + 1) * 25], Xr)


def test_a_later_call_continues_the_chain(eng, nd):
    G = small_graph(nd, 50, 70, seed=5)
    g = eng.ndl_graph(*G.csr())
    B = nd.path_adj(0, 6)
    e1 = eng.ndl_init_embeddings(g, B, 3, seed=21)
    e2 = e1.clone()
    Xa = eng.ndl_sample_patches(g, B, e1, 40, glauber=True, seed=21, step_offset=0)
    Xb1 = eng.ndl_sample_patches(g, B, e2, 15, glauber=True, seed=21, step_offset=0)
    Xb2 = eng.ndl_sample_patches(g, B, e2, 25, glauber=True, seed=21, step_offset=15)
    assert np.array_equal(e1.cpu().numpy(), e2.cpu().numpy())
    Xa, Xb1, Xb2 = Xa.cpu().numpy(), Xb1.cpu().numpy(), Xb2.cpu().numpy()
    for c in range(3):
        assert np.array_equal(Xa[:, c * 40:c * 40 + 15], Xb1[:, c * 15:(c + 1) * 15])
        assert np.array_equal(Xa[:, c * 40 + 15:(c + 1) * 40], Xb2[:, c * 25:(c + 1) * 25])


def test_patches_are_induced_adjacency_of_valid_path_embeddings(eng, nd):
    G = small_graph(nd, 200, 500, seed=8)
    g = eng.ndl_graph(*G.csr())
    k = 21
    B = nd.path_adj(0, k - 1)
    emb = eng.ndl_init_embeddings(g, B, 64, seed=2)
    for glauber in (True, False):
        X = eng.ndl_sample_patches(g, B, emb, 30, glauber=glauber, seed=2).cpu().numpy()
        assert set(np.unique(X)) <= {0.0, 1.0}
        P = X.reshape(k, k, -1)
        assert np.array_equal(P, P.transpose(1, 0, 2))               # undirected graph
        assert np.all(P[np.arange(k - 1), np.arange(1, k), :] == 1)  # consecutive path nodes are adjacent
        e = emb.cpu().numpy()
        for c in range(64):
            assert np.array_equal(P[:, :, c * 30 + 29], nd.patch(G, list(e[c])))


def test_pivot_visits_nodes_uniformly(eng, nd):
    """RW_update's Metropolis rule makes the uniform law stationary for the pivot (connected graph)."""
    G = small_graph(nd, 12, 10, seed=4)
    g = eng.ndl_graph(*G.csr())
    B = nd.path_adj(0, 2)
    chains, steps = 256, 400
    emb = eng.ndl_init_embeddings(g, B, chains, seed=77)
    eng.ndl_sample_patches(g, B, emb, 50, glauber=False, seed=77)              # burn-in
    counts = np.zeros(12)
    for rep in range(20):
        eng.ndl_sample_patches(g, B, emb, 5, glauber=False, seed=77, step_offset=50 + 5 * rep)
        counts += np.bincount(emb.cpu().numpy()[:, 0], minlength=12)
    freq = counts / counts.sum()
    assert np.abs(freq - 1 / 12).max() < 0.02, freq


def test_network_reconstructor_facade(eng, nd):
    from onmf_ontf_ndl_amd.network_reconstruction import Network_Reconstructor
    rs = np.random.RandomState(1)
    labels = rs.permutation(1000)[:80] + 5                    # arbitrary integer node names
    ring = [(labels[i], labels[(i + 1) % 80]) for i in range(80)]
    extra = [(labels[a], labels[b]) for a, b in rs.randint(80, size=(150, 2)) if a != b]
    edges = np.array(ring + extra)
    np.random.seed(0)
    rec = Network_Reconstructor(edges, n_components=6, MCMC_iterations=3, sub_iterations=5, sample_size=64,
                                batch_size=16, k1=0, k2=6, alpha=1, seed=5)
    assert list(rec.nodes[:2]) == [labels[0], labels[1]]
    B = rec.path_adj(0, 6)
    emb = rec.tree_sample(B, labels[3])
    assert emb[0] == labels[3] and len(emb) == 7
    names = set(map(tuple, np.sort(edges, axis=1)))
    assert all(tuple(sorted((emb[i], emb[i + 1]))) in names for i in range(6))
    X, emb2 = rec.get_patches_glauber(B, emb)
    assert X.shape == (49, 64) and X.dtype == np.float64
    hom, emb3 = rec.chd_gen_mx(B, emb2, iterations=50, is_glauber=False)
    assert hom.shape == (7, 7) and np.allclose(np.diag(hom, 1), 1)
    W = rec.train_dict()
    assert W.shape == (49, 6) and W.min() >= 0 and np.linalg.norm(W, axis=0).max() <= 1 + 1e-5
    many = Network_Reconstructor(edges, n_components=6, sample_size=64, k1=0, k2=6, n_chains=8, seed=5)
    e8 = many.initial_embedding(B)
    X8, e8b = many.get_patches_glauber(B, e8)
    assert e8.shape == (8, 7) and X8.shape == (49, 64) and e8b.shape == (8, 7)


def test_reconstruct_network_equals_the_sequential_loop(eng, nd):
    """Chain -> the reference's coder per patch (scikit-learn's positive LassoLars with alpha = 0,
    network_reconstruction_nx.py:466-473) -> edge-wise running mean, against the oracle's sequential
    restatement fed the same draws and scikit-learn itself as its coder (means to 1e-4; the thresholded edge
    sets equal wherever a mean is not within 1e-3 of one half); then the coordinate-descent coder against the
    oracle's."""
    from onmf_ontf_ndl_amd.network_reconstruction import Network_Reconstructor
    from oracle import onmf_oracle as orc
    G = small_graph(nd, 40, 70, seed=12)
    edges = np.array([(a, b) for a in range(G.n) for b in G.nbrs[a] if a < b])
    rs = np.random.RandomState(3)
    k1, k2, r, T = 0, 5, 8, 300
    rec = Network_Reconstructor(edges, n_components=r, k1=k1, k2=k2, is_glauber_recons=False, seed=17)
    idx = {int(v): i for i, v in enumerate(rec.nodes)}           # node label -> oracle index is the identity here
    W = rs.rand((k1 + k2 + 1) ** 2, r)
    W /= np.linalg.norm(W, axis=0)
    rec.W = W
    out, mean, count = rec.reconstruct_network(recons_iter=T, return_weights=True)
    # the same chain on the CPU: graph in the facade's node order
    Gf = nd.Graph(len(rec.nodes), [(idx[int(a)], idx[int(b)]) for a, b in edges])
    B = nd.path_adj(k1, k2)
    emb0 = nd.init_embedding(Gf, B, 17 + 0, 0)
    for solver in ("sklearn", "cd"):
        if solver == "cd":
            rec = Network_Reconstructor(edges, n_components=r, k1=k1, k2=k2, is_glauber_recons=False, seed=17)
            rec.W = W
            out, mean, count = rec.reconstruct_network(recons_iter=T, return_weights=True, coder="cd")
        coder = lambda x, Wm: orc.lasso_nonneg_code(x, Wm, alpha=0, solver=solver).T
        m_ref, c_ref, e_ref, _ = nd.reconstruct_network(Gf, B, W, emb0, False, T, 17, chain=0, step_offset=1,
                                                        coder=coder)
        assert int(count.sum()) == T * (k1 + k2 + 1) ** 2
        worst = 0.0
        for (a, b), v in m_ref.items():
            assert count[a, b] == c_ref[(a, b)]
            worst = max(worst, abs(mean[a, b] - v))
        print("reconstruct_network, coder %s: largest difference of an edge mean %.2e" % (solver, worst))
        assert worst < 1e-4
        got = {tuple(sorted((idx[int(a)], idx[int(b)]))) for a, b in out.edges}
        shaky = {tuple(sorted(key)) for key, v in m_ref.items() if abs(v - 0.5) < 1e-3}
        assert got - shaky == e_ref - shaky
