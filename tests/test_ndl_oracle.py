"""The chain oracle on the CPU: Philox4x32-10 against the generator's published known-answer vectors,
and invariants of the restated transition rules."""
import numpy as np

from oracle import ndl_oracle as nd


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert nd.philox4x32([0, 0, 0, 0], 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert nd.philox4x32([0xffffffff] * 4, 0xffffffffffffffff) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert nd.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0x299f31d0 << 32) | 0xa4093822) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def ring_graph(n, extra, seed):
    rs = np.random.RandomState(seed)
    edges = [(i, (i + 1) % n) for i in range(n)] + [tuple(rs.randint(n, size=2)) for _ in range(extra)]
    return nd.Graph(n, [(a, b) for a, b in edges if a != b])


def test_path_motifs_and_parents():
    B = nd.path_adj(2, 3)
    assert [nd.find_parent(B, i) for i in range(1, 6)] == [0, 1, 0, 3, 4]
    assert np.array_equal(nd.path_adj(0, 3), np.eye(4, k=1, dtype=int))
    assert np.array_equal(nd.path_adj(3, 0), np.eye(4, k=1, dtype=int))


def test_chains_keep_valid_homomorphisms():
    G = ring_graph(40, 60, 0)
    for B in (nd.path_adj(0, 5), nd.path_adj(2, 3)):
        k = B.shape[0]
        emb = nd.init_embedding(G, B, 7, 0)
        for glauber in (True, False):
            X, e = nd.sample_patches(G, B, emb, glauber, 60, 7)
            for i in range(1, k):
                assert G.has_edge(e[nd.find_parent(B, i)], e[i])
            P = X.reshape(k, k, -1)
            for i in range(1, k):
                assert np.all(P[nd.find_parent(B, i), i, :] == 1)


def test_metropolis_rule_accepts_moves_to_lower_degree():
    G = nd.Graph(4, [(0, 1), (1, 2), (1, 3)])            # star: centre 1
    assert nd.rw_update(G, 1, 0, 0xFFFFFFFF) == 0        # from the centre every proposal is accepted
    assert nd.rw_update(G, 0, 0, 0xFFFFFFFF) == 0        # leaf -> centre accepted with probability 1/3
    assert nd.rw_update(G, 0, 0, 0x10000000) == 1
    lonely = nd.Graph(3, [(0, 1)])
    assert nd.rw_update(lonely, 2, 0x80000000, 0) == 1   # isolated: uniform jump, (x * 3) >> 32
