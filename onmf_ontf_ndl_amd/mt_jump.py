"""Jump-ahead tables for NumPy's legacy generator (MT19937), so that a GPU can draw the reference's
``H0 = np.random.rand(r, b)`` (src/onmf.py:245-246) out of NumPy's own global stream, many chunks at once.

MT19937 is a linear recurrence over GF(2) on a window of 624 words y[n .. n+623]:
    y[n+624] = y[n+397] ^ (((y[n] & 0x80000000) | (y[n+1] & 0x7fffffff)) >> 1) ^ (0x9908b0df if y[n+1] & 1 else 0)
whose effective state has 19937 bits (all of the window but the low 31 bits of its first word).  With phi
the minimal polynomial of the one-word shift T and g_J = x^J mod phi, the window J words ahead is g_J(T)
applied to the current one, i.e.

    y[J + k] = XOR over the set bits i of g_J of y[i + k]                (k >= 1; k = 0: top bit only)

- a correlation of 19937 bits against the first 20 560 words of the sequence, which a GPU evaluates with
all lanes busy (``mt_jump_kernel`` in csrc/mt19937.hip).  The tables hold g_J for J = j * chunk - 1,
j = 1 .. n (one word early, so that the window's untrusted first word is not one that is drawn).

Everything here is integer arithmetic on Python ints used as bit sets; it runs once per table (a few
seconds) and the result is cached next to the library.  ``tests/test_mt_jump_cpu.py`` holds the tables to
NumPy's own stream.
"""
from __future__ import annotations

import os

import numpy as np

N, M = 624, 397
DEGREE = 19937
UPPER, LOWER, MATRIX_A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF
CHUNK_BLOCKS = 210                       # blocks of 624 words per chunk of the device generator
CHUNK_WORDS = N * CHUNK_BLOCKS
TABLE_CHUNKS = 256                       # jumps held: up to 256 chunks (33.5 M words) per round

_HERE = os.path.dirname(os.path.abspath(__file__))


def next_block(mt):
    """The 624 words after the block ``mt`` (uint32 array): NumPy's / the reference generator's twist, in
    its three dependency-free phases."""
    mt = np.asarray(mt, dtype=np.uint32)
    new = np.empty(N, dtype=np.uint32)

    def mix(a, b, c):
        yv = (a & np.uint32(UPPER)) | (b & np.uint32(LOWER))
        return c ^ (yv >> np.uint32(1)) ^ np.where(b & np.uint32(1), np.uint32(MATRIX_A), np.uint32(0))
    new[0:227] = mix(mt[0:227], mt[1:228], mt[397:624])
    new[227:454] = mix(mt[227:454], mt[228:455], new[0:227])
    new[454:623] = mix(mt[454:623], mt[455:624], new[227:396])
    new[623] = mix(mt[623:624], new[0:1], new[396:397])[0]
    return new


def raw_sequence(mt, n_words):
    """y[0 .. n_words) starting with the block ``mt`` itself."""
    out = [np.asarray(mt, dtype=np.uint32)]
    while N * len(out) < n_words:
        out.append(next_block(out[-1]))
    return np.concatenate(out)[:n_words]


def temper(y):
    y = np.asarray(y, dtype=np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
    y ^= y >> np.uint32(18)
    return y


# ---------------------------------------------------------------------------------------------
# GF(2)[x] on Python ints (bit i = coefficient of x^i)
# ---------------------------------------------------------------------------------------------
def _berlekamp_massey(bits):
    """Connection polynomial C (C_0 = 1) of the shortest recurrence s_n = sum_{i>=1} C_i s_{n-i}, and its
    length L."""
    C, B, L, m = 1, 1, 0, 1
    window = 0                                        # bit i = s_{n-i}
    for n, s in enumerate(bits):
        window = (window << 1) | int(s)
        d = (C & window).bit_count() & 1
        if d:
            T = C
            C ^= B << m
            if 2 * L <= n:
                L, B, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def minimal_polynomial():
    """phi(x): minimal polynomial of the one-word shift, from the top-bit sequence of a generic stream."""
    rs = np.random.RandomState(20240229)
    mt = rs.get_state()[1].astype(np.uint32)
    y = raw_sequence(mt, 2 * DEGREE + 200)
    bits = (y >> np.uint32(31)).astype(np.uint8)
    C, L = _berlekamp_massey(bits)
    if L != DEGREE:
        raise AssertionError("unexpected linear complexity %d" % L)
    # s_n = sum_{i=1..L} C_i s_{n-i}  <=>  x^L = sum C_i x^{L-i}: phi = reciprocal of C
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    return phi


def _spread(a, nbits):
    """Bits of ``a`` put 16 apart, so that an integer product cannot carry from one coefficient into the next."""
    raw = np.frombuffer(a.to_bytes((nbits + 7) // 8, "little"), dtype=np.uint8)
    bits = np.unpackbits(raw, bitorder="little")[:nbits]
    return int.from_bytes(bits.astype("<u2").tobytes(), "little")


def _gather(p, nbits):
    raw = np.frombuffer(p.to_bytes(2 * nbits, "little"), dtype="<u2")
    bits = (raw & 1).astype(np.uint8)
    return int.from_bytes(np.packbits(bits, bitorder="little").tobytes(), "little")


def poly_mul(a, b):
    na, nb = max(a.bit_length(), 1), max(b.bit_length(), 1)
    prod = _spread(a, na) * _spread(b, nb)             # coefficient sums < 2^16: no carries between slots
    return _gather(prod, na + nb)


class _Reducer:
    """Reduction modulo phi with two products instead of 20 000 shift-and-xor steps (Barrett): with
    mu = floor(x^(2d) / phi), q = floor(floor(a / x^d) * mu / x^d) is the quotient of a by phi for
    deg a < 2d, and a - q * phi the remainder."""

    def __init__(self, phi):
        self.phi = phi
        self.d = phi.bit_length() - 1
        # mu = floor(x^(2d) / phi) by long division (once)
        num, mu = 1 << (2 * self.d), 0
        while num.bit_length() - 1 >= self.d:
            sh = num.bit_length() - 1 - self.d
            mu |= 1 << sh
            num ^= phi << sh
        self.mu = mu

    def __call__(self, a):
        d = self.d
        if a.bit_length() <= d:
            return a
        q = poly_mul(a >> d, self.mu) >> d
        r = a ^ poly_mul(q, self.phi)
        r &= (1 << (2 * d)) - 1
        while r.bit_length() - 1 >= d:                 # (at most a step or two)
            r ^= self.phi << (r.bit_length() - 1 - d)
        return r


def x_power(e, reduce):
    """x^e mod phi."""
    result, base = 1, 2
    while e:
        if e & 1:
            result = reduce(poly_mul(result, base))
        e >>= 1
        if e:
            base = reduce(poly_mul(base, base))
    return result


def jump_table(chunk_words=CHUNK_WORDS, n_chunks=TABLE_CHUNKS):
    """uint32 array (n_chunks, 624): bit i of row j-1 (word i // 32, bit i % 32) is coefficient i of
    x^(j * chunk_words - 1) mod phi, j = 1 .. n_chunks."""
    phi = minimal_polynomial()
    red = _Reducer(phi)
    step = x_power(chunk_words, red)
    g = x_power(chunk_words - 1, red)
    rows = []
    for _ in range(n_chunks):
        rows.append(np.frombuffer(g.to_bytes(N * 4, "little"), dtype="<u4").astype(np.uint32))
        g = red(poly_mul(g, step))
    return np.stack(rows)


def apply_jump(row, y):
    """Window [y[J-1] (top bit only), y[J], ..., y[J+622]] from the first 19937 + 623 words of the sequence
    and the bits of x^(J-1) mod phi - the arithmetic of the device kernel, in NumPy."""
    bits = np.unpackbits(np.asarray(row, dtype="<u4").view(np.uint8), bitorder="little")[:DEGREE]
    out = np.zeros(N, dtype=np.uint32)
    for i in np.nonzero(bits)[0]:
        out ^= y[i:i + N]
    return out


def table_path():
    return os.environ.get("ONMF_MT_JUMP_TABLE") or os.path.join(_HERE, "mt19937_jump_%d_%d.npy" % (CHUNK_WORDS, TABLE_CHUNKS))


def load_table(build_if_missing=True):
    """The cached table (built by ``__graft_entry__.build()``); computed and saved when missing."""
    path = table_path()
    if os.path.exists(path):
        t = np.load(path, allow_pickle=False)
        if t.shape == (TABLE_CHUNKS, N) and t.dtype == np.uint32:
            return t
    if not build_if_missing:
        raise FileNotFoundError(path)
    t = jump_table()
    try:
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            np.save(f, t)
        os.replace(tmp, path)
    except OSError:
        pass
    return t
