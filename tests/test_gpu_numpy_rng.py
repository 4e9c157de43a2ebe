"""NumPy's legacy global generator continued on the GPU (onmf_numpy_rand): bit-identical draws, and NumPy's own
state left exactly where np.random.rand would have left it - the reference draws H0 = np.random.rand(r, b) in
every solve (src/onmf.py:245-246) and a drop-in has to stay in that stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from onmf_ontf_ndl_amd.engine import get_engine
    return get_engine()


def _check(eng, seed, skip, rows, cols):
    np.random.seed(seed)
    if skip:
        np.random.rand(skip)                                  # start somewhere inside a block (odd and even)
    st = np.random.get_state()
    want = np.random.rand(rows, cols).astype(np.float32)
    after = np.random.rand(5)
    np.random.set_state(st)
    got = eng.numpy_rand(rows, cols)
    assert got.shape == (rows, cols)
    got = got.cpu().numpy()
    assert np.array_equal(got, want), (seed, skip, rows, cols, int((got != want).sum()))
    assert np.array_equal(np.random.rand(5), after)           # NumPy's stream goes on where it would have


@pytest.mark.parametrize("seed,skip,rows,cols", [
    (0, 0, 64, 512),               # one chunk, from a fresh seed (position 624: block exhausted)
    (1, 7, 100, 1000),             # odd start inside a block
    (2, 311, 16, 2049),            # odd number of doubles
    (3, 1, 256, 1024),             # three chunks: jumps
    (4, 623, 256, 8192),           # 17 chunks, start at the last word of a block
    (5, 0, 33, 9929),              # odd sizes
])
def test_device_draw_equals_numpy(eng, seed, skip, rows, cols):
    _check(eng, seed, skip, rows, cols)


def test_headline_h0_and_a_second_round(eng):
    # r = 256, b = 65536: 128.02 chunks; 260 x 65536 doubles need a second round of chunks
    _check(eng, 6, 3, 256, 65536)
    _check(eng, 7, 0, 260, 65536)


def test_small_draws_take_the_host_path_and_stay_in_the_stream(eng):
    _check(eng, 8, 5, 16, 1000)
    assert 16 * 1000 < eng.NUMPY_RAND_ON_DEVICE_FROM


def test_without_a_jump_table_chunks_run_one_after_another(eng):
    """A context that was never given a table produces the same stream, chunk by chunk."""
    import ctypes as C
    from onmf_ontf_ndl_amd import _lib
    from onmf_ontf_ndl_amd.engine import Engine, _ptr
    other = Engine(eng.index)                                  # its own context, no table
    np.random.seed(9)
    np.random.rand(10)
    st = np.random.get_state()
    rows, cols = 128, 4100                                     # 5 chunks
    want = np.random.rand(rows, cols).astype(np.float32)
    key = np.ascontiguousarray(st[1], dtype=np.uint32)
    key_out = np.empty(624, dtype=np.uint32)
    pos_out = C.c_int(0)
    out = other.empty(rows, cols)
    rc = other.lib.onmf_numpy_rand(other.ctx, key.ctypes.data_as(C.c_void_p), int(st[2]), rows * cols, _ptr(out),
                                   key_out.ctypes.data_as(C.c_void_p), C.byref(pos_out), other.stream())
    _lib.check(rc, other.ctx, "onmf_numpy_rand")
    assert np.array_equal(out.cpu().numpy(), want)
    end = np.random.get_state()
    assert np.array_equal(key_out, end[1]) and pos_out.value == end[2]
