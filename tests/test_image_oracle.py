"""The image <-> patch oracle against its golden vectors (scikit-learn's patch functions and the
reference coder called once per window), and the facade's host-side helpers.  CPU only."""
import numpy as np

from conftest import load_golden
from oracle import image_oracle as io


def test_grid_patches_equal_sklearn_extract_patches_2d():
    g = load_golden("g10_image")
    assert np.array_equal(io.grid_patches(g["grey"], int(g["k"])), g["X_grey"])


def test_running_mean_fold_equals_sklearn_reconstruct_from_patches_2d():
    g = load_golden("g10_image")
    img, count = io.fold_patches(g["vals"], g["grey"].shape, int(g["k"]))
    assert np.abs(img - g["fold_grey"]).max() < 1e-14
    assert count.min() == 1 and count.max() == int(g["k"]) ** 2


def test_code_each_equals_one_reference_call_per_window():
    g = load_golden("g10_image")
    k, s = int(g["k"]), int(g["stride"])
    X = io.grid_patches(g["color"], k, s, drop_last=True)
    assert X.shape == (k * k * 3, g["H0"].shape[1])
    for tol, code, sweeps in ((0.01, g["code"], g["sweeps"]), (float(g["tol_loose"]), g["code_loose"], g["sweeps_loose"])):
        sw = []
        H = io.code_each(X, g["W"], g["H0"], alpha=1, sub_iter=10, stopping_diff=tol, sweeps=sw)
        assert np.abs(H - code).max() < 1e-12
        assert sw == list(sweeps)


def test_random_patches_follow_the_corner_stream():
    rs = np.random.RandomState(5)
    img = rs.rand(12, 13, 3)
    np.random.seed(3)
    corners = io.random_corners(img.shape, 4, 7)
    np.random.seed(3)
    for a, b in corners:
        assert a == np.random.choice(12 - 4) and b == np.random.choice(13 - 4)
    X = io.random_patches(img, 4, corners)
    assert X.shape == (48, 7)
    a, b = corners[2]
    assert X[(1 * 4 + 2) * 3 + 1, 2] == img[a + 1, b + 2, 1]


def test_fold_drop_last_grid_leaves_uncovered_pixels_zero():
    rs = np.random.RandomState(6)
    k, s = 3, 2
    shape = (9, 8, 3)
    gr, gc = io.grid_shape(9, 8, k, s, drop_last=True)
    assert (gr, gc) == (3, 3)
    R = rs.rand(k * k * 3, gr * gc)
    img, count = io.fold_patches(R, shape, k, s, drop_last=True)
    assert count[8, :].max() == 0 and np.all(img[8] == 0)
    assert count[2, 2] == 4          # corner shared by four windows


def test_downscale_local_mean_pads_with_zeros():
    from onmf_ontf_ndl_amd.image_reconstruction import downscale_local_mean
    a = np.arange(15, dtype=float).reshape(3, 5)
    out = downscale_local_mean(a, (2, 2))
    assert out.shape == (2, 3)
    assert out[0, 0] == (0 + 1 + 5 + 6) / 4
    assert out[1, 2] == 14 / 4       # one real value, three padded zeros
