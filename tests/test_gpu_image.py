"""GPU parity of the image <-> patch kernels and the per-column coder (through the C-ABI) against the
CPU oracle and the golden vectors.

Tolerances: patch extraction is a copy - bit-exact (after the float32 cast of the image); overlap
average 1e-6 relative (fp32 sums of at most k*k terms), counts exact; per-column codes 2e-5 relative
Frobenius and equal sweep counts wherever the oracle's deciding dist is not within 1e-4 (relative)
of the tolerance.
"""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu

TOL_H = 2e-5


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from onmf_ontf_ndl_amd.engine import get_engine
    e = get_engine()
    yield e
    e.set_option(3, 0)


@pytest.fixture(scope="module")
def io():
    from oracle import image_oracle
    return image_oracle


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("shape,k", [((11, 9), 4), ((20, 31, 3), 7), ((5, 5, 2), 5), ((300, 17, 3), 3)])
def test_extract_patches_at_given_corners_is_a_copy(eng, io, shape, k):
    rs = np.random.RandomState(sum(shape) + k)
    img = rs.rand(*shape).astype(np.float32)
    n = 333
    corners = np.stack([rs.randint(shape[0] - k + 1, size=n), rs.randint(shape[1] - k + 1, size=n)], axis=1)
    X = eng.extract_patches(eng.dev(img), k, corners=corners)
    assert np.array_equal(host(X), io.random_patches(img, k, corners).astype(np.float64))


@pytest.mark.parametrize("shape,k,stride,drop_last", [((11, 9), 4, 1, False), ((9, 10, 3), 4, 2, True),
                                                      ((40, 33, 3), 7, 3, True), ((16, 16), 16, 1, False),
                                                      ((64, 50, 3), 5, 1, True)])
def test_extract_grid_and_fold_back(eng, io, shape, k, stride, drop_last):
    rs = np.random.RandomState(sum(shape) + k + stride)
    img = rs.rand(*shape).astype(np.float32)
    grid = io.grid_shape(shape[0], shape[1], k, stride, drop_last)
    X = eng.extract_patches(eng.dev(img), k, grid=grid, stride=stride)
    assert np.array_equal(host(X), io.grid_patches(img, k, stride, drop_last).astype(np.float64))
    # averaging the image's own patches gives the image back wherever a patch lies
    back, count = eng.fold_patches(X, shape, k, grid, stride=stride)
    ref, ref_count = io.fold_patches(host(X), shape, k, stride, drop_last)
    assert np.array_equal(host(count), ref_count)
    assert np.abs(host(back) - ref).max() <= 1e-6
    covered = ref_count > 0
    assert np.abs(host(back)[covered] - img.astype(np.float64)[covered]).max() <= 1e-6
    # and arbitrary patch values average like the oracle's running mean
    R = rs.rand(*X.shape).astype(np.float32)
    out, _ = eng.fold_patches(eng.dev(R), shape, k, grid, stride=stride)
    ref, _ = io.fold_patches(R, shape, k, stride, drop_last)
    assert relerr(host(out), ref) <= 1e-6


def test_golden_patch_matrix_and_fold(eng):
    g = load_golden("g10_image")
    k = int(g["k"])
    h, w = g["grey"].shape
    grid = (h - k + 1, w - k + 1)
    X = eng.extract_patches(eng.dev(g["grey"]), k, grid=grid)
    assert np.array_equal(host(X), g["X_grey"].astype(np.float32).astype(np.float64))
    out, _ = eng.fold_patches(eng.dev(g["vals"]), (h, w), k, grid)
    assert relerr(host(out), g["fold_grey"]) <= 1e-6


def test_extract_rejects_bad_geometry(eng):
    from onmf_ontf_ndl_amd._lib import OnmfLibraryError
    img = eng.zeros(8, 8)
    with pytest.raises(OnmfLibraryError):
        eng.extract_patches(img, 9, grid=(1, 1))
    with pytest.raises(OnmfLibraryError):
        eng.extract_patches(img, 4, grid=(6, 5))            # 5 + 4 > 8
    with pytest.raises(ValueError):
        eng.extract_patches(img, 4, corners=np.array([[5, 0]]))


def check_each(eng, io, X, W, H0, alpha, tol, sub_iter=10):
    from oracle import onmf_oracle as orc
    H, sweeps = eng.sparse_code_each(eng.dev(X), eng.dev(W), H0=eng.dev(H0), alpha=alpha, sub_iter=sub_iter,
                                     stopping_diff=tol, want_sweeps=True)
    H, sweeps = host(H), sweeps.cpu().numpy()
    bad = 0
    for p in range(X.shape[1]):
        tr = []
        ref = orc.code_sweeps(X[:, p:p + 1].astype(np.float64), W.astype(np.float64), H0=H0[:, p:p + 1].astype(np.float64),
                              alpha=alpha, sub_iter=sub_iter, stopping_diff=tol, trace=tr)
        borderline = any(abs(dv - tol) <= 1e-4 * tol for _, dv in tr)
        if borderline:
            continue
        assert sweeps[p] == len(tr), (p, sweeps[p], tr)
        if relerr(H[:, p], ref[:, 0]) > TOL_H:
            bad += 1
    assert bad == 0
    return H, sweeps


def test_golden_one_solve_per_window(eng, io):
    g = load_golden("g10_image")
    k, s = int(g["k"]), int(g["stride"])
    X = io.grid_patches(g["color"], k, s, drop_last=True)
    for tol, code, sweeps in ((0.01, g["code"], g["sweeps"]), (float(g["tol_loose"]), g["code_loose"], g["sweeps_loose"])):
        H, sw = eng.sparse_code_each(eng.dev(X), eng.dev(g["W"]), H0=eng.dev(g["H0"]), alpha=1, sub_iter=10,
                                     stopping_diff=tol, want_sweeps=True)
        assert list(sw.cpu().numpy()) == list(sweeps)
        assert relerr(host(H), code) <= TOL_H


@pytest.mark.parametrize("d,r,b,alpha,tol", [(48, 6, 70, 1.0, 0.03), (147, 100, 300, 1.0, 0.06), (75, 25, 257, 0.0, 0.05),
                                             (300, 256, 130, 1.0, 0.08), (27, 17, 1, 0.5, 0.03)])
@pytest.mark.parametrize("kernel", [0, 1])
def test_one_solve_per_column_against_the_oracle(eng, io, d, r, b, alpha, tol, kernel):
    rs = np.random.RandomState(d + r + b)
    W = rs.rand(d, r).astype(np.float32)
    W /= np.maximum(1, np.linalg.norm(W, axis=0))
    X = rs.rand(d, b).astype(np.float32)
    H0 = rs.rand(r, b).astype(np.float32)
    eng.set_option(3, kernel)
    try:
        check_each(eng, io, X, W, H0, alpha, tol)
    finally:
        eng.set_option(3, 0)


def test_per_column_solve_is_independent_of_the_batch(eng):
    """A column's result does not depend on which other columns are solved with it."""
    rs = np.random.RandomState(99)
    d, r, b = 108, 64, 200
    W = rs.rand(d, r).astype(np.float32)
    X = rs.rand(d, b).astype(np.float32)
    H0 = rs.rand(r, b).astype(np.float32)
    Hall, sall = eng.sparse_code_each(eng.dev(X), eng.dev(W), H0=eng.dev(H0), alpha=1, stopping_diff=0.03, want_sweeps=True)
    pick = [3, 17, 64, 199]
    Hsub, ssub = eng.sparse_code_each(eng.dev(X[:, pick]), eng.dev(W), H0=eng.dev(H0[:, pick]), alpha=1,
                                      stopping_diff=0.03, want_sweeps=True)
    assert np.array_equal(host(Hall)[:, pick], host(Hsub))
    assert np.array_equal(sall.cpu().numpy()[pick], ssub.cpu().numpy())


def test_reconstruct_image_color_equals_the_per_window_loop(eng, io):
    from onmf_ontf_ndl_amd.image_reconstruction import Image_Reconstructor
    rs = np.random.RandomState(12)
    A = rs.rand(14, 17, 3)
    k, res = 5, 2
    np.random.seed(1)
    rec = Image_Reconstructor(A, n_components=9, patch_size=k, is_color=True)
    W = rec.W.copy()
    W /= np.maximum(1, np.linalg.norm(W, axis=0))
    rec.W = W
    np.random.seed(2)
    out = rec.reconstruct_image_color(A, recons_resolution=res)
    np.random.seed(2)
    gr, gc = io.grid_shape(14, 17, k, res, drop_last=True)
    H0 = np.random.rand(gr * gc, 9).T
    code, ref, count = io.reconstruct_image_color(A, W, k, H0, recons_resolution=res)
    assert out.shape == A.shape and out.dtype == np.float64
    assert np.array_equal(rec.A_overlap_count, count)
    assert relerr(out, ref) <= 2e-5


def test_reconstruct_image_grey_equals_one_batched_solve(eng, io):
    from onmf_ontf_ndl_amd.image_reconstruction import Image_Reconstructor
    rs = np.random.RandomState(13)
    img = rs.rand(24, 20)
    k = 6
    rec = Image_Reconstructor(img, n_components=12, patch_size=k, is_color=False, downscale_factor=1)
    W = rs.rand(k * k, 12)
    W /= np.linalg.norm(W, axis=0)
    rec.W = W
    np.random.seed(4)
    code = rec.reconstruct_image(img, downscale_factor=1, patch_size=k)
    np.random.seed(4)
    H0 = np.random.rand(12, (24 - k + 1) * (20 - k + 1))
    ref_code, ref_img = io.reconstruct_image(img, W, k, H0)
    assert relerr(code, ref_code) <= TOL_H
    assert relerr(rec.img_recons, ref_img) <= 2e-5


def test_train_dict_from_random_patches_keeps_the_constraint_set(eng):
    from onmf_ontf_ndl_amd.image_reconstruction import Image_Reconstructor
    rs = np.random.RandomState(14)
    A = rs.rand(30, 28, 3)
    np.random.seed(8)
    rec = Image_Reconstructor(A, n_components=8, iterations=3, sub_iterations=4, num_patches=50, batch_size=10,
                              patch_size=4, is_color=True)
    X = rec.extract_random_patches()
    assert X.shape == (48, 50) and X.min() >= 0 and X.max() <= 1
    W = rec.train_dict()
    assert W.shape == (48, 8)
    assert W.min() >= 0 and np.linalg.norm(W, axis=0).max() <= 1 + 1e-5
    assert rec.nmf.history > 0


def test_tensor_reconstructor_patch_tensors_and_color_reconstruction(eng, io):
    """image_reconstruction_tensor.py: patch tensors, and the per-window coder (scikit-learn's positive LassoLars
    with alpha = 1, image_reconstruction_tensor.py:309-312) as one batched solve, against scikit-learn itself
    called window by window + running-mean fold."""
    from onmf_ontf_ndl_amd.image_reconstruction_tensor import Image_Reconstructor
    from oracle import onmf_oracle as orc
    rs = np.random.RandomState(31)
    A = rs.rand(13, 15, 3)
    k, res = 4, 2
    rec = Image_Reconstructor(A, n_components=7, patch_size=k, num_patches=20, is_color=True)
    np.random.seed(6)
    T = rec.extract_random_patches()
    np.random.seed(6)
    corners = io.random_corners(A.shape, k, 20)
    ref = io.random_patches(A, k, corners).reshape(k * k, 3, 20)
    assert T.shape == (16, 3, 20)
    assert np.array_equal(T, ref.astype(np.float32).astype(np.float64))
    W = rs.rand(k * k * 3, 7)
    W /= np.linalg.norm(W, axis=0)
    rec.W = W
    out = rec.reconstruct_image_color(A, recons_resolution=res)
    X = io.grid_patches(A, k, res, drop_last=True)
    H = orc.lasso_nonneg_code(X, W, alpha=1, solver="sklearn").T
    ref_img, count = io.fold_patches(W @ H, A.shape, k, res, drop_last=True)
    assert np.array_equal(rec.A_overlap_count, count)
    assert relerr(out, ref_img) <= 2e-5


def test_tensor_reconstructor_trains_along_a_mode(eng):
    from onmf_ontf_ndl_amd.image_reconstruction_tensor import Image_Reconstructor
    rs = np.random.RandomState(32)
    A = rs.rand(20, 22, 3)
    np.random.seed(9)
    rec = Image_Reconstructor(A, n_components=5, iterations=2, sub_iterations=3, block_iterations=5, batch_size=8,
                              num_patches=30, patch_size=3, is_color=True)
    W = rec.train_dict(mode=2, learn_joint_dict=True)
    assert W.shape == (27, 5) and W.min() >= 0 and np.linalg.norm(W, axis=0).max() <= 1 + 1e-5


def test_image_to_patches_grey_and_unfolded_colour_against_sklearn(eng):
    """image_to_patches (image_reconstruction.py:117-171) for reconstruction (every window, in order): the grey
    branch, and the colour branch, which unfolds the RGB image along its channel mode and transposes it
    (tl_unfold(img, mode=2).T, :139-142) before the windows are taken - against scikit-learn's
    extract_patches_2d on the same matrices."""
    from sklearn.feature_extraction.image import extract_patches_2d
    from onmf_ontf_ndl_amd.image_reconstruction import Image_Reconstructor, downscale_local_mean
    rs = np.random.RandomState(41)
    grey = rs.rand(18, 14).astype(np.float32).astype(np.float64)
    rec = Image_Reconstructor(grey, n_components=4, patch_size=5, is_color=False)
    X = rec.image_to_patches(grey, patch_size=5, downscale_factor=2, is_recons=True)
    small = downscale_local_mean(grey, (2, 2))
    want = extract_patches_2d(small, (5, 5))
    want = want.reshape(len(want), -1).T
    assert X.shape == want.shape and relerr(X, want) < 1e-7
    rgb = rs.rand(9, 7, 3).astype(np.float32).astype(np.float64)
    rec = Image_Reconstructor(rgb, n_components=4, patch_size=2, is_color=True)
    X = rec.image_to_patches(rgb, patch_size=2, color=True, downscale_factor=1, is_recons=True)
    unfolded = np.reshape(np.moveaxis(rgb, 2, 0), (3, -1)).T            # (h*w) x 3
    want = extract_patches_2d(unfolded, (2, 2))
    want = want.reshape(len(want), -1).T
    assert X.shape == want.shape == (4, (9 * 7 - 1) * 2) and np.array_equal(X, want)
    with pytest.raises(ValueError):
        rec.image_to_patches(rgb, patch_size=7, color=True, downscale_factor=1, is_recons=True)
