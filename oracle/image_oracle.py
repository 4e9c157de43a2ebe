"""CPU oracle for the image <-> patch steps around the coder (TEST INFRASTRUCTURE, not product code).

NumPy restatement of the parts of the reference's ``image_reconstruction.py`` that sit either
side of the ONMF hot path:

* ``random_patches``          <- ``Image_Reconstructor.extract_random_patches`` (image_reconstruction.py:173-206)
* ``grid_patches``            <- the all-patches call of sklearn ``extract_patches_2d`` in
                                 ``image_to_patches`` (:161-166) and the window loop of
                                 ``reconstruct_image_color`` (:375-378)
* ``fold_patches``            <- sklearn ``reconstruct_from_patches_2d`` (:354) and the painting
                                 loop of ``reconstruct_image_color`` (:386-392)
* ``code_each``               <- one ``update_code_within_radius`` call per patch (:384)
* ``reconstruct_image``       <- ``Image_Reconstructor.reconstruct_image`` (:340-356), array in
* ``reconstruct_image_color`` <- ``Image_Reconstructor.reconstruct_image_color`` (:358-406), array in

``image_reconstruction.py`` itself cannot be imported in the build container (``skimage``,
``tensorly`` and ``PIL``-side helpers are not all installed), so this file is pinned piecewise by
``tests/golden/g10_image.npz``: the patch matrix and the fold against scikit-learn's own
``extract_patches_2d`` / ``reconstruct_from_patches_2d``, the per-patch codes against the
reference's ``update_code_within_radius`` imported from ``src/onmf.py``.  The painting loop of
the colour path (:388-392, a running mean) is restated from the text only - "parity unpinned"
for that loop beyond its agreement with the sklearn fold on a stride-1 grid.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU baseline may import this.
"""
import numpy as np

from . import onmf_oracle as orc


def random_corners(shape, k, num_patches):
    """Top-left corners in the order the reference draws them (image_reconstruction.py:185-186):
    per patch one ``np.random.choice(rows - k)`` then one ``np.random.choice(cols - k)`` from
    NumPy's global legacy generator."""
    out = np.empty((num_patches, 2), dtype=np.int64)
    for i in range(num_patches):
        out[i, 0] = np.random.choice(shape[0] - k)
        out[i, 1] = np.random.choice(shape[1] - k)
    return out


def random_patches(data, k, corners):
    """Patch matrix (k*k*ch x num_patches) for given corners (image_reconstruction.py:187-193,
    :199-205): every window flattened in C order, one column per patch."""
    data = np.asarray(data)
    cols = [data[a:a + k, b:b + k].reshape(-1) for a, b in np.asarray(corners)]
    return np.stack(cols, axis=1)


def grid_shape(h, w, k, stride=1, drop_last=False):
    """Patches per column / row of the window grid.  ``drop_last``: the colour path loops over
    ``range(0, h - k, stride)`` (image_reconstruction.py:375-376), which leaves out the window
    that touches the last row / column; sklearn's all-patches grid includes it."""
    if drop_last:
        return len(range(0, h - k, stride)), len(range(0, w - k, stride))
    return (h - k) // stride + 1, (w - k) // stride + 1


def grid_patches(data, k, stride=1, drop_last=False):
    """All windows of the grid, row by row, as columns (d x b)."""
    data = np.asarray(data)
    gr, gc = grid_shape(data.shape[0], data.shape[1], k, stride, drop_last)
    cols = [data[i * stride:i * stride + k, j * stride:j * stride + k].reshape(-1)
            for i in range(gr) for j in range(gc)]
    return np.stack(cols, axis=1)


def fold_patches(R, shape, k, stride=1, drop_last=False):
    """Running-mean painting of patch values onto a zero canvas (image_reconstruction.py:388-392).
    R: d x b on the grid of ``grid_patches``; returns (image, overlap count)."""
    shape = tuple(shape)
    ch = shape[2] if len(shape) == 3 else 1
    h, w = shape[0], shape[1]
    gr, gc = grid_shape(h, w, k, stride, drop_last)
    canvas = np.zeros((h, w, ch))
    count = np.zeros((h, w))
    p = 0
    for i in range(gr):
        for j in range(gc):
            patch = np.asarray(R[:, p]).reshape(k, k, ch)
            ys, xs = slice(i * stride, i * stride + k), slice(j * stride, j * stride + k)
            c = count[ys, xs][:, :, None]
            canvas[ys, xs, :] = (c * canvas[ys, xs, :] + patch) / (c + 1)
            count[ys, xs] += 1
            p += 1
    return (canvas if len(shape) == 3 else canvas[:, :, 0]), count


def code_each(X, W, H0, alpha=0, sub_iter=10, stopping_diff=0.01, sweeps=None):
    """One coder call per column, each with its own stopping test (image_reconstruction.py:384).
    H0: r x b, column p being the start of patch p.  ``sweeps`` (list) receives the sweep counts."""
    X = np.asarray(X, dtype=np.float64)
    H = np.empty((W.shape[1], X.shape[1]))
    for p in range(X.shape[1]):
        tr = []
        H[:, p:p + 1] = orc.code_sweeps(X[:, p:p + 1], W, H0=np.asarray(H0)[:, p:p + 1], r=None, alpha=alpha,
                                        sub_iter=sub_iter, stopping_diff=stopping_diff, trace=tr)
        if sweeps is not None:
            sweeps.append(len(tr))
    return H


def reconstruct_image(data, W, k, H0, alpha=0):
    """Grey image: all patches coded in ONE call with the global stopping test, then averaged back
    (image_reconstruction.py:347-354; ``sparse_code`` = alpha or 0, sub_iter 10, stopping_diff 0.01)."""
    X = grid_patches(data, k)
    code = orc.code_sweeps(X, W, H0=H0, r=None, alpha=alpha, sub_iter=10, stopping_diff=0.01)
    img, _ = fold_patches(W @ code, np.asarray(data).shape, k)
    return code, img


def reconstruct_image_color(A, W, k, H0, recons_resolution=1):
    """Colour image: one coder call per window (alpha=1, sub_iter=10, stopping_diff=0.01), windows
    on the ``range(0, n - k, resolution)`` grid, painted with the running mean
    (image_reconstruction.py:375-392)."""
    X = grid_patches(A, k, recons_resolution, drop_last=True)
    code = code_each(X, W, H0, alpha=1, sub_iter=10, stopping_diff=0.01)
    img, count = fold_patches(W @ code, np.asarray(A).shape, k, recons_resolution, drop_last=True)
    return code, img, count
