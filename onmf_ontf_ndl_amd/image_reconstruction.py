"""The patch-side callers of the ONMF step from the reference's ``image_reconstruction.py``:
``Image_Reconstructor`` with its data path only (patch extraction, dictionary training,
reconstruction by overlap-averaging) - no plotting, no files written.

Same constructor arguments and method names as the reference (image_reconstruction.py:14-406);
images may also be handed over as arrays (``path`` may be an ``ndarray`` / tensor, h x w[ x 3],
already scaled to [0, 1]).  Everything numeric runs on the GPU through libonmf_hip.so:

* ``extract_random_patches``  (:173-206)  one gather kernel instead of ``np.append`` in a loop;
* ``image_to_patches``        (:117-171)  all windows (``is_recons``) or ``max_patches`` random ones;
* ``train_dict``              (:277-319)  the warm-restarted ``Online_NMF`` chain, state kept on the GPU;
* ``reconstruct_image``       (:340-356)  one batched solve + product + overlap average;
* ``reconstruct_image_color`` (:358-406)  the per-window solves (each with its own stopping test,
  as the reference's one-call-per-window loop has) as ONE launch, then product + overlap average.

Extra keyword-only arguments: ``rng`` ("numpy": corners, dictionaries and starting codes are drawn
from NumPy's global generator in the reference's order; "device": drawn on the GPU), ``device``.
"""
from __future__ import annotations

import numpy as np

from .engine import get_engine
from .legacy.onmf import Online_NMF
from .onmf import _is_torch, _out

__all__ = ["Image_Reconstructor", "downscale_local_mean"]


def downscale_local_mean(image, factors):
    """Block mean with zero padding up to a multiple of the block (what skimage's function of this
    name does, image_reconstruction.py:157, :218, :274); host NumPy, 2-D or 3-D input."""
    image = np.asarray(image, dtype=np.float64)
    factors = tuple(factors) + (1,) * (image.ndim - len(factors))
    pad = [(0, (-n) % f) for n, f in zip(image.shape, factors)]
    image = np.pad(image, pad, mode="constant")
    shape = []
    for n, f in zip(image.shape, factors):
        shape += [n // f, f]
    return image.reshape(shape).mean(axis=tuple(range(1, 2 * image.ndim, 2)))


class Image_Reconstructor():
    def __init__(self,
                 path,
                 n_components=100,
                 iterations=200,
                 sub_iterations=20,
                 num_patches=1000,
                 batch_size=20,
                 downscale_factor=2,
                 patch_size=7,
                 patches_file='',
                 patches=None,
                 is_matrix=False,
                 is_stack=False,
                 is_color=True,
                 *,
                 rng="numpy",
                 device=None):
        self.path = path
        self.n_components = n_components
        self.iterations = iterations
        self.sub_iterations = sub_iterations
        self.num_patches = num_patches
        self.batch_size = batch_size
        self.downscale_factor = downscale_factor
        self.patch_size = patch_size
        self.patches = patches
        self.patches_file = patches_file
        self.is_matrix = is_matrix
        self.is_stack = is_stack
        self.is_color = is_color
        if rng not in ("numpy", "device"):
            raise ValueError("rng must be 'numpy' or 'device'")
        self.rng = rng
        self._device = device
        self.W = np.zeros(shape=(patch_size ** 2, n_components))
        if is_color:
            self.W = np.random.rand(3 * patch_size ** 2, n_components)      # image_reconstruction.py:50
        self.code = np.ones(shape=(n_components, iterations * batch_size))
        self.A_recons = []
        self.nmf = None
        self.data = self.read_img_as_array(path=self.path, is_matrix=self.is_matrix, is_color=self.is_color)
        self._data_dev = None

    # -- plumbing -------------------------------------------------------------------------
    def _engine(self):
        return get_engine(self._device)

    def _image_dev(self, data=None):
        eng = self._engine()
        if data is None:
            if self._data_dev is None:
                self._data_dev = eng.dev(self.data)
            return self._data_dev
        return eng.dev(data)

    # -- reading --------------------------------------------------------------------------
    def read_img_as_array(self, path, is_matrix=False, is_color=True):
        """Image as an array in [0, 1] (image_reconstruction.py:73-92).  Arrays pass through."""
        if not isinstance(path, (str, bytes)) and hasattr(path, "shape"):
            return path
        if is_matrix:
            img = np.load(path)
            return (img + 1) / 2                                  # a +-1 matrix becomes 0-1
        try:
            from PIL import Image
        except ImportError as e:                                  # pragma: no cover
            raise ImportError("reading image files needs Pillow; pass the image as an array instead") from e
        img = Image.open(path)
        img = img.convert('RGB') if is_color else img.convert('L')
        return np.asarray(img) / 255

    def get_downscaled_dims(self, path, downscale_factor=None, is_matrix=False):
        if downscale_factor is None:
            downscale_factor = self.downscale_factor
        img = self.read_img_as_array(path, is_matrix=False, is_color=False) if not is_matrix else np.load(path)
        data = downscale_local_mean(_to_numpy(img), (downscale_factor, downscale_factor))
        return data.shape[0], data.shape[1]

    # -- patches --------------------------------------------------------------------------
    def extract_random_patches(self):
        """``num_patches`` random k x k windows as columns: (3k^2 x n) colour, (k^2 x n) grey
        (image_reconstruction.py:173-206).  Corners are drawn per patch, row then column, like the
        reference's two ``np.random.choice`` calls (:185-186)."""
        return _out(self._random_patches_dev(), _is_torch(self.data))

    def _random_patches_dev(self):
        eng = self._engine()
        img = self._image_dev()
        k = self.patch_size
        h, w = img.shape[0], img.shape[1]
        n = int(self.num_patches)
        if self.rng == "numpy":
            corners = np.empty((n, 2), dtype=np.int32)
            for i in range(n):
                corners[i, 0] = np.random.choice(h - k)
                corners[i, 1] = np.random.choice(w - k)
        else:
            torch = eng.torch
            corners = torch.stack([torch.randint(h - k, (n,), device=eng.device),
                                   torch.randint(w - k, (n,), device=eng.device)], dim=1)
        return eng.extract_patches(img, k, corners=corners)

    def image_to_patches(self, path, patch_size=7, iterations=500, batch_size=20, color=False, downscale_factor=2,
                         is_matrix=False, is_recons=False):
        """Image -> (k^2 x n) patch matrix (image_reconstruction.py:117-171): every window when
        ``is_recons`` (their order matters for putting them back), else ``iterations*batch_size`` random
        ones drawn the way sklearn's ``extract_patches_2d(max_patches=...)`` draws them."""
        data = _to_numpy(self.read_img_as_array(path, is_matrix=is_matrix, is_color=bool(color)))
        if color and not is_matrix:
            # the reference unfolds the RGB image along its channel mode and transposes
            # (image_reconstruction.py:139-142: tl_unfold(img, mode=2).T): an (h*w) x 3 matrix, whose
            # windows are then taken like a grey image's - only patch sizes up to 3 fit into it
            if data.ndim != 3:
                raise ValueError("image_to_patches(color=True) needs an (h, w, channels) image")
            data = np.reshape(np.moveaxis(data, 2, 0), (data.shape[2], -1)).T
        if downscale_factor > 1:
            data = downscale_local_mean(data, (downscale_factor, downscale_factor))
        eng = self._engine()
        img = eng.dev(data)
        h, w = data.shape[0], data.shape[1]
        k = patch_size
        if k > h or k > w:
            # (scikit-learn's extract_patches_2d refuses the same way)
            raise ValueError("image_to_patches: patch size %d does not fit into the %d x %d array" % (k, h, w))
        all_patches = (h - k + 1) * (w - k + 1)
        if is_recons or iterations * batch_size >= all_patches:
            X = eng.extract_patches(img, k, grid=(h - k + 1, w - k + 1), stride=1)
        else:
            n = iterations * batch_size
            i_s = np.random.randint(h - k + 1, size=n)
            j_s = np.random.randint(w - k + 1, size=n)
            X = eng.extract_patches(img, k, corners=np.stack([i_s, j_s], axis=1).astype(np.int32))
        return _out(X, False)

    # -- training -------------------------------------------------------------------------
    def train_dict(self, is_stack=False):
        """``iterations`` rounds of: fresh random patches -> an ``Online_NMF`` of ``sub_iterations``
        warm-started from the previous round's dictionary, aggregates and history
        (image_reconstruction.py:277-319).  Dictionary and aggregates stay on the GPU between rounds."""
        eng = self._engine()
        W = At = Bt = Ct = None
        history = 0
        want_torch = _is_torch(self.data)
        for t in range(self.iterations):
            Xd = self._random_patches_dev()
            self.nmf = Online_NMF(Xd, n_components=self.n_components, iterations=self.sub_iterations,
                                  batch_size=self.batch_size, ini_dict=W, ini_A=At, ini_B=Bt, ini_C=Ct,
                                  history=history, alpha=None, rng=self.rng, device=self._device)
            W, At, Bt, Ct, H = self.nmf.train_dict()
            history = self.nmf.history
        self.W = _out(W, want_torch) if _is_torch(W) else W
        return self.W

    # -- reconstruction -------------------------------------------------------------------
    def reconstruct_image(self, path, downscale_factor=None, patch_size=10, is_matrix=False):
        """Grey image: all windows coded in one solve, averaged back; returns the code and keeps the
        image in ``self.img_recons`` (image_reconstruction.py:340-356, which shows it instead)."""
        if downscale_factor is None:
            downscale_factor = self.downscale_factor
        data = _to_numpy(self.read_img_as_array(path, is_matrix=is_matrix, is_color=False))
        if downscale_factor > 1:
            data = downscale_local_mean(data, (downscale_factor, downscale_factor))
        eng = self._engine()
        img = eng.dev(data)
        h, w = data.shape[0], data.shape[1]
        k = patch_size
        grid = (h - k + 1, w - k + 1)
        X = eng.extract_patches(img, k, grid=grid, stride=1)
        self.nmf = Online_NMF(X, self.n_components, self.iterations, self.batch_size, rng=self.rng,
                              device=self._device)
        code = self.nmf.sparse_code(X, eng.dev(self.W))
        R = eng.patch_product(eng.dev(self.W), code)
        out, _ = eng.fold_patches(R, (h, w), k, grid, stride=1)
        self.img_recons = _out(out, False)
        return _out(code, False)

    def reconstruct_image_color(self, path, recons_resolution=1):
        """Colour image: every window on the ``range(0, n - k, resolution)`` grid is coded on its own
        (alpha=1, ten sweeps at most, its own stopping test at 0.01) and painted back with the
        running mean over overlaps (image_reconstruction.py:358-406).  Returns the image."""
        eng = self._engine()
        A = self.read_img_as_array(path)
        like_torch = _is_torch(A)
        Ad = eng.dev(A)
        if Ad.dim() != 3:
            raise ValueError("reconstruct_image_color needs an (h, w, channels) image")
        k = self.patch_size
        s = int(recons_resolution)
        h, w = Ad.shape[0], Ad.shape[1]
        grid = (len(range(0, h - k, s)), len(range(0, w - k, s)))
        if grid[0] == 0 or grid[1] == 0:
            out = eng.zeros(*Ad.shape)
            self.A_recons = _out(out, like_torch)
            return self.A_recons
        X = eng.extract_patches(Ad, k, grid=grid, stride=s)
        Wd = eng.dev(self.W)
        r, b = Wd.shape[1], X.shape[1]
        H0 = None
        if self.rng == "numpy":
            H0 = eng.dev(np.random.rand(b, r).T)        # one rand(r, 1) per window, in window order
        code = eng.sparse_code_each(X, Wd, H0=H0, alpha=1, sub_iter=10, stopping_diff=0.01)
        R = eng.patch_product(Wd, code)
        out, count = eng.fold_patches(R, tuple(Ad.shape), k, grid, stride=s)
        self.A_overlap_count = _out(count, like_torch)
        self.A_recons = _out(out, like_torch)
        return self.A_recons


def _to_numpy(a):
    if _is_torch(a):
        return a.detach().cpu().numpy()
    return np.asarray(a)
