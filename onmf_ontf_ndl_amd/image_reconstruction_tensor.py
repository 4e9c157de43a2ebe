"""The tensor-patch callers of ``Online_NTF`` from the reference's ``image_reconstruction_tensor.py``:
``Image_Reconstructor`` with its data path only (patch tensors, dictionary training along a mode,
reconstruction by overlap-averaging) - no plotting, no files written.

Same constructor arguments and method names as the reference (image_reconstruction_tensor.py:15-328);
``path`` may also be an (h, w[, 3]) array already scaled to [0, 1].

* ``extract_random_patches`` (:87-124)  (k^2, 3, n) colour / (k^2, n, 1) grey patch tensors, one
  gather kernel instead of ``np.append`` in a loop;
* ``train_dict(mode, learn_joint_dict)`` (:220-262)  the warm-restarted ``Online_NTF`` chain;
* ``reconstruct_image`` (:269-285), ``reconstruct_image_color`` (:287-328)  the per-window LassoLars
  solves (positive codes, alpha = 1) as ONE batched launch of the same path, one window per wavefront
  (``Online_NTF``'s default coder) - then product and overlap average on the GPU.
"""
from __future__ import annotations

import numpy as np

from .engine import get_engine
from .image_reconstruction import downscale_local_mean, _to_numpy
from .onmf import _is_torch, _out
from .ontf import Online_NTF

__all__ = ["Image_Reconstructor"]


class Image_Reconstructor():
    def __init__(self,
                 path,
                 n_components=100,
                 iterations=50,
                 sub_iterations=20,
                 batch_size=20,
                 block_iterations=20,
                 num_patches=1000,
                 sub_num_patches=10000,
                 downscale_factor=2,
                 patch_size=7,
                 patches_file='',
                 learn_joint_dict=False,
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
        self.block_iterations = block_iterations
        self.num_patches = num_patches
        self.sub_num_patches = sub_num_patches
        self.batch_size = batch_size
        self.downscale_factor = downscale_factor
        self.patch_size = patch_size
        self.patches_file = patches_file
        self.learn_joint_dict = learn_joint_dict
        self.is_matrix = is_matrix
        self.is_stack = is_stack
        self.is_color = is_color
        self.rng = rng
        self._device = device
        self.W = np.zeros(shape=(patch_size, n_components))
        self.code = np.zeros(shape=(n_components, iterations * batch_size))
        self.ntf = None
        self.data = self.read_img_as_array()
        self._data_dev = None

    def _engine(self):
        return get_engine(self._device)

    def read_img_as_array(self, path=None):
        """Image as an array in [0, 1] (image_reconstruction_tensor.py:68-85).  Arrays pass through."""
        path = self.path if path is None else path
        if not isinstance(path, (str, bytes)) and hasattr(path, "shape"):
            return path
        if self.is_matrix:
            return (np.load(path) + 1) / 2
        try:
            from PIL import Image
        except ImportError as e:                                  # pragma: no cover
            raise ImportError("reading image files needs Pillow; pass the image as an array instead") from e
        img = Image.open(path)
        img = img.convert('RGB') if self.is_color else img.convert('L')
        return np.asarray(img) / 255

    # -- patches --------------------------------------------------------------------------
    def _random_patches_dev(self, n):
        eng = self._engine()
        if self._data_dev is None:
            self._data_dev = eng.dev(self.data)
        img = self._data_dev
        k = self.patch_size
        h, w = img.shape[0], img.shape[1]
        if self.rng == "numpy":
            corners = np.empty((n, 2), dtype=np.int32)
            for i in range(n):                                   # (:102-103): row, then column, per patch
                corners[i, 0] = np.random.choice(h - k)
                corners[i, 1] = np.random.choice(w - k)
        else:
            torch = eng.torch
            corners = torch.stack([torch.randint(h - k, (n,), device=eng.device),
                                   torch.randint(w - k, (n,), device=eng.device)], dim=1)
        return eng.extract_patches(img, k, corners=corners)

    def extract_random_patches(self, second_factor=False):
        """(k^2, 3, n) for a colour image, (k^2, n, 1) for a grey one (image_reconstruction_tensor.py:87-124)."""
        n = int(self.sub_num_patches if second_factor else self.num_patches)
        X = self._random_patches_dev(n)
        k2 = self.patch_size ** 2
        T = X.reshape(k2, 3, n) if np.ndim(self.data) == 3 else X.reshape(k2, n, 1)
        return _out(T, _is_torch(self.data))

    # -- training -------------------------------------------------------------------------
    def train_dict(self, mode, learn_joint_dict):
        """``iterations`` rounds of: fresh patch tensor -> ``Online_NTF`` of ``sub_iterations`` along
        ``mode``, warm-started from the previous round (image_reconstruction_tensor.py:220-262)."""
        eng = self._engine()
        W = At = Bt = None
        k2 = self.patch_size ** 2
        for t in range(self.iterations):
            n = int(self.num_patches)
            X = self._random_patches_dev(n)
            T = X.reshape(k2, 3, n) if np.ndim(self.data) == 3 else X.reshape(k2, n, 1)
            kw = dict(iterations=self.sub_iterations, sub_iterations=self.block_iterations,
                      batch_size=self.batch_size, learn_joint_dict=learn_joint_dict, mode=mode, rng=self.rng,
                      device=self._device)
            if t == 0:
                self.ntf = Online_NTF(T, self.n_components, **kw)
            else:
                self.ntf = Online_NTF(T, self.n_components, ini_dict=W, ini_A=At, ini_B=Bt,
                                      history=self.ntf.history, **kw)
            W, At, Bt, H = self.ntf.train_dict_single()
        self.W = _out(eng.dev(W), False)
        return self.W

    # -- reconstruction -------------------------------------------------------------------
    def _code_windows(self, X, alpha):
        eng = self._engine()
        coder = Online_NTF(X, self.n_components, alpha=alpha, rng=self.rng, device=self._device)
        return coder._code_dev(eng, X, eng.dev(self.W))           # r x b

    def reconstruct_image(self, path, downscale_factor=None, patch_size=10, is_matrix=False):
        """Grey image: every window coded (alpha = 2, the coder's default), averaged back; returns the
        code (windows x atoms) and keeps the image in ``self.img_recons``
        (image_reconstruction_tensor.py:269-285; its ``np.dot(self.W, code)`` is read as W code^T)."""
        if downscale_factor is None:
            downscale_factor = self.downscale_factor
        data = _to_numpy(self.read_img_as_array(path))
        if data.ndim == 3:
            data = data.mean(axis=2)
        if downscale_factor > 1:
            data = downscale_local_mean(data, (downscale_factor, downscale_factor))
        eng = self._engine()
        h, w = data.shape
        k = patch_size
        grid = (h - k + 1, w - k + 1)
        X = eng.extract_patches(eng.dev(data), k, grid=grid, stride=1)
        code = self._code_windows(X, None)
        out, _ = eng.fold_patches(eng.patch_product(eng.dev(self.W), code), (h, w), k, grid, stride=1)
        self.img_recons = _out(out, False)
        return _out(code.t().contiguous(), False)

    def reconstruct_image_color(self, path, recons_resolution=1):
        """Colour image: windows on the ``range(0, n - k, resolution)`` grid, each coded with alpha = 1,
        painted back with the running mean over overlaps (image_reconstruction_tensor.py:287-328)."""
        eng = self._engine()
        A = self.read_img_as_array(path)
        like_torch = _is_torch(A)
        Ad = eng.dev(A)
        if Ad.dim() != 3:
            raise ValueError("reconstruct_image_color needs an (h, w, channels) image")
        k, s = self.patch_size, int(recons_resolution)
        h, w = Ad.shape[0], Ad.shape[1]
        grid = (len(range(0, h - k, s)), len(range(0, w - k, s)))
        if grid[0] == 0 or grid[1] == 0:
            self.A_recons = _out(eng.zeros(*Ad.shape), like_torch)
            return self.A_recons
        X = eng.extract_patches(Ad, k, grid=grid, stride=s)
        code = self._code_windows(X, 1)
        out, count = eng.fold_patches(eng.patch_product(eng.dev(self.W), code), tuple(Ad.shape), k, grid, stride=s)
        self.A_overlap_count = _out(count, like_torch)
        self.A_recons = _out(out, like_torch)
        return self.A_recons
