"""Adapter for the OLDER ``Online_NMF`` calling convention that the reference's driver scripts
still use: keyword arguments ``ini_A / ini_B / ini_C`` and a five-tuple from ``train_dict``
(image_reconstruction.py:289-312, network_reconstruction_nx.py:360-377,
ising_reconstruction.py:116-127).  The shipped ``src/onmf.py`` no longer accepts that form; this
module maps it onto the current engine:

    from onmf_ontf_ndl_amd.legacy.onmf import Online_NMF, update_code_within_radius

The third aggregate C (X X^T) is always tracked, as those callers expect.
"""
from __future__ import annotations

from .. import onmf as _current
from ..onmf import update_code_within_radius  # noqa: F401  (image_reconstruction.py:1 imports it)

__all__ = ["Online_NMF", "update_code_within_radius"]


class Online_NMF(_current.Online_NMF):

    def __init__(self, X, n_components=100, iterations=500, batch_size=20, ini_dict=None,
                 ini_A=None, ini_B=None, ini_C=None, history=0, alpha=None, beta=None,
                 subsample=False, **kw):
        agg = None
        if ini_dict is not None:
            if ini_A is None or ini_B is None or ini_C is None:
                raise TypeError("ini_dict needs ini_A, ini_B and ini_C")
            agg = [ini_A, ini_B, ini_C]
        super().__init__(X, n_components=n_components, iterations=iterations, batch_size=batch_size,
                         ini_dict=ini_dict, ini_agg=agg, history=history, alpha=alpha, beta=beta,
                         subsample=subsample, **kw)

    def train_dict(self):
        W, agg, code = super().train_dict(full_code=True)
        return W, agg[0], agg[1], agg[2], code
