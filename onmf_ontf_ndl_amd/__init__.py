"""MI355X-native online NMF / NTF minibatch step behind the reference's class surface.

    from onmf_ontf_ndl_amd.onmf import Online_NMF, update_code_within_radius   # src/onmf.py
    from onmf_ontf_ndl_amd.ontf import Online_NTF                              # src/ontf.py
    from onmf_ontf_ndl_amd.legacy.onmf import Online_NMF                       # older call form

The arithmetic lives in libonmf_hip.so (hand-written HIP for gfx950, C-ABI in include/onmf_hip.h).
Importing the package does not need a GPU; creating an engine does.
"""
from .onmf import Online_NMF, update_code_within_radius, sparse_code_update, dict_update  # noqa: F401
from .ontf import Online_NTF  # noqa: F401

__version__ = "0.1.0"
