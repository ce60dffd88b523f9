"""dsw_amd - MI355X (gfx950) native hot path of DeepSphere-Weather.

Chebyshev graph convolution (ConvCheb / conv_cheb) and sparse interpolation pooling
(RemapBlock) behind the reference's own nn.Module API; see ``modules/`` next to this package for
the drop-in import path (``modules.layers`` etc.) and DESIGN.md at the repo root.
"""
from . import _native  # noqa: F401
from .functional import (  # noqa: F401
    CsrOperator,
    cheb_basis,
    cheb_conv,
    get_operator,
    set_test_backend,
    sparse_remap,
)

__version__ = "0.1.0"
