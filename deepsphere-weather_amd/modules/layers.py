"""Drop-in ``modules.layers`` for the DeepSphere-Weather hot path, backed by gfx950 HIP kernels.

Mirrors the public names, signatures, parameter / buffer names and error behaviour of the hot-path
subset of ``/root/reference/modules/layers.py`` so that the reference's model definitions
(``modules/my_models_graph.py``) and configs run unchanged on top of it:

=============================  ==========================================================
this file                      reference ``modules/layers.py``
=============================  ==========================================================
``estimate_lmax``              ``:57-69``
``scale_operator``             ``:72-79``
``prepare_torch_laplacian``    ``:82-106``
``conv_cheb``                  ``:113-180``
``ConvCheb``                   ``:183-376``
``build_pooling_matrices``     ``:576-581`` (xsphere/CDO if present, else dsw_amd.sphere)
``convert_to_torch_sparse``    ``:584-594``
``RemapBlock``                 ``:948-968``
``GeneralAvgPool/Unpool``      ``:971-987``
``GeneralMaxAreaPool/Unpool``  ``:991-1036`` (same SpMM with a 0/1 matrix)
``GeneralMaxValPool/Unpool``   ``:1040-1103`` (segmented arg-max / gather kernels, dsw_pool.hip)
``PoolUnpoolBlock``            ``:1139-1191``
``get_conv_fun``               ``:1198-1201``
``GeneralConvBlock``           ``:1204-1242``
=============================  ==========================================================

The arithmetic (``torch.sparse.mm`` at ``:164,167,962`` and ``matmul`` at ``:177`` plus the layout
copies around them) is replaced by ``dsw_amd.functional`` -> ``libdsw_hip.so``.  Everything that is
not on that path (image convolutions, HEALPix/equiangular max/avg pooling, cotan Laplacian) is out of scope and raises ``NotImplementedError`` with a pointer to DESIGN.md.
"""
import math
from abc import ABC, abstractmethod

import numpy as np
import torch
from scipy import sparse
from scipy.sparse import linalg as sparse_linalg

from dsw_amd import functional as _F

_OUT_OF_SCOPE = (
    "{} is outside the ConvCheb / interpolation-pooling hot path rebuilt for MI355X "
    "(see DESIGN.md, 'Out of scope'); use the reference implementation for it."
)

_RELU_LIKE = {
    "relu", "celu", "selu", "prelu", "hardswish", "mish", "silu", "gelu", "softplus", "softmax",
    "logsigmoid", "relu6", "rrlu", "leaky_relu", "elu",
}
_LINEAR_LIKE = {"linear", "hardshrink ", "sigmoid", "hardsigmoid", "tanh", "hardtanh", "softsign"}


# ----------------------------------------------------------------------------------------------
# Operator preparation
# ----------------------------------------------------------------------------------------------
def estimate_lmax(laplacian, tol=5e-3):
    """Largest eigenvalue (ARPACK, one Ritz value) with the reference's (1 + 2 tol) safety margin.

    The reference (layers.py:57-69) lets ARPACK draw its start vector: at ``tol = 5e-3`` the estimate then jitters by
    ~3e-4 from call to call, i.e. two builds of one model - or the replicas of a data-parallel job - convolve with
    slightly different operators.  Here the start vector is a fixed seeded draw: same algorithm and tolerance, a value
    inside the reference's own scatter, but the SAME value every time."""
    n = laplacian.shape[0]
    v0 = np.random.default_rng(20210317).standard_normal(n).astype(laplacian.dtype if laplacian.dtype.kind == "f" else np.float64)
    ev = sparse_linalg.eigs(laplacian, k=1, tol=tol, ncv=min(n, 10), return_eigenvectors=False, v0=v0)
    return float(np.real(ev[0])) * (1 + 2 * tol)


def scale_operator(laplacian, lmax, scale=1):
    """Map the spectrum from [0, lmax] to [-scale, scale]: ``L * (2 scale / lmax) - I``."""
    eye = sparse.identity(laplacian.shape[0], format=laplacian.format, dtype=laplacian.dtype)
    laplacian *= 2 * scale / lmax
    laplacian -= eye
    return laplacian


def _scipy_to_coalesced_coo(mat, dtype):
    mat = sparse.coo_matrix(mat)
    index = torch.from_numpy(np.stack((mat.row, mat.col)).astype(np.int64))
    out = torch.sparse_coo_tensor(index, mat.data, mat.shape, dtype=dtype, device=index.device)
    return out.coalesce()


def prepare_torch_laplacian(laplacian, lmax=None):
    """scipy Laplacian -> rescaled, coalesced torch sparse COO (default dtype, int64 indices).

    ``lmax`` may be given to bypass the (nondeterministic) ARPACK estimate; the default follows
    the reference exactly.
    """
    laplacian = laplacian.astype(np.float32)
    if lmax is None:
        lmax = estimate_lmax(laplacian)
    laplacian = scale_operator(laplacian, lmax)
    laplacian = sparse.coo_matrix(laplacian, laplacian.dtype)
    return _scipy_to_coalesced_coo(laplacian, torch.get_default_dtype())


def convert_to_torch_sparse(mat):
    """scipy sparse matrix -> coalesced torch sparse COO in the default dtype."""
    return _scipy_to_coalesced_coo(mat, torch.get_default_dtype())


def compute_cotan_laplacian(graph, return_mass=False):
    raise NotImplementedError(_OUT_OF_SCOPE.format("compute_cotan_laplacian (needs igl)"))


# ----------------------------------------------------------------------------------------------
# Chebyshev graph convolution
# ----------------------------------------------------------------------------------------------
def conv_cheb(laplacian, inputs, weight):
    """Chebyshev convolution ``[B, V, Fin] -> [B, V, Fout]`` with ``weight [Fin, K, Fout]``.

    ``laplacian`` is the torch sparse COO operator (as produced by ``prepare_torch_laplacian``) or
    an already derived ``dsw_amd.CsrOperator``.
    """
    _, _, fin_x = inputs.shape
    fin_w, _, _ = weight.shape
    if fin_x != fin_w:
        raise ValueError(
            "Input tensor shape does not match the expected shape: \n"
            + "- Input tensor shape :{} \n".format(fin_x)
            + "- Expected tensor shape :{} \n".format(fin_w)
        )
    op = laplacian if isinstance(laplacian, _F.CsrOperator) else _F.get_operator(laplacian)
    return _F.cheb_conv(op, inputs, weight)


class _Fp32OperatorMixin:
    """Keeps the sparse operator buffers of a module at (at least) fp32 across ``.to(dtype)`` / ``.bfloat16()`` /
    ``.half()`` casts.  The reference rounds the Laplacian / remap matrix to the activation dtype on such a cast
    (its CPU path then also accumulates in bf16); here the kernels keep operator and accumulators in fp32 whatever
    the activation storage type is, so rounding the buffer to an 8-bit mantissa would only lose accuracy (pooling rows
    would no longer sum to 1, the Laplacian spectrum would move by ~4e-3 per hop).  Device moves and ``.double()``
    behave as in the reference."""

    _operator_buffers = ()

    def _apply(self, fn, *args, **kwargs):
        before = {n: self._buffers.get(n) for n in self._operator_buffers}
        out = super()._apply(fn, *args, **kwargs)
        for name, old in before.items():
            new = self._buffers.get(name)
            if old is None or new is None or new is old:
                continue
            if new.dtype in (torch.bfloat16, torch.float16) and old.dtype not in (torch.bfloat16, torch.float16):
                self._buffers[name] = old.to(new.device)
        return out


class ConvCheb(_Fp32OperatorMixin, torch.nn.Module):
    """Graph convolution with Chebyshev polynomials of the rescaled Laplacian (Defferrard 2016).

    Input ``(sample, node, feature)``; parameters ``weight [in, kernel_size, out]``, ``bias [out]``
    (or ``None``); buffer ``laplacian`` (sparse COO, part of the state_dict).  ``kernel_size - 1`` is
    the polynomial order (1 = no neighbourhood, 2 = one hop, ...).
    """

    _operator_buffers = ("laplacian",)

    def __init__(self, in_channels, out_channels, kernel_size, laplacian, bias=True, conv=conv_cheb, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self._conv = conv
        self.register_buffer("laplacian", laplacian)
        self.weight = torch.nn.Parameter(torch.empty(in_channels, kernel_size, out_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self, activation="relu", fan="in", distribution="normal"):
        """He (relu / fan-in / normal), Glorot (linear / avg / uniform) or LeCun (linear / in) init."""
        widths = {
            "in": self.in_channels,
            "out": self.out_channels,
            "avg": (self.in_channels + self.out_channels) / 2,
        }
        if fan not in widths:
            raise ValueError("unknown fan")
        fan = widths[fan] * self.kernel_size
        if activation in _RELU_LIKE:
            gain = 2  # half of the activations are zeroed
        elif activation in _LINEAR_LIKE:
            gain = 1
        else:
            raise ValueError("Unknown activation")
        if distribution == "normal":
            self.weight.data.normal_(0, math.sqrt(gain / fan))
        elif distribution == "uniform":
            bound = math.sqrt(3 * gain / fan)
            self.weight.data.uniform_(-bound, bound)
        else:
            raise ValueError("Unknown distribution")
        if self.bias is not None:
            self.bias.data.fill_(0)

    def set_parameters(self, weight, bias=None):
        """Replace weight ``[in, kernel_size, out]`` (and bias ``[out]``) by the given arrays."""
        self.weight = torch.nn.Parameter(torch.as_tensor(weight))
        if bias is not None:
            self.bias = torch.nn.Parameter(torch.as_tensor(bias))

    def extra_repr(self):
        return "{} -> {}, kernel_size={}, bias={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.bias is not None
        )

    supports_fused_activation = True   # forward(x, activation="relu") (extension; ConvBlock uses it)

    def forward_activated(self, inputs, activation="relu"):
        """``activation(forward(inputs))`` with the activation applied in the epilogue of the channel-mix kernel
        (one pass less over the output than ``F.relu(conv(x))``, my_models_graph.py:108-114).  Only for the built-in
        ``conv_cheb``; a custom ``conv=`` callable gets the plain two-step evaluation.  Same as
        ``self(inputs, activation=...)`` minus the module hooks."""
        if self._conv is conv_cheb and activation == "relu" and inputs.shape[2] == self.weight.shape[0]:
            return _F.cheb_conv(_F.get_operator(self.laplacian), inputs, self.weight, self.bias, activation="relu")
        return getattr(torch.nn.functional, activation)(self.forward(inputs))

    def forward(self, inputs, activation=None):
        """``inputs``: n_signals x n_vertices x n_features.  ``activation`` (extension, default None = the reference's
        signature): name of a ``torch.nn.functional`` activation applied to the result - "relu" rides in the
        channel-mix epilogue."""
        if activation is not None:
            return self.forward_activated(inputs, activation)
        # weight / bias / laplacian are read by attribute on every call (SWAG re-assigns them)
        if self._conv is conv_cheb:
            # fused path: the bias add rides in the channel-mix epilogue (reference: layers.py:375)
            if inputs.shape[2] != self.weight.shape[0]:
                return conv_cheb(self.laplacian, inputs, self.weight)  # raises the reference's error
            return _F.cheb_conv(_F.get_operator(self.laplacian), inputs, self.weight, self.bias)
        outputs = self._conv(self.laplacian, inputs, self.weight)
        if self.bias is not None:
            outputs += self.bias
        return outputs


class Conv2dEquiangular(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(_OUT_OF_SCOPE.format("Conv2dEquiangular (conv_type='image')"))


# ----------------------------------------------------------------------------------------------
# Interpolation pooling
# ----------------------------------------------------------------------------------------------
def build_pooling_matrices(src_graph, dst_graph):
    """(pool, unpool) scipy matrices between two samplings (``src`` finer than ``dst``).

    The reference derives them from CDO conservative-remap weights via ``xsphere``
    (``layers.py:531-581``).  When that tool-chain is importable it is used; otherwise the
    self-contained builder of ``dsw_amd.sphere`` supplies matrices with the same invariants
    (row-stochastic pool / unpool) - value parity with CDO is unpinned, see DESIGN.md.
    """
    try:
        from xsphere.remapping import compute_interpolation_weights  # noqa: F401
    except Exception:
        from dsw_amd import sphere

        return sphere.build_pooling_matrices(src_graph, dst_graph)
    ds = compute_interpolation_weights(
        src_graph=src_graph, dst_graph=dst_graph, method="conservative", normalization="fracarea"
    )
    rows = np.array(ds.dst_address) - 1  # CDO is 1-based
    cols = np.array(ds.src_address) - 1
    weights = sparse.csr_matrix((ds.remap_matrix.squeeze(), (rows, cols)))
    assert weights.shape == (dst_graph.n_vertices, src_graph.n_vertices)
    np.testing.assert_allclose(weights.sum(axis=1), 1)
    weights = weights.multiply(ds.dst_grid_area.values[:, np.newaxis])
    pool = weights.multiply(1 / weights.sum(1))
    unpool = weights.multiply(1 / weights.sum(0)).T
    return pool, unpool


class RemapBlock(_Fp32OperatorMixin, torch.nn.Module):
    """Mesh-based pooling / unpooling: ``y[b, d, f] = sum_v M[d, v] x[b, v, f]``.

    ``remap_matrix`` (sparse COO ``[V_dst, V_src]``) is a registered buffer as in the reference.
    The result is returned contiguous ``[B, V_dst, F]`` (the reference returns a permuted view with
    the same values).
    """

    _operator_buffers = ("remap_matrix",)
    supports_out = True    # forward(..., out=slice) writes into a preallocated channel slice (concat-free decoder)

    def __init__(self, remap_matrix):
        super().__init__()
        self.register_buffer("remap_matrix", self.process_remap_matrix(remap_matrix))

    def forward(self, x, *args, out=None, add=None, **kwargs):
        """``out`` (extension): a preallocated ``[B, V_dst, F]`` channel slice of a wider tensor to write into.
        ``add`` (extension): a ``[B, V_dst, F]`` tensor added to the product in the kernel's epilogue (see ``forward_add``)."""
        if add is not None:
            if out is not None:
                raise ValueError("RemapBlock: `out` and `add` cannot be combined")
            return _F.sparse_remap_add(_F.get_operator(self.remap_matrix), x, add)
        return _F.sparse_remap(_F.get_operator(self.remap_matrix), x, out=out)

    def forward_fork(self, x):
        """``(x_again, forward(x))`` (extension) for an input that has a second consumer, e.g. the skip connection of a
        U-Net: hand ``x_again`` to that consumer and its gradient is added inside this layer's backward product."""
        return _F.sparse_remap_fork(_F.get_operator(self.remap_matrix), x)

    def forward_add(self, x, addend):
        """``forward(x) + addend`` in one launch (extension): the addend rides in the epilogue of the product - an unpooled
        coarse result added to a tensor of the fine level costs no separate pass."""
        return self(x, add=addend)

    def process_remap_matrix(self, mat):
        return convert_to_torch_sparse(mat)


class _IndexlessPool(RemapBlock):
    """Pooling layers of the remap family return ``(x, None)`` (there are no source indices to hand to the unpooling)."""

    def forward(self, x, *args, **kwargs):
        return super().forward(x, *args, **kwargs), None

    def forward_fork(self, x):
        x_again, y = super().forward_fork(x)
        return x_again, (y, None)


class GeneralAvgPool(_IndexlessPool):
    """Interpolation (area-average) pooling; returns ``(x, None)`` (no source indices)."""


class GeneralAvgUnpool(RemapBlock):
    """Interpolation unpooling; extra positional arguments (pool indices) are ignored."""

    def forward(self, x, *args, **kwargs):
        return super().forward(x, *args, **kwargs)


def _argmax_selector(mat, axis):
    """0/1 selection matrix keeping, per row (axis=1) or per column (axis=0), the largest weight."""
    mat = sparse.csr_matrix(mat)
    if axis == 1:
        cols = np.asarray(mat.argmax(axis=1)).ravel()
        rows = np.arange(mat.shape[0])
    else:
        rows = np.asarray(mat.argmax(axis=0)).ravel()
        cols = np.arange(mat.shape[1])
    index = torch.from_numpy(np.stack((rows, cols)).astype(np.int64))
    sel = torch.sparse_coo_tensor(
        index, torch.ones(index.shape[1]), mat.shape, dtype=torch.get_default_dtype()
    )
    return sel.coalesce()


class GeneralMaxAreaPool(_IndexlessPool):
    """Pooling that copies, per coarse cell, the fine cell with the largest overlap area."""

    def process_remap_matrix(self, mat):
        return _argmax_selector(mat, axis=1)


class GeneralMaxAreaUnpool(RemapBlock):
    """Unpooling counterpart of ``GeneralMaxAreaPool`` (constructed from ``pool_mat.T``)."""

    def process_remap_matrix(self, mat):
        return _argmax_selector(mat, axis=0)


class GeneralMaxValPool(RemapBlock):
    """Max-value pooling: per coarse cell, sample and channel the value of the overlapping fine cell whose
    area-WEIGHTED value is largest (reference ``layers.py:1040-1079``).

    Returns ``(x_pooled [B, Vd, F], index)``.  ``index_format = "reference"`` (the default: what code written against the
    reference sees): the reference's ``[2, B*F*Vd]`` int64 tensor; ``"compact"`` (set by this package's own
    ``UNetSpherical``, which only hands the index to the unpooling): the int32 ``[B, Vd, F]`` selection the kernel
    produces (the chosen fine cell per output element) - the same information at a sixteenth of the size and without the
    conversion pass.  :meth:`reference_index` converts; :class:`GeneralMaxValUnpool` accepts either form."""

    supports_out = False
    forward_fork = None    # the selection gradient is a scatter, not a product with an epilogue
    index_format = "reference"

    def forward(self, x, *args, **kwargs):
        y, sel = _F.maxval_pool(_F.get_operator(self.remap_matrix), x)
        if self.index_format == "compact":
            return y, sel
        return y, _F.maxval_reference_index(sel)

    @staticmethod
    def reference_index(index):
        return _F.maxval_reference_index(index)


class GeneralMaxValUnpool(RemapBlock):
    """Max-value unpooling: coarse values go back to the fine cells they were pooled from, zeros elsewhere
    (reference ``layers.py:1082-1103``; only the SHAPE of ``remap_matrix`` is used, as in the reference)."""

    supports_out = False

    def forward(self, x, index, *args, **kwargs):
        B, D, F_ = x.shape
        if index.dim() == 2:   # the reference's [2, B*F*Vd] form
            index = _F.maxval_compact_index(index, B, D, F_)
        return _F.maxval_unpool(x, index, self.remap_matrix.shape[0])


class PoolUnpoolBlock(torch.nn.Module):
    """Factories of (pooling, unpooling) layer pairs."""

    @staticmethod
    def getPoolUnpoolLayer(sampling: str, pool_method: str, **kwargs):
        sampling = sampling.lower()
        pool_method = pool_method.lower()
        assert sampling in ("healpix", "equiangular")
        assert pool_method in ("max", "avg")
        raise NotImplementedError(_OUT_OF_SCOPE.format(f"{sampling} '{pool_method}' pooling"))

    @staticmethod
    def getGeneralPoolUnpoolLayer(src_graph, dst_graph, pool_method: str):
        if src_graph.n_vertices < dst_graph.n_vertices:  # src is always the finer sampling
            src_graph, dst_graph = dst_graph, src_graph
        pool_mat, unpool_mat = build_pooling_matrices(src_graph, dst_graph)
        if pool_method == "interp":
            return GeneralAvgPool(pool_mat), GeneralAvgUnpool(unpool_mat)
        if pool_method == "maxarea":
            return GeneralMaxAreaPool(pool_mat), GeneralMaxAreaUnpool(pool_mat.T)
        if pool_method == "learn":
            raise NotImplementedError()
        if pool_method == "maxval":
            return GeneralMaxValPool(pool_mat), GeneralMaxValUnpool(unpool_mat)
        raise ValueError(f"{pool_method} is not supoorted.")


# ----------------------------------------------------------------------------------------------
# Convolution factory (the plugin boundary the model files call)
# ----------------------------------------------------------------------------------------------
def get_conv_fun(conv_type):
    return {"image": Conv2dEquiangular, "graph": ConvCheb}[conv_type]


class GeneralConvBlock(ABC, torch.nn.Module):
    @abstractmethod
    def forward(self, *args, **kwargs):
        pass

    @staticmethod
    def getConvLayer(in_channels: int, out_channels: int, kernel_size: int, conv_type: str = "graph", **kwargs):
        conv_type = conv_type.lower()
        if conv_type == "graph":
            assert "laplacian" in kwargs
            kwargs.pop("lonlat_ratio")
            kwargs.pop("periodic_padding")
        elif conv_type == "image":
            assert "lonlat_ratio" in kwargs
            kwargs.pop("laplacian")
        else:
            raise ValueError(
                "{} conv_type is not supported. Choose either 'graph' or 'image'".format(conv_type)
            )
        return get_conv_fun(conv_type)(in_channels, out_channels, kernel_size, **kwargs)
