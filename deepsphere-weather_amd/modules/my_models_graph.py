"""Spherical residual U-Net on the HIP ConvCheb / interpolation-pooling layers.

Counterpart of ``/root/reference/modules/my_models_graph.py`` (ConvBlock ``:26-118``, ResBlock
``:121-216``, UNetSpherical ``:220-564``): same constructor signatures, config keys, sub-module /
parameter / buffer names (``conv1.convblock1.conv.weight``, ``conv1.rezero_weight``,
``conv1.res_connection.weight``, ``pool1.remap_matrix`` ...) and forward semantics, so that
checkpoints and the JSON configs of the reference load unchanged.  Pinned by fixture G5.
"""
from typing import Dict

import numpy as np
import torch
from torch.nn import BatchNorm1d, Identity, Linear
from torch.nn import functional as F

from dsw_amd import functional as dsw_functional
from modules.layers import GeneralConvBlock, PoolUnpoolBlock
from modules.models import UNet
from modules.utils_models import (
    check_conv_type,
    check_pool_method,
    check_sampling,
    check_skip_connection,
    pygsp_graph_coarsening,
)

_CANONICAL_DIMS = ("sample", "node", "time", "feature")


class ConvBlock(GeneralConvBlock):
    """conv -> [batch-norm] -> activation -> [batch-norm] on ``(sample, node, channel)`` tensors.

    With ``batch_norm`` the convolution carries no bias; ``batch_norm_before_activation`` selects on
    which side of the activation the normalisation sits.
    """

    def __init__(
        self,
        in_channels,
        out_channels,
        laplacian,
        kernel_size=3,
        conv_type="graph",
        bias=True,
        batch_norm=False,
        batch_norm_before_activation=False,
        activation=True,
        activation_fun="relu",
        periodic_padding=True,
        lonlat_ratio=2,
    ):
        super().__init__()
        self.conv = GeneralConvBlock.getConvLayer(
            in_channels=in_channels,
            out_channels=out_channels,
            kernel_size=kernel_size,
            laplacian=laplacian,
            conv_type=conv_type,
            bias=bias and not batch_norm,
            periodic_padding=periodic_padding,
            lonlat_ratio=lonlat_ratio,
        )
        if batch_norm:
            self.bn = BatchNorm1d(out_channels)
        self.norm = batch_norm
        self.bn_before_act = batch_norm_before_activation
        self.act = activation
        self.act_fun = getattr(F, activation_fun)

    def _normalise(self, x):
        return self.bn(x.transpose(1, 2)).transpose(1, 2)  # BatchNorm1d wants (sample, channel, node)

    def forward(self, x):
        if self.act and self.act_fun is F.relu and not (self.norm and self.bn_before_act) \
                and getattr(self.conv, "supports_fused_activation", False):
            x = self.conv(x, activation="relu")             # conv + bias + relu in one epilogue; through __call__, so
                                                            # forward (pre-)hooks on the conv fire as on the plain path
        else:
            x = self.conv(x)
            if self.norm and self.bn_before_act:
                x = self._normalise(x)
            if self.act:
                x = self.act_fun(x)
        if self.norm and not self.bn_before_act:
            x = self._normalise(x)
        return x


class _NodeLinear(Linear):
    """``torch.nn.Linear`` (same parameters / state_dict entries, same initial values) whose per-node map runs on the
    path's own MFMA GEMM kernels (the K = 1 channel mix) instead of a rocBLAS call.

    ``weight`` keeps ``Linear``'s shape ``[out, in]`` but is laid out column-major (strides ``(1, out)``): the kernels read
    ``weight.t()`` = a dense ``[in, out]`` matrix, and the gradient they produce already has the parameter's strides, so
    neither direction launches a transpose copy (two 5 us launches per residual block and step otherwise)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._lay_out_weight()

    def _lay_out_weight(self):
        # only a REGISTERED parameter is re-laid, and only in place of its data: SWAG (reference modules/swag.py:33-48)
        # pops every entry of `_parameters` before `.to(device)` and later assigns plain tensors by attribute - there is
        # nothing to lay out then, nothing may be re-registered, and `forward` reads whatever layout it is given.
        # Swapping `.data` (not the Parameter object) keeps an optimizer's reference to the parameter valid.
        w = self._parameters.get("weight")
        if w is not None and w.dim() == 2 and w.stride() != (1, w.shape[0]):
            w.data = w.data.t().contiguous().t()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)     # .to() / .cuda() keep dense strides; anything that did not is re-laid
        self._lay_out_weight()
        return out

    def forward(self, x):
        # `weight.t()` is dense [in, out] for the column-major parameter; a row-major tensor (SWAG sample, user
        # assignment) is made contiguous inside dense_mix - same values either way
        # (the parameter's gradient buffer, if a GradBucket in direct mode attached one: memory order [in][out], which is
        # the layout of the gradient the kernels produce for `weight.t()`)
        acc = dsw_functional.grad_accumulators(self.weight, self.bias) if self.weight.stride() == (1, self.weight.shape[0]) else None
        return dsw_functional.dense_mix(x, self.weight.t(), self.bias, acc=acc)   # raises on CPU tensors like every layer here


class ResBlock(torch.nn.Module):
    """Stack of ConvBlocks (``convblock1..n``, no activation on the last) with a ReZero-scaled
    residual branch: ``out = rezero_weight * convs(x) + res_connection(x)``."""

    def __init__(self, in_channels, out_channels, laplacian, convblock_kwargs, **kwargs):
        super().__init__()
        self.rezero = True
        if not isinstance(out_channels, (int, tuple, list)):
            raise TypeError("'output_channels' must be int or list/tuple of int.")
        widths = list(out_channels) if isinstance(out_channels, (tuple, list)) else [out_channels]
        self.conv_names_list = []
        width_in = in_channels
        for pos, width_out in enumerate(widths, start=1):
            opts = dict(convblock_kwargs)
            if pos == len(widths):
                opts["activation"] = False  # a relu here would only allow positive increments
            name = f"convblock{pos}"
            setattr(self, name, ConvBlock(width_in, width_out, laplacian=laplacian, **opts))
            self.conv_names_list.append(name)
            width_in = width_out
        self.res_connection = Identity() if in_channels == widths[-1] else _NodeLinear(in_channels, widths[-1])
        if self.rezero:
            self.rezero_weight = torch.nn.Parameter(torch.zeros(1), requires_grad=True)
        if convblock_kwargs["batch_norm"]:  # start each block as an identity mapping
            last = getattr(self, self.conv_names_list[-1])
            torch.nn.init.constant_(last.bn.weight, 0)
            torch.nn.init.constant_(last.bn.bias, 0)

    # Block-level fusion is IMPLEMENTED and OFF: measured same-box on the U-Net workload
    # (nside 32, B 8: 3.93 ms plain) every variant was slower - backward only (ReZero scale inside the dgrad kernels,
    # dscale = <W, dW_raw> + <b, db_raw>) +0.4 %, + forward epilogue +0.9 %, + fork +0.8 %, everything +1.8 %: the
    # passes they remove run out of the Infinity Cache at these sizes (50 MB tensors), while a residual operand in the
    # MFMA epilogue costs the GEMM 40-60 registers (its second wave per SIMD).  Worth switching on where the tensors
    # exceed the cache (nside 64 models); fp32 only.
    fuse_tail = False           # True: last convolution takes ReZero scale + residual add (see _fusable_tail)
    fuse_tail_epilogue = True   # forward: scale + residual in the GEMM epilogue (False: one pass behind the convolution)
    fuse_fork = True            # the residual map collects the convolution stack's input gradient in its dgrad GEMM

    def _fusable_tail(self, x):
        """The last convolution can take the block's tail - ReZero scale and residual add - into its GEMM epilogue: a plain
        ConvCheb (no norm, no activation behind it), and nobody listening on the modules this shortcut does not call."""
        from modules.layers import ConvCheb, conv_cheb

        if not (self.fuse_tail and self.rezero and x.dim() == 3):
            return False
        last = getattr(self, self.conv_names_list[-1])
        conv = last.conv
        if last.norm or last.act or not isinstance(conv, ConvCheb) or conv._conv is not conv_cheb:
            return False
        if conv.kernel_size * conv.out_channels <= 16:
            return False     # a few output columns (the model's 64 -> 2 layer): the vector-ALU kernels of dsw_narrow.hip
                             # beat a matrix-core tile with 3-9 % live columns, and they have no epilogue operands
        watched = [last, conv, self.res_connection]
        if any(m._forward_hooks or m._forward_pre_hooks for m in watched):
            return False
        if not isinstance(self.res_connection, (_NodeLinear, Identity)):
            return False
        return x.dtype == torch.float32

    def forward(self, x, out=None):
        """``out`` (extension): a preallocated channel slice to write the block's output into (the encoder blocks of the
        U-Net write straight into the decoder's concatenation buffer, ``dsw_functional.skip_slot``)."""
        if self._fusable_tail(x):
            # out = rezero_weight * convs(x) + res_connection(x), the last two steps in the epilogue of the last
            # convolution's channel mix (no separate pass over the output, nothing of the unscaled convolution is kept);
            # the residual map hands the block input on to the convolutions and collects their gradient in its own
            # backward GEMM (no autograd `add` over the input's gradient)
            lin = self.res_connection
            if isinstance(lin, _NodeLinear):
                if self.fuse_fork and x.requires_grad:
                    y, r = dsw_functional.dense_mix_fork(x, lin.weight.t(), lin.bias)
                else:
                    y, r = x, dsw_functional.dense_mix(x, lin.weight.t(), lin.bias)
            else:
                y, r = x, x
            for name in self.conv_names_list[:-1]:
                y = getattr(self, name)(y)
            conv = getattr(self, self.conv_names_list[-1]).conv
            if y.shape[2] == conv.weight.shape[0] and r.shape[2] == conv.weight.shape[2]:
                epi = self.fuse_tail_epilogue
                if epi and out is not None and not dsw_functional.cheb_conv_res_takes_out(*conv.weight.shape[::2], conv.weight.shape[1]):
                    epi = False
                return dsw_functional.cheb_conv_res(dsw_functional.get_operator(conv.laplacian), y, conv.weight, conv.bias,
                                                    self.rezero_weight, r, out=out, epilogue=epi)
            return dsw_functional.rezero_residual(getattr(self, self.conv_names_list[-1])(y), r, self.rezero_weight, out=out)
        y = x
        for name in self.conv_names_list:
            y = getattr(self, name)(y)
        if self.rezero:   # rezero_weight * y + residual in one pass (the reference: in-place mul, then in-place add)
            return dsw_functional.rezero_residual(y, self.res_connection(x), self.rezero_weight, out=out)
        y += self.res_connection(x)
        return y if out is None else out.copy_(y)


# (attribute, input width, output widths, level) of the six residual blocks; "in" / "out" stand for the model's
# merged time-feature widths, `level` selects the U-Net level whose Laplacian the block convolves with
_UNET_BLOCKS = (
    ("conv1", "in", (64, 128), 0),
    ("conv2", 128, (192, 256), 1),
    ("conv3", 256, (512, 256), 2),
    ("uconv2", 512, (256, 128), 1),      # input = unpooled level 3 stacked with the level-2 skip
    ("uconv1", 256, (128, 64), 0),       # input = unpooled level 2 stacked with the level-1 skip
    ("uconv1_final", 64, "out", 0),
)


class UNetSpherical(UNet, torch.nn.Module):
    """Three-level spherical U-Net of residual ConvCheb blocks.

    The options (and their defaults) are the keys of ``configs/UNetSpherical/*/*.json`` in the
    reference; see its docstring for their meaning.  ``tensor_info`` carries the dimension order and
    the feature / time / node counts of inputs and outputs.
    """

    def __init__(
        self, tensor_info: Dict, sampling: str, sampling_kwargs: Dict,
        kernel_size_conv: int = 3, conv_type: str = "graph", graph_type: str = "knn", knn: int = 20,
        periodic_padding: bool = True,
        bias: bool = True, batch_norm: bool = False, batch_norm_before_activation: bool = False,
        activation: bool = True, activation_fun: str = "relu",
        pool_method: str = "max", kernel_size_pooling: int = 4,
        skip_connection: str = "stack", increment_learning: bool = False,
    ):
        super().__init__()
        self.dim_names = tensor_info["dim_order"]["dynamic"]
        for side in ("input", "output"):
            setattr(self, side + "_n_feature", tensor_info[side + "_n_feature"])
            setattr(self, side + "_n_time", tensor_info[side + "_n_time"])
            setattr(self, side + "_n_node", tensor_info[side + "_shape_info"]["dynamic"]["node"])
        # ConvCheb mixes the merged (time, feature) axis
        self.input_channels = self.input_n_time * self.input_n_feature
        self.output_channels = self.output_n_time * self.output_n_feature
        self.increment_learning = increment_learning

        sampling = check_sampling(sampling)
        conv_type = check_conv_type(conv_type, sampling)
        pool_method = check_pool_method(pool_method)
        check_skip_connection(skip_connection)
        lonlat_ratio = None
        if sampling == "equiangular":
            lonlat_ratio = sampling_kwargs["nlon"] / sampling_kwargs["nlat"]
        block_opts = dict(
            kernel_size=kernel_size_conv, conv_type=conv_type, bias=bias, batch_norm=batch_norm,
            batch_norm_before_activation=batch_norm_before_activation, activation=activation,
            activation_fun=activation_fun, periodic_padding=periodic_padding, lonlat_ratio=lonlat_ratio,
        )

        # one graph per level, each `step` times coarser (linear) than the one above
        n_levels, step = 3, int(np.sqrt(kernel_size_pooling))
        sampling_kwargs["k"] = knn
        per_level = [sampling_kwargs]
        while len(per_level) < n_levels:
            per_level.append(pygsp_graph_coarsening(sampling, per_level[-1], step))
        self.init_graph_and_laplacians(sampling_list=[sampling] * n_levels, sampling_kwargs_list=per_level,
                                       graph_type=graph_type, conv_type=conv_type)

        if pool_method in ("interp", "maxval", "maxarea", "learn"):
            assert conv_type == "graph"
            for lvl in (1, 2):
                pool, unpool = PoolUnpoolBlock.getGeneralPoolUnpoolLayer(
                    src_graph=self.graphs[lvl - 1], dst_graph=self.graphs[lvl], pool_method=pool_method)
                if hasattr(pool, "index_format"):
                    pool.index_format = "compact"     # this model only passes the index on to the unpooling (decode())
                setattr(self, f"pool{lvl}", pool)
                setattr(self, f"unpool{lvl}", unpool)
        elif pool_method in ("max", "avg"):
            assert sampling in ["healpix", "equiangular"]
            for lvl in (1, 2):
                pool, unpool = PoolUnpoolBlock.getPoolUnpoolLayer(
                    sampling=sampling, pool_method=pool_method, kernel_size=kernel_size_pooling,
                    lonlat_ratio=lonlat_ratio)
                setattr(self, f"pool{lvl}", pool)
                setattr(self, f"unpool{lvl}", unpool)
        elif pool_method is not None:
            raise ValueError("Not valid pooling method provided.")

        widths = {"in": self.input_channels, "out": self.output_channels}
        for name, w_in, w_out, level in _UNET_BLOCKS:
            block = ResBlock(widths.get(w_in, w_in), widths.get(w_out, w_out) if isinstance(w_out, str) else w_out,
                             laplacian=self.laplacians[level], convblock_kwargs=block_opts)
            setattr(self, name, block)
        if self.increment_learning:
            self.res_increment = torch.nn.Parameter(torch.zeros(1), requires_grad=True)

    def encode(self, x):
        """``x`` in ``self.dim_names`` order -> (x_enc3, x_enc2, x_enc1, idx2, idx1, x_last_timestep)."""
        batch = x.shape[0]
        x_last_timestep = x[:, -1, :, -2:].unsqueeze(dim=1)
        order = [self.dim_names.index(d) for d in _CANONICAL_DIMS]
        x = x.permute(*order).reshape(batch, self.input_n_node, self.input_channels)
        # the skip tensors are born inside the decoder's concatenation buffers (their right-hand channel slice; the
        # unpooling fills the left one in `decode`): no `torch.cat` copy forward, no slice copies backward
        x_enc1 = self.conv1(x, out=self._skip_slot(x, self.unpool1, self.uconv1, self.conv1))
        x_enc1, (x_enc2_ini, idx1) = self._pool_and_skip(self.pool1, x_enc1)
        x_enc2 = self.conv2(x_enc2_ini, out=self._skip_slot(x_enc2_ini, self.unpool2, self.uconv2, self.conv2))
        x_enc2, (x_enc3_ini, idx2) = self._pool_and_skip(self.pool2, x_enc2)
        x_enc3 = self.conv3(x_enc3_ini)
        return x_enc3, x_enc2, x_enc1, idx2, idx1, x_last_timestep

    @staticmethod
    def _pool_and_skip(pool, x):
        """``(x for the skip connection, pool(x))``.  The tensor has two consumers; pooling layers with ``forward_fork``
        add the skip connection's gradient inside their own backward product instead of leaving it to an autograd
        ``add`` (same values; the sum is formed in the other order)."""
        fork = getattr(pool, "forward_fork", None)
        if fork is not None and x.requires_grad and torch.is_grad_enabled():
            return fork(x)
        return x, pool(x)

    concat_in_place = True   # False: skip tensors are ordinary tensors and the decoder calls torch.cat (A/B runs)

    @classmethod
    def _skip_slot(cls, x, unpool, decoder_block, encoder_block):
        """Where `encoder_block` should write: the right-hand slice of the buffer `decoder_block` will read (None: plain
        tensor + `torch.cat`, e.g. for unpooling layers that cannot write into a slice)."""
        if not (cls.concat_in_place and getattr(unpool, "supports_out", False)):
            return None
        width_in = getattr(decoder_block, decoder_block.conv_names_list[0]).conv.in_channels
        width_skip = getattr(encoder_block, encoder_block.conv_names_list[-1]).conv.out_channels
        if width_in <= width_skip:
            return None
        return dsw_functional.skip_slot(x, x.shape[1], width_in - width_skip, width_skip)

    @staticmethod
    def _unpool_concat(unpool, x, idx, skip):
        """`torch.cat((unpool(x, idx), skip), dim=2)` (my_models_graph.py:528-545), without the copy when `skip` lives
        in a concatenation buffer."""
        if getattr(unpool, "supports_out", False):
            width_left = x.shape[2]
            buf = dsw_functional.skip_buffer(skip, width_left)
            if buf is not None:
                left = unpool(x, idx, out=dsw_functional.left_slot(buf, width_left))
                return dsw_functional.concat_in_place(left, skip, buf)
        return torch.cat((unpool(x, idx), skip), dim=2)

    def decode(self, x_enc3, x_enc2, x_enc1, idx2, idx1, x_last_timestep):
        x = self.uconv2(self._unpool_concat(self.unpool2, x_enc3, idx2, x_enc2))
        x = self.uconv1(self._unpool_concat(self.unpool1, x, idx1, x_enc1))
        x = self.uconv1_final(x)
        batch = x.shape[0]
        x = x.reshape(batch, self.output_n_node, self.output_n_time, self.output_n_feature)
        order = [_CANONICAL_DIMS.index(d) for d in self.dim_names]
        x = x.permute(*order)
        if self.increment_learning:
            x *= self.res_increment
            x += x_last_timestep
        return x
