"""CPU restatement of the reference's UNetSpherical forward pass (interp pooling, ReZero residual blocks).
TEST INFRASTRUCTURE ONLY - the checker of the UNet-level parity tests and the ``cpu_baseline`` of
``bench.py --workload unet``; never imported by the product.

A *functional* restatement: it takes a state_dict (the 53 entries of the reference model: 38 parameters, 11
``laplacian`` and 4 ``remap_matrix`` sparse buffers) and evaluates, with the torch op sequence of the reference,

* ``ConvBlock.forward``   /root/reference/modules/my_models_graph.py:104-118   (conv -> relu; batch_norm = False)
* ``ResBlock.forward``    :205-216   (conv stack, ``*= rezero_weight``, ``+= res_connection(x)``)
* ``UNetSpherical.encode``:492-525,  ``decode`` :528-564,  ``UNet.forward`` modules/models.py:112-116

on top of ``cheb_oracle.conv_cheb_layer_torch`` (layers.py:113-180,365-376) and ``cheb_oracle.remap_torch``
(layers.py:956-964).  Autograd supplies the backward, as in the reference.

Pinned: ``tests/test_oracle_golden.py::test_unet_oracle_matches_reference_fixture`` checks output, loss and the
gradient fingerprints of all 38 parameter tensors against fixture G5, which was produced by the imported reference
model itself (tests/golden/make_golden.py).
"""
from __future__ import annotations

import torch
from torch.nn import functional as F

from . import cheb_oracle as orc

_CANONICAL = ("sample", "node", "time", "feature")
# residual blocks of the U-Net in evaluation order: name -> number of ConvBlocks (my_models_graph.py:438-485)
_BLOCKS = {"conv1": 2, "conv2": 2, "conv3": 2, "uconv2": 2, "uconv1": 2, "uconv1_final": 1}


def _conv_block(sd, prefix, x, activation):
    """ConvBlock.forward (my_models_graph.py:104-118) without batch norm."""
    x = orc.conv_cheb_layer_torch(sd[prefix + "conv.laplacian"], x, sd[prefix + "conv.weight"],
                                  sd.get(prefix + "conv.bias"))
    return F.relu(x) if activation else x


def _res_block(sd, name, x):
    """ResBlock.forward (my_models_graph.py:205-216): last ConvBlock has no activation (:163-164)."""
    n = _BLOCKS[name]
    out = x
    for i in range(1, n + 1):
        out = _conv_block(sd, f"{name}.convblock{i}.", out, activation=(i < n))
    out = out * sd[name + ".rezero_weight"]                       # x_out *= self.rezero_weight
    if name + ".res_connection.weight" in sd:                     # Linear(in, out) unless in == out (Identity)
        res = F.linear(x, sd[name + ".res_connection.weight"], sd[name + ".res_connection.bias"])
    else:
        res = x
    return out + res                                              # x_out += self.res_connection(x)


def unet_forward(sd, x, dim_names=("sample", "time", "node", "feature"), output_n_time=1, output_n_feature=2,
                 increment_learning=False):
    """``UNet.forward`` = decode(*encode(x)) (models.py:112-116) for ``x`` in ``dim_names`` order."""
    batch = x.shape[0]
    x_last = x[:, -1, :, -2:].unsqueeze(dim=1)                    # my_models_graph.py:500
    x = x.permute(*[dim_names.index(d) for d in _CANONICAL])
    n_node = x.shape[1]
    x = x.reshape(batch, n_node, -1)                              # :509-511
    e1 = _res_block(sd, "conv1", x)
    e2 = _res_block(sd, "conv2", orc.remap_torch(sd["pool1.remap_matrix"], e1))
    e3 = _res_block(sd, "conv3", orc.remap_torch(sd["pool2.remap_matrix"], e2))
    y = orc.remap_torch(sd["unpool2.remap_matrix"], e3)           # decode, :532-545
    y = _res_block(sd, "uconv2", torch.cat((y, e2), dim=2))
    y = orc.remap_torch(sd["unpool1.remap_matrix"], y)
    y = _res_block(sd, "uconv1", torch.cat((y, e1), dim=2))
    y = _res_block(sd, "uconv1_final", y)
    y = y.reshape(batch, n_node, output_n_time, output_n_feature)
    y = y.permute(*[_CANONICAL.index(d) for d in dim_names])
    if increment_learning:                                        # :556-561
        y = y * sd["res_increment"] + x_last
    return y


def leaf_state(model_state_dict, device="cpu"):
    """Detached fp32 CPU copy of a model's state_dict with the parameters turned into autograd leaves."""
    sd = {}
    for k, v in model_state_dict.items():
        v = v.detach().to(device)
        if v.is_sparse:
            sd[k] = v.float().coalesce()
        else:
            sd[k] = v.float().clone().requires_grad_(True)
    return sd


def unet_fwd_bwd(sd, x, target):
    """One fwd + bwd with the MSE loss of the parity fixtures; returns (y, loss, {name: grad})."""
    for v in sd.values():
        if not v.is_sparse and v.grad is not None:
            v.grad = None
    y = unet_forward(sd, x)
    loss = ((y - target) ** 2).mean()
    loss.backward()
    return y.detach(), loss.item(), {k: v.grad for k, v in sd.items() if not v.is_sparse and v.grad is not None}
