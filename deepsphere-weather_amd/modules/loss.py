"""Loss of the training harness: counterpart of the hot-path subset of ``/root/reference/modules/loss.py``.

``reshape_tensors_4_loss`` (``:31-54``) and ``WeightedMSELoss`` (``:118-156``) keep the reference's names, signatures,
reductions and error messages.  ``AreaWeights`` (``:60-68``) needs CDO in the reference; here the cell areas come from
``dsw_amd.sphere`` (HEALPix pixels are equal-area by construction; other samplings use their Voronoi-cell areas).
Plotting helpers are out of scope.
"""
import numpy as np
import torch
from torch import nn


def reshape_tensors_4_loss(Y_pred, Y_obs, dim_info_dynamic):
    """``(data_points, node, feature)`` views of prediction and observation: every dimension other than ``node`` and
    ``feature`` is flattened, in the tensors' own dimension order (``dim_info_dynamic``: name -> axis)."""
    names = [k for k, _ in sorted(dim_info_dynamic.items(), key=lambda item: item[1])]
    lead = [i for i, n in enumerate(names) if n not in ("node", "feature")]
    order = lead + [names.index("node"), names.index("feature")]

    def flat(t):
        t = t.permute(*order)
        return t.reshape(-1, t.shape[-2], t.shape[-1])

    return flat(Y_pred), flat(Y_obs)


def AreaWeights(graph):
    """Per-node area fractions (sum to 1) as a float32 tensor."""
    from dsw_amd import sphere

    area = sphere.cell_areas(graph)
    return torch.from_numpy((area / np.sum(area)).astype(np.float32))


class WeightedMSELoss(nn.MSELoss):
    """Squared error weighted per node.  ``pred`` / ``label``: ``(data_points, node, feature)``.

    ``reduction='mean'``: ``sum(w * err^2) / sum(w) / n_data_points / n_feature``; ``'sum'``: ``sum(w * err^2) * n_node``;
    ``'none'``: the weighted element-wise errors."""

    def __init__(self, reduction="mean", weights=None):
        super().__init__(reduction="none")
        if not isinstance(reduction, str) or reduction not in ("mean", "sum", "none"):
            raise ValueError("{} is not a valid value for reduction".format(reduction))
        self.weighted_mse_reduction = reduction
        if weights is not None:
            self.check_weights(weights)
        self.weights = weights

    def _node_weights(self, like, n_nodes):
        """The node weights on ``like``'s device in THEIR OWN dtype (fp32 area fractions stay fp32 under bf16 / fp16
        predictions, as in the reference: ``mse * weights`` then promotes, and the reduction and the returned loss are
        fp32).  ``self.weights`` is never overwritten; the device copy is cached beside it."""
        w = self.weights
        if w is None:
            return torch.ones(n_nodes, dtype=like.dtype, device=like.device)
        if len(w) != n_nodes:
            raise ValueError(
                "The number of weights does not match the the number of pixels. {} != {}".format(len(w), n_nodes)
            )
        if w.device == like.device:
            return w
        cached = getattr(self, "_weights_on_device", None)
        if cached is None or cached[0] is not w or cached[1].device != like.device:
            cached = self._weights_on_device = (w, w.to(device=like.device))   # moved once: graph captures see a resident tensor
        return cached[1]

    def forward(self, pred, label):
        err2 = super().forward(pred, label)                  # element-wise (reduction="none" in the base class)
        n_points, n_nodes, n_feat = err2.shape
        w = self._node_weights(err2, n_nodes)
        weighted = err2 * w.view(1, n_nodes, 1)
        how = self.weighted_mse_reduction
        if how == "none":
            return weighted
        total = weighted.sum()
        return total * n_nodes if how == "sum" else total / w.sum() / n_points / n_feat

    def check_weights(self, weights):
        if not isinstance(weights, torch.Tensor):
            raise TypeError("Weights type is not a torch.Tensor. Got {}".format(type(weights)))
        if len(weights.shape) != 1:
            raise ValueError("Weights is a 1D vector. Got {}".format(weights.shape))
