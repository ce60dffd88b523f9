"""Abstract bases of the spherical architectures: they own the per-level graphs and the prepared
(rescaled, sparse-COO) Laplacians that the ConvCheb layers register as buffers.

API-compatible counterpart of the reference's ``modules/models.py`` (class and method names,
arguments, attributes ``graphs`` / ``laplacians``)."""
from abc import ABC, abstractmethod

from modules import layers as _layers
from modules.utils_models import get_pygsp_graph_fun


def _make_graph(sampling, kwargs):
    return get_pygsp_graph_fun(sampling)(lap_type="normalized", **kwargs)


class DeepSphere(ABC):
    @abstractmethod
    def forward(self, x):
        ...

    @staticmethod
    def build_pygsp_graphs(sampling_list, sampling_kwargs_list):
        """One graph per resolution level."""
        if not isinstance(sampling_list, list):
            raise TypeError("sampling_list must be a list specifying the sampling of each graph.")
        if len(sampling_kwargs_list) != len(sampling_list):
            raise ValueError("sampling_list must have same length of sampling_kwargs_list.")
        return [_make_graph(s, kw) for s, kw in zip(sampling_list, sampling_kwargs_list)]

    @staticmethod
    def get_laplacian_kernels(graphs, graph_type="knn"):
        """Prepared Laplacian of every graph ('knn': the graph's own L; 'voronoi': cotan Laplacian)."""
        assert graph_type in ["knn", "voronoi"]
        out = []
        for graph in graphs:
            lap = graph.L if graph_type == "knn" else _layers.compute_cotan_laplacian(graph, return_mass=False)
            out.append(_layers.prepare_torch_laplacian(lap))
        return out

    def init_graph_and_laplacians(self, sampling_list, sampling_kwargs_list, graph_type="knn", conv_type="graph"):
        self.graphs = DeepSphere.build_pygsp_graphs(sampling_list, sampling_kwargs_list)
        if conv_type == "image":
            self.laplacians = [None for _ in sampling_list]
        elif conv_type == "graph":
            self.laplacians = DeepSphere.get_laplacian_kernels(self.graphs, graph_type=graph_type)


class UNet(DeepSphere):
    """encode -> decode with skip connections handed over as a tuple."""

    @abstractmethod
    def encode(self, *args, **kwargs):
        ...

    @abstractmethod
    def decode(self, *args, **kwargs):
        ...

    def forward(self, x):
        encoded = self.encode(x)
        return self.decode(*encoded)


class ConvNet(DeepSphere):
    @abstractmethod
    def forward(self, x):
        ...


class DownscalingNet(DeepSphere):
    @abstractmethod
    def decode(self, *args, **kwargs):
        ...

    def forward(self, x):
        return self.decode(x)
