"""Model base classes: graph + Laplacian set-up shared by the spherical architectures
(counterpart of ``/root/reference/modules/models.py``)."""
from abc import ABC, abstractmethod
from typing import Dict, List

from modules.layers import compute_cotan_laplacian, prepare_torch_laplacian
from modules.utils_models import get_pygsp_graph_fun


class DeepSphere(ABC):
    """Owns ``self.graphs`` (one per resolution level) and ``self.laplacians`` (prepared operators)."""

    @abstractmethod
    def forward(self, x):
        pass

    @staticmethod
    def build_pygsp_graphs(sampling_list: List[str], sampling_kwargs_list: List[Dict]):
        if not isinstance(sampling_list, list):
            raise TypeError("sampling_list must be a list specifying the sampling of each graph.")
        if len(sampling_list) != len(sampling_kwargs_list):
            raise ValueError("sampling_list must have same length of sampling_kwargs_list.")
        return [
            get_pygsp_graph_fun(name)(**kwargs, lap_type="normalized")
            for name, kwargs in zip(sampling_list, sampling_kwargs_list)
        ]

    @staticmethod
    def get_laplacian_kernels(graphs, graph_type: str = "knn"):
        assert graph_type in ["knn", "voronoi"]
        raw = [g.L if graph_type == "knn" else compute_cotan_laplacian(g, return_mass=False) for g in graphs]
        return [prepare_torch_laplacian(lap) for lap in raw]

    def init_graph_and_laplacians(self, sampling_list, sampling_kwargs_list, graph_type="knn", conv_type="graph"):
        self.graphs = DeepSphere.build_pygsp_graphs(sampling_list, sampling_kwargs_list)
        if conv_type == "graph":
            self.laplacians = DeepSphere.get_laplacian_kernels(self.graphs, graph_type=graph_type)
        elif conv_type == "image":
            self.laplacians = [None] * len(sampling_list)


class UNet(DeepSphere):
    @abstractmethod
    def encode(self, *args, **kwargs):
        pass

    @abstractmethod
    def decode(self, *args, **kwargs):
        pass

    def forward(self, x):
        return self.decode(*self.encode(x))


class ConvNet(DeepSphere):
    @abstractmethod
    def forward(self, x):
        pass


class DownscalingNet(DeepSphere):
    @abstractmethod
    def decode(self, *args, **kwargs):
        pass

    def forward(self, x):
        return self.decode(x)
