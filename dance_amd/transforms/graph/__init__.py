from .cell_feature_graph import CellFeatureGraph, PCACellFeatureGraph
from .feature_feature_graph import FeatureFeatureGraph
from .heteronet_graph import HeteronetGraph
from .neighbor_graph import NeighborGraph
from .spatial_graph import SpaGCNGraph, SpaGCNGraph2D, StagateGraph

__all__ = ["CellFeatureGraph", "PCACellFeatureGraph", "FeatureFeatureGraph", "HeteronetGraph", "NeighborGraph", "SpaGCNGraph",
           "SpaGCNGraph2D", "StagateGraph"]
