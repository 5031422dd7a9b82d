from .base import BaseTransform
from .cell_feature import CellPCA, WeightedFeaturePCA
from .misc import Compose, SetConfig

__all__ = ["BaseTransform", "CellPCA", "WeightedFeaturePCA", "Compose", "SetConfig"]
