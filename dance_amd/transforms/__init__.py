from .base import BaseTransform
from .cell_feature import CellPCA, WeightedFeaturePCA
from .misc import Compose, SetConfig
from .normalize import ColumnSumNormalize, Log1P, NormalizeTotal, NormalizeTotalLog1P, Scale

__all__ = ["BaseTransform", "CellPCA", "WeightedFeaturePCA", "Compose", "SetConfig", "ColumnSumNormalize", "Log1P", "NormalizeTotal",
           "NormalizeTotalLog1P", "Scale"]
