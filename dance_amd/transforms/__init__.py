from .base import BaseTransform
from .cell_feature import CellPCA, WeightedFeaturePCA
from .filter import (FilterCellsScanpy, FilterCellsScanpyOrder, FilterGenesCommon, FilterGenesPercentile, FilterGenesScanpyOrder,
                     FilterGenesTopK, FilterCellsType, FilterGenesMatch, FilterGenesScanpy, HighlyVariableGenesLogarithmizedByMeanAndDisp,
                     HighlyVariableGenesLogarithmizedByTopGenes, HighlyVariableGenesRawCount)
from .misc import Compose, SaveRaw, SetConfig
from .normalize import ColumnSumNormalize, Log1P, NormalizeTotal, NormalizeTotalLog1P, Scale, UpdateSizeFactors

__all__ = ["BaseTransform", "CellPCA", "WeightedFeaturePCA", "Compose", "SaveRaw", "SetConfig", "ColumnSumNormalize", "Log1P", "NormalizeTotal",
           "NormalizeTotalLog1P", "Scale", "FilterCellsScanpy", "FilterGenesScanpy", "HighlyVariableGenesLogarithmizedByMeanAndDisp",
           "HighlyVariableGenesLogarithmizedByTopGenes", "HighlyVariableGenesRawCount", "FilterCellsType", "FilterCellsScanpyOrder", "FilterGenesCommon", "FilterGenesPercentile", "FilterGenesScanpyOrder", "FilterGenesTopK", "FilterGenesMatch", "UpdateSizeFactors"]
