from .base import BaseTransform
from .cell_feature import CellPCA, WeightedFeaturePCA
from .filter import (FilterCellsScanpy, FilterCellsScanpyOrder, FilterGenesCommon, FilterGenesPercentile, FilterGenesScanpyOrder,
                     FilterGenesTopK, FilterCellsType, FilterGenesMatch, FilterGenesScanpy, HighlyVariableGenesLogarithmizedByMeanAndDisp,
                     HighlyVariableGenesLogarithmizedByTopGenes, HighlyVariableGenesRawCount)
from .interface import AnnDataTransform
from .mask import CellwiseMaskData
from .misc import Compose, RemoveSplit, SaveRaw, SetConfig, UpdateRaw
from .normalize import ColumnSumNormalize, Log1P, NormalizeTotal, NormalizeTotalLog1P, Scale, UpdateSizeFactors

__all__ = ["BaseTransform", "CellPCA", "WeightedFeaturePCA", "Compose", "SaveRaw", "SetConfig", "UpdateRaw", "RemoveSplit", "CellwiseMaskData", "AnnDataTransform", "ColumnSumNormalize", "Log1P", "NormalizeTotal",
           "NormalizeTotalLog1P", "Scale", "FilterCellsScanpy", "FilterGenesScanpy", "HighlyVariableGenesLogarithmizedByMeanAndDisp",
           "HighlyVariableGenesLogarithmizedByTopGenes", "HighlyVariableGenesRawCount", "FilterCellsType", "FilterCellsScanpyOrder", "FilterGenesCommon", "FilterGenesPercentile", "FilterGenesScanpyOrder", "FilterGenesTopK", "FilterGenesMatch", "UpdateSizeFactors"]
