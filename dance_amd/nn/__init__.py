from .gnn import AdaptiveSAGE

__all__ = ["AdaptiveSAGE"]
