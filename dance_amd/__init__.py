"""dance_amd — MI355X-native GNN message-passing hot path behind DANCE's transform / method API.

Scope: the GCN / GraphSAGE layers of dance.modules / dance.models and the graph builders of
dance.transforms.graph, computed by hand-written HIP kernels (libdancehip.so, C ABI in include/dance_hip.h).
"""
__version__ = "0.1.0"
