"""CPU oracle for the DANCE GNN message-passing hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``dance_amd/`` may import this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the CPU reference being timed — never as
the product path.

Every function restates, on the CPU (numpy / scipy / torch-CPU), the arithmetic the reference performs at the
cited file:line of OmicsML/dance (paths relative to the reference tree).

Pinning status (see DESIGN.md "Oracle"):
* ``oracle.layers``  (GNNLayer / GraphConvolution)  — PINNED: tests/golden/gcn_layers.npz holds outputs and
  gradients produced by the reference's own classes, AST-extracted from /root/reference and executed on
  torch-CPU by tests/golden/make_golden.py.
* ``oracle.matrix``  (pairwise_distance / normalize) — PINNED by the reference's known-answer tests
  (tests/utils/test_matrix.py:9-65, vectors copied into tests/golden/matrix_known_answers.json).
* ``oracle.graphs.cell_feature_graph / heteronet_edges / stagate_* / spagcn_xyz``, ``oracle.sage.*`` (AdaptiveSAGE
  message + mean, WeightedGraphConv), ``oracle.spagcn.*`` — PINNED: tests/golden/graph_builders.npz holds outputs of
  the reference's own methods (CellFeatureGraph.__call__, AdaptiveSAGE.message_func / forward,
  WeightedGraphConv.forward, HeteronetGraph.build_graph, SpaGCNGraph.__call__, StagateGraph.__call__, calculate_p,
  search_l, refine), lifted method by method by oracle.ref_extract.extract_method and executed on torch-CPU; where
  that code calls dgl, oracle.ref_extract.DGLStubGraph supplies DGL's storage semantics (edge ids in insertion
  order, mean over in-edges with 0 for isolated nodes) — those semantics are the only part not executed from the
  reference tree.  sklearn is the installed 1.7 (reference pins 1.3.2).
* ``oracle.graphs.knn_exact / fuzzy_simplicial_set / neighbor_graph`` (scanpy ``sc.pp.neighbors`` → pynndescent /
  umap-learn) — PARITY UNPINNED by reference output (scanpy, numba, umap-learn are not installable here): kNN is
  checked against scikit-learn's exact search, the fuzzy simplicial set is restated from umap-learn's published
  algorithm.
"""
