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
* ``oracle.graphs`` / ``oracle.sage`` (DGL / scanpy / sklearn call sites) — the reference cannot run here
  (dgl, scanpy, anndata, numba are not installed): PARITY UNPINNED by reference output; restated from the
  cited lines and from the published algorithms of dgl 1.1.3 / umap-learn / scikit-learn.
"""
