"""Lift torch-only classes/functions verbatim out of the reference tree by AST extraction.

Works ONLY where /root/reference exists (the build container).  Used by tests/golden/make_golden.py to produce
the committed golden vectors and by tests that pin the oracle restatement against the reference's own code;
never imported by the product and never needed on the GPU box.
"""
import ast
import math
import os

REFERENCE_ROOT = os.environ.get("DANCE_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "dance"))


def extract(rel_path: str, name: str, extra_ns=None):
    """exec the top-level class/function ``name`` of ``rel_path`` in a minimal torch namespace."""
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn.parameter import Parameter
    path = os.path.join(REFERENCE_ROOT, rel_path)
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name == name:
            seg = ast.get_source_segment(src, node)
            # decorators (e.g. numba.njit) are not part of get_source_segment for FunctionDef bodies we
            # want to run as plain python; ast gives the def line onward.
            ns = {"torch": torch, "nn": nn, "F": F, "Parameter": Parameter, "np": np, "math": math}
            ns.update(extra_ns or {})
            exec(compile(seg, f"<reference:{rel_path}:{node.lineno}>", "exec"), ns)
            return ns[name]
    raise KeyError(f"{name} not found in {path}")
