"""CPU: the C-ABI library loads and exports exactly the entry points include/dance_hip.h declares (no compute
calls here — there is no GPU), argument validation returns error codes, and the product refuses to run
without a device instead of falling back."""
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "dance_hip.h")).read()
    return sorted(set(re.findall(r"DH_API\s+[\w\s\*]+?\b(dh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from dance_amd import _lib
    lib = _lib.load()
    declared = _header_symbols()
    assert len(declared) >= 11
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (dh_[a-z0-9_]+)", out)))
    assert exported == declared, (set(exported) ^ set(declared))
    assert _lib.exported_symbols() == declared  # the ctypes binding covers the whole header
    assert lib.dh_version() >= 100
    assert lib.dh_last_error_string() is not None


def test_each_declaration_cites_a_reference_site():
    text = open(os.path.join(ROOT, "include", "dance_hip.h")).read()
    assert text.count("dance/") >= 3 and "scdsc.py:498" in text and "spagcn.py:359" in text


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_no_cpu_fallback_without_device():
    from dance_amd import _lib, kernels
    assert _lib.load().dh_device_count() == 0
    with pytest.raises(_lib.DanceHipError):
        kernels.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    with pytest.raises(_lib.DanceHipError):
        _lib.require_device()


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "dance_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(d, f)).read(), re.M):
                bad.append(os.path.join(d, f))
    assert not bad, bad
