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


def test_workspace_size_queries_need_no_device():
    """The *_workspace_bytes entry points are pure host arithmetic: callable on a CPU-only box, consistent with the
    contracts in include/dance_hip.h."""
    from dance_amd import _lib
    lib = _lib.load()
    n, d, k = 1_000_000, 50, 15
    scan = lib.dh_knn_bruteforce_f32_workspace_bytes(n, d, n, k, 1)
    filt = lib.dh_knn_bruteforce_f32_workspace_bytes(n, d, n, k, 2)
    assert scan == n * 52 * 4                      # only the zero-padded copy (d = 50 -> 52 columns)
    assert filt > scan + 2 * n * 176 * 2           # + the two bf16x3 operand matrices (K3 = 176) and the survivor lists
    assert lib.dh_knn_bruteforce_f32_workspace_bytes(n, d, n, k, 0) == filt        # auto -> filter at this size
    assert lib.dh_knn_bruteforce_f32_workspace_bytes(1000, d, 1000, k, 0) == lib.dh_knn_bruteforce_f32_workspace_bytes(1000, d, 1000, k, 1)
    assert lib.dh_knn_bruteforce_f32_workspace_bytes(0, d, 0, k, 0) == 0
    assert lib.dh_relu_mask_bytes(n, 512) == n * 512 // 8 and lib.dh_relu_mask_bytes(n, 50) == 0
    # headline dW GEMM: 16 tiles x 64 K-slices of fp32 slabs; the forward GEMM needs none
    assert lib.dh_gemm_f32_workspace_bytes(2000, 512, n, 1, 0) == 64 * 2000 * 512 * 4
    assert lib.dh_gemm_f32_workspace_bytes(n, 512, 2000, 0, 0) == 0
    # bf16: the native NT form needs no workspace, K-strided operands are repacked
    assert lib.dh_gemm_bf16_workspace_bytes(n, 200, 400, 0, 1) == 0
    assert lib.dh_gemm_bf16_workspace_bytes(n, 400, 200, 0, 0) >= 400 * 200 * 2


def test_bench_and_smoke_fail_loudly_without_a_gpu():
    """bench.py and __graft_entry__.smoke() are GPU programs: on a box without a HIP device they raise instead of timing
    (or checking) some fallback."""
    import subprocess
    import sys

    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "no HIP device" in r.stderr and "metric" not in r.stdout
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "metric" not in r.stdout
    # a multi-rank request without the launcher is refused before anything is initialised
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_library_never_records_memset_nodes():
    """No hipMemsetAsync / hipMemcpyAsync in the kernels' translation units (comm.hip's host-side staging apart): under a hipGraph
    capture they become memset / memcpy NODES, which ROCm 7.2 replayed with another operation's extent and pattern (round 5,
    profiles/r05_replay_fault.md).  dh::zero_async (a kernel) is the library's only way to clear memory."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dance_amd", "csrc")
    bad = []
    for path in sorted(glob.glob(os.path.join(root, "*.hip"))):
        if os.path.basename(path) == "comm.hip":
            continue
        for i, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if re.search(r"\bhipMemsetAsync\s*\(|\bhipMemset2DAsync\s*\(|\bhipMemcpyAsync\s*\(|\bhipMemset\s*\(", code):
                bad.append(f"{os.path.basename(path)}:{i}")
    assert not bad, bad


def test_small_gemm_is_opt_in_and_scoped():
    """dh_gemm_f32_small sums in another order than dh_gemm_f32, so ``kernels.gemm`` takes it only inside
    ``kernels.mini_batch_products()`` (entered by GraphSC.fit / ScDeepSort.fit): the full-batch layers keep row results that do not
    depend on the row count of the call.  The context nests and always restores."""
    from dance_amd import kernels
    depth = kernels._mini_batch_depth.get  # a context variable: per thread / task
    assert depth() == 0
    with kernels.mini_batch_products():
        assert depth() == 1
        try:
            with kernels.mini_batch_products():
                assert depth() == 2
                raise RuntimeError("leave by exception")
        except RuntimeError:
            pass
        assert depth() == 1
        import threading
        seen = []
        t = threading.Thread(target=lambda: seen.append(depth()))  # another thread's gemm() calls are not switched by this fit
        t.start()
        t.join()
        assert seen == [0]
    assert depth() == 0
    import inspect
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    assert "mini_batch_products" in inspect.getsource(GraphSC.fit) and "mini_batch_products" in inspect.getsource(ScDeepSort.fit)
    assert "mini_batch_products" not in inspect.getsource(ScDSC.fit)   # full batch, shardable: stays on dh_gemm_f32
