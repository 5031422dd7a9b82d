"""CPU: static guard on the built library — the register / scratch figures of every kernel, read from the gfx950 code objects
(scripts/kernel_resources.py).  A kernel that starts spilling vector registers to scratch is a performance regression the GPU tests
would not notice; the known cases are listed with their reason."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

LIB = os.path.join(ROOT, "dance_amd", "libdancehip.so")

# kernel name fragment -> why its spill is tolerated for now
KNOWN_VGPR_SPILLS = {
    "gemm_f32_kernel<Cfg<2, 4, 4, 2>, true, true,": "the transposed-transposed GEMM variant: no caller on the model paths",
    "sage_bcm_kernel<false, false, 4,": "fp32 features, 13 - 16 column tiles: 8 dwords of loop-invariant addresses spilled ONCE before the loop "
                                        "(the allocation is dictated by the 4-tile mover path, which holds 8 feature pieces)",
    "sage_bcm_kernel<false, true, 4,": "same kernel, bf16 output",
    "zinb_heads_fused_kernel<": "asked for four waves per SIMD (__launch_bounds__(256, 4)): 128 registers with 18 dwords spilled in the rare "
                                "gamma-function path of the float64 count terms, measured FASTER than 151 registers at three waves "
                                "(12.3 against 12.9 ms at 1M x 2000; zinb.hip)",
}


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_no_unexpected_scratch_spills():
    import kernel_resources as kr
    if not os.path.exists(kr.READELF):
        pytest.skip("llvm-readelf not available")
    rows = []
    for elf in kr.code_objects(open(LIB, "rb").read()):
        rows.extend(kr.kernels_of(elf))
    names = kr.demangle([r["name"] for r in rows])
    own = [(n, r) for n, r in zip(names, rows) if not n.startswith(("rocprim::", "hipcub::"))]
    assert len(own) > 200                                   # the table really was read
    spilled = [n for n, r in own if r["vspill"] > 0]
    unexpected = [n for n in spilled if not any(k in n for k in KNOWN_VGPR_SPILLS)]
    assert not unexpected, f"kernels spilling VGPRs to scratch: {unexpected}"
    for n, r in own:  # the tolerated sage_bcm spill stays what it is: a handful of dwords (36 bytes of scratch), two waves per SIMD
        if n.startswith("sage_bcm_kernel<"):  # (round 6: 28 spill / reload instructions in the bf16-output variant; the build measured faster, sage_bcm.hip)
            assert r["vspill"] <= 32 and r["scratch"] <= 64 and r["vgpr"] <= 256, (n, r)
    # the headline kernels keep their occupancy: the fp32 GEMM at most 256 registers (2 waves per SIMD), the SpMM at most 64 (8 waves)
    for n, r in own:
        if n.startswith("gemm_f32_kernel<Cfg<2, 4, 4, 2>"):
            assert r["vgpr"] <= 256, (n, r["vgpr"])
        if n.startswith("spmm_csr_kernel<") and n.rstrip(">").endswith(", 4"):
            assert r["vgpr"] <= 64, (n, r["vgpr"])
        if "knn_fold_filter_kernel<" in n:  # 8 waves per workgroup = 2 per SIMD, no scratch
            assert r["vgpr"] <= 256 and r["scratch"] == 0, (n, r)
        if n.startswith("zinb_heads_fused_kernel<"):  # the tolerated spill stays small and buys the fourth wave
            assert r["vgpr"] <= 128 and r["vspill"] <= 24 and r["scratch"] <= 128, (n, r)
