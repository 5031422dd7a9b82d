"""CPU (hipcc cross-compiles): the kernels whose gathers were found issuing ONE LOAD AT A TIME in round 5 — every load behind a divergent
guard, closed by ``s_waitcnt vmcnt(0)`` — keep their loads batched.  ``scripts/isa_load_audit.py`` compiles a source to
gfx950 assembly and reports runs of consecutive load -> wait-for-all pairs inside loops; the kernels named here must have none of
length >= 3 (what remains in these files are the first-form Student-t kernels kept for c > 32 and rare-path loops)."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")

CLEAN = {
    "gcn_narrow.hip": ("gcn_narrow_forward_kernel", ),
    "sddmm.hip": (),                      # (its 4-wide scale / column prefetch shows as a run of 4: checked below by an upper bound)
    "student_t.hip": ("student_t_forward_fast_kernel", "student_t_backward_fast_kernel"),
    "gemm_small.hip": (),
}


@pytest.mark.parametrize("src", sorted(CLEAN))
def test_gathers_stay_batched(src):
    import isa_load_audit as audit
    path = os.path.join(ROOT, "dance_amd", "csrc", src)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        audit.compile_to_asm(path, out)
        found = audit.audit(out, 3)
    by_kernel = {}
    for longest, mangled, runs in found:
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        by_kernel[name] = longest
    for needle in CLEAN[src]:
        bad = {k: v for k, v in by_kernel.items() if needle in k}
        assert not bad, f"{src}: loads serialised again in {bad}"
    # nothing in these files may come near the 15 - 50 dependent round trips the round-4 forms had
    assert all(v <= 6 for k, v in by_kernel.items() if "student_t_backward_kernel" not in k), by_kernel
