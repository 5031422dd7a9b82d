"""Instruction mix of a kernel's loops, from the gfx950 code objects embedded in libdancehip.so (static, no GPU):

    python scripts/isa_loop_mix.py 'sage_mfma_kernel<false, true, 7, false>' > profiles/<tag>_sage_mfma_isa.md

Every backward branch closes a loop; the loops are listed largest first with their instruction counts by class (matrix-core, vector
ALU, scalar ALU, LDS, global / buffer memory, waitcnt, barrier).  A loop that contains another is reported with the inner one inside."""
import bisect
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def klass(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    return "valu" if op.startswith("v_") else "other"


def main():
    want = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dance_amd", "libdancehip.so")
    for elf in code_objects(open(path, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        text = subprocess.run(["c++filt"], input=text, capture_output=True, text=True).stdout
        body, inside = [], False
        for line in text.splitlines():
            if re.match(r"^[0-9a-f]+ <.*>:$", line):
                if inside:
                    break
                inside = want in line
                continue
            if inside:
                body.append(line)
        if body:
            break
    else:
        raise SystemExit(f"no kernel matching {want!r}")
    ins = []
    for line in body:
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addrs = [a for a, _, _ in ins]
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")) and args.split():
            off = int(args.split()[0])
            off -= 65536 if off >= 32768 else 0
            if off < 0:
                nxt = ins[i + 1][0] if i + 1 < len(ins) else a + 4
                loops.append((i - bisect.bisect_left(addrs, nxt + off * 4) + 1, bisect.bisect_left(addrs, nxt + off * 4), i))
    loops.sort(reverse=True)
    cols = ("mfma", "valu", "salu", "lds", "vmem", "waitcnt", "barrier", "accvgpr")
    print(f"# Loops of `{want}` (gfx950 ISA, static)\n")
    total = collections.Counter(klass(o) for _, o, _ in ins)
    print(f"{len(ins)} instructions in the kernel: " + ", ".join(f"{total[c]} {c}" for c in cols if total[c]) + ".\n")
    print("| loop (instruction index range) | instructions | " + " | ".join(cols) + " |")
    print("|---|---:|" + "---:|" * len(cols))
    seen = set()
    for n, lo, hi in loops[:12]:
        if (lo, hi) in seen or n < 12:
            continue
        seen.add((lo, hi))
        c = collections.Counter(klass(o) for _, o, _ in ins[lo:hi + 1])
        print(f"| {lo}..{hi} | {n} | " + " | ".join(str(c[k]) for k in cols) + " |")


if __name__ == "__main__":
    main()
