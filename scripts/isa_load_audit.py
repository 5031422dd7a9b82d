"""Which kernels issue their global loads ONE AT A TIME?  (CPU-only tool: hipcc cross-compiles, nothing runs.)

A load whose value is consumed inside a divergent `if` sits in its own exec-masked basic block, and the compiler closes that block with
`s_waitcnt vmcnt(0)`: N guarded loads in a row are N dependent memory round trips, whatever the source says about "N in flight"
(round 5: gcn_narrow_forward, sddmm_csr, the tile loops of student_t, colsum_bf16 — DESIGN.md §3.3).  This script compiles every
dance_amd/csrc/*.hip to gfx950 assembly and reports, per kernel, the longest run of consecutive `load -> s_waitcnt vmcnt(0)` pairs
with no second load issued in between, inside loops (the label comments `in Loop: ... Depth=`).  A long run in a hot loop is a suspect;
prologues (fragment preloads) and boundary epilogues show up too and are harmless — read the ISA before acting.

    python scripts/isa_load_audit.py [--min-run 3] [file.hip ...]
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dance_amd", "csrc")


def compile_to_asm(src, out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDH_BUILDING", "-ffp-contract=off", "-S", "--cuda-device-only",
           f"-I{CSRC}", f"-I{os.path.join(ROOT, 'include')}", src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def audit(asm_path, min_run):
    lines = open(asm_path).read().split("\n")
    found, i = [], 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*; @", lines[i])
        if not m:
            i += 1
            continue
        j, body = i + 1, []
        while j < len(lines) and "s_endpgm" not in lines[j]:
            body.append(lines[j])
            j += 1
        depth, pending, run, run_depth, runs = 0, None, 0, 0, []
        for line in body:
            ls = line.strip()
            if line.startswith(".LBB"):
                mm = re.search(r"Depth=(\d+)", line)
                depth = int(mm.group(1)) if mm else (depth if "in Loop" in line else 0)
            if re.match(r"(global|buffer|flat)_load", ls):
                if pending is not None:  # the previous load was not waited for: loads are batched here
                    if run >= min_run:
                        runs.append((run, run_depth))
                    run = 0
                pending = True
            elif ls.startswith("s_waitcnt") and "vmcnt(0)" in ls:
                if pending is not None:
                    run += 1
                    run_depth = depth if run == 1 else max(run_depth, depth)
                pending = None
        if run >= min_run:
            runs.append((run, run_depth))
        in_loops = [r for r in runs if r[1] > 0]
        if in_loops:
            found.append((max(r[0] for r in in_loops), m.group(1), in_loops))
        i = j
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-run", type=int, default=3)
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
            try:
                compile_to_asm(f, out)
            except subprocess.CalledProcessError:
                print(f"(could not compile {f})", file=sys.stderr)
                continue
            rows += [(n, os.path.basename(f), demangle(k)[:140], r) for n, k, r in audit(out, a.min_run)]
    rows.sort(reverse=True)
    print("| longest serial run | file | kernel | runs (length, loop depth) |\n|---|---|---|---|")
    for n, f, k, r in rows:
        print(f"| {n} | {f} | `{k}` | {r[:6]} |")


if __name__ == "__main__":
    main()
