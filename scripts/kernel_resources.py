"""Static resource table of every gfx950 kernel in dance_amd/libdancehip.so (no GPU needed): registers, LDS, scratch / spills and the
occupancy they allow, read from the AMDGPU metadata notes of the embedded code objects.

    python scripts/kernel_resources.py [path/to/libdancehip.so] > profiles/<tag>_kernel_resources.md

The .so carries one clang offload bundle per translation unit in .hip_fatbin; each bundle's gfx950 entry is an ELF whose
NT_AMDGPU_METADATA note ``llvm-readelf --notes`` prints as YAML.  ``.vgpr_count`` is the wave's whole allocation in the unified
512-entry register file of CDNA3/4 (architectural VGPRs plus the ``.agpr_count`` accumulation registers behind them), so occupancy
per SIMD = min(8, floor(512 / ceil8(vgpr_count))), further limited by LDS: floor(160 KB / lds) workgroups per CU."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob: bytes):
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        n, = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[base + off:base + off + size]


def kernels_of(elf: bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    for block in re.split(r"\n\s*- \.agpr_count:", "\n" + text)[1:]:
        block = ".agpr_count:" + block
        get = lambda key, d=None: (re.search(rf"\.{key}:\s*(\S+)", block) or [None, d])[1]
        name = get("name")
        if name is None:
            continue
        yield dict(name=name.strip("'\""), vgpr=int(get("vgpr_count", 0)), agpr=int(get("agpr_count", 0)), sgpr=int(get("sgpr_count", 0)),
                   lds=int(get("group_segment_fixed_size", 0)), scratch=int(get("private_segment_fixed_size", 0)),
                   vspill=int(get("vgpr_spill_count", 0)), sspill=int(get("sgpr_spill_count", 0)), wg=int(get("max_flat_workgroup_size", 0)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    short = []
    for full in out:
        s = re.sub(r"^void ", "", full)
        s = re.sub(r"\((?:[^()]|\([^()]*\))*\)(?: \[clone[^\]]*\])?$", "", s)   # drop the argument list
        s = re.sub(r"\(anonymous namespace\)::", "", s)
        short.append(s)
    return short


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dance_amd", "libdancehip.so")
    rows = []
    for elf in code_objects(open(path, "rb").read()):
        rows.extend(kernels_of(elf))
    for r, short in zip(rows, demangle([r["name"] for r in rows])):
        r["short"] = short
        regs = -(-r["vgpr"] // 8) * 8   # .vgpr_count already contains the AGPRs
        waves_wg = max(1, r["wg"] // 64)
        by_regs = min(8, 512 // max(regs, 8))
        wg_by_lds = (160 * 1024) // r["lds"] if r["lds"] else 10**6
        # waves per SIMD the LDS allows: workgroups per CU * waves per workgroup / 4 SIMDs
        by_lds = wg_by_lds * waves_wg / 4
        r["occ"] = min(by_regs, by_lds)
        r["limit"] = "regs" if by_regs <= by_lds else "LDS"
    library = [r for r in rows if r["short"].startswith(("rocprim::", "hipcub::"))]
    rows = sorted((r for r in rows if r not in library), key=lambda r: r["short"])
    print(f"# Kernel resources of `{os.path.basename(path)}` (gfx950, static: llvm-readelf --notes)\n")
    spilled = [r for r in rows if r["vspill"] or r["sspill"] or r["scratch"]]
    lib_scratch = sum(1 for r in library if r["scratch"])
    print(f"{len(library)} rocPRIM instantiations (radix sort / scan of the block and transpose builders; {lib_scratch} of them use scratch, "
          f"at most {max((r['scratch'] for r in library), default=0)} B) are not listed.\n")
    print(f"{len(rows)} kernels of this repo; {len(spilled)} use scratch or spill registers (sgpr spills go to VGPR lanes, not to memory)"
          + (": " + ", ".join(f"`{r['short']}` (scratch {r['scratch']} B, vgpr spills {r['vspill']}, sgpr spills {r['sspill']})" for r in spilled) if spilled else "") + ".\n")
    print("`occ` = resident waves per SIMD the registers / LDS allow (max 8); `wg` = the launch bound the kernel was compiled for.\n")
    print("| kernel | vgpr (total) | of which agpr | sgpr | LDS B | scratch B | wg | occ | limited by |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---|")
    for r in rows:
        occ = f"{r['occ']:.2g}" if r["occ"] < 8 else "8"
        print(f"| `{r['short']}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['lds']} | {r['scratch']} | {r['wg']} | {occ} | {r['limit'] if r['occ'] < 8 else '-'} |")


if __name__ == "__main__":
    main()
