#!/usr/bin/env python3
"""Per-kernel resources (VGPRs, SGPRs, LDS, scratch, spills) of every kernel in the SHIPPED liblmc_atomi.so, read from the code-object
metadata notes.  The .so carries one clang offload bundle per translation unit in its .hip_fatbin section; each bundle is split off,
its gfx950 code object unbundled and its AMDGPU metadata note parsed.

    python scripts/kernel_resources.py [path/to/liblmc_atomi.so] [--json]

Used by tests/test_kernel_resources.py (the scratch fence: DESIGN section 3.0p "Scratch finding") and for the tables in DESIGN.md."""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "lmc_atomi_amd", "lib", "liblmc_atomi.so")


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return [re.sub(r"^void ", "", d).replace("lmc::", "") for d in out]


def kernel_resources(lib=DEFAULT_LIB):
    """-> list of dicts {name, vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill, wg_max}"""
    res = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "unused.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else len(blob)
            part = os.path.join(d, f"b{i}.bin")
            co = os.path.join(d, f"b{i}.co")
            open(part, "wb").write(blob[s:e])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
            if os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            if "amdhsa.kernels:" not in notes:
                continue
            md = notes[notes.index("amdhsa.kernels:"):]
            for blk in re.split(r"\n\s+- \.agpr_count:", md)[1:]:
                blk = ".agpr_count:" + blk

                def g(key, default=0):
                    m = re.search(r"\.%s:\s+(\d+)" % key, blk)
                    return int(m.group(1)) if m else default
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                res.append(dict(name=name, vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                                scratch=g("private_segment_fixed_size"), vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"),
                                wg_max=g("max_flat_workgroup_size")))
    for r, dm in zip(res, _demangle([r["name"] for r in res])):
        r["demangled"] = dm
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rs = kernel_resources(args[0] if args else DEFAULT_LIB)
    if "--json" in sys.argv:
        print(json.dumps(rs, indent=1))
    else:
        for r in sorted(rs, key=lambda r: (-r["scratch"], r["demangled"])):
            print(f"{r['demangled'][:110]:110s} vgpr {r['vgpr']:4d} sgpr {r['sgpr']:4d} lds {r['lds']:6d} scratch {r['scratch']:5d} spill {r['vgpr_spill']:3d}")
        print(f"{len(rs)} kernels; max scratch {max(r['scratch'] for r in rs)} B per lane")
