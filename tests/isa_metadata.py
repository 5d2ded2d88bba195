"""Kernel resource metadata of the gfx950 code object (test infrastructure): parses the amdhsa.kernels notes of the assembly
`hipcc -save-temps` leaves behind -- per kernel: VGPRs, SGPRs, spilled registers, scratch (private segment) bytes, LDS bytes,
workgroup size limit -- with names demangled by llvm-cxxfilt."""
import os
import re
import subprocess


FIELDS = {".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill",
          ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds", ".max_flat_workgroup_size": "max_threads",
          ".agpr_count": "agpr"}


def parse(asm_path):
    """{demangled kernel name: {vgpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, max_threads, agpr}}"""
    kernels, cur = [], None
    in_notes = False
    for line in open(asm_path, errors="replace"):
        if "amdhsa.kernels:" in line:
            in_notes = True
            continue
        if not in_notes:
            continue
        if line.startswith("amdhsa.") and "amdhsa.kernels" not in line:
            in_notes = False
            continue
        if re.match(r"\s*- \.", line):
            # a new list entry at kernel level starts with "  - .agpr_count" (args entries are deeper: "      - .")
            if re.match(r"  - \.", line):
                cur = {}
                kernels.append(cur)
        m = re.match(r"\s*(?:- )?(\.[a-z_]+):\s*(.+)$", line)
        if m and cur is not None and len(line) - len(line.lstrip()) <= 4:
            key, val = m.group(1), m.group(2).strip()
            if key == ".name":
                cur["mangled"] = val.strip("'\"")
            elif key in FIELDS:
                cur[FIELDS[key]] = int(val)
    names = [k["mangled"] for k in kernels if "mangled" in k]
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    table = {}
    for k, name in zip([k for k in kernels if "mangled" in k], out.stdout.splitlines()):
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        table[name] = {f: k.get(f, 0) for f in FIELDS.values()}
    return table


if __name__ == "__main__":
    import sys
    t = parse(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for name in sorted(t):
        if pat in name:
            r = t[name]
            print(f"{name:70s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} spill {r['vgpr_spill']:3d} scratch {r['scratch']:4d} lds {r['lds']:6d} threads {r['max_threads']}")
