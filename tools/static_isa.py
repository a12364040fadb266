#!/usr/bin/env python3
"""Developer tool, CPU only: what the compiler made of the kernels, read off the device assembly.

    python tools/static_isa.py > profiles/<tag>_static_isa.md

For every kernel of the device library: registers, scratch, LDS and occupancy (hipcc's kernel-resource-usage remarks);
for the free-schedule process kernels the instruction mix of every innermost loop of the assembly hipcc emits for
gfx950 (vector ALU, of which packed FP32 and division-sequence instructions; scalar ALU; scalar and vector memory;
LDS; waits; branches).  Nothing is run: the counts are static, per trip through the loop body with every branch inside
it counted once.  Loops that touch no memory (the hit network, the merge's rank loops) are left out."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LOOP_KERNELS = ("_ZN3rtd14k_process_aabbILb1ELb1EEEvNS_11ProcessArgsEPj", "_ZN3rtd14k_process_meshILb1ELb1EEEvNS_11ProcessArgsE",
                "_ZN3rtd12k_fused_rankEPK13rt_index_pairPNS_9DevGlobalEPjS5_j", "_ZN3rtd13k_fused_placeEPK13rt_index_pairPKjPS0_NS_9PlacementE")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {n: re.sub(r"\(.*", "", d).replace("rtd::", "").replace("void ", "") for n, d in zip(names, out)}


def resources(source):
    err = subprocess.run([HIPCC] + FLAGS + ["-c", "-Rpass-analysis=kernel-resource-usage", source, "-o", os.devnull],
                         capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = {"name": m.group(2)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(1).split(" ")[0]] = m.group(2)
    return rows


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    return "other"


def innermost_loops(asm, kernel):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels = {m.group(1): i for i in range(start, end) for m in [re.match(r"^(\.LBB\d+_\d+):", lines[i])] if m}
    loops = set()
    for i in range(start, end):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and labels.get(m.group(1), end) < i:
            loops.add((labels[m.group(1)], i))
    heads = {}
    for a, b in loops:  # a loop = its header with the farthest back edge
        heads[a] = max(heads.get(a, 0), b)
    spans = sorted(heads.items())
    inner = [(a, b) for a, b in spans if not any(a < c and d <= b and (c, d) != (a, b) for c, d in spans)]
    out = []
    for a, b in inner:
        mix = {}
        for l in lines[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = l.split()[0]
            mix[kind(op)] = mix.get(kind(op), 0) + 1
            if op.startswith("v_pk_"):
                mix["packed"] = mix.get("packed", 0) + 1
            if op.startswith(("v_rcp", "v_div_")):
                mix["division"] = mix.get("division", 0) + 1
            if op.startswith(("v_writelane", "v_readlane")):
                mix["lanes"] = mix.get("lanes", 0) + 1
        out.append((a - start, b - start, mix))
    return out


def main():
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    rows = [r for f in sources for r in resources(os.path.join(CSRC, f))]
    names = demangle([r["name"] for r in rows])
    print("# Static view of the device library (gfx950, flags of the Makefile)\n")
    print("Registers, scratch, LDS, occupancy: `hipcc -Rpass-analysis=kernel-resource-usage`.  Template arguments: "
          "`<FREE, MAD>` = free schedule, fused arithmetic.\n")
    print("| kernel | VGPRs | SGPRs | scratch B/lane | LDS B/block | waves/SIMD |\n|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{names[r['name']]}` | {r.get('VGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize')} | {r.get('LDS')} | {r.get('Occupancy')} |")
    with tempfile.TemporaryDirectory() as tmp:
        for f in sources:
            asm_path = os.path.join(tmp, f + ".s")
            subprocess.run([HIPCC] + FLAGS + ["-S", os.path.join(CSRC, f), "-o", asm_path], check=True, capture_output=True)
            asm = open(asm_path).read()
            for k in LOOP_KERNELS:
                if ("\n" + k + ":") not in asm:
                    continue
                print(f"\n## Innermost loops of `{demangle([k])[k]}` (free schedule, fused arithmetic)\n")
                print("| assembly lines | VALU | of which packed | of which division sequence | of which scalar-register spill moves | SALU | scalar loads | vector memory | LDS | waits | branches |\n|---|---|---|---|---|---|---|---|---|---|---|")
                for a, b, m in innermost_loops(asm, k):
                    if not m.get("smem", 0) and not m.get("vmem", 0):  # the scan loops and the loops that touch memory
                        continue
                    print(f"| {a}-{b} | {m.get('valu', 0)} | {m.get('packed', 0)} | {m.get('division', 0)} | {m.get('lanes', 0)} | {m.get('salu', 0)} | {m.get('smem', 0)} | "
                          f"{m.get('vmem', 0)} | {m.get('lds', 0)} | {m.get('wait', 0)} | {m.get('branch', 0)} |")


if __name__ == "__main__":
    sys.exit(main())
