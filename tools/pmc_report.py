#!/usr/bin/env python3
"""Developer tool: turns gpurun_out/pmc_summary.json (tools/pmc.py) into the files kept under profiles/:
<tag>_pmc.json (per kernel and dispatch: counters + derived figures) and traffic.json (HBM bytes per launch,
read by bench.py for roofline.traffic).  usage: python3 tools/pmc_report.py <tag>"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
schedule = sys.argv[2] if len(sys.argv) > 2 else "free"   # which bench.py --schedule the counters were collected on
arith = sys.argv[3] if len(sys.argv) > 3 else "mad"       # ... and which --arith
src = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_summary.json")))
report, traffic = {}, {}
for name, c in sorted(src.items()):
    if "rtd::k_" not in name:
        continue
    k = name.split("::")[-1]
    if "<" in k:
        # template arguments: <schedule is free, multiply-adds fused> for the visiting kernels, <fused> for the others;
        # the other instances of a kernel were not launched in this run, or only by set-up calls
        targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
        if len(targs) == 2 and targs != [str(schedule == "free").lower(), str(arith == "mad").lower()]:
            continue
        if len(targs) == 1 and targs[0] in ("true", "false") and targs[0] != str(arith == "mad").lower():
            continue
    k = k.split("<")[0]
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes,
    # so the read side is doubled (calibrated for wide streaming reads; other widths are uncalibrated).
    rd = 2 * c.get("FETCH_SIZE", 0.0) * 1024
    wr = c.get("WRITE_SIZE", 0.0) * 1024
    simd_cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8 * 1024  # GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
    report[k] = {
        "dispatches_averaged": c.get("_dispatches"),
        "hbm_read_bytes": rd, "hbm_write_bytes": wr,
        "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_REQ_sum", 0.0)),
        "waves": c.get("SQ_WAVES"),
        "valu_wave_instructions": c.get("SQ_INSTS_VALU"),
        "salu_wave_instructions": c.get("SQ_INSTS_SALU"),
        "smem_wave_instructions": c.get("SQ_INSTS_SMEM"),
        "lds_wave_instructions": c.get("SQ_INSTS_LDS"),
        "vmem_read_wave_instructions": c.get("SQ_INSTS_VMEM_RD"),
        "vmem_write_wave_instructions": c.get("SQ_INSTS_VMEM_WR"),
        # share of the SIMDs' cycles in which a vector instruction executes (rocprof's VALUBusy: SQ_ACTIVE_INST_VALU is
        # in units of four cycles and covers what an instruction really takes -- packed and transcendental ones more
        # than the two cycles of a plain wave64 operation), and the share of lanes active in those instructions
        "valu_busy": 4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles if simd_cycles else None,
        "valu_lanes_active": c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64 * c.get("SQ_ACTIVE_INST_VALU", 0.0)),
        # lower bound of the same: every instruction counted as a plain wave64 operation of two cycles
        "valu_utilisation": 2 * c.get("SQ_INSTS_VALU", 0.0) / simd_cycles if simd_cycles else None,
        "wave_quad_cycles": c.get("SQ_WAVE_CYCLES"),
        "share_waiting": c.get("SQ_WAIT_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0)),
        "share_issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0)),
        "share_issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0)),
        "gpu_busy_cycles_per_xcd": c.get("GRBM_GUI_ACTIVE", 0.0) / 8,
    }
    traffic[k] = rd + wr
json.dump(report, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
import hashlib
lib = os.path.join(ROOT, "raytracer_amd", "csrc", "librt_hip.so")
stamp = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "schedule": schedule, "arithmetic": arith,
         "unit": "HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies a 128-byte read request as 64 bytes; calibrated on "
                 "wide streaming reads, so the scattered 64-byte requests of these kernels may be counted up to twice: the read side lies "
                 "between kernels_read_once and this figure)",
         "kernels_fetch_counted_once": {k: r["hbm_read_bytes"] / 2 + r["hbm_write_bytes"] for k, r in report.items()},
         "kernels": traffic}
json.dump(stamp, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for k, r in report.items():
    print(f"{k:18s} rd {r['hbm_read_bytes']/1e6:7.1f} MB wr {r['hbm_write_bytes']/1e6:7.1f} MB  L2 hit {r['l2_hit_rate']:.2f}  "
          f"VALU busy {r['valu_busy'] or 0:.2f} (lanes {r['valu_lanes_active']:.2f})  wait {r['share_waiting']:.2f} stall {r['share_issue_stalled']:.2f} issue {r['share_issuing']:.2f}")
