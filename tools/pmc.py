#!/usr/bin/env python3
"""Developer tool: hardware counters of the hot-path kernels, one rocprofv3 --pmc pass per counter set
(each pass is its own run of a short bench.py), aggregated per kernel into gpurun_out/pmc_summary.json.
Run on the GPU box:  python3 tools/pmc.py [--steps 16]"""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "sq_a": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM",
             "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"],
    "sq_b": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
             "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"],
    "grbm": ["GRBM_GUI_ACTIVE"],
    # the vector memory path (not part of the default report: --passes ta,ta2,tcp_a,tcp_b,tcp_c,tcp_d)
    "ta": ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"],
    "ta2": ["TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TOTAL_WAVEFRONTS_sum"],
    "tcp_a": ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    "tcp_b": ["TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
    "tcp_c": ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum",
              "TCP_UTCL1_STALL_INFLIGHT_MAX_sum"],
    "tcp_d": ["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum", "TCP_LFIFO_STALL_CYCLES_sum",
              "TCP_RFIFO_STALL_CYCLES_sum"],
    # the scalar path (--passes sqc_a,sqc_b)
    "sqc_a": ["SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQC_DCACHE_MISSES_DUPLICATE", "SQC_DCACHE_BUSY_CYCLES",
              "SQC_TC_DATA_READ_REQ", "SQC_TC_STALL", "SQ_INST_LEVEL_SMEM"],
    "sqc_b": ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_BUSY_CYCLES", "SQ_IFETCH", "SQ_IFETCH_LEVEL",
              "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
}
DEFAULT_PASSES = "sq_a,sq_b,fetch,write,l2,grbm"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=96)
    ap.add_argument("--passes", default=DEFAULT_PASSES)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
    ap.add_argument("--summary", default="pmc_summary.json", help="file name of the summary under gpurun_out/")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    summary = {}
    for name in args.passes.split(","):
        d = os.path.join(args.out, name)
        cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc", *PASSES[name], "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--cpu-seconds", "0", "--no-extras", *args.extra.split()]
        print("pass", name, flush=True)
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode:
            print("pass failed", name, r.returncode, r.stderr[-2000:])
            sys.exit(1)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"].split("(")[0]
                    c = row["Counter_Name"]
                    e = summary.setdefault(k, {}).setdefault(c, [0.0, 0])
                    e[0] += float(row["Counter_Value"])
                    e[1] += 1
    out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in summary.items()}
    for k, cs in summary.items():
        out[k]["_dispatches"] = max(v[1] for v in cs.values())
    path = os.path.join(os.path.dirname(args.out), args.summary)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        if "process" in k or "sort" in k:
            print(k, json.dumps(out[k], sort_keys=True))


if __name__ == "__main__":
    main()
