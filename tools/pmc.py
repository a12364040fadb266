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
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=96)
    ap.add_argument("--passes", default=",".join(PASSES))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
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
    path = os.path.join(os.path.dirname(args.out), "pmc_summary.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        if "process" in k or "sort" in k:
            print(k, json.dumps(out[k], sort_keys=True))


if __name__ == "__main__":
    main()
