#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 600 python bench.py --cpu-seconds 0 --arith mad > gpurun_out/r3/bench_mad_full.json 2> gpurun_out/r3/bench_mad_full.err || { tail -5 gpurun_out/r3/bench_mad_full.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r3/bench_mad_full.json"))
print(round(d["value"],1), round(d["ms_per_step"],4))
print(json.dumps(d["extra"].get("fixed_work"))[:3000])
print(d["extra"].get("error"))
PY
