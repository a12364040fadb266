#!/bin/bash
# tests of the arithmetic modes + the whole GPU suite + a bench line per mode
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests -x -q -s -m gpu --timeout 600 --timeout-method thread > gpurun_out/r3/t_all.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r3/t_all.log
for a in strict mad; do
  timeout -k 10 300 python bench.py --no-extras --cpu-seconds 0 --arith $a > gpurun_out/r3/bench_$a.json 2> gpurun_out/r3/bench_$a.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r3/bench_$a.json"))
print("$a", round(d["value"],1), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_step"].items()})
PY
done
