#!/bin/bash
# usage: r3_ab.sh "<label>|<env assignments>" ...   -- one bench line (mad arithmetic) per configuration
mkdir -p gpurun_out/r3
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  env $envs timeout -k 10 300 python bench.py --no-extras --cpu-seconds 0 --arith ${ARITH:-mad} > gpurun_out/r3/ab_$label.json 2> gpurun_out/r3/ab_$label.err || { echo "$label failed"; tail -3 gpurun_out/r3/ab_$label.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r3/ab_$label.json"))
print("$label", round(d["value"],1), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_step"].items()})
PY
done
