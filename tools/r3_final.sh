#!/bin/bash
# what the driver runs at the end of a round: smoke, the GPU tests, the default bench line (timed)
mkdir -p gpurun_out/r3f
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3f/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3f/smoke.log
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 600 --timeout-method thread > gpurun_out/r3f/t_all.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r3f/t_all.log
s=$(date +%s.%N)
timeout -k 10 900 python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc=$?"
e=$(date +%s.%N); echo "bench wall: $(echo "$e - $s" | bc) s"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3f/bench_driver.json 2> gpurun_out/r3f/bench_driver.err
python - <<PY
import json
for f in ("bench","bench_driver"):
    d=json.loads([l for l in open("gpurun_out/r3f/%s.json"%f) if l.startswith("{")][-1])
    print(f, round(d["value"],1), round(d["ms_per_step"],4), d["roofline"]["frac"], d["roofline"]["traffic"], d.get("cpu_baseline",{}).get("value"))
PY
