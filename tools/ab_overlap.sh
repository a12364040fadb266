#!/bin/bash
# Developer tool: A/B of the side-stream overlap (run on the GPU box).
set -e
for v in 1 2; do
  timeout -k 10 300 python -u bench.py --steps 64 --warmup 96 --cpu-seconds 0 > gpurun_out/b_ov.json
  python -c "import json; d=json.load(open('gpurun_out/b_ov.json')); print('overlap', d['value'], d['ms_per_step'])"
  RT_NO_OVERLAP=1 timeout -k 10 300 python -u bench.py --steps 64 --warmup 96 --cpu-seconds 0 > gpurun_out/b_no.json
  python -c "import json; d=json.load(open('gpurun_out/b_no.json')); print('serial ', d['value'], d['ms_per_step'])"
done
