#!/bin/bash
# Developer tool: bench several builds of the device library back to back on one box.
# usage: bash tools/ab.sh <rounds> <lib-suffix>...   ("" = the default library)
rounds=$1; shift
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for v in "$@"; do
    lib=raytracer_amd/csrc/librt_hip${v:+_var_$v}.so
    RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 -u bench.py --cpu-seconds 0 --no-extras > gpurun_out/ab_$v.json || exit 1
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab_$v.json')); s=d['stage_ms_per_step']
print('%-12s %7.1f Mrays/s %.4f ms/step  '%('${v:-default}', d['value'], d['ms_per_step'])+' '.join('%s %.1f'%(k,v*1e3) for k,v in s.items()))"
  done
done
