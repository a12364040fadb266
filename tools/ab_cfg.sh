#!/bin/bash
# Developer tool: one bench configuration on several builds of the device library.  usage: bash tools/ab_cfg.sh "<bench args>" <lib-suffix>...
args=$1; shift
for v in "$@"; do
  lib=raytracer_amd/csrc/librt_hip${v:+_var_$v}.so
  RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 -u bench.py --cpu-seconds 0 --no-extras $args > gpurun_out/abc_$v.json || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/abc_$v.json')); s=d['stage_ms_per_step']
print('%-10s %7.1f Mrays/s %.4f ms/step  '%('${v:-default}', d['value'], d['ms_per_step'])+' '.join('%s %.1f'%(k,v*1e3) for k,v in s.items()))"
done
