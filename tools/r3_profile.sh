#!/bin/bash
# Bench line, rocprofv3 kernel statistics (overlapped and serial), the other configurations, PMC passes; files are named r3_*.
set -e
o=$PWD/gpurun_out/r3p
mkdir -p $o
B="--cpu-seconds 0 --no-extras"
timeout -k 10 600 python3 -u bench.py > $o/r3_bench.json 2> $o/r3_bench.err
timeout -k 10 120 python3 bench.py $B --arith strict > $o/x_strict.json
timeout -k 10 120 python3 bench.py $B --schedule reference > $o/x_reference.json
timeout -k 10 120 python3 bench.py $B --steps 20 --warmup 5 > $o/x_driver.json
timeout -k 10 120 python3 bench.py $B --serial > $o/x_serial.json
timeout -k 10 200 python3 bench.py $B --width 3840 --height 2160 --shading diffuse --bounces 2 > $o/x_4k_diffuse.json
timeout -k 10 200 python3 bench.py $B --scene dragon --grid 5 --shading diffuse > $o/x_grid5.json
timeout -k 10 200 python3 bench.py $B --gpus 1 --collective > $o/x_collective.json
for f in r3_bench x_strict x_reference x_driver x_serial x_4k_diffuse x_grid5 x_collective; do python3 -c "
import json; d=json.loads([l for l in open('$o/$f.json') if l.startswith('{')][-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],4), d['roofline'].get('frac'), d['roofline'].get('traffic'), d['stage_ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o r3 -- python3 -u /root/repo/bench.py $B > $o/prof_bench.json 2> $o/prof.err
find $o/prof -name "*kernel_stats.csv" -exec cp {} $o/r3_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_serial -o r3s -- python3 -u /root/repo/bench.py $B --serial > $o/prof_serial_bench.json 2> $o/prof_serial.err
find $o/prof_serial -name "*kernel_stats.csv" -exec cp {} $o/r3_serial_kernel_stats.csv \;
find $o -name "*_kernel_trace.csv" -delete; find $o -name "*.db" -delete
head -8 $o/r3_serial_kernel_stats.csv
cd /root/repo && timeout -k 10 900 python3 tools/pmc.py --out $o/pmc --summary pmc_summary.json > $o/pmc.log 2>&1 || { tail -5 $o/pmc.log; exit 1; }
tail -3 $o/pmc.log | cut -c1-300
