#!/bin/bash
# Bench line + rocprofv3 kernel statistics of the same command, for profiles/.
# usage (on the GPU box): bash tools/profile_round.sh <tag>
set -e
tag=${1:-run}
out=$PWD/gpurun_out
mkdir -p $out
timeout -k 10 300 python3 -u bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 -u /root/repo/bench.py --cpu-seconds 0 --no-extras > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err
find $out/${tag}_prof -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_serial -o ${tag}_serial -- python3 -u /root/repo/bench.py --cpu-seconds 0 --no-extras --serial > $out/${tag}_serial_bench.json 2> $out/${tag}_prof_serial.err
find $out/${tag}_prof_serial -name "*kernel_stats.csv" -exec cp {} $out/${tag}_serial_kernel_stats.csv \;
find $out/${tag}_prof_serial -name "*_kernel_trace.csv" -delete
find $out/${tag}_prof -name "*_kernel_trace.csv" -delete
find $out/${tag}_prof -name "*.db" -delete
head -12 $out/${tag}_kernel_stats.csv
cat $out/${tag}_bench.json
