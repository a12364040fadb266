#!/bin/bash
# Developer tool: kernel timeline of a few steps (start/end per dispatch) to look at idle gaps.
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 -u /root/repo/bench.py --cpu-seconds 0 --no-extras --steps 16 --warmup 96 ${BENCH_FLAGS} > /dev/null 2> $out/trace.err
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get('OUT', '/root/repo/gpurun_out'), 'trace', '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1], r.get('Stream_Id', r.get('Queue_Id', '')))
        for r in csv.DictReader(open(f))]
rows.sort()
# steps 100..104 of the timed region: find k_process_aabb starts
starts = [i for i, r in enumerate(rows) if r[2] == 'k_process_aabb']
lo, hi = starts[100], starts[103]
t0 = rows[lo][0]
for s, e, n, q in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {n} [{q}]")
PY
