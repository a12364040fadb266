set -e
o=gpurun_out
timeout -k 10 300 python3 bench.py > $o/r2_bench.json 2> $o/r2_bench.err
B="--cpu-seconds 0 --no-extras"
timeout -k 10 120 python3 bench.py $B --schedule reference > $o/x_reference.json
timeout -k 10 120 python3 bench.py $B --steps 20 --warmup 5 > $o/x_driver.json
timeout -k 10 120 python3 bench.py $B --serial > $o/x_serial.json
timeout -k 10 200 python3 bench.py $B --width 3840 --height 2160 --shading diffuse --bounces 2 > $o/x_4k_diffuse.json
timeout -k 10 200 python3 bench.py $B --scene dragon --grid 5 --shading diffuse > $o/x_grid5.json
timeout -k 10 200 python3 bench.py $B --scene dragon --grid 7 --shading diffuse > $o/x_grid7.json
for f in r2_bench x_reference x_driver x_serial x_4k_diffuse x_grid5 x_grid7; do python3 -c "
import json; d=json.load(open('$o/$f.json')); print('$f', round(d['value'],1), round(d['ms_per_step'],4), d['roofline'].get('traffic'), d['stage_ms_per_step'])"; done
