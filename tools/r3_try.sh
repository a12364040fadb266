#!/bin/bash
# usage: r3_try.sh "<env assignments>" [pytest -k expression]: free-schedule tests under the given environment, then a bench line
mkdir -p gpurun_out/r3
env $1 timeout -k 10 600 python -u -m pytest tests/test_gpu_pipeline.py tests/test_bounce_shading.py tests/test_gpu_configs.py tests/test_arith_modes.py -x -q -m gpu -k "${2:-free or configs or bounce or mad}" --timeout 300 --timeout-method thread > gpurun_out/r3/t_try.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3/t_try.log
bash tools/r3_ab.sh "base|X=1" "try|$1"
