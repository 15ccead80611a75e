#!/bin/bash
# Round 6 session 32: A/B of the Winograd policy at C = 32, K = 11 (dilated convs on Winograd launches too) at the
# zero-product kernels: contract line with engine.wino_policy forced to True, against the default.
#   gpurun --timeout 900 -- 'bash scripts/gpu_r6_s32.sh'   ->  gpurun_out/r6s32/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s32; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>$O/bench.err | tail -1 > $O/bench_default.json; cut -c1-330 $O/bench_default.json
timeout 300 python -c "
import sys, runpy
import openvoice_amd.engine as e
e.wino_policy = lambda C, K, dil: True
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-opt-in']
runpy.run_path('bench.py', run_name='__main__')" 2>$O/bench_c32_dilated_wino.err | tail -1 > $O/bench_c32_dilated_wino.json; cut -c1-330 $O/bench_c32_dilated_wino.json
