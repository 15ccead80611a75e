#!/bin/bash
# Round 6 session 34: the opt-in split-precision path with stage 2 (C = 64) left on the fp32 Winograd launches instead of the split
# convs (which tie there, and cost two layout passes over the stage's tensors): bench.py --split-bf16x3 as is, and with stage 2 off.
#   gpurun --timeout 900 -- 'bash scripts/gpu_r6_s34.sh'   ->  gpurun_out/r6s34/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s34; mkdir -p $O
timeout 300 python bench.py --split-bf16x3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/split_all.err | tail -1 > $O/split_all.json; cut -c1-330 $O/split_all.json
timeout 300 python -c "
import sys, runpy
import openvoice_amd.engine as e
orig = e.ConverterEngine.use_split_bf16x3
def patched(self, enable=True, products=6):
    r = orig(self, enable, products)
    if self.split_resblocks is not None and len(self.split_resblocks) > 2:
        self.split_resblocks[2] = None
    return r
e.ConverterEngine.use_split_bf16x3 = patched
sys.argv = ['bench.py', '--split-bf16x3', '--steps', '6', '--warmup', '2', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>$O/split_stages01.err | tail -1 > $O/split_stages01.json; cut -c1-330 $O/split_stages01.json
