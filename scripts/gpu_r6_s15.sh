#!/bin/bash
# Round 6 session 15: where do the Winograd-domain launches lose time?  Phase timers at K = 3 / 7 / 11 (C = 128), then the
# staggered-start A/B (tools/wino_stagger_ab.py) at 0 / 0.25 / 0.5 / 1 item periods.
#   gpurun --timeout 900 -- 'bash scripts/gpu_r6_s15.sh'   ->  gpurun_out/r6s15/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s15; mkdir -p $O
for k in 3 7 11; do timeout 120 python tools/wino_phases.py --K $k 2>&1 | tail -1 | tee -a $O/phases.jsonl; done
timeout 120 python tools/wino_phases.py --K 3 --res 2>&1 | tail -1 | tee -a $O/phases.jsonl
for s in -1 16 32 64 -1; do timeout 300 python tools/wino_stagger_ab.py --stagger $s 2>&1 | tail -1 | tee -a $O/stagger.jsonl; done
