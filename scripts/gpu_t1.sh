#!/bin/bash
# tests + phase timers + per-shape table + bench (development loop)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/t1; mkdir -p $O
echo "== tests"; timeout 600 python -m pytest ${TESTS:-tests} -q -m gpu -x --timeout 200 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests.log
for v in "" n; do
  lib=$PWD/tools/micro/bin/libopenvoice_amd_exp3$v.so
  if [ -f $lib ]; then echo "== phases exp3$v"; EXP3_LIB=$lib bash scripts/conv_phases.sh > $O/phases$v.log 2>&1; grep -A4 "C=128 k=3\|C=64 k=3 plain\|C=128 k=11 plain" $O/phases$v.log; fi
done
echo "== convs"; timeout 300 python tools/bench_convs.py --reps 4 ${CONVS:-} 2>&1 | grep -v amdgpu.ids | tee $O/convs.log | tail -50
echo "== bench"; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $O/bench.log | cut -c1-1500
