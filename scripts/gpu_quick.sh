#!/bin/bash
# Quick GPU check: parity tests, then the bench line.  OUT=<name> picks the log directory under gpurun_out/;
# CONVS="<bench_convs.py arguments>" adds a per-shape table; TESTS="<pytest paths>" narrows the tests.
#   gpurun --timeout 900 -- 'OUT=r2s2 bash scripts/gpu_quick.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-quick}; mkdir -p $O
echo "== tests"; timeout 600 python -m pytest ${TESTS:-tests} -q -m gpu -x --timeout 200 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests.log
if [ -n "${CONVS:-}" ]; then
  echo "== convs $CONVS"; timeout 500 python tools/bench_convs.py $CONVS 2>&1 | grep -v amdgpu.ids | tee $O/convs.log | tail -${CONVS_TAIL:-60}
fi
if [ "${BENCH:-1}" = "1" ]; then
  echo "== bench"; timeout 400 python bench.py --steps ${STEPS:-5} --warmup 2 --cpu-budget ${CPU_BUDGET:-12} ${BENCH_ARGS:-} 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/bench.log
fi
