#!/bin/bash
# Round 5, closing check at HEAD (after the length-aware work lists in the split conv): whole GPU suite, smoke, the contract
# bench line with the driver's defaults.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5close; mkdir -p $O
echo "== gpu suite"; timeout 1200 python -m pytest tests -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
