#!/bin/bash
# Round 3, session 1: the whole GPU suite on the torch binding (new default) incl. the RCCL world-size-1 tests, smoke,
# bench under both bindings (A/B of the binding's host cost), bf16 generator counters.  Outputs: gpurun_out/r3s1/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3s1; mkdir -p $O
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/gpu_tests.txt
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== bench (torch binding, default)"; timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 10 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bench_torch.log | cut -c1-400
echo "== bench (ctypes binding)"; OPENVOICE_AMD_BINDING=ctypes timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bench_ctypes.log | cut -c1-300
echo "== latency B=1 both bindings"; timeout 200 python tools/bench_latency.py 2>&1 | tail -6 | tee $O/latency_torch.log
OPENVOICE_AMD_BINDING=ctypes timeout 200 python tools/bench_latency.py 2>&1 | tail -6 | tee $O/latency_ctypes.log
echo "== bf16 counters"; bash scripts/profile_bf16.sh 2>&1 | tail -60
mkdir -p $O/bf16; cp gpurun_out/bf16_* $O/bf16/ 2>/dev/null
