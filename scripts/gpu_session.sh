#!/bin/bash
# One gpurun call = tests + microbench + bench (+ rocprof / PMC when PROFILE=1), logged under gpurun_out/.
#   gpurun --timeout 900 -- 'bash scripts/gpu_session.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
(nproc; python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|^CPU\(s\)|Socket") > gpurun_out/host_info.txt 2>&1
echo "== kernel tests" ; timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout 120 2>&1 | tail -15 | tee gpurun_out/test_kernels.log
echo "== e2e + tts + bf16 tests" ; timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_tts.py tests/test_gpu_bf16.py -q -m gpu --timeout 250 2>&1 | tail -6 | tee gpurun_out/test_e2e.log
echo "== smoke" ; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
if [ "${SQ:-0}" = "1" ]; then
  echo "== SQ counters on the k=3 C=128 conv"
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d "$OLDPWD/gpurun_out/pmc_sq" -o r1 --output-format csv -- python "$OLDPWD/tools/bench_convs.py" --channels 128 32 --kernels 3 11 --reps 2 > "$OLDPWD/gpurun_out/pmc_sq.log" 2>&1)
  python tools/pmc_summary.py gpurun_out/pmc_sq > gpurun_out/pmc_sq_summary.txt 2>&1; head -70 gpurun_out/pmc_sq_summary.txt
  find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
fi
if [ "${SWEEP:-0}" = "2" ]; then
  echo "== conv microbench: chunk sweep on k=3" ; timeout 300 python tools/bench_convs.py --kernels 3 --chunks 0 32 --channels 256 128 64 --reps 3 2>&1 | tee gpurun_out/convs_chunk.log | tail -30
  echo "== conv microbench: defaults" ; timeout 300 python tools/bench_convs.py --reps 3 --wn 2>&1 | tee gpurun_out/convs.log | tail -12
elif [ "${SWEEP:-0}" = "1" ]; then
  echo "== conv microbench sweep" ; timeout 300 python tools/bench_convs.py --tiles 1 2 3 4 --loaders 1 2 4 --tpw 1 --reps 3 --wn 2>&1 | tee gpurun_out/convs_sweep.log | tail -8
else
  echo "== conv microbench" ; timeout 200 python tools/bench_convs.py --reps 3 --wn --loaders 0 2>&1 | tee gpurun_out/convs.log | tail -8
fi
echo "== bench" ; timeout 300 python bench.py --steps 3 --warmup 1 --cpu-budget 12 2>&1 | tail -5 | tee gpurun_out/bench.log
if [ "${PROFILE:-0}" = "1" ]; then
  echo "== rocprof" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r1 --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
  grep '"metric"' gpurun_out/rocprof.log | cut -c1-400
  f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f"
  find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
  echo "== pmc fetch" ; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o r1 --output-format csv -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --pmc-calibration > "$OLDPWD/gpurun_out/pmc_fetch.log" 2>&1)
  echo "== pmc write" ; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OLDPWD/gpurun_out/pmc_write" -o r1 --output-format csv -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --pmc-calibration > "$OLDPWD/gpurun_out/pmc_write.log" 2>&1)
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.txt 2>&1; tail -30 gpurun_out/pmc_summary.txt
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_traffic.json 2>gpurun_out/pmc_traffic.err; cat gpurun_out/pmc_traffic.json
  find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
fi
