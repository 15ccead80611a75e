#!/bin/bash
# A/B of the XCD-contiguous tile order (OV_F_NO_XCD_MAP = 8 restores the round-robin order): kernel tests, per-shape
# table, bench line and one FETCH_SIZE pass each way.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/xcd; mkdir -p $O; A=$PWD
echo "== tests"; timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -x --timeout 200 2>&1 | tail -3 | tee $O/tests.log
for f in 0 8; do
  export OPENVOICE_AMD_CONV_FLAGS=$f
  echo "== convs flags=$f"; timeout 300 python tools/bench_convs.py --reps 4 2>&1 | grep -v amdgpu.ids | tee $O/convs_$f.log | tail -40
  echo "== bench flags=$f"; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $O/bench_$f.log | cut -c1-1200
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch_$f" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --pmc-calibration > "$A/$O/pmc_fetch_$f.log" 2>&1)
  python tools/pmc_summary.py $O/pmc_fetch_$f > $O/pmc_fetch_summary_$f.txt 2>&1
  find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
done
