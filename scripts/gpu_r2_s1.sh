#!/bin/bash
# Round 2, GPU session 1: accumulator-preload rewrite (parity + per-shape A/B), bench line, origin of the copyBuffer dispatches.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2s1; mkdir -p $O
echo "== tests"; timeout 500 python -m pytest tests -q -m gpu -x --timeout 200 2>&1 | tail -8 | tee $O/tests.log
echo "== convs (defaults, + persistent forced on the residual launches)"
timeout 400 python tools/bench_convs.py --reps 5 --wn --tpw 0 -1 --modes plain1 res res+add 2>&1 | grep -v amdgpu.ids | tee $O/convs.log | tail -70
echo "== bench"; timeout 300 python bench.py --steps 5 --warmup 2 --cpu-budget 10 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/bench.log
echo "== copy trace"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --hip-trace -d "$OLDPWD/$O/trace" -o r1 --output-format csv -- python "$OLDPWD/tools/trace_copies.py" --run > "$OLDPWD/$O/trace.log" 2>&1)
python tools/trace_copies.py $O/trace 2>&1 | tee $O/trace_summary.txt | tail -60
find $O/trace -name '*.csv' -size +8M -delete
