#!/bin/bash
# Round 5, session 6: split-precision conv with products ordered by weight plane + all weight requests in a pair's first
# block + one epilogue barrier for the single-round geometry (C = 64); the MP3 -> HIP path end to end.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s6; mkdir -p $O
echo "== split3 tests + mp3 e2e"; timeout 500 python -m pytest tests/test_gpu_split3.py tests/test_gpu_split3_e2e.py tests/test_gpu_e2e.py::test_convert_and_extract_se_from_an_mp3_file -q -m gpu --timeout 300 -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests_split3.log
echo "== split3 table"; timeout 400 python tools/bench_split3.py --shapes --reps 6 --dbg --products3 --out $O/split3_table.json 2>&1 | grep -v amdgpu.ids | grep -v "^{" | cut -c1-420 | tee $O/split3_table.log
echo "== bench, opt-in split line"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --no-cpu-baseline 2>$O/bench_split.err | tail -1 | tee $O/bench_split.json | cut -c1-200
echo "== bench, opt-in split line, 3 products"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --split-products 3 --no-cpu-baseline 2>$O/bench_split3p.err | tail -1 | tee $O/bench_split_3products.json | cut -c1-200
