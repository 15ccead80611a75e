# rocprofv3 kernel stats of the V1 TTS path (BASELINE.json configs[3]): is it launch-bound?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_tts" -o r1 --output-format csv -- python "$OLDPWD/tools/bench_tts.py" --steps 3 --warmup 1 > "$OLDPWD/gpurun_out/prof_tts.log" 2>&1)
grep workload gpurun_out/prof_tts.log | cut -c1-300
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_tts/r1_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernel time per infer (4 runs): %.2f ms, launches per infer: %d' % (tot/4e6, calls/4))
for r in rows[:12]: print(r['Calls'], round(int(r['TotalDurationNs'])/4e6,3), r['Name'][:90])
PY
find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
