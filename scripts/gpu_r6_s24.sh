#!/bin/bash
# Round 6 session 24: where the k = 11 Winograd launches lose their 20 % (mfma_busy 0.80 at C = 128, 0.67 at C = 64, 0.49 at
# C = 32): measurement builds of conv1d_wino_k11.hip (scripts/exp_wino.sh; results of these builds are meaningless) --
# 2 = no transform, 3 = no weight stream, 4 = no epilogue stores, 5 = no B-operand reads in the k-loop, 6 = helpers idle.
#   gpurun --timeout 900 -- 'bash scripts/gpu_r6_s24.sh'   ->  gpurun_out/r6s28/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s28; mkdir -p $O
cat > /tmp/t.py <<'P'
import sys, json, torch
sys.path.insert(0, ".")
from openvoice_amd import wino
from tools.bench_split3 import ramp, timed
DEV = "cuda:0"
rows = []
ramp()
for C, L in ((128, 55104), (64, 110208), (32, 220416), (256, 6888)):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(C, C, 11, generator=g) * (C * 11) ** -0.5
    layer = wino.PackedConvWino(w, torch.zeros(C), DEV)
    x = torch.randn(32, C, L, generator=g).to(DEV); out = torch.empty_like(x); res = torch.randn(32, C, L, generator=g).to(DEV)
    bs = C * L
    t0 = min(timed(lambda: wino.launch_conv_wino(layer, x, bs, out, bs, 32, L, in_slope=0.1), 10) for _ in range(3))
    t1 = min(timed(lambda: wino.launch_conv_wino(layer, x, bs, out, bs, 32, L, in_slope=0.1, res=res, res_bs=bs), 10) for _ in range(3))
    rows.append(dict(C=C, ms_plain=round(t0, 4), ms_res=round(t1, 4)))
print(json.dumps(rows))
P
echo "exp 0 $(python /tmp/t.py 2>&1 | tail -1)" | tee -a $O/exp_k11.txt
for n in 8; do
  [ -f openvoice_amd/libopenvoice_amd_wexp$n.so ] || bash scripts/exp_wino.sh $n >/dev/null 2>&1
  echo "exp $n $(OPENVOICE_AMD_LIB=$PWD/openvoice_amd/libopenvoice_amd_wexp$n.so OPENVOICE_AMD_ALLOW_EXPERIMENT=1 python /tmp/t.py 2>&1 | tail -1)" | tee -a $O/exp_k11.txt
done
