#!/bin/bash
# Round 5, session 9: V1 TTS (configs[3]) with the generator's MRF stages on the split-precision kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s9; mkdir -p $O
for m in 0 6 3; do echo "== tts, split $m, skip_padding"; timeout 300 python tools/bench_tts.py --split-bf16x3 $m 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_split$m.json | cut -c1-500; done
for m in 0 6; do echo "== tts, split $m, full padding"; timeout 300 python tools/bench_tts.py --split-bf16x3 $m --full-padding 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_full_split$m.json | cut -c1-500; done
echo "== tts tests with split"; timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tts_parity_split.txt
import torch, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from openvoice_amd.models import SynthesizerTrn
from openvoice_amd.params import synthetic_tts_state_dict
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
from oracle import tts_oracle
sd = synthetic_tts_state_dict(CFG)
m = SynthesizerTrn(68, 513, n_speakers=10, **CFG); m.load_state_dict(sd, strict=True); m = m.to("cuda:0").eval()
gen = torch.Generator().manual_seed(0)
B, Tx = 3, 23
tok = torch.randint(0, 68, (B, Tx), generator=gen); lens = torch.tensor([23, 17, 9]); sid = torch.tensor([1, 4, 7])
nw = torch.randn(B, 2, Tx, generator=gen); nz = torch.randn(B, 192, 16 * Tx, generator=gen)
run = lambda: m.infer(tok.cuda(), lens.cuda(), sid=sid.cuda(), noise_scale=0.667, noise_scale_w=0.6, length_scale=1.0, noise_w=nw.cuda(), noise_z=nz.cuda())[0].cpu()
o32 = run()
m.engine().core.use_split_bf16x3(True); o6 = run()
m.engine().core.use_split_bf16x3(True, products=3); o3 = run()
print("tts infer: split6 vs fp32 kernels", (o6 - o32).abs().max().item(), "split3 vs fp32", (o3 - o32).abs().max().item(), "|o|max", o32.abs().max().item())
PY
