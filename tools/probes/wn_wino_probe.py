"""Probe (measurement only): what would a Winograd-domain gate conv of the WaveNet layer cost?  Times the existing K = 3 and
K = 7 Winograd instances (6 and 16 products per tile) on the layer's shape (192 -> 384 rows, 32 x 864 frames) with one and two
fragments per wave, interpolates K = 5 (11 products per tile: (w0 w1 w2)(w3 w4 0)), and adds the 1x1 res/skip conv as its own
direct launch -- against the fused direct layer (ov_wn_layer_f32: ~195 us).   python tools/probes/wn_wino_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openvoice_amd import wino  # noqa: E402
from openvoice_amd.engine import PackedConv, launch_conv  # noqa: E402
from tools.bench_split3 import ramp, timed  # noqa: E402

DEV = "cuda:0"
H, B, L = 192, 32, 864
gen = torch.Generator().manual_seed(1)
x = torch.randn(B, H, L, generator=gen).to(DEV)
out = torch.empty(B, 2 * H, L, device=DEV)
res = {}
ramp()
for K in (3, 7):
    w = torch.randn(2 * H, H, K, generator=gen) * (H * K) ** -0.5
    layer = wino.PackedConvWino(w, torch.zeros(2 * H), DEV)
    for frags in (1, 2):
        run = lambda: wino.launch_conv_wino(layer, x, H * L, out, 2 * H * L, B, L, frags=frags)
        run()
        torch.cuda.synchronize()
        res[f"wino_k{K}_frags{frags}_us"] = min(timed(run, 50) for _ in range(3)) * 1e3
acts = torch.randn(B, H, L, generator=gen).to(DEV)
w1 = torch.randn(2 * H, H, 1, generator=gen) * H ** -0.5
l1 = PackedConv(w1, torch.zeros(2 * H), DEV, K=1)
run = lambda: launch_conv(l1, acts, 0, H * L, out, 0, 2 * H * L, B, L)
run()
torch.cuda.synchronize()
res["direct_1x1_us"] = min(timed(run, 50) for _ in range(3)) * 1e3
w5 = torch.randn(2 * H, H, 5, generator=gen) * (H * 5) ** -0.5
l5 = PackedConv(w5, torch.zeros(2 * H), DEV, K=5)
run = lambda: launch_conv(l5, x, 0, H * L, out, 0, 2 * H * L, B, L)
run()
torch.cuda.synchronize()
res["direct_k5_us"] = min(timed(run, 50) for _ in range(3)) * 1e3
for frags in (1, 2):
    t3, t7 = res[f"wino_k3_frags{frags}_us"], res[f"wino_k7_frags{frags}_us"]
    res[f"wino_k5_interpolated_frags{frags}_us"] = t3 + (t7 - t3) * (11 - 6) / (16 - 6)
print(json.dumps(res))
