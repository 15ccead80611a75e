"""Golden vectors for the fused ResBlock kernels from the UNMODIFIED reference module (ORACLE tooling; build container
only):  python oracle/make_resblock_golden.py  ->  tests/golden/resblock1_*.pt

``ResBlock1(channels, kernel_size, dilation=(1, 3, 5)).forward(x, x_mask=None)`` (/root/reference/openvoice/modules.py
:221-309) with random weight-norm parameters on a seeded input: three (c1_d, c2) pairs in sequence -- exactly what three
``ov_resblock_pair_f32`` / ``ov_resblock_pair_bf16cl`` launches replace.  Recorded: x, the module's state dict, the
output, and the intermediate after the first pair (recomputed with the module's own convs)."""
import os
import sys
import warnings

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.make_golden import GOLDEN_DIR, import_reference  # noqa: E402

CASES = [dict(name="resblock1_c32_k3", C=32, K=3, B=2, L=516), dict(name="resblock1_c32_k11", C=32, K=11, B=1, L=700),
         dict(name="resblock1_c64_k7", C=64, K=7, B=1, L=388)]


def main():
    warnings.filterwarnings("ignore")
    import_reference()
    from openvoice import modules as ref_modules
    for case in CASES:
        gen = torch.Generator().manual_seed(len(case["name"]) + case["K"])
        block = ref_modules.ResBlock1(case["C"], case["K"], (1, 3, 5)).eval()
        with torch.no_grad():
            for name, prm in block.named_parameters():
                fan = case["C"] * case["K"]
                if name.endswith("weight_v"):
                    prm.copy_(torch.randn(prm.shape, generator=gen) * (0.5 if "convs2" in name else 1.0) * fan ** -0.5)
                elif name.endswith("weight_g"):
                    prm.copy_(0.5 + torch.rand(prm.shape, generator=gen))          # exercises weight-norm folding
                else:
                    prm.copy_(0.05 * torch.randn(prm.shape, generator=gen))
            for i in range(3):                     # weight_g scaled by ||v|| so activations stay O(1)
                for convs in (block.convs1, block.convs2):
                    v = convs[i].weight_v
                    convs[i].weight_g.mul_(v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1))
            x = torch.randn(case["B"], case["C"], case["L"], generator=gen)
            out = block(x)
            inter, cur = [], x
            for c1, c2 in zip(block.convs1, block.convs2):
                cur = c2(F.leaky_relu(c1(F.leaky_relu(cur, ref_modules.LRELU_SLOPE)), ref_modules.LRELU_SLOPE)) + cur
                inter.append(cur.clone())
            assert torch.equal(inter[-1], out)
        rec = dict(case=case, x=x, out=out, after_pair0=inter[0], state_dict={k: v.clone() for k, v in block.state_dict().items()})
        torch.save(rec, os.path.join(GOLDEN_DIR, case["name"] + ".pt"))
        print(case["name"], tuple(out.shape), f"|out|max {out.abs().max():.3f} |out - x|max {(out - x).abs().max():.3f}")


if __name__ == "__main__":
    main()
