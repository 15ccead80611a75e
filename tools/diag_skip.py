import sys, torch
sys.path.insert(0, '/root/repo')
from openvoice_amd.models import SynthesizerTrn
from openvoice_amd.params import synthetic_state_dict
from openvoice_amd.utils import CONVERTER_MODEL_CONFIG as CFG
DEV='cuda:0'
sd = synthetic_state_dict(CFG, 513, seed=1234)
m = SynthesizerTrn(0, 513, n_speakers=0, zero_g=True, **CFG); m.load_state_dict(sd, strict=True); m = m.to(DEV).eval()
B, T = 4, 300
gen = torch.Generator().manual_seed(31)
spec = (torch.rand(B, 513, T, generator=gen) * torch.linspace(3, 0.05, 513)[None, :, None]).to(DEV)
lengths = torch.tensor([300, 171, 40, 1], device=DEV)
g1, g2 = (0.3 * torch.randn(1, 256, 1, generator=gen)).to(DEV), (0.3 * torch.randn(B, 256, 1, generator=gen)).to(DEV)
noise = torch.randn(B, 192, T, generator=gen).to(DEV)
eng = m.engine()
def run(fuse, skip):
    eng.fuse_pairs = fuse
    return m.voice_conversion(spec, lengths, g1, g2, tau=0.3, noise=noise, skip_padding=skip)[0].clone()
a = run(True, False); b = run(False, False); c = run(False, True); d = run(True, True)
torch.cuda.synchronize()
print("fused vs unfused (no skip) equal:", torch.equal(a, b), (a-b).abs().max().item())
for i, n in enumerate(lengths.tolist()):
    print(i, n, "unfused skip vs unfused full:", torch.equal(c[i,:,:256*n], b[i,:,:256*n]), (c[i,:,:256*n]-b[i,:,:256*n]).abs().max().item() if n else 0,
          "| skip vs fused full:", torch.equal(d[i,:,:256*n], a[i,:,:256*n]))
    if not torch.equal(c[i,:,:256*n], b[i,:,:256*n]):
        diff = (c[i,0,:256*n]-b[i,0,:256*n]).abs()
        nz = diff.nonzero()
        print("   first/last differing sample", nz.min().item(), nz.max().item(), "count", nz.numel())
