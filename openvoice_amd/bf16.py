"""Host helpers of the bf16 channels-last generator kernels (csrc/conv1d_bf16.hip): weight packing and launch.
Activations are ``torch.bfloat16`` tensors of shape (B, L, C) -- channels contiguous."""
import ctypes

import torch

from . import _lib
from ._lib import ConvBf16Params


class PackedConvBf16:
    """One conv layer in bf16 kernel-ready form: B-fragment-ordered bf16 weights + fp32 bias."""

    def __init__(self, w_dense, bias, device, dil=1):
        w = w_dense.detach().to(torch.float32).cpu().contiguous()
        self.cout, self.cin, self.K = w.shape
        self.dil = dil
        n = _lib.call("ov_conv1d_bf16_pack_size", self.cout, self.cin, self.K)
        if n == 0:
            raise _lib.OvError(f"bf16 conv needs Cin % 32 == 0 (got {self.cin})")
        packed = torch.empty(n, dtype=torch.int16)
        _lib.call("ov_conv1d_bf16_pack", w, self.cout, self.cin, self.K, packed)
        self.w = packed.to(device)
        # the second-generation fused pair runs on 16x16x32 fragments: its own record order (ov_conv1d_bf16_pack16)
        self.w16 = None
        if self.cout == self.cin and _lib.call("ov_resblock_pair2_bf16_supported", self.cin, self.K, dil):
            packed16 = torch.empty(n, dtype=torch.int16)
            _lib.call("ov_conv1d_bf16_pack16", w, self.cout, self.cin, self.K, packed16)
            self.w16 = packed16.to(device)
        self.bias = None if bias is None else bias.detach().float().contiguous().to(device)


def launch_conv_bf16(layer, x, out, in_slope=1.0, scale=1.0, res=None, add=None, layout=0, out_slope=1.0, dbg=None):
    """out = (conv1d(lrelu(x, in_slope)) + bias [+ res] [+ add]) * scale on torch's current stream.
    x (B, L, Cin), out / res / add (B, L, Cout), all contiguous bfloat16."""
    B, L, cin = x.shape
    assert cin == layer.cin and out.shape == (B, L, layer.cout)
    for t in (x, out, res, add):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous())
    if _lib.use_torch_binding():
        _lib.torch_op("conv1d_bf16cl", x, layer.w, layer.bias, out, res, add, dbg,
                      [B, L, cin, layer.cout, layer.K, layer.dil, 0, 0, layout], [in_slope, scale, out_slope])
        return
    p = ConvBf16Params()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w, p.bias, p.out, p.res, p.add = vp(x), vp(layer.w), vp(layer.bias), vp(out), vp(res), vp(add)
    p.B, p.L, p.Cin, p.Cout, p.K, p.dil = B, L, cin, layer.cout, layer.K, layer.dil
    p.in_slope, p.scale, p.layout, p.out_slope = in_slope, scale, layout, out_slope
    p.dbg = vp(dbg)
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_conv1d_bf16cl(ctypes.byref(p), stream), "ov_conv1d_bf16cl")


def launch_pair_bf16(c1, c2, x, out, add=None, scale=1.0, slope=0.1, nwg=0, dbg=None):
    """One fused ResBlock1 iteration on bf16 channels-last tensors (``ov_resblock_pair_bf16cl``):
    out = bf16((c2(lrelu(bf16(c1(lrelu(x))))) + x [+ add]) * scale).  x / out / add (B, L, C) contiguous bfloat16;
    ``out`` must not alias ``x``."""
    B, L, C = x.shape
    assert c1.cin == c1.cout == c2.cin == c2.cout == C and c1.K == c2.K and c2.dil == 1 and out.shape == x.shape
    for t in (x, out, add):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous())
    if _lib.use_torch_binding():
        _lib.torch_op("resblock_pair_bf16cl", x, c1.w, c1.bias, c2.w, c2.bias, out, add, dbg,
                      [B, L, C, c1.K, c1.dil, nwg], [slope, scale])
        return
    p = _lib.RespairBf16Params()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w1, p.b1, p.w2, p.b2, p.out, p.add = vp(x), vp(c1.w), vp(c1.bias), vp(c2.w), vp(c2.bias), vp(out), vp(add)
    p.B, p.L, p.C, p.K, p.dil, p.nwg = B, L, C, c1.K, c1.dil, nwg
    p.slope, p.scale = slope, scale
    p.dbg = vp(dbg)
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_resblock_pair_bf16cl(ctypes.byref(p), stream), "ov_resblock_pair_bf16cl")


def launch_pair2_bf16(c1, c2, x, out, add=None, scale=1.0, slope=0.1, out_slope=1.0, nwg=0, dbg=None, exp_flags=0):
    """One fused ResBlock1 iteration of the matrix-bound stages (``ov_resblock_pair2_bf16cl``, C in {64, 128}) on
    bf16 channels-last tensors stored ACTIVATED: ``x`` = bf16(lrelu(x_raw, slope));
    out = bf16(lrelu((c2(bf16(lrelu(c1(x) + b1))) + b2 + x_raw [+ add]) * scale, out_slope)), ``add`` raw.
    x / out / add (B, L, C) contiguous bfloat16; ``out`` must not alias ``x``."""
    B, L, C = x.shape
    assert c1.cin == c1.cout == c2.cin == c2.cout == C and c1.K == c2.K and c2.dil == 1 and out.shape == x.shape
    for t in (x, out, add):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous())
    if c1.w16 is None or c2.w16 is None:
        raise _lib.OvError(f"ov_resblock_pair2_bf16cl: no 16x16x32 weight stream for C={C} K={c1.K} dil={c1.dil}")
    if _lib.use_torch_binding():
        _lib.torch_op("resblock_pair2_bf16cl", x, c1.w16, c1.bias, c2.w16, c2.bias, out, add, dbg,
                      [B, L, C, c1.K, c1.dil, nwg, exp_flags], [slope, scale, out_slope])
        return
    p = _lib.Respair2Bf16Params()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w1, p.b1, p.w2, p.b2, p.out, p.add = vp(x), vp(c1.w16), vp(c1.bias), vp(c2.w16), vp(c2.bias), vp(out), vp(add)
    p.B, p.L, p.C, p.K, p.dil, p.nwg = B, L, C, c1.K, c1.dil, nwg
    p.slope, p.scale, p.out_slope, p.exp_flags = slope, scale, out_slope, exp_flags
    p.dbg = vp(dbg)
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_resblock_pair2_bf16cl(ctypes.byref(p), stream), "ov_resblock_pair2_bf16cl")


_pair_supported = {}
_pair2_supported = {}


def pair2_bf16_supported(C, K, dil):
    key = (C, K, dil)
    if key not in _pair2_supported:
        _pair2_supported[key] = bool(_lib.call("ov_resblock_pair2_bf16_supported", C, K, dil))
    return _pair2_supported[key]


def pair_bf16_supported(C, K, dil):
    key = (C, K, dil)
    if key not in _pair_supported:
        _pair_supported[key] = bool(_lib.call("ov_resblock_pair_bf16_supported", C, K, dil))
    return _pair_supported[key]


def _launch(layer, x, out, L, in_slope=1.0, scale=1.0, res=None, add=None, phase_s=0, bias=None, bias_bstride=0,
            out_slope=1.0):
    B = x.shape[0]
    bias = layer.bias if bias is None else bias
    if _lib.use_torch_binding():
        _lib.torch_op("conv1d_bf16cl", x, layer.w, bias, out, res, add, None,
                      [B, L, layer.cin, layer.cout, layer.K, layer.dil, phase_s, bias_bstride, 0],
                      [in_slope, scale, out_slope])
        return
    p = ConvBf16Params()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    p.x, p.w, p.bias, p.out, p.res, p.add = vp(x), vp(layer.w), vp(bias), vp(out), vp(res), vp(add)
    p.B, p.L, p.Cin, p.Cout, p.K, p.dil = B, L, layer.cin, layer.cout, layer.K, layer.dil
    p.phase_s, p.bias_bstride, p.in_slope, p.scale, p.out_slope = phase_s, bias_bstride, in_slope, scale, out_slope
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().ov_conv1d_bf16cl(ctypes.byref(p), stream), "ov_conv1d_bf16cl")


def generator_alg_bytes(cfg, B, T, esize=2, z_channels=192, fused=True):
    """Algorithmic HBM bytes of one generator pass: every tensor pass of the launch sequence (conv_pre; per stage
    the ups read + write and the MRF -- per ResBlock pair 2 passes when it is one fused launch (read x, write out),
    5 as two launches (conv1 r + w, conv2 r + res + w) -- plus the 2 running-sum reads; conv_post)."""
    ch, L = cfg["upsample_initial_channel"], T
    total = B * T * (z_channels + ch) * esize
    kernels, dils = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    for u in cfg["upsample_rates"]:
        total += B * L * ch * esize
        ch //= 2
        L *= u
        tensor = B * L * ch * esize
        passes = 1                                                  # the ups write
        for j, (k, rd) in enumerate(zip(kernels, dils)):
            one = fused and all(pair_bf16_supported(ch, k, d) or pair2_bf16_supported(ch, k, d) for d in rd)
            passes += len(rd) * (2 if one else 5) + (1 if j > 0 else 0)
        total += tensor * passes
    return total + B * L * ch * esize + B * L * 4


class GeneratorBf16:
    """HiFi-GAN generator (reference: openvoice/models.py:272-291) with bf16 activations in HBM, channels-last,
    fp32 accumulation -- BASELINE.json configs[4].  Same load-time algebra as the fp32 engine (weight-norm folded,
    ConvTranspose as a 3-tap phase conv, MRF mean folded into the last conv's scale); the residual and the MRF
    running sum are added on the matrix pipe (identity rounds, csrc/conv1d_bf16.hip)."""

    def __init__(self, state_dict, model_cfg, device):
        from .engine import conv_transpose_as_conv
        from .params import effective_weight
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        cfg = dict(model_cfg.items()) if hasattr(model_cfg, "items") else dict(model_cfg)
        self.cfg, self.device = cfg, torch.device(device)
        dev = self.device
        self.conv_pre = PackedConvBf16(sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], dev)
        self.cond_w = sd["dec.cond.weight"][:, :, 0].contiguous().to(dev)
        self.cond_b = (sd["dec.cond.bias"] + sd["dec.conv_pre.bias"]).contiguous().to(dev)   # conv_pre bias folded in
        self.ups, self.resblocks = [], []
        ch = cfg["upsample_initial_channel"]
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            wc = conv_transpose_as_conv(effective_weight(sd, f"dec.ups.{i}"), u)      # rows co*u + ph
            co = ch // 2
            wc = wc.reshape(co, u, ch, 3).transpose(0, 1).reshape(u * co, ch, 3)       # rows ph*co + c
            bias = sd[f"dec.ups.{i}.bias"].repeat(u)
            self.ups.append(dict(conv=PackedConvBf16(wc, bias, dev), stride=u))
            ch = co
            stage = []
            for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
                rb = f"dec.resblocks.{i * len(cfg['resblock_kernel_sizes']) + j}"
                stage.append([(PackedConvBf16(effective_weight(sd, f"{rb}.convs1.{n}"), sd[f"{rb}.convs1.{n}.bias"], dev, dil=d),
                               PackedConvBf16(effective_weight(sd, f"{rb}.convs2.{n}"), sd[f"{rb}.convs2.{n}.bias"], dev, dil=1))
                              for n, d in enumerate(rd)])
            self.resblocks.append(stage)
        self.final_channels = ch
        self.post_w = sd["dec.conv_post.weight"][0].contiguous().to(dev)              # [C, 7] fp32
        self._ws = {}
        # ResBlock pairs of the HBM-bound stages as ONE launch each (csrc/conv1d_bf16_pair.hip): C = 32 every kernel
        # size, C = 64 K = 3 -- 1.7-2.0x faster than two launches on MI355X.  The fused kernel is bit-identical to two
        # PLAIN launches (t = bf16(lrelu(bf16(v)))); the unfused pairs below store t activated (out_slope: one rounding,
        # t = bf16(lrelu(v))), which can differ from that by one bf16 ulp of t on negative values -- fuse_pairs on / off
        # therefore agree within bf16 rounding, not bit for bit (tests/test_gpu_bf16_pair.py)
        self.fuse_pairs = True
        # Stages whose every ResBlock pair has a second-generation fused instance (csrc/conv1d_bf16_pair2.hip: C = 64 /
        # 128, every kernel size) keep their tensors ACTIVATED in HBM: the ConvTranspose stores lrelu(u), each pair
        # stores lrelu(x_n), the residual add inverts it in fp32, the chains' sums stay raw (round 4).  False = the
        # round-3 launch sequence (first-generation pairs where they exist, two launches elsewhere).
        self.act_hbm = True
        # independent ResBlock chains of a stage on this many HIP streams (1 = one stream, the serial order).  Measured
        # (round 3, batch 64): 3 streams 45.46 ms vs 45.86 ms on one -- a launch of 2 workgroups per CU owns the chip,
        # so kernels of different streams overlap only at their ramps and tails -- 0.9 %, not worth a default that makes
        # per-kernel profiles harder to read: off unless asked for
        self.chain_streams = 1
        self._streams = []

    def _side_streams(self, n):
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(self.device))
        return self._streams[:n]

    def _workspace(self, B, T):
        key = (B, T, self.chain_streams > 1)
        if key not in self._ws:
            self._ws.clear()
            ch, L, biggest = self.cfg["upsample_initial_channel"], T, 0
            for u in self.cfg["upsample_rates"]:
                ch //= 2
                L *= u
                biggest = max(biggest, ch * L)
            f = lambda n: torch.empty(n, dtype=torch.bfloat16, device=self.device)
            # stage input, ups output, running sum + (t1, ra) per concurrent chain: ONE pair for the default serial order
            # (chain_streams == 1: 5 buffers, 4.5 GB at batch 64), one per chain only when concurrency is on (9, 8.1 GB)
            nbuf = 3 + 2 * (max(1, len(self.cfg["resblock_kernel_sizes"])) if self.chain_streams > 1 else 1)
            self._ws[key] = dict(pre=f(B * T * self.cfg["upsample_initial_channel"]), dec=[f(B * biggest) for _ in range(nbuf)])
        return self._ws[key]

    @torch.no_grad()
    def decode(self, z, g):
        """``z`` [B, inter, T] fp32 (channels-first, as the flow produces it), ``g`` [B or 1, gin, 1] ->
        waveform [B, 1, T * prod(upsample_rates)] fp32."""
        from .engine import FINAL_LRELU_SLOPE, LRELU_SLOPE
        dev = self.device
        B, C, T = z.shape
        ws = self._workspace(B, T)
        x = z.to(dev, torch.float32).transpose(1, 2).to(torch.bfloat16).contiguous()        # [B, T, C] bf16
        g2 = g.to(dev, torch.float32).reshape(g.shape[0], -1)
        g2 = g2.contiguous()
        cond = torch.empty(g2.shape[0], self.cond_w.shape[0], dtype=torch.float32, device=dev)
        _lib.call("ov_linear_f32", g2, self.cond_w, self.cond_b, cond, g2.shape[0], self.cond_w.shape[0],
                  self.cond_w.shape[1])                                                      # dec.cond, T = 1 GEMV
        cond = cond.expand(B, -1).contiguous()                                               # [B, 512] fp32
        ch = self.cfg["upsample_initial_channel"]
        pre = ws["pre"][: B * T * ch].view(B, T, ch)
        # every tensor a ConvTranspose reads is stored ACTIVATED by its producer (conv_pre here, the MRF mean below):
        # its loaders then copy instead of unpacking / activating / re-packing every vector
        _launch(self.conv_pre, x, pre, T, bias=cond, bias_bstride=ch, out_slope=LRELU_SLOPE)
        cur_x, L, cur_act = pre, T, True                  # cur_act: cur_x holds lrelu(x) already
        free = list(ws["dec"])
        nk = len(self.cfg["resblock_kernel_sizes"])
        # The three ResBlocks of a stage (k = 3, 7, 11) are independent chains until the MRF sum.  In bf16 they bound
        # DIFFERENT resources -- the k = 3 convs HBM (3.4 TB/s at 33 % matrix-busy), the k = 11 convs the power-limited
        # matrix pipe (60 % busy at 1.5 TB/s) -- so with chain_streams > 1 they are issued on separate HIP streams;
        # the sum keeps its order (chain j's last launch waits for chain j - 1's), so the result is bit-identical to
        # the serial order.  Not under graph capture (the engine captures on one stream).
        concurrent = self.chain_streams > 1 and nk > 1 and not torch.cuda.is_current_stream_capturing()
        main = torch.cuda.current_stream(dev)
        side = self._side_streams(nk) if concurrent else None
        for i, up in enumerate(self.ups):
            s = up["stride"]
            cin, ch = ch, ch // 2
            u = free.pop()[: B * L * s * ch].view(B, L * s, ch)
            act = (self.act_hbm and self.fuse_pairs and
                   all(pair2_bf16_supported(ch, c1.K, c1.dil) for pairs in self.resblocks[i] for c1, _ in pairs))
            _launch(up["conv"], cur_x, u, L, in_slope=1.0 if cur_act else LRELU_SLOPE, phase_s=s,
                    out_slope=LRELU_SLOPE if act else 1.0)
            L *= s
            acc = free.pop()[: B * L * ch].view(B, L, ch)
            nchains = nk if concurrent else 1
            scratch = [tuple(free.pop()[: B * L * ch].view(B, L, ch) for _ in range(2)) for _ in range(nchains)]
            cur = [u] * nk
            fused = [self.fuse_pairs and all(pair_bf16_supported(ch, c1.K, c1.dil) for c1, _ in pairs)
                     for pairs in self.resblocks[i]]
            # can the launch that writes the MRF mean apply an activation?  (the first-generation fused pair cannot)
            mean_act = i + 1 < len(self.ups) and (act or not fused[nk - 1])
            npairs = len(self.resblocks[i][0])

            def pair(j, n):
                c1, c2 = self.resblocks[i][j][n]
                t1, ra = scratch[j if concurrent else 0]
                last = n == npairs - 1
                add = acc if (last and j > 0) else None
                scale = 1.0 / nk if (last and j == nk - 1) else 1.0
                # the MRF mean feeds the next stage's ConvTranspose, which wants it activated; the last stage's feeds
                # conv_post (its own slope, applied there); the chains' partial sums stay raw
                mean_slope = LRELU_SLOPE if (last and j == nk - 1 and mean_act) else 1.0
                if act:            # activated tensors between the launches
                    dst = acc if last else (t1 if cur[j] is ra else ra)
                    launch_pair2_bf16(c1, c2, cur[j], dst, add=add, scale=scale, slope=LRELU_SLOPE,
                                      out_slope=mean_slope if last else LRELU_SLOPE)
                elif fused[j]:       # one launch per pair, intermediate in LDS; out must not alias x: ra / t1 ping-pong
                    dst = acc if last else (t1 if cur[j] is ra else ra)
                    launch_pair_bf16(c1, c2, cur[j], dst, add=add, scale=scale, slope=LRELU_SLOPE)
                else:
                    # t1 is consumed by c2 only, which activates it: store it activated (one rounding instead of
                    # two) and let c2's loaders copy it as is
                    _launch(c1, cur[j], t1, L, in_slope=LRELU_SLOPE, out_slope=LRELU_SLOPE)
                    dst = acc if last else ra
                    _launch(c2, t1, dst, L, in_slope=1.0, res=cur[j], add=add, scale=scale, out_slope=mean_slope)
                cur[j] = dst

            if not concurrent:
                for j in range(nk):
                    for n in range(npairs):
                        pair(j, n)
            else:
                fork = torch.cuda.Event()
                fork.record(main)
                done = [None] * nk
                for n in range(npairs):                 # round-robin over the chains: every queue has work early
                    for j in range(nk):
                        with torch.cuda.stream(side[j]):
                            if n == 0:
                                side[j].wait_event(fork)
                            if n == npairs - 1 and j > 0:
                                side[j].wait_event(done[j - 1])      # the running sum is accumulated in chain order
                            pair(j, n)
                            if n == npairs - 1:
                                done[j] = torch.cuda.Event()
                                done[j].record(side[j])
                main.wait_event(done[nk - 1])
            # every scratch buffer except the one holding this stage's output is free again
            free = [buf for buf in ws["dec"] if buf.data_ptr() != acc.data_ptr()]
            cur_x, cur_act = acc, mean_act
        o = torch.empty(B, 1, L, dtype=torch.float32, device=dev)
        _lib.call("ov_conv_post_tanh_bf16", cur_x, self.post_w, o, B, ch, L, self.post_w.shape[1], FINAL_LRELU_SLOPE)
        return o
