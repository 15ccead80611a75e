"""The fused WaveNet layer (``ov_wn_layer_f32``, openvoice_amd/csrc/wn_layer.hip) against plain fp32 PyTorch on CPU
(reference: openvoice/modules.py:192-209, commons.py:100-107), through the C ABI.

Tolerance: 2e-5 of the output scale -- fp32 MFMA is an exact fmaf chain (summation order differs from the CPU's);
the gate uses v_exp_f32 / v_rcp_f32 (1 ulp) instead of libm tanhf / expf.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openvoice_amd import _lib  # noqa: E402
from openvoice_amd.engine import launch_wn_layer, wn_fused_row_order, wn_pack  # noqa: E402

DEV = "cuda:0"
H, K = 192, 5


def _rand(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=gen)


def _close(got, ref, rel=2e-5, what=""):
    err = (got.cpu() - ref).abs().max().item()
    bound = rel * max(1.0, ref.abs().max().item())
    assert err <= bound, f"{what}: max-abs err {err:.3e} > {bound:.3e}"


def _layer(seed, last=False):
    w_in, b_in = _rand(2 * H, H, K, seed=seed, scale=(K * H) ** -0.5), _rand(2 * H, seed=seed + 1, scale=0.1)
    rows = H if last else 2 * H
    w_rs, b_rs = _rand(rows, H, 1, seed=seed + 2, scale=H ** -0.5), _rand(rows, seed=seed + 3, scale=0.1)
    return w_in, b_in, w_rs, b_rs


def _packed(w_in, b_in, w_rs, b_rs):
    order = wn_fused_row_order(H)
    if w_rs.shape[0] == H:
        w_rs = torch.cat([torch.zeros_like(w_rs), w_rs])
        b_rs = torch.cat([torch.zeros_like(b_rs), b_rs])
    return dict(hidden=H, K=K, w_in=wn_pack(w_in[order], DEV), b_in=b_in[order].contiguous().to(DEV),
                w_rs=wn_pack(w_rs, DEV), b_rs=b_rs.contiguous().to(DEV))


def _reference(x, g, mask, skip0, w_in, b_in, w_rs, b_rs, first, last):
    x_in = F.conv1d(x, w_in, b_in, padding=(K - 1) // 2) + g[:, :, None]
    acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
    rs = F.conv1d(acts, w_rs, b_rs)
    if last:
        return None, (0 if first else skip0) + rs
    return (x + rs[:, :H]) * mask[:, None], (0 if first else skip0) + rs[:, H:]


def _run(B, T, lengths, width, first=False, last=False, per_item_cond=True, seed=10, row_split=1, with_acts=False):
    ld = (T + 3) // 4 * 4
    x, skip0 = _rand(B, H, T, seed=seed), _rand(B, H, T, seed=seed + 1)
    g = _rand(B if per_item_cond else 1, 2 * H, seed=seed + 2, scale=0.3)
    mask = (torch.arange(T)[None, :] < torch.tensor(lengths)[:, None]).float()
    x = x * mask[:, None]                 # the WN input is always masked (modules.py:207, models.py:216)
    lw = _layer(seed + 3, last)
    x_ref, skip_ref = _reference(x, g.expand(B, -1), mask, skip0, *lw, first, last)
    pad = lambda t: F.pad(t, (0, ld - T)).contiguous().to(DEV)
    xd, skipd, maskd = pad(x), pad(skip0 if not first else torch.full_like(skip0, float("nan"))), pad(mask)
    outd = torch.full((B, H, ld), float("nan"), device=DEV)
    gd = g[:, wn_fused_row_order(H)].contiguous().to(DEV)
    acts = torch.full((B, H, ld), float("nan"), device=DEV) if (with_acts or row_split == 3) else None
    launch_wn_layer(_packed(*lw), xd, outd, skipd, maskd, B, T, ld, cond=gd, cond_bs=2 * H if per_item_cond else 0,
                    first=first, last=last, width=width, mask_bs=ld, acts=acts, row_split=row_split)
    torch.cuda.synchronize()
    if not last:
        _close(outd[:, :, :T], x_ref, what=f"h' (T={T}, width={width})")
        assert torch.isnan(outd[:, :, T:]).all(), "columns >= T must not be written"
    _close(skipd[:, :, :T], skip_ref, what=f"skip (T={T}, width={width})")
    assert (skipd[:, :, T:].cpu() == 0).all()
    return outd, skipd, acts


@pytest.mark.parametrize("width", [16, 32, 48, 64, 80, 96, 112, 128])
def test_every_tile_width(width):
    """Each instantiated tile width (16 .. 128 columns), three tiles and a ragged last one, ragged utterance lengths."""
    T = 2 * width + max(1, width // 2 - 3)
    _run(2, T, [T, max(1, T - 5)], width)


@pytest.mark.parametrize("T", [1, 3, 17, 200, 861, 1000])
def test_launcher_chosen_width(T):
    """width = 0: the launcher's rule; T = 1 (one column), T not a multiple of 4, the benchmark length and beyond."""
    _run(3, T, [T, max(1, T - 5), max(1, T // 2)], 0)


def test_first_and_last_layer_forms():
    """Layer 0 initialises the skip accumulator (NaN-poisoned here); the last layer has H skip rows only and does not
    write h' (modules.py:170-173, :203-207)."""
    _run(2, 150, [150, 99], 0, first=True)
    _run(2, 150, [150, 99], 0, last=True)
    _run(2, 150, [150, 99], 64, first=True, last=True)


def test_broadcast_conditioning_row():
    _run(3, 77, [77, 50, 1], 0, per_item_cond=False)


@pytest.mark.parametrize("B,T", [(1, 861), (2, 861), (1, 1), (1, 17), (3, 64), (1, 65), (2, 127), (1, 1000)])
@pytest.mark.parametrize("form", ["middle", "first", "last"])
def test_row_split_pair_is_bit_identical_to_the_fused_launch(B, T, form):
    """The two-launch row-split form (gate rows -> ``acts`` scratch, then res/skip rows; three workgroups per 16-column
    tile, one 16-row fragment per wave) against plain PyTorch AND bit for bit against the fused launch on the same tile
    width: same weight records, same summation order per output element.  Ragged lengths, T = 1, T not a multiple of 4 /
    16, the first layer (skip initialised over NaN) and the last (skip rows only, h' untouched)."""
    lengths = [max(1, T - 7 * i) for i in range(B)]
    kw = dict(first=form == "first", last=form == "last")
    o3, s3, acts = _run(B, T, lengths, 0, row_split=3, **kw)
    o1, s1, _ = _run(B, T, lengths, 16, row_split=1, **kw)
    assert torch.equal(s3[:, :, :T], s1[:, :, :T])
    if form != "last":
        assert torch.equal(o3[:, :, :T], o1[:, :, :T])
    assert torch.isfinite(acts[:, :, :T]).all() and torch.isnan(acts[:, :, T:]).all()   # scratch: columns < T only


def test_row_split_with_a_broadcast_conditioning_row():
    """What a batch-1 ``convert`` issues: one conditioning row for the whole batch (cond_bstride = 0), split pair vs fused."""
    o3, s3, _ = _run(2, 333, [333, 200], 0, per_item_cond=False, row_split=3)
    o1, s1, _ = _run(2, 333, [333, 200], 16, per_item_cond=False, row_split=1)
    assert torch.equal(o3[:, :, :333], o1[:, :, :333]) and torch.equal(s3[:, :, :333], s1[:, :, :333])


def test_row_split_is_the_launchers_choice_for_one_or_two_utterances_only():
    """row_split = 0 with a scratch: one or two utterances at frame rate take the split pair (the scratch is written),
    batches whose 16-column tiles fill the compute units without it stay fused (the scratch is untouched); without a
    scratch, or with a forced tile width, always fused.  Forcing the split without a scratch is an argument error."""
    for B, T, expect in ((1, 861, True), (1, 1300, True), (2, 861, True), (3, 861, False), (32, 861, False)):
        _, _, acts = _run(B, T, [T] * B, 0, row_split=0, with_acts=True)
        assert bool(torch.isfinite(acts[:, :, :T]).all()) == expect and bool(torch.isnan(acts).all()) == (not expect)
    _, _, acts = _run(1, 200, [200], 32, row_split=0, with_acts=True)
    assert torch.isnan(acts).all()
    layer = _packed(*_layer(1))
    x = torch.zeros(1, H, 16, device=DEV)
    mask = torch.ones(1, 16, device=DEV)
    with pytest.raises(_lib.OvError, match="BADARG"):
        launch_wn_layer(layer, x, torch.zeros_like(x), torch.zeros_like(x), mask, 1, 16, 16, row_split=3)
    with pytest.raises(_lib.OvError, match="BADARG"):
        launch_wn_layer(layer, x, torch.zeros_like(x), torch.zeros_like(x), mask, 1, 16, 16, row_split=2)
    with pytest.raises(_lib.OvError, match="BADARG"):
        launch_wn_layer(layer, x, torch.zeros_like(x), torch.zeros_like(x), mask, 1, 16, 16, acts=x, row_split=3)
    with pytest.raises(_lib.OvError, match="BADARG"):                     # phase timers exist for the fused launch only
        launch_wn_layer(layer, x, torch.zeros_like(x), torch.zeros_like(x), mask, 1, 16, 16, acts=torch.zeros_like(x),
                        row_split=3, dbg=torch.zeros(8 * 8, dtype=torch.int64, device=DEV))


def test_tile_rule():
    lib = _lib.load()
    assert lib.ov_wn_layer_tile(32, 861, 0) == 112      # 8 tiles per utterance = 256 tiles on 256 CUs
    assert lib.ov_wn_layer_tile(1, 861, 0) == 16        # 54 tiles: as many CUs as possible
    assert lib.ov_wn_layer_tile(4, 100, 24) == 0 and lib.ov_wn_layer_tile(4, 100, 144) == 0
    assert lib.ov_wn_layer_tile(4, 100, 96) == 96


def test_argument_checks():
    B, T = 1, 16
    layer = _packed(*_layer(1))
    x = torch.zeros(B, H, T, device=DEV)
    mask = torch.ones(B, T, device=DEV)
    with pytest.raises(_lib.OvError, match="BADARG|bad"):
        launch_wn_layer(layer, x, x, torch.zeros_like(x), mask, B, T, T)          # out aliases x
    bad = dict(layer, hidden=128)
    with pytest.raises(_lib.OvError):
        launch_wn_layer(bad, x, torch.zeros_like(x), torch.zeros_like(x), mask, B, T, T)


def test_mask_rows_must_be_vector_aligned():
    """The loaders read the mask as 16-byte vectors: a mask whose rows are T (not a multiple of 4) floats apart, or that
    starts off a 16-byte boundary, is refused (OV_E_ALIGN) instead of being read misaligned / past its last row."""
    B, T, ld = 2, 18, 20
    layer = _packed(*_layer(1))
    x = torch.zeros(B, H, ld, device=DEV)
    out, skip = torch.zeros_like(x), torch.zeros_like(x)
    tight = torch.ones(B, T, device=DEV)                      # rows 18 floats apart
    with pytest.raises(_lib.OvError, match="ALIGN"):
        launch_wn_layer(layer, x, out, skip, tight, B, T, ld, mask_bs=T)
    shifted = torch.ones(B * ld + 4, device=DEV)[1:]          # 4-byte aligned only
    with pytest.raises(_lib.OvError, match="ALIGN"):
        launch_wn_layer(layer, x, out, skip, shifted, B, T, ld, mask_bs=ld)
    short = torch.ones(B, 16, device=DEV)                     # aligned rows that do not own the last vector of T = 18
    with pytest.raises(_lib.OvError, match="BADARG"):
        launch_wn_layer(layer, x, out, skip, short, B, T, ld, mask_bs=16)
    launch_wn_layer(layer, x, out, skip, torch.ones(B, ld, device=DEV), B, T, ld, mask_bs=ld)     # the valid form
    torch.cuda.synchronize()
