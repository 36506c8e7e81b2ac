"""GPU parity of the Stable-Diffusion feature tower: primitives against plain torch fp32, the composed
tower against the reference-generated golden (tests/golden/sd_tiny.npz) with the bf16 CPU oracle as the error yardstick."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(__file__))
from test_oracle_golden import SD_TAGS, load_sd_case  # noqa: E402

from law_of_vision_representation_in_mllms_amd import _lib, engine, sd_engine as SE  # noqa: E402
from law_of_vision_representation_in_mllms_amd import sd_weights as SW  # noqa: E402
from oracle import diffusion as OD  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).norm() / want.norm().clamp_min(1e-12)).item()


def tokens(x):                      # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def untokens(y, B, H, W):           # [B*H*W, C] -> [B,C,H,W]
    return y.view(B, H, W, -1).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ primitives
@pytest.mark.parametrize("B,HW,C,groups,silu", [(2, 64, 64, 32, True), (3, 100, 320, 32, False), (1, 2304, 640, 32, True),
                                                (2, 37, 192, 32, True), (1, 9216, 128, 32, True)])
def test_groupnorm(B, HW, C, groups, silu):
    g = torch.Generator().manual_seed(HW + C)
    x = bf(torch.randn(B, HW, C, generator=g) * 2 + 0.7)
    gam, bet = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    want = F.group_norm(x.float().permute(0, 2, 1), groups, gam, bet, 1e-5)
    want = (F.silu(want) if silu else want).permute(0, 2, 1)
    got = SE.groupnorm(x.view(B * HW, C).to(DEV), gam.to(DEV), bet.to(DEV), B, groups, 1e-5, silu).view(B, HW, C)
    assert (got.float().cpu() - want).abs().max().item() < 4e-2
    assert rel_err(got, want) < 5e-3


@pytest.mark.parametrize("stride,pad_mode,up", [(1, 0, False), (2, 0, False), (2, 1, False), (1, 0, True)])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 8, 64, 128), (1, 13, 10, 8, 64), (1, 24, 24, 320, 320)])
def test_conv3x3_as_gather_plus_gemm(B, H, W, Ci, Co, stride, pad_mode, up):
    g = torch.Generator().manual_seed(H * 100 + Ci + stride)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.1
    xin = x.float()
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if pad_mode == 1:
        want = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), b, stride=stride, padding=0)
    else:
        want = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    wp = w.float().permute(0, 2, 3, 1).reshape(Co, 9 * Ci)
    kp = (9 * Ci + 63) // 64 * 64
    wpk = torch.zeros(Co, kp)
    wpk[:, : 9 * Ci] = wp
    cols, Ho, Wo = SE.im2col3x3(tokens(x).to(DEV), B, H, W, kp, stride, pad_mode, up)
    assert (Ho, Wo) == tuple(want.shape[2:])
    got = engine.gemm(cols, bf(wpk).to(DEV), b.to(DEV), _lib.EPI_F32)
    got = untokens(got, B, Ho, Wo)
    assert (got.cpu() - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("stride,pad_mode,up", [(1, 0, False), (2, 0, False), (2, 1, False), (1, 0, True)])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 8, 64, 128), (1, 13, 10, 128, 64), (1, 24, 24, 320 + 64, 320), (3, 12, 12, 1280, 1280),
                                         (1, 96, 96, 128, 128)])
def test_conv3x3_implicit_gemm(B, H, W, Ci, Co, stride, pad_mode, up):
    """The gather folded into the GEMM's A-operand DMA: vs F.conv2d, vs the explicit im2col path, all epilogues, split-K shapes."""
    g = torch.Generator().manual_seed(H * 100 + Ci + stride)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.1
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    if pad_mode == 1:
        want = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), b, stride=stride, padding=0)
    else:
        want = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    wp = bf(w.float().permute(0, 2, 3, 1).reshape(Co, 9 * Ci)).to(DEV)
    SE.ensure_scratch(torch.device(DEV))
    xt = tokens(x).to(DEV)
    got, Ho, Wo = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, up, _lib.EPI_F32)
    assert (Ho, Wo) == tuple(want.shape[2:])
    assert (untokens(got, B, Ho, Wo).cpu() - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())
    cols, _, _ = SE.im2col3x3(xt, B, H, W, 9 * Ci, stride, pad_mode, up)
    explicit = engine.gemm(cols, wp, b.to(DEV), _lib.EPI_F32)
    assert (got - explicit).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())
    res = bf(torch.randn(B * Ho * Wo, Co, generator=g)).to(DEV)
    r, _, _ = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, up, _lib.EPI_RESID, resid=res)
    assert rel_err(r, res.float().cpu() + tokens(want)) < 5e-3
    o, _, _ = SE.conv3x3(xt, B, H, W, wp, None, stride, pad_mode, up, _lib.EPI_BIAS)
    assert rel_err(o, tokens(want - b[None, :, None, None])) < 5e-3


@pytest.mark.parametrize("B,H,W,Ci,Co,stride,pad_mode", [(4, 192, 192, 64, 256, 1, 0),      # 576 tiles: 2 full rounds + a 64-tile remainder (tail launch, row offset)
                                                        (9, 128, 120, 128, 256, 1, 0),     # 540 tiles, W % 8 == 0 but rows wrap inside a lane's four pieces
                                                        (8, 301, 223, 64, 256, 2, 1),      # stride 2, (0,1,0,1) padding, odd sizes, M % 256 != 0
                                                        (12, 150, 147, 64, 512, 2, 0),     # two column tiles, stride 2 symmetric
                                                        (2, 260, 259, 128, 256, 1, 0)])
def test_conv3x3_in_the_256_kernel(B, H, W, Ci, Co, stride, pad_mode):
    """Shapes with >= 2 rounds of 256x256 tiles take the persistent ping-pong kernel (gemm_bf16_256q<EPI, false, CONV>): equal to the 128x128
    kernel's result BITWISE (same K order, same MFMA shape, fp32 accumulation) and to F.conv2d within the bf16 tolerance; BIAS and RESID."""
    g = torch.Generator().manual_seed(H + W + Ci)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.1
    if pad_mode == 1:
        want = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b, stride=stride, padding=0)
    else:
        want = F.conv2d(x.float(), w.float(), b, stride=stride, padding=1)
    wp = bf(w.float().permute(0, 2, 3, 1).reshape(Co, 9 * Ci)).to(DEV)
    xt = tokens(x).to(DEV)
    lib = _lib.load()
    got, Ho, Wo = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_BIAS)
    assert (Ho, Wo) == tuple(want.shape[2:]) and (B * Ho * Wo + 255) // 256 * (Co // 256) >= 512
    res = bf(torch.randn(B * Ho * Wo, Co, generator=g)).to(DEV)
    got_r, _, _ = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_RESID, resid=res)
    lib.visrep_set_gemm_variant(1)
    try:
        small, _, _ = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_BIAS)
        small_r, _, _ = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_RESID, resid=res)
    finally:
        lib.visrep_set_gemm_variant(5)
    # the rows of the whole tile rounds are bitwise the 128x128 kernel's; the remainder rows (a 128x128 tail launch, split-K when K >= 1024:
    # another summation order) agree to rounding
    ntn, ntm = Co // 256, (B * Ho * Wo + 255) // 256
    head = (ntm * ntn) // 256 * 256 // ntn * 256
    assert head >= 2 * 256 * 256 // ntn
    assert torch.equal(got[:head], small[:head]) and torch.equal(got_r[:head], small_r[:head])
    assert (got.float() - small.float()).abs().max().item() < 2e-2 and (got_r.float() - small_r.float()).abs().max().item() < 4e-2
    assert rel_err(got, tokens(want)) < 5e-3
    assert rel_err(got_r, res.float().cpu() + tokens(want)) < 5e-3


@pytest.mark.parametrize("B,H,W,Ci,Co,stride,pad_mode", [(2, 32, 32, 64, 128, 1, 0),        # 128-channel layers: 4 channels per group
                                                        (3, 48, 32, 128, 256, 1, 0),       # 8 per group
                                                        (2, 33, 65, 64, 512, 2, 1),        # 16 per group, stride 2 with (0,1,0,1) padding -> 16 x 32
                                                        (9, 128, 120, 64, 128, 1, 0),      # 1080 tiles of 128 x 128
                                                        (2, 64, 64, 128, 64, 1, 0)])       # one 64-column wave tile per block row (N edge)
def test_conv3x3_emits_groupnorm_partials(B, H, W, Ci, Co, stride, pad_mode):
    """visrep_conv3x3_bf16_gn: the convolution result is the plain call's (bitwise when both take the same route), and GroupNorm from its partial sums equals GroupNorm of the
    stored tensor (statistics of the fp32 outputs vs of their bf16 roundings: far inside the bf16 output tolerance); BIAS and RESID."""
    g = torch.Generator().manual_seed(H + W + Co)
    G = 32 if Co >= 128 else 16
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.5 + 0.3
    wp = bf(w.float().permute(0, 2, 3, 1).reshape(Co, 9 * Ci)).to(DEV)
    xt = tokens(x).to(DEV)
    gam, bet = (torch.randn(Co, generator=g) * 0.3 + 1).to(DEV), (torch.randn(Co, generator=g) * 0.2).to(DEV)
    plain, Ho, Wo = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_BIAS)
    assert SE.conv_gn_supported(B, Ho * Wo, Co, G) and not SE.conv_gn_supported(B, Ho * Wo + 64, Co, 32) and not SE.conv_gn_supported(B, Ho * Wo, 320, 32)
    # the 256x256 kernel's shapes: partial sums for EPI_BIAS (round 5); a residual convolution keeps the separate statistics pass (round 6: opt-in only)
    assert SE.conv_gn_supported(16, 384 * 384, 256, 32) == (os.environ.get("VISREP_GN_RESID_256", "0") == "1") and SE.conv_gn_supported(16, 384 * 384, 256, 32, _lib.EPI_BIAS)
    out, _, _, part = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_BIAS, gn_groups=G)
    # same kernel, same order of accumulation - unless the plain call splits K (few tiles, K >= 1024: partial sums in another order)
    same = (lambda a, c: torch.equal(a, c)) if 9 * Ci < 1024 else (lambda a, c: (a.float() - c.float()).abs().max().item() < 2e-2)
    assert same(out, plain)
    for silu in (True, False):
        want = SE.groupnorm(plain, gam, bet, B, G, 1e-6, silu)
        got = SE.groupnorm_from_partials(out, gam, bet, B, G, 1e-6, silu, part)
        assert (got.float() - want.float()).abs().max().item() < 4e-2 and rel_err(got, want.float().cpu()) < 3e-3
    res = bf(torch.randn(B * Ho * Wo, Co, generator=g)).to(DEV)
    plain_r, _, _ = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_RESID, resid=res)
    out_r, _, _, part_r = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), stride, pad_mode, False, _lib.EPI_RESID, resid=res, gn_groups=G)
    assert same(out_r, plain_r)
    want = SE.groupnorm(plain_r, gam, bet, B, G, 1e-6, True)
    got = SE.groupnorm_from_partials(out_r, gam, bet, B, G, 1e-6, True, part_r)
    assert (got.float() - want.float()).abs().max().item() < 4e-2 and rel_err(got, want.float().cpu()) < 3e-3
    # against torch's GroupNorm of the fp32 convolution
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b, stride=stride) if pad_mode == 1 else F.conv2d(x.float(), w.float(), b, stride=stride, padding=1)
    ref = F.silu(F.group_norm(ref, G, gam.cpu(), bet.cpu(), 1e-6))
    got = SE.groupnorm_from_partials(out, gam, bet, B, G, 1e-6, True, part)
    assert rel_err(got, tokens(ref)) < 1e-2


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 48, 80), (1, 16, 16), (2, 256, 272)])
def test_conv3x3_c8_equals_conv2d_and_the_im2col_route(B, H, W):
    """visrep_conv3x3_c8_bf16 (the VAE encoder's conv_in: 3 channels as 8-channel tokens -> 128, no im2col): against F.conv2d in fp32, against the
    im2col + GEMM route it replaces (same products; the order of the fp32 sum inside an MFMA differs), zero padding at all four borders, and the
    GroupNorm partial sums of its output against float64 statistics of the stored tensor."""
    g = torch.Generator().manual_seed(H * 7 + W)
    x = bf(torch.randn(B, 3, H, W, generator=g))
    w = bf(torch.randn(128, 3, 3, 3, generator=g) / math.sqrt(27))
    b = torch.randn(128, generator=g) * 0.3
    want = F.conv2d(x.float(), w.float(), b, padding=1)
    Wp = torch.zeros(128, 9, 8)
    Wp[:, :, :3] = w.float().permute(0, 2, 3, 1).reshape(128, 9, 3)                  # K order (ky, kx, c8): sd_engine.SdEngine._conv3
    wp = torch.zeros(128, 128)
    wp[:, :72] = Wp.reshape(128, 72)
    wp = bf(wp).to(DEV)
    xt = SE.nchw_to_tokens(x.to(DEV), 8)
    assert SE.conv_c8_supported(B, H, W, 128) and not SE.conv_c8_supported(B, H, W + 8, 128) and not SE.conv_c8_supported(B, H, W, 256)
    _lib.routes(reset=True)
    got = SE.conv3x3_c8(xt, B, H, W, wp, b.to(DEV))
    assert _lib.routes()["conv_c8"] == 1
    cols, Ho, Wo = SE.im2col3x3(xt, B, H, W, 128)
    old = SE.gemm(cols, wp, b.to(DEV))
    assert (Ho, Wo) == (H, W) and got.shape == old.shape == (B * H * W, 128)
    assert (got.float() - old.float()).abs().max().item() < 2e-2 and rel_err(got, tokens(want)) < 5e-3
    if (H * W) % 128 == 0:
        out, part = SE.conv3x3_c8(xt, B, H, W, wp, b.to(DEV), gn_groups=32)
        assert torch.equal(out, got)
        st = SE.groupnorm_stats(out, B, 32, 1e-6, partial=part).double().cpu()
        o = out.double().cpu().reshape(B, H * W, 32, 4)
        mean, var = o.mean(dim=(1, 3)), o.var(dim=(1, 3), unbiased=False)
        assert (st[..., 0] - mean).abs().max().item() < 2e-3 and ((st[..., 1] - (var + 1e-6).rsqrt()) / (var + 1e-6).rsqrt()).abs().max().item() < 2e-3
    with pytest.raises(RuntimeError, match="conv3x3_c8"):
        SE.conv3x3_c8(xt, B, H, W, wp[:, :64].contiguous(), b.to(DEV))              # fewer than 96 weight columns


@pytest.mark.parametrize("B,H,W,Ci,Co", [(4, 192, 192, 64, 256),       # 576 tiles: two rounds in the 256x256 kernel + a 64-tile tail in the 128x128 one (row offset)
                                         (9, 128, 120, 128, 256),      # 540 tiles, rows wrap inside a lane's pieces; 8 channels per group
                                         (8, 128, 128, 64, 512)])      # two column tiles, 16 channels per group, whole rounds only
def test_conv3x3_emits_groupnorm_partials_from_the_256_kernel(B, H, W, Ci, Co, monkeypatch):
    """EPI_BIAS convolutions with whole rounds of 256x256 tiles keep the ping-pong kernel when asked for the GroupNorm partial sums
    (gemm_bf16_256q<EPI_BIAS, false, true, true>: a pass over the accumulators in front of the plain epilogue): the convolution result is the plain
    call's BITWISE, GroupNorm from the partial sums equals GroupNorm of the stored tensor, the routing counters show the 256x256 kernel.  Since
    round 6 a RESID convolution of the same shape CAN emit them too (gemm_bf16_256q<EPI_RESID, false, true, true>: the pre-pass reads the residual
    tile as well; opt-in with VISREP_GN_RESID_256=1 - measured 0.5 ms slower on the SD1.5 forward than the separate statistics pass, so by
    default such a convolution is still refused the sums) - same checks under the opt-in."""
    g = torch.Generator().manual_seed(H + W + Co + 5)
    G = 32
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.5 + 0.3
    wp = bf(w.float().permute(0, 2, 3, 1).reshape(Co, 9 * Ci)).to(DEV)
    xt = tokens(x).to(DEV)
    gam, bet = (torch.randn(Co, generator=g) * 0.3 + 1).to(DEV), (torch.randn(Co, generator=g) * 0.2).to(DEV)
    monkeypatch.delenv("VISREP_GN_RESID_256", raising=False)
    assert SE.conv_gn_supported(B, H * W, Co, G, _lib.EPI_BIAS) and not SE.conv_gn_supported(B, H * W, Co, G, _lib.EPI_RESID)
    res = bf(torch.randn(B * H * W, Co, generator=g) * 1.5 + 0.4).to(DEV)
    with pytest.raises(RuntimeError, match="conv_gn_supported"):
        SE.conv3x3(xt, B, H, W, wp, b.to(DEV), 1, 0, False, _lib.EPI_RESID, resid=res, gn_groups=G)
    monkeypatch.setenv("VISREP_GN_RESID_256", "1")
    assert SE.conv_gn_supported(B, H * W, Co, G, _lib.EPI_RESID)
    for epi, rs in ((_lib.EPI_BIAS, None), (_lib.EPI_RESID, res)):
        plain, Ho, Wo = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), 1, 0, False, epi, resid=rs)
        _lib.routes(reset=True)
        out, _, _, part = SE.conv3x3(xt, B, H, W, wp, b.to(DEV), 1, 0, False, epi, resid=rs, gn_groups=G)
        r = _lib.routes()
        tail = ((B * H * W + 255) // 256 * (Co // 256)) % 256 != 0
        assert r["conv_256"] == 1 and r["conv_128_gn"] == (1 if tail else 0) and r["conv_128"] == 0, (epi, r)
        # the rows of the whole tile rounds: the same kernel body, bitwise; the tail rows: the 128x128 kernel, which splits K for the plain call when
        # K >= 1024 and cannot when it emits partial sums (another order of summation)
        ntn, ntm = Co // 256, (B * H * W + 255) // 256
        head = (ntm * ntn) // 256 * 256 // ntn * 256
        assert torch.equal(out[:head], plain[:head]) and (out.float() - plain.float()).abs().max().item() < 4e-2, epi
        if 9 * Ci < 1024:
            assert torch.equal(out, plain)
        for silu in (True, False):
            want = SE.groupnorm(plain, gam, bet, B, G, 1e-6, silu)
            got = SE.groupnorm_from_partials(out, gam, bet, B, G, 1e-6, silu, part)
            assert (got.float() - want.float()).abs().max().item() < 4e-2 and rel_err(got, want.float().cpu()) < 3e-3, epi
        # the statistics themselves against float64 sums of the stored tensor (per image and group)
        st = SE.groupnorm_stats(out, B, G, 1e-6, partial=part).double().cpu()
        o = out.double().cpu().reshape(B, H * W, G, Co // G)
        mean, var = o.mean(dim=(1, 3)), o.var(dim=(1, 3), unbiased=False)
        assert (st[..., 0] - mean).abs().max().item() < 2e-3 and ((st[..., 1] - (var + 1e-6).rsqrt()) / (var + 1e-6).rsqrt()).abs().max().item() < 2e-3, epi


@pytest.fixture(params=[8, 16], ids=["tile16x8", "tile16x16"])
def halo_tile(request):
    """Both tile shapes of conv3x3_halo's Cout = 128 kernel: 16 x 16 pixels (default: 8 waves, one workgroup per CU) and 16 x 8 (4 waves, two)."""
    lib = _lib.load()
    old = lib.visrep_set_conv_halo_tile(request.param)
    assert old in (8, 16)
    yield request.param
    lib.visrep_set_conv_halo_tile(old)


@pytest.mark.parametrize("B,H,W,Co,resid,exact", [(4, 128, 128, 128, False, True), (4, 128, 128, 128, True, True), (2, 128, 128, 256, True, True),
                                                  (2, 32, 48, 128, True, False), (1, 16, 64, 256, False, False), (3, 16, 16, 128, False, False)])
def test_conv3x3_halo_equals_groupnorm_apply_plus_implicit_gemm(B, H, W, Co, resid, exact, halo_tile):
    """conv3x3_halo (GroupNorm + SiLU of the input fused into a halo-resident 3x3 convolution, csrc/conv_halo.hip) against the two launches it
    replaces - groupnorm (statistics + apply pass) and the implicit-GEMM convolution.  Same normalisation arithmetic, same K order, same MFMA
    chains: BIT-IDENTICAL outputs where the reference convolution is not split along K (`exact`: enough tiles to fill the chip); the small
    shapes (image borders on every side of every tile, one-tile images) take the reference through split-K, whose fp32 summation order
    differs - compared at bf16 rounding.  The output's GroupNorm partial sums feed groupnorm_from_partials like the 128x128 kernel's."""
    g = torch.Generator().manual_seed(B * 1000 + H + W + Co)
    C = 128
    x = (torch.randn(B * H * W, C, generator=g) * (1 + torch.rand(1, C, generator=g)) + torch.randn(1, C, generator=g)).to(torch.bfloat16).to(DEV)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV)
    w = (torch.randn(Co, 9 * C, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
    bias = (0.1 * torch.randn(Co, generator=g)).to(DEV)
    res = (torch.randn(B * H * W, Co, generator=g)).to(torch.bfloat16).to(DEV) if resid else None
    epi = _lib.EPI_RESID if resid else _lib.EPI_BIAS
    assert SE.conv_halo_supported(B, H, W, C, Co) and not SE.conv_halo_supported(B, H + 8, W, C, Co) and not SE.conv_halo_supported(B, H, W, 256, Co)
    h = SE.groupnorm(x, gamma, beta, B, 32, 1e-6, True)
    want, _, _ = SE.conv3x3(h, B, H, W, w, bias, 1, 0, False, epi, res)
    stats = SE.groupnorm_stats(x, B, 32, 1e-6)
    tab = SE.groupnorm_table(stats, gamma, beta)
    _lib.routes(reset=True)
    got, part = SE.conv3x3_halo(x, B, H, W, w, bias, epi, res, tab, True, 32)
    assert _lib.routes()["conv_halo"] == 1
    if exact:
        assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()
    else:
        assert rel_err(got, want) < 2e-3 and (got.float() - want.float()).abs().max().item() <= 2.0 ** -6 * want.float().abs().max().item()
    # no GroupNorm in front (gn_table = None): the plain convolution
    plain_want, _, _ = SE.conv3x3(x, B, H, W, w, bias, 1, 0, False, epi, res)
    plain = SE.conv3x3_halo(x, B, H, W, w, bias, epi, res, None, False, 0)
    assert torch.equal(plain, plain_want) if exact else rel_err(plain, plain_want) < 2e-3
    # the output's partial sums: the following GroupNorm without a statistics pass
    g2, b2 = (1 + 0.3 * torch.randn(Co, generator=g)).to(DEV), (0.2 * torch.randn(Co, generator=g)).to(DEV)
    a = SE.groupnorm_from_partials(got, g2, b2, B, 32, 1e-6, True, part)
    b_ = SE.groupnorm(got, g2, b2, B, 32, 1e-6, True)
    assert rel_err(a, b_) < 1e-3
    st_a, st_b = SE.groupnorm_stats(got, B, 32, 1e-6, partial=part), SE.groupnorm_stats(got, B, 32, 1e-6)
    # partial sums are over the fp32 (unrounded) outputs, the statistics pass reads the bf16 tensor: equal to bf16 rounding of the inputs
    assert (st_a[..., 0] - st_b[..., 0]).abs().max().item() < 2e-3 and ((st_a[..., 1] - st_b[..., 1]).abs() / st_b[..., 1]).max().item() < 2e-3


def test_vae_encoder_with_and_without_the_fused_halo_convolution(monkeypatch):
    """SdEngine.vae_moments with the 128-channel layers on conv3x3_halo against the same engine with VISREP_CONV_HALO=0 (apply pass +
    implicit-GEMM convolution): the same arithmetic up to the summation order of the GroupNorm statistics (partial sums per 16 x 4 patch
    instead of per 64 raster pixels), on a 128-px image of the real SD1.5 VAE widths."""
    sp = SW.SD_SPECS['runwayml/stable-diffusion-v1-5']
    wv = SW.synthetic_vae(sp.vae, 22)
    wu = SW.synthetic_unet(SW.tiny_sd_spec().unet, 21, n_up_blocks=1)
    spec = SW.SdSpec("vae-halo", unet=SW.tiny_sd_spec().unet, vae=sp.vae, sched=sp.sched, text_len=sp.text_len)
    img = torch.from_numpy(np.random.RandomState(8).uniform(-1, 1, (2, 3, 128, 128)).astype(np.float32)).to(DEV)
    monkeypatch.setenv("VISREP_CONV_HALO", "1")
    on = SE.SdEngine(spec, wu, wv, DEV, up_ft_index=0)
    _lib.routes(reset=True)
    m_on, h, w = on.vae_moments(img)
    r = _lib.routes(reset=True)
    assert r["conv_halo"] == 4, r                      # down block 0: 2 resnets x 2 convolutions at 128 channels (128 -> 256 stays on the 256x256 kernel)
    monkeypatch.setenv("VISREP_CONV_HALO", "0")
    off = SE.SdEngine(spec, wu, wv, DEV, up_ft_index=0)
    m_off, _, _ = off.vae_moments(img)
    assert _lib.routes(reset=True)["conv_halo"] == 0
    Z = sp.vae.latent_channels
    e = rel_err(m_on[:, : 2 * Z], m_off[:, : 2 * Z])
    m32, l32 = OD.vae_encode_moments(sp.vae, wv, img.cpu())
    mean_on = untokens(m_on[:, :Z], 2, h, w).cpu()
    print(f"VAE moments, halo on vs off: rel {e:.3e}; halo on vs fp32 oracle (mean): {rel_err(mean_on, m32):.3e}")
    assert e < 2e-2 and rel_err(mean_on, m32) < 3e-2


def test_conv3x3_gn_rejects_unsupported_shapes():
    x = torch.zeros(2 * 24 * 24, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(320, 576, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="conv_gn_supported"):
        SE.conv3x3(x, 2, 24, 24, w, None, gn_groups=32)


def test_conv3x3_rejects_narrow_channels():
    x = torch.zeros(64, 8, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        SE.conv3x3(x, 1, 8, 8, torch.zeros(64, 72, dtype=torch.bfloat16, device=DEV), None)


@pytest.mark.parametrize("M,N,K", [(144, 1280, 11520), (576, 320, 2880), (64, 128, 1024), (300, 1280, 23040)])
def test_gemm_split_k_small_m_deep_k(M, N, K):
    """Few tiles, deep reduction: with the scratch attached the K loop is split over CUs and reduced in slice order."""
    g = torch.Generator().manual_seed(M + K)
    a = bf(torch.randn(M, K, generator=g)).to(DEV)
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    bias, ls = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    x = bf(torch.randn(M, N, generator=g)).to(DEV)
    lib = _lib.load()
    assert lib.visrep_set_scratch(None, 0) == 0
    SE._SCRATCH.clear()
    plain = [engine.gemm(a, w, bias, _lib.EPI_F32), engine.gemm(a, w, bias, _lib.EPI_ACT, act="gelu"),
             engine.gemm(a, w, bias, _lib.EPI_RESID, resid=x, ls=ls), engine.gemm(a, w, None, _lib.EPI_BIAS)]
    SE.ensure_scratch(torch.device(DEV))
    split = [engine.gemm(a, w, bias, _lib.EPI_F32), engine.gemm(a, w, bias, _lib.EPI_ACT, act="gelu"),
             engine.gemm(a, w, bias, _lib.EPI_RESID, resid=x, ls=ls), engine.gemm(a, w, None, _lib.EPI_BIAS)]
    want = a.float() @ w.float().t() + bias
    assert (split[0] - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    for p, q in zip(plain, split):
        assert rel_err(q, p) < 4e-3
    x2 = x.clone()
    engine.gemm(a, w, bias, _lib.EPI_RESID, resid=x2, out=x2)                 # in place
    assert rel_err(x2, x.float() + want) < 4e-3
    again = engine.gemm(a, w, bias, _lib.EPI_F32)
    assert torch.equal(again, split[0])                                       # slice order is fixed: bit-reproducible


def test_geglu_and_mean_groups_and_softmax_rows():
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(70, 512, generator=g) * 2)
    want = x[:, :256].float() * F.gelu(x[:, 256:].float())
    assert rel_err(SE.geglu(x.to(DEV)), want) < 4e-3
    y = bf(torch.randn(3, 4, 50, 64, generator=g))
    got = SE.mean_groups(y.to(DEV), 3, 4).view(3, 50, 64)
    assert rel_err(got, y.float().mean(1)) < 4e-3
    s = torch.randn(33, 100, generator=g) * 20
    p = SE.softmax_rows(s.to(DEV), 100, 128, 0.25)
    assert p.shape == (33, 128) and (p[:, 100:] == 0).all()
    assert (p[:, :100].float().cpu() - torch.softmax(s * 0.25, -1)).abs().max().item() < 4e-3


def test_nchw_to_tokens_and_noisy_latents():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 5, 7, generator=g)
    for src in (x, bf(x)):
        t = SE.nchw_to_tokens(src.to(DEV), 8).cpu().float()
        assert torch.equal(t[:, :3], tokens(bf(src).float())) and (t[:, 3:] == 0).all()
    sp = SW.tiny_sd_spec()
    B, Z, h, w = 2, 4, 6, 5
    mean, logvar = torch.randn(B, Z, h, w, generator=g), torch.randn(B, Z, h, w, generator=g) * 3 - 1
    post, ddim = torch.randn(B, Z, h, w, generator=g), torch.randn(B, Z, h, w, generator=g)
    mom = torch.zeros(B * h * w, 64)
    mom[:, :Z], mom[:, Z: 2 * Z] = tokens(mean), tokens(logvar)
    lat = torch.empty(B * h * w, 8, dtype=torch.bfloat16, device=DEV)
    ac = float(sp.sched.alphas_cumprod()[261])
    mom_d, post_d, ddim_d = mom.to(DEV), post.to(DEV), ddim.to(DEV)        # keep the device copies alive across the raw-pointer call
    rc = _lib.load().visrep_sd_noisy_latents(_lib.ptr(mom_d), 64, _lib.ptr(post_d), _lib.ptr(ddim_d), _lib.ptr(lat), B, Z,
                                              h * w, 8, sp.vae.scaling_factor, math.sqrt(ac), math.sqrt(1 - ac), _lib.stream_ptr())
    assert rc == 0
    want = OD.noisy_latents(sp, mean, logvar.clamp(-30, 20), post, ddim, 261)
    got = lat.cpu().float()
    assert (got[:, Z:] == 0).all()
    tw = tokens(want)
    assert ((got[:, :Z] - tw).abs() / tw.abs().clamp_min(1.0)).max().item() < 1e-2


def ref_attn(q, k, v, B, Tq, Tk, H, dh, shared):
    q = q.float().view(B, Tq, H, dh).transpose(1, 2)
    k = (k.float().view(1, Tk, H, dh).expand(B, -1, -1, -1) if shared else k.float().view(B, Tk, H, dh)).transpose(1, 2)
    v = (v.float().view(1, Tk, H, dh).expand(B, -1, -1, -1) if shared else v.float().view(B, Tk, H, dh)).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * Tq, H * dh)


@pytest.mark.parametrize("B,Tq,Tk,H,dh,shared", [(2, 144, 144, 2, 40, False), (1, 576, 576, 3, 80, False), (2, 300, 300, 2, 160, False),
                                                 (3, 256, 77, 2, 40, True), (2, 100, 11, 4, 160, True), (2, 64, 64, 2, 64, False),
                                                 (1, 2304, 2304, 2, 80, False)])
def test_attention_padded_heads_and_cross(B, Tq, Tk, H, dh, shared):
    """Head widths of SD1.5 (40 / 80 / 160) zero-padded to 64 / 128 / 192 by the weight packer; prompt K/V shared."""
    g = torch.Generator().manual_seed(Tq + dh)
    d = H * dh
    dp = (dh + 63) // 64 * 64
    din = 128                                                               # GEMM K granule; head widths come from the weights
    xq = bf(torch.randn(B * Tq, din, generator=g)).to(DEV)
    xk = xq if not shared and Tk == Tq else bf(torch.randn((1 if shared else B) * Tk, din, generator=g)).to(DEV)
    wq, wk, wv = [bf(torch.randn(d, din, generator=g) * (1.3 / math.sqrt(din))) for _ in range(3)]
    pad = SE.SdEngine._pad_heads_out
    q = engine.gemm(xq, bf(pad(wq.float(), H, dp)).to(DEV))
    k = engine.gemm(xk, bf(pad(wk.float(), H, dp)).to(DEV))
    vt = engine.linear_vt(xk, bf(pad(wv.float(), H, dp)).to(DEV), None)
    out = SE.attention(q, k, vt, H * dp, B, Tq, Tk, H, dp, dh ** -0.5, shared)
    out = out.view(B * Tq, H, dp)[:, :, :dh].reshape(B * Tq, d)
    assert (SE.attention(q, k, vt, H * dp, B, Tq, Tk, H, dp, dh ** -0.5, shared).view(B * Tq, H, dp)[:, :, dh:] == 0).all()
    qr, kr, vr = bf(xq.float().cpu() @ wq.float().t()), bf(xk.float().cpu() @ wk.float().t()), bf(xk.float().cpu() @ wv.float().t())
    want = ref_attn(qr, kr, vr, B, Tq, Tk, H, dh, shared)
    assert rel_err(out, want) < 1e-2


def test_attention_rejects_unsupported_head_width():
    z = torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="head_dim"):
        SE.attention(z, z, torch.zeros(256, 128, dtype=torch.bfloat16, device=DEV), 256, 1, 64, 64, 1, 256, 1.0, False)


# ------------------------------------------------------------------------------------------------ composed tower
@pytest.mark.parametrize("tag", SD_TAGS)
def test_sd_tower_matches_reference_golden(tag):
    sp, wu, wv, inp, want = load_sd_case(tag)
    eng = SE.SdEngine(sp, wu, wv, DEV, up_ft_index=inp["up_ft_index"])
    # stage 1: VAE posterior moments against the reference's latent_dist
    E = inp["ensemble_size"]
    img = inp["img"].repeat_interleave(E, dim=0)
    mom, h, w = eng.vae_moments(img.to(DEV))
    Z = sp.vae.latent_channels
    B = img.shape[0]
    mean = untokens(mom[:, :Z], B, h, w).cpu()
    logvar = untokens(mom[:, Z: 2 * Z], B, h, w).cpu()
    mref, lref = OD.vae_encode_moments(sp.vae, {k: bf(v) for k, v in wv.items()}, bf(img))       # bf16 reference run
    assert rel_err(mean, inp["mean"]) < max(2.0 * rel_err(mref, inp["mean"]), 2e-2)
    assert rel_err(logvar, inp["logvar"]) < max(2.0 * rel_err(lref, inp["logvar"]), 2e-2)
    # stage 2: the whole SDFeaturizer.forward + DiffVisionTower.forward
    got = eng.forward(inp["img"], inp["prompt_embeds"], t=inp["t"], ensemble_size=E, post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"])
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    ref_bf16 = OD.sd_features(sp, wu, wv, inp["img"], inp["prompt_embeds"], inp["post_noise"], inp["ddim_noise"], t=inp["t"],
                              up_ft_index=inp["up_ft_index"], ensemble_size=E, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert e_hip < max(2.0 * e_ref, 2e-2), (tag, e_hip, e_ref)


def test_sd_tower_graph_replay_equals_eager():
    sp, wu, wv, inp, _ = load_sd_case("conv_up1_ens2")
    kw = dict(t=inp["t"], ensemble_size=inp["ensemble_size"], post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"])
    eager = SE.SdEngine(sp, wu, wv, DEV, up_ft_index=1, graph=False).forward(inp["img"], inp["prompt_embeds"], **kw)
    eng = SE.SdEngine(sp, wu, wv, DEV, up_ft_index=1, graph=True)
    pe = inp["prompt_embeds"]
    first = eng.forward(inp["img"], pe, **kw)                    # warm-up + capture + replay
    assert len(eng._graphs) == 1
    again = eng.forward(inp["img"], pe, **kw)                    # replay only
    other = eng.forward(inp["img"].flip(-1), pe, **kw)           # same graph, new input
    assert len(eng._graphs) == 1
    assert torch.equal(first, eager) and torch.equal(again, eager)            # every reduction has a fixed order: bit-reproducible
    assert rel_err(other, eager) > 0.05
    eng.forward(inp["img"], pe, **{**kw, "t": 5})                # a new timestep re-folds the conv1 biases: graphs dropped
    assert len(eng._graphs) == 1 and list(eng._graphs)[0][2] == 5


def test_sd_tower_random_noise_path_and_state_errors():
    sp, wu, wv, inp, _ = load_sd_case("conv_up0")
    eng = SE.SdEngine(sp, wu, wv, DEV)
    with pytest.raises(RuntimeError, match="set_prompt"):
        eng.forward(inp["img"])
    a = eng.forward(inp["img"], inp["prompt_embeds"], t=100)
    assert torch.isfinite(a.float()).all() and a.shape == (2, 64, 128)
    with pytest.raises(ValueError, match="noise tensors"):
        eng.forward(inp["img"], t=100, post_noise=torch.zeros(1, 4, 3, 3))


# ------------------------------------------------------------------------------------------------ prompt encoder
@pytest.mark.parametrize("tag", ["quick", "gelu"])
def test_clip_text_engine_matches_hf_golden(tag):
    from test_oracle_golden import load_text_case
    from law_of_vision_representation_in_mllms_amd.text_engine import ClipTextEngine
    from oracle import text as OT
    ts, w, ids, want = load_text_case(tag)
    got = ClipTextEngine(ts, w, DEV).forward(ids)
    ref_bf16 = OT.clip_text_hidden({k: bf(v) if v.is_floating_point() else v for k, v in w.items()}, ids, ts.heads, ts.act).float()
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert got.shape == want.shape and e_hip < max(2.0 * e_ref, 1e-2), (e_hip, e_ref)
    pen = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "text_tiny.npz"))[f"{tag}.penultimate"])
    assert rel_err(ClipTextEngine(ts, w, DEV).forward(ids, hidden_state=-2), pen) < max(2.0 * e_ref, 1e-2)     # SDXL's hidden_states[-2]


# ------------------------------------------------------------------------------------------------ full-width SD1.5
def test_sd15_full_width_parity_and_tower_api(monkeypatch):
    """The real SD1.5 architecture (320/640/1280/1280, heads of 40/80/160, VAE 128..512 with the 512-wide single-head
    attention) on a small image against the fp32 CPU oracle; then the DiffVisionTower drop-in around the same engine."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import builder as B
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    args = SimpleNamespace(vision_tower='runwayml/stable-diffusion-v1-5', up_ft_index=0, t=261, prompt="a photo of a cat",
                           ensemble_size=1, img_size=128)
    tower = B.build_diffusion_vision_tower(args)
    assert tower.is_loaded and tower.hidden_size == 1280 and tower.dtype == torch.bfloat16
    feat = tower.vision_tower
    sp = feat.spec
    rs = np.random.RandomState(5)
    img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, 128, 128)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((1, 4, 16, 16)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((1, 4, 16, 16)).astype(np.float32))
    pe = feat.encode_prompt(args.prompt)
    assert pe.shape == (1, 77, 768)
    got = feat.forward(img, args.prompt, t=261, up_ft_index=0, ensemble_size=1, post_noise=post, ddim_noise=ddim)     # [c, h, w]
    assert got.shape == (1280, 4, 4)
    wu = {k: v for k, v in feat._wu.items() if not k.startswith(("up_blocks.1", "up_blocks.2", "up_blocks.3"))}
    want = OD.sd_features(sp, wu, feat._wv, img, pe.float().cpu(), post, ddim, t=261)                                 # [1, 16, 1280]
    ref_bf16 = OD.sd_features(sp, wu, feat._wv, img, pe.float().cpu(), post, ddim, t=261, dtype=torch.bfloat16)
    got_tok = got.permute(1, 2, 0).reshape(1, 16, 1280)
    e_hip, e_ref = rel_err(got_tok, want), rel_err(ref_bf16, want)
    assert e_hip < max(2.0 * e_ref, 3e-2), (e_hip, e_ref)
    # tower.forward: tensor batch and a single [3,H,W] image, random noise drawn on the device
    out = tower(torch.cat([img, img.flip(-1)], 0))
    assert out.shape == (2, 16, 1280) and torch.isfinite(out.float()).all()
    assert tower(img[0]).shape == (1, 16, 1280)


def test_sdxl_full_width_parity(monkeypatch):
    """The real SDXL-base UNet (320/640/1280, attention-free first block, 2 / 10 transformer layers per attention, Linear proj_in /
    proj_out, 64-wide heads, 2048-wide two-encoder prompt) on a small image against the fp32 CPU oracle.  2.4 B parameters: the
    deterministic weights are drawn on the device (VISREP_FAST_SYNTHETIC) and the same tensors feed the engine and the oracle."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import builder as B
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("VISREP_FAST_SYNTHETIC", "cuda")
    args = SimpleNamespace(vision_tower='stabilityai/stable-diffusion-xl-base-1.0', up_ft_index=0, t=261, prompt="a photo of a cat",
                           ensemble_size=1, img_size=128)
    tower = B.build_diffusion_vision_tower(args)
    assert tower.is_loaded and tower.hidden_size == 1280
    feat = tower.vision_tower
    sp = feat.spec
    assert sp.unet.tlayers == (1, 2, 10) and sp.unet.linear_projection
    rs = np.random.RandomState(6)
    img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, 128, 128)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((1, 4, 16, 16)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((1, 4, 16, 16)).astype(np.float32))
    pe = feat.encode_prompt(args.prompt)
    assert pe.shape == (1, 77, 2048)                                     # hidden_states[-2] of both text encoders, concatenated
    got = feat.forward(img, args.prompt, t=261, up_ft_index=0, ensemble_size=1, post_noise=post, ddim_noise=ddim)     # [c, h, w]
    assert got.shape == (1280, 8, 8)
    wu = {k: v.float().cpu() for k, v in feat._wu.items() if not k.startswith(("up_blocks.1", "up_blocks.2"))}
    wv = {k: v.float().cpu() for k, v in feat._wv.items()}
    want = OD.sd_features(sp, wu, wv, img, pe.float().cpu(), post, ddim, t=261)                                       # [1, 64, 1280]
    ref_bf16 = OD.sd_features(sp, wu, wv, img, pe.float().cpu(), post, ddim, t=261, dtype=torch.bfloat16)
    got_tok = got.permute(1, 2, 0).reshape(1, 64, 1280)
    e_hip, e_ref = rel_err(got_tok, want), rel_err(ref_bf16, want)
    assert e_hip < max(2.0 * e_ref, 3e-2), (e_hip, e_ref)


# ------------------------------------------------------------------------------------------------ DiT tower
@pytest.mark.parametrize("tag", ["last", "first_other_res"])
def test_dit_tower_matches_reference_golden(tag):
    from test_oracle_golden import load_dit_case
    from law_of_vision_representation_in_mllms_amd.dit_engine import DiTEngine
    from oracle import dit as ODT
    sp, wd, wv, inp, want = load_dit_case(tag)
    eng = DiTEngine(sp, wd, wv, DEV, up_ft_index=inp["up_ft_index"])
    kw = dict(t=inp["t"], post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"])
    got = eng.forward(inp["img"], **kw)
    ref_bf16 = ODT.dit_features(sp, wd, wv, inp["img"], inp["post_noise"], inp["ddim_noise"], t=inp["t"], up_ft_index=inp["up_ft_index"],
                                dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert got.shape == want.shape and e_hip < max(2.0 * e_ref, 2e-2), (tag, e_hip, e_ref)
    assert torch.equal(eng.forward(inp["img"], **kw), got)                           # graph replay, bit-reproducible
    with pytest.raises(ValueError, match="ensemble"):
        eng.forward(inp["img"], ensemble_size=2, **kw)


def test_dit_and_sd3_graphs_survive_a_batch_size_change():
    """Regression (round 2): the position-table cache of the DiT / SD3 engines dropped the table a captured HIP graph of ANOTHER batch
    size still pointed at, so batch 2 -> batch 1 -> batch 2 replayed the first graph on freed memory (wrong features, then a memory
    fault in the 13-setting sweep).  The same inputs must give the same bits before and after other batch sizes ran."""
    from test_oracle_golden import load_dit_case, load_sd3_case
    from law_of_vision_representation_in_mllms_amd.dit_engine import DiTEngine
    from law_of_vision_representation_in_mllms_amd.sd3_engine import Sd3Engine
    sp, wd, wv, inp, _ = load_dit_case("last")
    eng = DiTEngine(sp, wd, wv, DEV, up_ft_index=inp["up_ft_index"])
    img = inp["img"]
    B = img.shape[0]
    big = lambda t, n: torch.cat([t] * n, 0)
    first = eng.forward(img, t=inp["t"], post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"])
    for n in (2, 3):
        junk = eng.forward(big(img, n), t=inp["t"], post_noise=big(inp["post_noise"], n), ddim_noise=big(inp["ddim_noise"], n))
        assert rel_err(junk[:B], first.float().cpu()) < 2e-2, n                     # (split-K choices depend on the batch: close, not bit-equal)
        torch.cuda.empty_cache()
        torch.randn(1 << 22, device=DEV)                                            # recycle whatever was freed
    assert torch.equal(eng.forward(img, t=inp["t"], post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"]), first)
    sp, wc, wv, inp, _ = load_sd3_case("last")
    eng = Sd3Engine(sp, wc, wv, DEV, up_ft_index=inp["up_ft_index"])
    run = lambda n: eng.forward(big(inp["img"], n), inp["prompt_embeds"], t=inp["t"], post_noise=big(inp["post_noise"], n),
                                ddim_noise=big(inp["noise"], n), pooled=inp["pooled"])
    first = run(1)
    for n in (2, 3):
        assert rel_err(run(n)[: first.shape[0]], first.float().cpu()) < 2e-2, n
        torch.cuda.empty_cache()
        torch.randn(1 << 22, device=DEV)
    assert torch.equal(run(1), first)


def test_dit_xl2_full_width_parity_and_tower_api(monkeypatch):
    """DiT-XL/2 widths (16 heads x 72 -> padded to 128, d = 1152, ff 4608) for the first 3 blocks on a 128-px image against the
    fp32 CPU oracle, then the DiffVisionTower drop-in (4608-channel tokens)."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.dit_engine import DiTEngine
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM import diffusion_encoder as DE
    from oracle import dit as ODT
    sp = SW.DIT_SPECS["facebook/DiT-XL-2-512"]
    wd, wv = SW.synthetic_dit(sp.core, 31, n_layers=3), SW.synthetic_vae(sp.vae, 32)
    rs = np.random.RandomState(8)
    img = torch.from_numpy(rs.uniform(-1, 1, (2, 3, 128, 128)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((2, 4, 16, 16)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((2, 4, 16, 16)).astype(np.float32))
    got = DiTEngine(sp, wd, wv, DEV, up_ft_index=2).forward(img, t=261, post_noise=post, ddim_noise=ddim)
    assert got.shape == (2, 16, 4608)
    want = ODT.dit_features(sp, wd, wv, img, post, ddim, t=261, up_ft_index=2)
    ref_bf16 = ODT.dit_features(sp, wd, wv, img, post, ddim, t=261, up_ft_index=2, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert e_hip < max(2.0 * e_ref, 2e-2), (e_hip, e_ref)
    # drop-in tower around a featurizer with these (3-block) weights
    class Feat(DE.DiTFeaturizer):
        def __init__(self, sd_id):
            self.sd_id, self.device, self.spec, self._wd, self._wv = sd_id, torch.device(DEV), sp, wd, wv
            self._engines, self.dtype = {}, torch.bfloat16
    monkeypatch.setitem(DE.build_featurelizer_mapping, 'facebook/DiT-XL-2-512', Feat)
    tower = DE.DiffVisionTower(SimpleNamespace(vision_tower='facebook/DiT-XL-2-512', up_ft_index=2, t=261, prompt="", ensemble_size=1, img_size=128))
    assert tower.hidden_size == 4608
    out = tower(img)
    assert out.shape == (2, 16, 4608) and torch.isfinite(out.float()).all()


# ------------------------------------------------------------------------------------------------ image-variation tower
@pytest.mark.parametrize("H,W,OH,OW", [(64, 64, 224, 224), (300, 200, 224, 224), (768, 768, 224, 224), (5, 7, 3, 4)])
def test_resize_bilinear_matches_torch(H, W, OH, OW):
    from law_of_vision_representation_in_mllms_amd.image_embed import resize_bilinear
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(2, 3, H, W, generator=g) * 2 - 1
    want = F.interpolate(x, size=(OH, OW), mode="bilinear")
    got = resize_bilinear(x.to(DEV), (OH, OW))
    assert got.dtype == torch.bfloat16 and (got.float().cpu() - want).abs().max().item() < 8e-3
    got_bf = resize_bilinear(bf(x).to(DEV), (OH, OW))
    assert (got_bf.float().cpu() - F.interpolate(bf(x).float(), size=(OH, OW), mode="bilinear")).abs().max().item() < 8e-3


def test_imsd_tower_matches_reference_golden(monkeypatch):
    from test_oracle_golden import load_imsd_case
    from law_of_vision_representation_in_mllms_amd.image_embed import ClipImageEmbedder
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models import dift_imsd as DI
    from oracle import vit as OV
    sp, wu, wv, vs, (w, g, b, p), inp, want = load_imsd_case()

    class Feat(DI.IMSDFeaturizer):                       # the featurizer around the tiny fixture models
        def __init__(self):
            self.sd_id, self.device, self.spec, self._wu, self._wv = "tiny-imsd", torch.device(DEV), sp, wu, wv
            self.embedder = ClipImageEmbedder(vs, w, g, b, p, DEV)
            self._engines, self.dtype = {}, torch.bfloat16
    feat = Feat()
    emb = feat.encode_image(inp["img"])
    assert emb.shape == (2, 1, sp.unet.cross_dim)
    px = F.interpolate(inp["img"], size=(224, 224), mode="bilinear")
    e_ref = rel_err(OV.clip_image_embeds(vs, w, g, b, p, bf(px).float(), dtype=torch.bfloat16), inp["image_embeds"])
    assert rel_err(emb[:, 0], inp["image_embeds"]) < max(2.0 * e_ref, 1e-2)
    got = feat.forward(inp["img"], "ignored", t=261, up_ft_index=0, ensemble_size=2, post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"])
    assert got.shape == (2, 128, 8, 8)                                        # [B, c, h, w] like the reference's squeeze()
    got_tok = got.permute(0, 2, 3, 1).reshape(2, 64, 128)
    ref_bf16 = OD.imsd_features(sp, wu, wv, inp["img"], inp["image_embeds"].unsqueeze(1), inp["post_noise"], inp["ddim_noise"], t=261,
                                up_ft_index=0, ensemble_size=2, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got_tok, want), rel_err(ref_bf16, want)
    assert e_hip < max(2.0 * e_ref, 2e-2), (e_hip, e_ref)
    again = feat.forward(inp["img"], "ignored", t=261, up_ft_index=0, ensemble_size=2, post_noise=inp["post_noise"], ddim_noise=inp["ddim_noise"])
    assert torch.equal(again, got)                                            # graph replay with the per-image context buffer


# ------------------------------------------------------------------------------------------------ SD3 (MMDiT) tower
@pytest.mark.parametrize("tag", ["last", "mid"])
def test_sd3_tower_matches_reference_golden(tag):
    from test_oracle_golden import load_sd3_case
    from law_of_vision_representation_in_mllms_amd.sd3_engine import Sd3Engine
    from oracle import sd3 as O3
    sp, wc, wv, inp, want = load_sd3_case(tag)
    eng = Sd3Engine(sp, wc, wv, DEV, up_ft_index=inp["up_ft_index"])
    kw = dict(t=inp["t"], post_noise=inp["post_noise"], ddim_noise=inp["noise"], pooled=inp["pooled"])
    pe = inp["prompt_embeds"]
    got = eng.forward(inp["img"], pe, **kw)
    ref_bf16 = O3.sd3_features(sp, wc, wv, inp["img"], pe, inp["pooled"], inp["post_noise"], inp["noise"], t=inp["t"],
                               up_ft_index=inp["up_ft_index"], dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert got.shape == want.shape and e_hip < max(2.0 * e_ref, 2e-2), (tag, e_hip, e_ref)
    assert torch.equal(eng.forward(inp["img"], pe, **kw), got)                       # graph replay, bit-reproducible
    with pytest.raises(ValueError, match="ensemble"):
        eng.forward(inp["img"], pe, ensemble_size=2, **kw)


def test_sd3_medium_width_blocks_and_featurizer_prompt(monkeypatch):
    """SD3-medium widths (24 heads x 64 = 1536, joint 4096, pooled 2048, 16-channel VAE) for the first 2 blocks on a 128-px
    image against the fp32 CPU oracle; the featurizer's prompt assembly (CLIP-L | CLIP-G | T5 zeros) with tiny text encoders."""
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.sd3_engine import Sd3Engine
    from law_of_vision_representation_in_mllms_amd.text_engine import ClipTextEngine
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder.diffLVLM.src.models import dift_sd3 as D3
    from oracle import sd3 as O3
    from oracle import text as OT
    sp = SW.SD3_SPECS["stabilityai/stable-diffusion-3-medium-diffusers"]
    wc, wv = SW.synthetic_sd3(sp.core, 61, n_layers=2), SW.synthetic_vae(sp.vae, 62)
    rs = np.random.RandomState(9)
    img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, 128, 128)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((1, 16, 16, 16)).astype(np.float32))
    noise = torch.from_numpy((0.05 * rs.standard_normal((1, 16, 16, 16))).astype(np.float32))
    pe = torch.from_numpy(rs.standard_normal((1, 77 + 256, 4096)).astype(np.float32))
    pe[:, 77:] = 0
    pooled = torch.from_numpy(rs.standard_normal((1, 2048)).astype(np.float32))
    got = Sd3Engine(sp, wc, wv, DEV, up_ft_index=1).forward(img, pe, t=2, post_noise=post, ddim_noise=noise, pooled=pooled)
    assert got.shape == (1, 16, 6144)
    want = O3.sd3_features(sp, wc, wv, img, pe, pooled, post, noise, t=2, up_ft_index=1)
    ref_bf16 = O3.sd3_features(sp, wc, wv, img, pe, pooled, post, noise, t=2, up_ft_index=1, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert e_hip < max(2.0 * e_ref, 2e-2), (e_hip, e_ref)
    # prompt assembly with two tiny text encoders in place of CLIP-L / CLIP-G
    tspecs = (SW.tiny_text_spec("quick_gelu", 2, 77), SW.tiny_text_spec("gelu", 3, 77))
    feat = D3.SD3Featurizer.__new__(D3.SD3Featurizer)
    feat.device, feat.spec, feat.tokenizers, feat._prompt_cache = torch.device(DEV), sp, [None, None], {}
    tw = [SW.synthetic_text(ts, 70 + i) for i, ts in enumerate(tspecs)]
    feat.text = [ClipTextEngine(ts, w, DEV) for ts, w in zip(tspecs, tw)]
    proj = [SW._synthetic([("text_projection.weight", (ts.d, ts.d))], 80 + i)["text_projection.weight"] for i, ts in enumerate(tspecs)]
    feat.text_proj = [p.to(DEV, torch.bfloat16) for p in proj]
    pe2, pooled2 = feat.encode_prompt("a photo of a cat")
    assert pe2.shape == (1, 77 + 256, 4096) and pooled2.shape == (1, 256)
    assert (pe2[:, 77:] == 0).all() and (pe2[:, :77, 256:] == 0).all()
    for i, (ts, w) in enumerate(zip(tspecs, tw)):
        ids = feat.tokenize("a photo of a cat", i)
        pen = OT.clip_text_hidden(w, ids, ts.heads, ts.act, hidden_state=-2)
        assert rel_err(pe2[:, :77, i * 128:(i + 1) * 128], pen) < 2e-2
        last = OT.clip_text_hidden(w, ids, ts.heads, ts.act)
        want_pool = last[0, int(ids[0].argmax())] @ proj[i].t()
        assert rel_err(pooled2[0, i * 128:(i + 1) * 128], want_pool) < 2e-2


# ------------------------------------------------------------------------------------------------ full depth
# One test per diffusion family at the architecture's FULL depth on a 256-px image against the fp32 CPU oracle: error accumulation over
# 28 DiT blocks / 24 MMDiT blocks, and the whole VAE encoder at a resolution where every down block has real spatial extent (the
# 9,216-token mid-block attention of the 768-px towers is 1,024 tokens here; the code path - GEMM scores -> softmax_rows -> P.V - is
# the same).  Tolerance = max(1.5 x the oracle's own bf16 error, 2e-2), like the ViT towers.
def _fast_cuda(monkeypatch):
    monkeypatch.setenv("VISREP_FAST_SYNTHETIC", "cuda")                   # multi-GB weight sets: drawn in HBM, the same tensors feed the oracle


def test_dit_xl2_full_depth_parity(monkeypatch):
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.dit_engine import DiTEngine
    from oracle import dit as ODT
    _fast_cuda(monkeypatch)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sp = SW.DIT_SPECS["facebook/DiT-XL-2-512"]
    assert sp.core.layers == 28
    wd, wv = SW.synthetic_dit(sp.core, 131), SW.synthetic_vae(sp.vae, 132)           # all 28 blocks
    rs = np.random.RandomState(18)
    img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, 256, 256)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((1, 4, 32, 32)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((1, 4, 32, 32)).astype(np.float32))
    got = DiTEngine(sp, wd, wv, DEV, up_ft_index=-1).forward(img, t=261, post_noise=post, ddim_noise=ddim)
    want = ODT.dit_features(sp, wd, wv, img, post, ddim, t=261, up_ft_index=-1)
    ref_bf16 = ODT.dit_features(sp, wd, wv, img, post, ddim, t=261, up_ft_index=-1, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert got.shape == want.shape and e_hip < max(1.5 * e_ref, 2e-2), (e_hip, e_ref)


def test_sd3_medium_full_depth_parity(monkeypatch):
    from law_of_vision_representation_in_mllms_amd import sd_weights as SW
    from law_of_vision_representation_in_mllms_amd.sd3_engine import Sd3Engine
    from oracle import sd3 as O3
    _fast_cuda(monkeypatch)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sp = SW.SD3_SPECS["stabilityai/stable-diffusion-3-medium-diffusers"]
    assert sp.core.layers == 24
    wc, wv = SW.synthetic_sd3(sp.core, 161), SW.synthetic_vae(sp.vae, 162)           # all 24 joint blocks (the last one context_pre_only)
    rs = np.random.RandomState(19)
    img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, 256, 256)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((1, 16, 32, 32)).astype(np.float32))
    noise = torch.from_numpy((0.05 * rs.standard_normal((1, 16, 32, 32))).astype(np.float32))
    pe = torch.from_numpy(rs.standard_normal((1, 77 + 256, 4096)).astype(np.float32))
    pe[:, 77:] = 0
    pooled = torch.from_numpy(rs.standard_normal((1, 2048)).astype(np.float32))
    got = Sd3Engine(sp, wc, wv, DEV, up_ft_index=-1).forward(img, pe, t=2, post_noise=post, ddim_noise=noise, pooled=pooled)
    want = O3.sd3_features(sp, wc, wv, img, pe, pooled, post, noise, t=2, up_ft_index=-1)
    ref_bf16 = O3.sd3_features(sp, wc, wv, img, pe, pooled, post, noise, t=2, up_ft_index=-1, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got, want), rel_err(ref_bf16, want)
    assert got.shape == want.shape and e_hip < max(1.5 * e_ref, 2e-2), (e_hip, e_ref)


def test_sd15_full_depth_256px_parity(monkeypatch):
    """SD1.5: the whole VAE encoder + every down block, the mid block and up block 0 of the real UNet at 256 px."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import builder as B
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args = SimpleNamespace(vision_tower='runwayml/stable-diffusion-v1-5', up_ft_index=0, t=261, prompt="a photo of a cat",
                           ensemble_size=1, img_size=256)
    feat = B.build_diffusion_vision_tower(args).vision_tower
    sp = feat.spec
    rs = np.random.RandomState(25)
    img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, 256, 256)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((1, 4, 32, 32)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((1, 4, 32, 32)).astype(np.float32))
    pe = feat.encode_prompt(args.prompt)
    got = feat.forward(img, args.prompt, t=261, up_ft_index=0, ensemble_size=1, post_noise=post, ddim_noise=ddim)     # [c, h, w]
    assert got.shape == (1280, 8, 8)
    wu = {k: v for k, v in feat._wu.items() if not k.startswith(("up_blocks.1", "up_blocks.2", "up_blocks.3"))}
    want = OD.sd_features(sp, wu, feat._wv, img, pe.float().cpu(), post, ddim, t=261)                                 # [1, 64, 1280]
    ref_bf16 = OD.sd_features(sp, wu, feat._wv, img, pe.float().cpu(), post, ddim, t=261, dtype=torch.bfloat16)
    got_tok = got.permute(1, 2, 0).reshape(1, 64, 1280)
    e_hip, e_ref = rel_err(got_tok, want), rel_err(ref_bf16, want)
    assert e_hip < max(1.5 * e_ref, 2e-2), (e_hip, e_ref)


def test_sd15_768px_sweep_launch_shape_parity_and_routes(monkeypatch):
    """SD1.5 at the shape the sweep LAUNCHES - 768-px images, 16 per launch - against the fp32 CPU oracle on one of the
    images (the other 15 are arbitrary), bounded by the oracle's own bf16 run; and the routes the dispatcher picks only at this shape are
    asserted, not assumed: the VAE's 256- / 512-channel 3x3 convolutions in the persistent 256x256 kernel (`conv_256`), its 128-channel
    768^2 layers in the fused GroupNorm + convolution kernel (`conv_halo`), the stride-2 downsample in the 128x128 kernel with GroupNorm
    partial sums from the epilogue (`conv_128_gn`), the 12^2 / 24^2 UNet
    convolutions through deterministic split-K, the VAE's mid-block attention in the wide-head flash kernel, head / tail splits with a row
    offset (`gemm_tail`).  The eager forward is measured (a HIP-graph replay does not pass the dispatcher); the replayed graph must give
    the same features bit for bit."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd import _lib
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import builder as B
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    NB, PX = 16, 768
    args = SimpleNamespace(vision_tower='runwayml/stable-diffusion-v1-5', up_ft_index=0, t=261, prompt="a photo of a cat",
                           ensemble_size=1, img_size=PX)
    feat = B.build_diffusion_vision_tower(args).vision_tower
    sp = feat.spec
    rs = np.random.RandomState(31)
    imgs = torch.from_numpy(rs.uniform(-1, 1, (NB, 3, PX, PX)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((NB, 4, PX // 8, PX // 8)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((NB, 4, PX // 8, PX // 8)).astype(np.float32))
    pe = feat.encode_prompt(args.prompt)
    eng = feat._engine(0)                                                                                             # the SdEngine behind SDFeaturizer.forward(up_ft_index=0)
    eng.graph = False
    _lib.routes(reset=True)
    got = eng.forward(imgs, pe, t=261, ensemble_size=1, post_noise=post, ddim_noise=ddim)                             # [16, 576, 1280], eager
    r = _lib.routes(reset=True)
    print(f"routes of one eager SD1.5 forward at {PX} px x {NB}: {r}")
    assert got.shape == (NB, 576, 1280) and torch.isfinite(got.float()).all()
    # what the sweep's launch shape is tuned for (profiles/round4_sd15_kernel_stats.md): asserted per route
    assert r["conv_256"] >= 16, r            # VAE 256- / 512-channel layers at 384^2 / 192^2 / 96^2: whole rounds of 256x256 tiles
    assert r["conv_halo"] == 4, r            # VAE 128-channel layers at 768^2: GroupNorm + SiLU + convolution in one kernel (conv3x3_halo)
    assert r["conv_c8"] == 1, r              # the VAE's conv_in straight from the pixel tokens (no im2col)
    assert r["conv_128_gn"] >= 1, r          # the stride-2 128-channel downsample: 128x128 kernel, statistics from the epilogue
    assert r["attn_wide"] == 1, r            # the VAE's 512-wide single head: one flash launch for the batch
    assert r["splitk"] >= 4, r               # 12^2 / 24^2 UNet convolutions and projections: few tiles, deep K
    assert r["gemm_tail"] + r["conv_256"] + r["gemm_256"] > 0 and r["attn"] >= 8, r
    eng.graph = True
    again = eng.forward(imgs, pe, t=261, ensemble_size=1, post_noise=post, ddim_noise=ddim)                           # warm-up + capture + replay
    assert torch.equal(again, got)
    k = 5                                                                                                            # the checked image: not the first of the launch
    wu = {kk: v for kk, v in feat._wu.items() if not kk.startswith(("up_blocks.1", "up_blocks.2", "up_blocks.3"))}
    want = OD.sd_features(sp, wu, feat._wv, imgs[k:k + 1], pe.float().cpu(), post[k:k + 1], ddim[k:k + 1], t=261)     # [1, 576, 1280]
    ref_bf16 = OD.sd_features(sp, wu, feat._wv, imgs[k:k + 1], pe.float().cpu(), post[k:k + 1], ddim[k:k + 1], t=261, dtype=torch.bfloat16)
    e_hip, e_ref = rel_err(got[k:k + 1], want), rel_err(ref_bf16, want)
    print(f"SD1.5 @768 px, image {k} of a 16-image launch: HIP {e_hip:.3e}  oracle bf16 {e_ref:.3e}  routes {r}")
    assert e_hip < max(1.5 * e_ref, 2e-2), (e_hip, e_ref)


@pytest.mark.parametrize("tower_id,PX,tol", [('stabilityai/stable-diffusion-2-1', 768, 1.5), ('stabilityai/stable-diffusion-xl-base-1.0', 512, 2.0),
                                            ('lambdalabs/sd-image-variations-diffusers', 768, 1.5)])
def test_sd_family_at_the_sweeps_launch_shape(monkeypatch, tower_id, PX, tol):
    """SD2.1 (Linear proj_in / proj_out, 64-wide heads, 1024-wide prompt), SDXL (2 / 10 transformer layers per attention,
    2048-wide two-encoder prompt) and the image-variation tower (per-image CLIP context) at FULL width and at the resolution and launch size the
    sweep runs them with (sweep.SETTINGS: 768 / 512 / 768 px, 16 images of a 32-image launch's two halves), one image of the launch against the fp32 CPU
    oracle, bounded by the oracle's own bf16 run.  Until round 5 these were pinned at tiny width / 128 px only."""
    from types import SimpleNamespace
    from law_of_vision_representation_in_mllms_amd import _lib
    from law_of_vision_representation_in_mllms_amd.llava.model.multimodal_encoder import builder as B
    monkeypatch.setenv("VISREP_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("VISREP_FAST_SYNTHETIC", "cuda")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    NB = 16
    args = SimpleNamespace(vision_tower=tower_id, up_ft_index=0, t=261, prompt="a photo of a cat", ensemble_size=1, img_size=PX)
    feat = B.build_diffusion_vision_tower(args).vision_tower
    sp = feat.spec
    rs = np.random.RandomState(37)
    imgs = torch.from_numpy(rs.uniform(-1, 1, (NB, 3, PX, PX)).astype(np.float32))
    post = torch.from_numpy(rs.standard_normal((NB, 4, PX // 8, PX // 8)).astype(np.float32))
    ddim = torch.from_numpy(rs.standard_normal((NB, 4, PX // 8, PX // 8)).astype(np.float32))
    _lib.routes(reset=True)
    got = feat.forward(imgs, args.prompt, t=261, up_ft_index=0, ensemble_size=1, post_noise=post, ddim_noise=ddim)     # [NB, c, h, w]
    r = _lib.routes(reset=True)
    assert got.shape[0] == NB and torch.isfinite(got.float()).all()
    C, h, w = got.shape[1:]
    k = 9                                                                                                               # the checked image: inside the launch
    drop = tuple(f"up_blocks.{i}" for i in range(1, 4))
    wu = {kk: v.float().cpu() for kk, v in feat._wu.items() if not kk.startswith(drop)}
    wv = {kk: v.float().cpu() for kk, v in feat._wv.items()}
    if hasattr(feat, "encode_image"):                                                                                   # image-variation tower: the image's own CLIP embedding is the context
        ctx = feat.encode_image(imgs[k:k + 1]).float().cpu()                                                            # [1, 1, cross_dim]
        want = OD.imsd_features(sp, wu, wv, imgs[k:k + 1], ctx, post[k:k + 1], ddim[k:k + 1], t=261, up_ft_index=0, ensemble_size=1)
        ref_bf16 = OD.imsd_features(sp, wu, wv, imgs[k:k + 1], ctx, post[k:k + 1], ddim[k:k + 1], t=261, up_ft_index=0, ensemble_size=1, dtype=torch.bfloat16)
    else:
        ctx = feat.encode_prompt(args.prompt).float().cpu()
        want = OD.sd_features(sp, wu, wv, imgs[k:k + 1], ctx, post[k:k + 1], ddim[k:k + 1], t=261)                      # [1, h w, C]
        ref_bf16 = OD.sd_features(sp, wu, wv, imgs[k:k + 1], ctx, post[k:k + 1], ddim[k:k + 1], t=261, dtype=torch.bfloat16)
    got_tok = got[k].permute(1, 2, 0).reshape(1, h * w, C)
    e_hip, e_ref = rel_err(got_tok, want), rel_err(ref_bf16, want)
    print(f"{tower_id} @{PX} px, image {k} of a {NB}-image launch: HIP {e_hip:.3e}  oracle bf16 {e_ref:.3e}  routes {r}")
    assert e_hip < max(tol * e_ref, 3e-2), (e_hip, e_ref)
    assert r["conv_256"] + r["conv_halo"] >= 8 and r["attn"] >= 4 and r["attn_wide"] >= 1, r       # the full-size routes (counted over the warm-up and the captured pass), not the small-image ones


def test_vae_mid_block_attention_at_9216_tokens():
    """the 768-px VAE's mid-block attention (96 x 96 = 9,216 latent pixels, ONE head of width 512: fp32 score GEMM ->
    softmax_rows -> role-swapped V^T GEMM -> P V GEMM, per image through HBM) was only ever compared at 1,024 tokens.  Here at the real token
    count and head width: a two-level VAE (64 -> 512 channels, one 2x downsample) on a 192-px image has exactly SD1.5's mid block at 768 px;
    posterior moments against the fp32 oracle, bounded by the oracle's own bf16 run."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    vae = SW.VaeSpec(block_out=(64, 512), layers_per_block=1)
    base = SW.tiny_sd_spec()
    sp = SW.SdSpec("vae-mid-9216", unet=base.unet, vae=vae, sched=base.sched, text_len=base.text_len)
    wv = SW.synthetic_vae(vae, 22)
    wu = SW.synthetic_unet(sp.unet, 21, n_up_blocks=1)
    eng = SE.SdEngine(sp, wu, wv, DEV, up_ft_index=0)
    img = torch.from_numpy(np.random.RandomState(3).uniform(-1, 1, (1, 3, 192, 192)).astype(np.float32))
    mom, h, w = eng.vae_moments(img.to(DEV))
    assert (h, w) == (96, 96) and h * w == 9216
    Z = vae.latent_channels
    mean, logvar = untokens(mom[:, :Z], 1, h, w).cpu(), untokens(mom[:, Z: 2 * Z], 1, h, w).cpu()
    m32, l32 = OD.vae_encode_moments(vae, wv, img)                                         # fp32 oracle
    m16, l16 = OD.vae_encode_moments(vae, {k: bf(v) for k, v in wv.items()}, bf(img))      # the same arithmetic with bf16 weights / input
    assert torch.isfinite(mean).all() and torch.isfinite(logvar).all()
    assert rel_err(mean, m32) < max(2.0 * rel_err(m16, m32), 2e-2), (rel_err(mean, m32), rel_err(m16, m32))
    assert rel_err(logvar, l32) < max(2.0 * rel_err(l16, l32), 2e-2), (rel_err(logvar, l32), rel_err(l16, l32))
