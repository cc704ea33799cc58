"""GPU parity tests (-m gpu) for the individual HIP kernels, called through the C ABI (ctypes), against fp32 CPU math."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from demonet_amd.plan import fragment_major

pytestmark = pytest.mark.gpu


def _lib():
    from demonet_amd import _lib as L
    return L, L.lib()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _act(x, act):
    return [lambda v: v, F.relu, F.relu6, F.hardswish][act](x)


PW_CASES = [
    # m, cin, cout, hw, act, residual, se, fp32
    (1000, 16, 16, 100, 0, True, False, False),
    (4096, 16, 64, 4096, 1, False, False, False),
    (777, 72, 24, 777, 0, False, False, False),
    (800, 72, 40, 400, 0, False, True, False),      # SE-scaled projection (two images)
    (513, 120, 40, 171, 0, True, True, False),
    (640, 40, 240, 320, 3, False, False, False),
    (400, 184, 80, 400, 0, True, False, False),
    (300, 112, 672, 100, 3, False, False, False),
    (200, 672, 112, 100, 0, True, True, False),
    (800, 672, 546, 400, 0, False, False, True),    # level-0 class head, fp32 out into [img][anchor][K]
    (50, 128, 24, 25, 0, False, False, True),
    (9, 256, 128, 9, 2, False, False, False),
    (1, 64, 128, 1, 2, False, False, False),
    (16640, 200, 80, 16640, 0, True, False, False),   # BP=64 strip kernel, K not a multiple of 16
    (25600, 112, 672, 400, 3, False, False, False),
    (19200, 480, 112, 400, 0, False, True, False),    # SE scale, 48 images
    (200000, 24, 72, 200000, 1, False, False, False), # big-M tiled kernel
    (150000, 72, 24, 150000, 0, True, False, False),
    # XCD-grouped mapping (>= 8 images: workgroups with equal index % 8 own one contiguous group of images)
    (3600, 672, 546, 400, 0, False, False, True),     # 9 images (groups of 2, last groups empty / half), fp32 head rows
    (1300, 80, 200, 100, 3, False, False, False),     # 13 images of 100 pixels: tiles end inside a group, ragged last group
    (1100, 480, 80, 100, 0, True, True, False),       # strip kernel, 11 images, SE scale + residual
    (17 * 25, 512, 128, 25, 2, False, False, False),  # 17 images of 25 pixels
    (8 * 6400, 24, 72, 6400, 1, False, False, False), # one image per XCD group
    (13 * 1000, 80, 480, 1000, 3, False, False, False),   # streaming kernel (round 3): 15 channel tiles in runs, ragged rows inside the image groups
    (12800, 72, 408, 12800, 1, False, False, False),      # ... K tail of 8 columns, a last channel tile of 24, one image (plain mapping)
]


@pytest.mark.parametrize("m,cin,cout,hw,act,res,se,fp32", PW_CASES)
def test_pointwise_conv(m, cin, cout, hw, act, res, se, fp32):
    L, lib = _lib()
    g = torch.Generator().manual_seed(m * 31 + cin)
    x = torch.randn(m, cin, generator=g).half()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half()
    b = torch.randn(cout, generator=g)
    r = torch.randn(m, cout, generator=g).half() if res else None
    nimg = m // hw
    s = torch.rand(nimg, cin, generator=g) if se else None
    xs = x.float()
    if se:
        xs = (xs.view(nimg, hw, cin) * s[:, None, :]).half().float().view(m, cin)   # kernel rounds the scaled input to fp16
    ref = xs @ w.float().t() + b
    ref = _act(ref, act)
    if res:
        ref = ref + r.float()
    dev = "cuda"
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    rd = r.to(dev) if res else None
    sd = s.to(dev) if se else None
    if fp32:
        extra = 7 * cout                        # emulate writing one level into a larger per-image anchor array
        stride = hw * cout + extra
        out = torch.full((nimg * stride,), -777.0, device=dev)
    else:
        stride = 0
        out = torch.zeros(m, cout, dtype=torch.half, device=dev)
    wfd = torch.from_numpy(fragment_major(w.numpy())).to(dev)     # lets the strip kernel be chosen
    rc = lib.dn_pointwise_conv(_ptr(xd), _ptr(wd), _ptr(wfd), _ptr(bd), _ptr(rd), _ptr(sd), _ptr(out), m, cin, cout, hw, act,
                               int(fp32), stride, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "dn_pointwise_conv")
    torch.cuda.synchronize()
    if fp32:
        o = out.cpu().view(nimg, stride)
        got = o[:, :hw * cout].reshape(m, cout)
        assert (o[:, hw * cout:] == -777.0).all()        # nothing written outside the level's slice
        torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)
    else:
        torch.testing.assert_close(out.cpu().float(), ref.half().float(), rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("m,cin,cout,hw,act", [(25600, 112, 672, 400, 3), (13 * 1000, 80, 480, 1000, 3), (12800, 72, 408, 12800, 1)])
def test_pointwise_streaming_bit_identical_to_direct(m, cin, cout, hw, act, monkeypatch):
    """pw_stream_kernel (weight tiles streamed through two register sets past resident pixel rows) runs pw_direct_kernel's per-tile
    arithmetic: same outputs, bit for bit."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(m + cin)
    x = torch.randn(m, cin, generator=g).half().cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half()
    b = torch.randn(cout, generator=g).cuda()
    wfd = torch.from_numpy(fragment_major(w.numpy())).cuda()
    wd = w.cuda()
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_PW_STREAM", flag)
        out = torch.zeros(m, cout, dtype=torch.half, device="cuda")
        L.check(lib.dn_pointwise_conv(_ptr(x), _ptr(wd), _ptr(wfd), _ptr(b), _ptr(None), _ptr(None), _ptr(out), m, cin, cout, hw, act, 0, 0,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dn_pointwise_conv")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and float(outs[0].float().abs().max()) > 0




@pytest.mark.parametrize("m,cin,cout,hw,act", [
    (25600, 112, 672, 400, 3), (64 * 100, 80, 480, 100, 3), (13 * 1600, 40, 120, 1600, 1), (9 * 6400, 24, 72, 6400, 1),     # the V3 expansions
    (25600 + 37, 112, 672, 25637, 3),                       # one "image": plain mapping, ragged last tile
    (13 * 400, 80, 480, 400, 3),                            # XCD groups of 2 images, the last group short, 25 tiles per image pair
    (8 * 5625, 16, 96, 5625, 2), (19 * 361, 64, 384, 361, 2), (40 * 100, 96, 576, 100, 2), (3300, 128, 264, 3300, 0),    # V2-like: odd maps, K = 128, cout % 32 == 8
    (7 * 1444, 32, 192, 1444, 2), (9 * 400, 48, 72, 400, 1),
])
def test_pointwise_wstat_bit_identical_to_direct(m, cin, cout, hw, act, monkeypatch):
    """pw_wstat_kernel (round 6: weight-stationary waves, x tiles by LDS-DMA, counted waits, 64-byte store pieces) computes every output with
    pw_direct_kernel's arithmetic -- same K order, bias in the reduction, one rounding -- so the two agree bit for bit; pw_direct_kernel is
    the path test_pointwise_conv holds against the fp32 reference. Also against that reference directly."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(m + cin)
    x = torch.randn(m, cin, generator=g).half()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half()
    b = torch.randn(cout, generator=g)
    ref = _act(x.float() @ w.float().t() + b, act).half().float()
    xd, bd, wd = x.cuda(), b.cuda(), w.cuda()
    wfd = torch.from_numpy(fragment_major(w.numpy())).cuda()
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_PW_WSTAT", flag)
        out = torch.full((m + 64, cout), -777.0, dtype=torch.half, device="cuda")     # 64 guard rows behind the tensor
        L.check(lib.dn_pointwise_conv(_ptr(xd), _ptr(wd), _ptr(wfd), _ptr(bd), _ptr(None), _ptr(None), _ptr(out), m, cin, cout, hw, act, 0, 0,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dn_pointwise_conv")
        torch.cuda.synchronize()
        assert bool((out[m:] == -777.0).all())
        outs.append(out[:m])
    assert torch.equal(outs[0], outs[1]) and float(outs[0].float().abs().max()) > 0
    torch.testing.assert_close(outs[1].cpu().float(), ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("m,cin,cout,hw,se,res", [(19200, 480, 112, 400, True, False), (25600 + 37, 672, 112, 25637, False, False),
                                                  (9 * 400, 672, 112, 400, True, True), (6400, 480, 256, 100, False, False),
                                                  (13 * 100 + 0, 960, 160, 100, True, False)])
def test_pointwise_fast_k_loop_bit_identical(m, cin, cout, hw, se, res, monkeypatch):
    """DN_PW_FASTK (round 3, default on): the tiled 1x1 kernel's K loop without bound tests (offsets formed once, clamped rows, named register ring) and, for
    SE-scaled inputs, the scales applied from LDS as the rows are staged instead of behind each load. Same products in the same order, the same
    (half)((float)x * s) rounding: outputs equal bit for bit, for ragged last tiles, XCD-grouped batches and tiles that span two images.
    (DN_PW_LONGK_PF routes the long-K case to the 4-stage form; the strip kernel is switched off so that every shape takes the tiled kernel.)"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(m + cin)
    nimg = m // hw
    x = torch.randn(m, cin, generator=g).half().cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half()
    b = torch.randn(cout, generator=g).cuda()
    s = torch.rand(nimg, cin, generator=g).cuda() if se else None
    r = torch.randn(m, cout, generator=g).half().cuda() if res else None
    wd = w.cuda()
    monkeypatch.setenv("DN_PW_XS", "0")
    monkeypatch.setenv("DN_PW_LONGK_PF", "1")
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_PW_FASTK", flag)
        out = torch.zeros(m, cout, dtype=torch.half, device="cuda")
        L.check(lib.dn_pointwise_conv(_ptr(x), _ptr(wd), _ptr(None), _ptr(b), _ptr(r), _ptr(s), _ptr(out), m, cin, cout, hw, 0, 0, 0,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dn_pointwise_conv")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and float(outs[0].float().abs().max()) > 0

DW_CASES = [
    # n, h, w, c, k, s, act
    (2, 20, 20, 16, 3, 1, 1),
    (2, 21, 19, 64, 3, 2, 1),
    (1, 40, 40, 72, 5, 2, 1),
    (3, 10, 10, 120, 5, 1, 1),
    (2, 5, 5, 256, 3, 2, 2),
    (2, 3, 3, 128, 3, 2, 2),
    (2, 2, 2, 128, 3, 2, 2),
    (4, 1, 1, 128, 3, 1, 2),
    (1, 20, 20, 672, 3, 1, 3),
    (1, 20, 20, 672, 5, 2, 3),
    (9, 10, 10, 120, 5, 1, 1),        # XCD-grouped mapping: 9 images (groups of 2)
    (16, 21, 19, 64, 3, 2, 1),
    (13, 3, 3, 128, 3, 2, 2),
    (3, 37, 23, 120, 5, 1, 3),        # (shapes of the opt-in LDS-tiled 5x5 kernel, round 3) ragged tiles on both axes, one 15-group chunk
    (2, 37, 23, 72, 5, 2, 1),         # ... stride 2, odd map
    (9, 20, 20, 480, 5, 1, 3),        # ... four 15-group chunks, XCD-grouped mapping
    (2, 7, 9, 40, 5, 1, 0),           # ... smaller than one tile, identity activation
]


@pytest.mark.parametrize("n,h,w,c,k,s,act", DW_CASES)
def test_depthwise_conv(n, h, w, c, k, s, act):
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 131 + c + k)
    x = torch.randn(n, c, h, w, generator=g).half()
    wt = (torch.randn(c, 1, k, k, generator=g) / k).half()
    b = torch.randn(c, generator=g)
    pad = (k - 1) // 2
    ref = _act(F.conv2d(x.float(), wt.float(), b, s, pad, 1, c), act)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = wt.view(c, k * k).t().contiguous().cuda()
    ho, wo = ref.shape[-2:]
    out = torch.zeros(n, ho, wo, c, dtype=torch.half, device="cuda")
    rc = lib.dn_depthwise_conv(_ptr(xd), _ptr(wd), _ptr(b.cuda()), _ptr(out), n, h, w, c, k, s, pad, act,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "dn_depthwise_conv")
    torch.cuda.synchronize()
    got = out.cpu().float().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref.half().float(), rtol=4e-3, atol=4e-3)


DENSE_CASES = [
    # n, h, w, cin, cout, k, s, pad, dil, act, zeros
    (2, 19, 19, 64, 256, 3, 1, 1, 1, 1, True),       # 256x256-tile kernel: M = 722 (3 tiles, ragged last), borders on every side
    (1, 38, 38, 128, 512, 3, 1, 1, 1, 1, True),      # two channel tiles, 18 K stages
    (3, 19, 19, 128, 256, 3, 1, 6, 6, 1, True),      # fc6-like: dilation 6, padding 6 (ssd_vgg16.py:86)
    (2, 19, 19, 256, 512, 3, 2, 1, 1, 1, True),      # extras-like: stride 2 (ssd_vgg16.py:60-72)
    (2, 10, 10, 256, 256, 1, 1, 0, 1, 1, True),      # fc7-like 1x1 through the conv entry
    (2, 5, 5, 128, 256, 3, 1, 0, 1, 1, True),        # valid 3x3 (extras tail): 5x5 -> 3x3
    (2, 19, 19, 64, 256, 3, 1, 1, 1, 1, False),      # same problem without the zero block -> the 128x128 / 64x64 tiles
    (2, 21, 17, 32, 96, 3, 1, 1, 1, 0, True),        # cin % 64 != 0, cout not a multiple of 256 -> general path, no activation
    (1, 24, 150, 128, 128, 3, 1, 1, 1, 1, True),     # conv2_2-like: 512 x 128 tile, run of 512 + 2 * 151 rows
    (1, 12, 256, 64, 128, 3, 1, 1, 1, 1, True),      # wide map, one 64-channel iteration: 256 x 128 tile
    (1, 10, 300, 64, 64, 3, 1, 1, 1, 1, True),       # conv1_2-like: 256 x 64 tile, W = 300
    (5, 9, 9, 64, 64, 3, 1, 1, 1, 1, True),          # tiles spanning several small images
    (9, 19, 19, 128, 512, 3, 1, 1, 1, 1, True),      # 13 pixel tiles x 2 channel tiles: XCD-grouped tile order (8 + remainder 5), two K iterations
    (16, 16, 16, 192, 256, 3, 1, 1, 1, 1, True),     # 16 pixel tiles, three K iterations (run buffers refilled twice)
    (3, 40, 56, 64, 128, 3, 1, 1, 1, 1, True),       # weights-resident patch kernel (round 5): 3 images x 12 blocks x 2 channel groups, ragged rows (40 = 2.5 blocks)
    (2, 20, 272, 128, 64, 3, 1, 1, 1, 1, True),      # the same with 128 input channels (two slices per block, 32-channel groups): conv2_2 of ssd512-like, W > 159
    (1, 12, 256, 128, 128, 3, 1, 1, 1, 1, True),     # four 32-channel groups
    (1, 33, 47, 64, 64, 3, 1, 1, 1, 0, True),        # no activation: the streamed patch kernel (the resident one is ReLU-only)
]


@pytest.mark.parametrize("n,h,w,cin,cout,k,s,pad,dil,act,zeros", DENSE_CASES)
def test_dense_conv(n, h, w, cin, cout, k, s, pad, dil, act, zeros):
    """Dense kxk convolution (VGG path) against fp32 F.conv2d on the fp16-rounded operands; result rounded to fp16 once."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 17 + cin + cout + k)
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).half()
    b = torch.randn(cout, generator=g)
    ref = _act(F.conv2d(x.float(), wt.float(), b, s, pad, dil), act)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = wt.permute(0, 2, 3, 1).contiguous().cuda()                 # [cout][ky][kx][cin]
    z = torch.zeros(64, dtype=torch.uint8, device="cuda") if zeros else None
    ho, wo = ref.shape[-2:]
    out = torch.full((n, ho, wo, cout), float("nan"), dtype=torch.half, device="cuda")
    rc = lib.dn_dense_conv(_ptr(xd), _ptr(wd), _ptr(b.cuda()), _ptr(z) if zeros else None, _ptr(out), n, h, w, cin, cout, k, s, pad, dil, act,
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "dn_dense_conv")
    torch.cuda.synchronize()
    got = out.cpu().float().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref.half().float(), rtol=4e-3, atol=4e-3)


def test_bad_arguments_report_errors():
    L, lib = _lib()
    x = torch.zeros(8, 12, dtype=torch.half, device="cuda")
    rc = lib.dn_pointwise_conv(_ptr(x), _ptr(x), None, _ptr(x), None, None, _ptr(x), 8, 12, 8, 8, 0, 0, 0, None)
    assert rc < 0 and b"multiple of 8" in lib.dn_last_error()


EXPDW_CASES = [
    # n, h, w, cin, cexp, cout, k, s, expand, project, pool, res, act1, act2
    (2, 36, 44, 16, 64, 0, 3, 2, True, False, False, False, 1, 1),      # b2-like (8x8 tiles), ragged tile edges
    (1, 40, 40, 24, 72, 0, 3, 1, True, False, False, False, 1, 1),      # b3-like (8x16 tiles), cin padded 24 -> 32, tail 72 = 64 + 8
    (2, 40, 40, 24, 72, 0, 5, 2, True, False, True, False, 1, 1),       # b4-like: 5x5 stride 2 + pooled sums
    (2, 20, 20, 40, 120, 0, 5, 1, True, False, True, False, 1, 1),      # 5x5 stride 1, pooled
    (1, 20, 20, 80, 200, 0, 3, 1, True, False, False, False, 3, 3),     # b8-like, hardswish, 4 chunks
    (2, 20, 20, 112, 672, 0, 3, 1, True, False, True, False, 3, 3),     # widest expand of the backbone, 7 K steps, 11 chunks
    (2, 10, 10, 80, 480, 0, 5, 1, True, False, True, False, 3, 3),      # 5x10 tiles
    (3, 5, 5, 40, 104, 0, 3, 2, True, False, False, False, 2, 2),       # tiny map (5x5 tile), channel tail
    (2, 19, 19, 64, 384, 0, 3, 1, True, False, False, False, 2, 2),     # MobileNetV2-like odd map, relu6
    (2, 3, 3, 128, 256, 0, 3, 2, True, False, False, False, 1, 1),      # extras-like: 3x3 -> 2x2, cin = 128
    (2, 40, 48, 16, 16, 16, 3, 1, False, True, False, True, 0, 1),      # b1-like: depthwise -> project + residual, no expand
    (2, 36, 44, 16, 64, 24, 3, 2, True, True, False, False, 1, 1),      # b2-like: expand -> dw s2 -> project, ragged tiles
    (1, 40, 40, 24, 72, 24, 3, 1, True, True, False, True, 1, 1),       # b3-like: full block with residual, two chunks (64 + 8)
    (2, 38, 38, 32, 96, 40, 3, 1, True, True, False, False, 2, 2),      # V2-like: 2 output channel tiles
    (9, 20, 20, 40, 120, 0, 5, 1, True, False, True, False, 1, 1),      # XCD-grouped mapping (9 images): pooled sums per image
    (12, 36, 44, 16, 64, 24, 3, 2, True, True, False, False, 1, 1),     # grouped, full block
    (8, 20, 20, 80, 200, 0, 3, 1, True, False, False, False, 3, 3),     # grouped, chunks split over workgroups
]


@pytest.mark.parametrize("n,h,w,cin,cexp,cout,k,s,expand,project,pool,res,act1,act2", EXPDW_CASES)
def test_expand_depthwise(n, h, w, cin, cexp, cout, k, s, expand, project, pool, res, act1, act2):
    """The fused inverted-residual stages against the same chain of fp32 CPU ops with the expanded and depthwise activations
    rounded to fp16 where the unfused path stores them."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 7 + cexp)
    x = torch.randn(n, cin, h, w, generator=g).half()
    w1 = (torch.randn(cexp, cin, generator=g) / cin ** 0.5).half() if expand else None
    b1 = torch.randn(cexp, generator=g) * 0.1 if expand else None
    wd = (torch.randn(cexp, 1, k, k, generator=g) / k).half()
    bd = torch.randn(cexp, generator=g) * 0.1
    w3 = (torch.randn(cout, cexp, generator=g) / cexp ** 0.5).half() if project else None
    b3 = torch.randn(cout, generator=g) * 0.1 if project else None
    pad = (k - 1) // 2
    y = x.float()
    if expand:
        y = _act(F.conv2d(y, w1.float()[:, :, None, None], b1), act1).half().float()
    y32 = _act(F.conv2d(y, wd.float(), bd, s, pad, 1, cexp), act2)
    pooled_ref = y32.sum(dim=(2, 3))
    ref = y32.half().float()
    if project:
        ref = F.conv2d(ref, w3.float()[:, :, None, None], b3)
        if res:
            ref = ref + x.float()
    ho, wo = y32.shape[-2:]
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wdd = wd.view(cexp, k * k).t().contiguous().cuda()
    out = torch.zeros(n, ho, wo, cout if project else cexp, dtype=torch.half, device="cuda")
    tiles = lib.dn_expand_depthwise_tiles(ho, wo, s)
    pp = torch.zeros(n, tiles, cexp, device="cuda") if pool else None
    dev = lambda t: t.cuda() if t is not None else None
    w1d, b1d, w3d, b3d, bdd = dev(w1), dev(b1), dev(w3), dev(b3), dev(bd)
    rc = lib.dn_expand_depthwise(_ptr(xd), _ptr(w1d), _ptr(b1d), _ptr(wdd), _ptr(bdd), _ptr(w3d), _ptr(b3d), _ptr(out), _ptr(pp),
                                 n, h, w, cin, cexp, cout, k, s, act1, act2, int(res),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "dn_expand_depthwise")
    torch.cuda.synchronize()
    got = out.cpu().float().permute(0, 3, 1, 2)
    tol = dict(rtol=1e-2, atol=2e-2) if project else dict(rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(got, ref.half().float(), **tol)
    if pool:
        torch.testing.assert_close(pp.cpu().sum(1), pooled_ref, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("n,h,w,cin,cexp,cout,k,s,res,act", [
    (12, 36, 44, 16, 64, 24, 3, 2, False, 1),       # b2-like, XCD-grouped (12 images: ragged last group), ragged tile edges
    (3, 160, 160, 16, 64, 24, 3, 2, False, 1),      # the 160 x 160 block itself, plain mapping: 300 tiles on 300 workgroups ... and
    (40, 160, 160, 16, 64, 24, 3, 2, False, 1),     # ... 4 000 tiles on 512: every workgroup walks 7 - 8 tiles
    (19, 80, 80, 24, 72, 24, 3, 1, True, 1),        # b3-like: 72 channels taken whole, residual from the staged region, 50 tiles per image
    (2, 40, 40, 24, 72, 24, 3, 1, True, 3),         # few tiles: fewer workgroups than resident slots, hardswish
    (9, 38, 38, 32, 64, 32, 3, 1, True, 2),         # another X row width (cin 32), relu6
])
def test_expdw_persistent_bit_identical(n, h, w, cin, cexp, cout, k, s, res, act, monkeypatch):
    """expdw_one_kernel (round 4, default for the single-chunk full blocks): 512 resident workgroups walk the tiles, the next tile's input
    region copied by LDS-DMA under the current tile's depthwise / project stages, the finished tile stored during the next one. Same
    arithmetic at the same rounding points as one workgroup per tile (expdw_kernel, DN_EXPDW_PERSIST=0): outputs equal bit for bit."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(h * 3 + cexp + n)
    x = torch.randn(n, h, w, cin, generator=g).half().cuda()
    w1 = (torch.randn(cexp, cin, generator=g) / cin ** 0.5).half().cuda()
    b1 = (torch.randn(cexp, generator=g) * 0.1).cuda()
    wd = (torch.randn(k * k, cexp, generator=g) / k).half().cuda()
    bd = (torch.randn(cexp, generator=g) * 0.1).cuda()
    w3 = (torch.randn(cout, cexp, generator=g) / cexp ** 0.5).half().cuda()
    b3 = (torch.randn(cout, generator=g) * 0.1).cuda()
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_EXPDW_PERSIST", flag)
        out = torch.full((n, ho, wo, cout), float("nan"), dtype=torch.half, device="cuda")
        for _ in range(2):          # (twice: the second run starts from a used LDS / L2 state)
            L.check(lib.dn_expand_depthwise(_ptr(x), _ptr(w1), _ptr(b1), _ptr(wd), _ptr(bd), _ptr(w3), _ptr(b3), _ptr(out), None,
                                            n, h, w, cin, cexp, cout, k, s, act, act, int(res),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dn_expand_depthwise")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0], outs[1])


def test_expand_depthwise_rejects_unsupported_shapes():
    L, lib = _lib()
    x = torch.zeros(4096, dtype=torch.half, device="cuda")
    f = torch.zeros(4096, device="cuda")
    rc = lib.dn_expand_depthwise(_ptr(x), _ptr(x), _ptr(f), _ptr(x), _ptr(f), None, None, _ptr(x), None, 1, 4, 4, 136, 64, 0, 3, 1, 0, 0, 0, None)
    assert rc < 0 and b"unsupported" in lib.dn_last_error()
