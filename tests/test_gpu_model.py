"""GPU parity tests (-m gpu): the whole HIP path against the oracle and the golden vectors of the real reference."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import ssd_oracle as so
from demonet_amd import models, synth

pytestmark = pytest.mark.gpu

# fp16 storage / fp32 accumulate vs the fp32 CPU path: tolerance on the class logits and box regression.
# north_star: "within 1e-3 on logits" holds for reference-initialised heads (|logit| < 0.3, test_model_heads_small_logits_within_1e3);
# the synthetic heads used here have |logit| up to 13-22, so the bound is stated as atol + rtol * |ref|. Every pair below is set from
# a MEASUREMENT (tools/layer_errors.py, profiles/r02_layer_errors.txt: per-layer table, error = fp16 rounding of every stored
# activation, 60+ layers deep; no layer adds more than the half-ulp of its output): the worst ratio err / (atol + rtol * |ref|)
# over all logits of the measured images is quoted per model and is >= 0.5, i.e. no tolerance is more than 2x what the kernels do.
#   model                      logits max|err| / mean|err|     regression max|err|     worst ratio at the tolerance below
#   ssdlite320_mobilenet_v3    0.059 / 5.3e-3 (max|ref| 13.3)   0.027                   0.86
#   ssd_lite_mobilenet_v2 320  0.078 / 8.6e-3 (13.3)            0.054                   0.86
#   ssd_lite_mobilenet_v2 300  0.109 / 8.5e-3 (16.8)            0.040                   (0.08 + 0.015 |ref| gives 1.21)
#   ssd300_vgg16               0.020 / 2.9e-3 (18.4)            0.013                   0.90
#   ssd512_vgg16               0.028 / 3.6e-3 (22.2)            0.016                   0.78
LOGIT_ATOL, LOGIT_RTOL = 6e-2, 1e-2
LOGIT_MEAN = 9e-3                        # mean|err| bound for the V3 model: 1.6x the measured 5.7e-3
HIT_MIN = 0.97                           # share of the reference's detections reproduced (same label, IoU > 0.9): measured 99.0 - 99.7 %


def _golden(golden_dir, name):
    p = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(p):
        pytest.skip("no golden for " + name)
    return np.load(p)


def _model(name, z=None, **kw):
    ncls = int(z["num_classes"]) if z is not None else kw.pop("num_classes", 91)
    m = getattr(models, name)(num_classes=ncls, **kw)
    models.load_synthetic(m, int(z["weight_seed"]) if z is not None else 0)
    return m.cuda()


def _images(g, seeds):
    W, H = g.size
    return [torch.from_numpy(synth.images(int(s), 1, H, W)[0]).cuda() for s in seeds]


def _postprocess(lib_mod, logits, reg, anchors, hw, st, nt, topk, dets):
    L = lib_mod.lib()
    n, A, K = logits.shape
    ws = torch.empty(L.dn_postprocess_workspace_bytes(n, A, K, topk, dets), dtype=torch.uint8, device="cuda")
    boxes = torch.empty(n, dets, 4, device="cuda")
    scores = torch.empty(n, dets, device="cuda")
    labels = torch.empty(n, dets, dtype=torch.int64, device="cuda")
    counts = torch.empty(n, dtype=torch.int32, device="cuda")
    kept = torch.empty(n, dets, dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = L.dn_postprocess(p(logits), p(reg), p(anchors), n, A, K, float(hw[0]), float(hw[1]), None, float(st), float(nt),
                          int(topk), int(dets), p(boxes), p(scores), p(labels), p(counts), p(kept), p(ws), ws.numel(),
                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
    lib_mod.check(rc, "dn_postprocess")
    torch.cuda.synchronize()
    return boxes.cpu().numpy(), scores.cpu().numpy(), labels.cpu().numpy(), counts.cpu().numpy(), kept.cpu().numpy()


@pytest.fixture(params=["1", "0"], ids=["cutoff-fast-path", "full-path"])
def pp_fast(request, monkeypatch):
    """DN_PP_FAST: 1 = histogram cut-off + fast per-class kernel with device-side fallback (default), 0 = the full per-class
    kernel for every image. The knob is read per call (csrc/common.h dn_knob), so both run in one process."""
    monkeypatch.setenv("DN_PP_FAST", request.param)
    return request.param


def test_postprocess_on_golden_logits_is_index_exact(golden_dir, pp_fast):
    """Isolation test: the reference's own logits/regression in -> kept-box indices bit-exact (fixture is tie-free)."""
    from demonet_amd import _lib
    z = _golden(golden_dir, "ssdlite320_mobilenet_v3_large")
    st, nt, dets, topk = z["post"]
    logits = torch.from_numpy(z["cls_logits_full_0"])[None].cuda()
    reg = torch.from_numpy(z["bbox_regression"][0:1]).cuda()
    anchors = torch.from_numpy(z["anchors"]).cuda()
    b, s, l, c, k = _postprocess(_lib, logits, reg, anchors, (320, 320), st, nt, int(topk), int(dets))
    n = int(c[0])
    assert n == z["det_labels_0"].shape[0]
    assert np.array_equal(l[0, :n], z["det_labels_0"])
    assert np.array_equal(k[0, :n], z["det_anchor_idx_0"])
    np.testing.assert_allclose(s[0, :n], z["det_scores_0"], rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(b[0, :n], z["det_boxes_0"], rtol=1e-5, atol=2e-3)
    assert (b[0, n:] == 0).all() and (s[0, n:] == 0).all()


@pytest.mark.parametrize("n,A,K,topk,dets,st", [
    (3, 500, 7, 40, 30, 0.05),        # ragged: few candidates per class, counts < dets possible
    (2, 3234, 21, 400, 100, 0.02),    # V2-like shape, topk > 320
    (1, 70, 3, 300, 300, 0.0),        # fewer anchors than topk
    (2, 200, 5, 64, 512, 0.01),       # dets at the cap
    (1, 128, 4, 100, 50, 0.9999),     # nothing passes the threshold -> empty output
    (11, 640, 6, 80, 40, 0.03),       # 11 images: the XCD-grouped mapping of the post-process launches
    (2, 5000, 3, 300, 300, 0.01),     # two foreground classes share the 4 x 300 scores above the cut-off: ~600 each > topk, capped in the fast kernel
    (1, 9000, 2, 400, 200, 0.01),     # one class, 800 scores above the cut-off, topk 400 (the ssd512 situation; 512-candidate kernel variant)
    (1, 20000, 2, 400, 100, 0.0),     # more scores above the cut-off than the fast kernel's list holds -> device-side fallback to the full kernel
])
def test_postprocess_random_vs_oracle(n, A, K, topk, dets, st, pp_fast):
    """Same scores/boxes semantics as the oracle on random inputs, including exact score ties (duplicated rows):
    the canonical tie-break (score desc, class asc, anchor asc) must make the index lists identical."""
    from demonet_amd import _lib
    rng = np.random.default_rng(A * 7 + K)
    logits = rng.normal(0, 2.0, (n, A, K)).astype(np.float32)
    reg = rng.normal(0, 1.0, (n, A, 4)).astype(np.float32)
    ctr = rng.uniform(20, 300, (A, 2)).astype(np.float32)
    wh = rng.uniform(10, 120, (A, 2)).astype(np.float32)
    anchors = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    # exact duplicates -> exact score ties AND identical boxes when anchors are duplicated too
    dup = rng.integers(0, A, A // 5)
    src = rng.integers(0, A, A // 5)
    logits[:, dup] = logits[:, src]
    reg[:, dup] = reg[:, src]
    anchors[dup[: len(dup) // 2]] = anchors[src[: len(dup) // 2]]
    nt = 0.5
    dets_o = so.postprocess_detections(torch.from_numpy(logits), torch.from_numpy(reg), torch.from_numpy(anchors), (320, 320),
                                       st, nt, dets, topk, return_intermediates=True)
    b, s, l, c, k = _postprocess(_lib, torch.from_numpy(logits).cuda(), torch.from_numpy(reg).cuda(),
                                 torch.from_numpy(anchors).cuda(), (320, 320), st, nt, topk, dets)
    for i, d in enumerate(dets_o):
        m = so.selection_margins(d["softmax"], d["decoded"], st, nt, topk, dets)
        cnt = int(c[i])
        assert cnt == d["labels"].shape[0]
        risky = min(m["thresh_gap"], m["iou_gap"]) < 1e-5 or (0 < m["topk_gap"] < 1e-5) or (0 < m["final_gap"] < 1e-5) \
            or (0 < m["order_gap"] < 1e-5)
        if not risky:        # exact ties (gap == 0) are fine: canonical order; near-ties could flip with 1-ulp softmax noise
            assert np.array_equal(l[i, :cnt], d["labels"])
            assert np.array_equal(k[i, :cnt], d["anchor_idx"])
        np.testing.assert_allclose(np.sort(s[i, :cnt])[::-1], np.sort(d["scores"])[::-1], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n,A,K,topk,dets,st", [(3, 20000, 2, 400, 100, 0.0), (5, 3234, 91, 300, 300, 0.001)])
def test_fallback_merge_in_the_selection_launch_equals_the_two_launch_form(n, A, K, topk, dets, st, monkeypatch):
    """Round 4 (DN_PP_FUSE_FALLBACK, default 1): behind the cut-off pass the flagged images (here: a class with more scores above the cut-off than the
    fast kernel's list holds; near-uniform scores, where fewer than `dets` boxes survive among the candidates above the cut-off) are redone by the full
    per-class kernel, and the workgroup that finishes an image's last class -- a device-scope ticket -- merges the image in the same launch. Outputs
    must equal the two-launch form (full selection, then a merge launch) exactly, and repeat (the ticket is left at zero)."""
    from demonet_amd import _lib
    rng = np.random.default_rng(A + K)
    scale = 2.0 if K == 2 else 0.01        # K = 91: near-uniform softmax -> every image falls back
    logits = torch.from_numpy(rng.normal(0, scale, (n, A, K)).astype(np.float32)).cuda()
    reg = torch.from_numpy(rng.normal(0, 1.0, (n, A, 4)).astype(np.float32)).cuda()
    ctr = rng.uniform(20, 300, (A, 2)).astype(np.float32)
    wh = rng.uniform(10, 120, (A, 2)).astype(np.float32)
    anchors = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)).cuda()
    res = {}
    for flag in ("0", "1", "1"):
        monkeypatch.setenv("DN_PP_FUSE_FALLBACK", flag)
        res.setdefault(flag, []).append(_postprocess(_lib, logits, reg, anchors, (320, 320), st, 0.5, topk, dets))
    ref = res["0"][0]
    for got in res["1"]:
        for x, y in zip(ref, got):
            assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("name,n,size", [("ssd300_vgg16", 70, 300), ("ssd512_vgg16", 9, 512)])
def test_vgg_head_tiles_of_128_channels_are_bit_identical(name, n, size, monkeypatch):
    """Round 4 (DN_CONV_HEAD_NARROW, default 1): the dense 3x3 heads of the large levels run on 512 x 128 / 256 x 128 tiles of conv_halo_kernel instead of
    256 x 256 (380 / 570 channels fill 3 / 5 narrow channel tiles to 99 % / 89 %, two / three wide ones to 74 %). Every output is ONE accumulator
    walking K in the same order (slice outer, tap inner) whatever the tile: logits and box regressions equal bit for bit."""
    imgs = torch.from_numpy(synth.images(31, n, size, size)).cuda()
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_CONV_HEAD_NARROW", flag)
        m = _model(name, num_classes=91)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
    assert torch.isfinite(res["1"][0]).all()
    assert torch.equal(res["0"][0], res["1"][0]), (res["0"][0] - res["1"][0]).abs().max().item()
    assert torch.equal(res["0"][1], res["1"][1]), (res["0"][1] - res["1"][1]).abs().max().item()


@pytest.mark.parametrize("name,n,size", [("ssd512_vgg16", 3, 512), ("ssd300_vgg16", 5, 300)])
def test_max_pool_in_the_conv_epilogue_is_bit_identical(name, n, size, monkeypatch):
    """DN_CONV_POOL (default 1): a 3x3 conv followed by MaxPool2d(2, 2) (ssd_vgg16.py:34-37 via torchvision vgg16 features) is one launch -- the
    16 x 16 block of conv_patch_kernel (conv1_2, conv2_2) or, since round 4, the row-pair tile of conv_halo_kernel<3,4,4> (conv3_3 of ssd512,
    DN_CONV_HALO_POOL) pools in its epilogue and the full-resolution conv output never reaches memory. A maximum is exact: the head outputs must
    equal the unfused form bit for bit."""
    imgs = torch.from_numpy(synth.images(47, n, size, size)).cuda()
    res = {}
    for key, env in (("unfused", {"DN_CONV_POOL": "0"}), ("patch only", {"DN_CONV_HALO_POOL": "0"}), ("all", {})):
        for k in ("DN_CONV_POOL", "DN_CONV_HALO_POOL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = _model(name, num_classes=91)
        res[key] = [t.clone() for t in m.forward_heads(imgs)]
    for key in ("patch only", "all"):
        assert torch.equal(res["unfused"][0], res[key][0]), (key, (res["unfused"][0] - res[key][0]).abs().max().item())
        assert torch.equal(res["unfused"][1], res[key][1]), key


@pytest.mark.parametrize("name,n,size", [("ssd300_vgg16", 5, 300), ("ssd512_vgg16", 3, 512), ("ssd300_vgg16", 33, 300)])
def test_weights_resident_patch_kernel_is_bit_identical_to_the_streamed_one(name, n, size, monkeypatch):
    """Round 5 (DN_PATCH_RESIDENT, default 1): conv1_2 (+ pool1) and conv2_1 of the VGG models (64 input channels, ssd_vgg16.py:33-37 via torchvision
    vgg16 features[2:7]) run on conv_patch_resident_kernel -- one persistent workgroup per CU with the nine taps resident in LDS, the next block's
    patch requests and the previous block's bias + ReLU + 2 x 2 max-pool (across lanes) + stores riding in the MFMA steps -- instead of the streamed
    conv_patch_kernel. Same products, same accumulation order (tap, then 16-channel step), one rounding: the head outputs are equal bit for bit;
    300 x 300 has ragged blocks on both edges (19 x 19 blocks of 16), 33 images run as two chains."""
    imgs = torch.from_numpy(synth.images(59, n, size, size)).cuda()
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_PATCH_RESIDENT", flag)
        m = _model(name, num_classes=91)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        again = m.forward_heads(imgs)
        assert torch.equal(res[flag][0], again[0]) and torch.equal(res[flag][1], again[1])
    assert torch.equal(res["0"][0], res["1"][0]), (res["0"][0] - res["1"][0]).abs().max().item()
    assert torch.equal(res["0"][1], res["1"][1])


@pytest.mark.parametrize("name,n,size", [("ssd300_vgg16", 5, 300), ("ssd512_vgg16", 3, 512), ("ssd300_vgg16", 33, 300)])
def test_pipelined_stem_is_bit_identical_to_the_sequential_one(name, n, size, monkeypatch):
    """Round 6 (DN_STEM_PIPE, default 1; the fallback behind DN_STEM_SPLIT): conv1_1 of the VGG models (3 -> 64 on the full-size image, ssd_vgg16.py:33 via torchvision vgg16
    features[0], behind the transform's normalisation) on stem_mfma64p_kernel -- the same 32-pixel tiles, fp32 products and accumulation order as
    stem_mfma64_kernel with the next tile's taps requested ahead and the previous tile's epilogue in the shadow of the MFMA chain. Head outputs
    equal bit for bit; 300 x 300 is ragged (2812.5 tiles per image: the last workgroup's tiles run off the image, rows wrap inside a tile)."""
    imgs = torch.from_numpy(synth.images(61, n, size, size)).cuda()
    monkeypatch.setenv("DN_STEM_SPLIT", "0")            # the fp32 kernels: what runs when a model's weights do not fit the split-fp16 kernel
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_STEM_PIPE", flag)
        m = _model(name, num_classes=91)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        again = m.forward_heads(imgs)
        assert torch.equal(res[flag][0], again[0]) and torch.equal(res[flag][1], again[1])
    assert torch.equal(res["0"][0], res["1"][0]), (res["0"][0] - res["1"][0]).abs().max().item()
    assert torch.equal(res["0"][1], res["1"][1])


@pytest.mark.parametrize("name,n,size", [("ssdlite320_mobilenet_v3_large", 3, 320), ("ssd_lite_mobilenet_v2", 3, 300), ("ssd300_vgg16", 3, 300),
                                         ("ssd512_vgg16", 2, 512), ("ssdlite320_mobilenet_v3_large", 9, 320)])
def test_split_fp16_stem_is_within_an_fp16_ulp_of_the_fp32_stem(name, n, size, monkeypatch):
    """Round 6 (DN_STEM_SPLIT, default 1): the first conv (3 input channels on the full-size image behind the transform's normalisation:
    mobilenetv3.py / mobilenetv2.py features[0], ssd_vgg16.py:33) on stem_split_kernel -- every operand as the sum of two fp16 numbers, three fp16
    matrix products with fp32 accumulation -- instead of exact fp32 products (stem3s2_kernel / stem_mfma64p_kernel). The sums differ by fp32
    rounding noise (~1e-7 relative; ~1e-7 absolute once an operand's low half is subnormal), so the fp16 OUTPUT may differ in the last place where
    the sum sits on a rounding boundary: never by more than one unit (or the noise floor of the fp32 sums themselves, 2^-22 of the largest output, for outputs near zero), in < 0.1 % of the values; the head outputs move by less than the tolerances against the CPU path. Image borders, the left-edge shift of row 0 and ragged tiles (300 x 300 -> 150 x 150, 22 500 / 32 tiles) included."""
    kw = {"image_size": size} if name == "ssd_lite_mobilenet_v2" else {}
    imgs = torch.from_numpy(synth.images(67, n, size, size)).cuda()
    monkeypatch.setenv("DN_WS_REUSE", "0")
    res, stem = {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_STEM_SPLIT", flag)
        m = _model(name, num_classes=91, **kw)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        tid = next(nd.out for nd in m.graph.nodes if nd.op == "stem")
        stem[flag] = m.tensor(imgs.shape, tid)
        again = m.forward_heads(imgs)
        assert torch.equal(res[flag][0], again[0]) and torch.equal(res[flag][1], again[1])
    a, b = stem["0"], stem["1"]
    assert torch.isfinite(b.float()).all()
    # distance in fp16 units in the last place: the bit patterns of same-sign fp16 numbers are ordered like the numbers
    ia, ib = a.view(torch.int16).int(), b.view(torch.int16).int()
    same_sign = (ia >= 0) == (ib >= 0)
    d = (ia - ib).abs()
    d = torch.where(same_sign, d, (ia & 0x7FFF) + (ib & 0x7FFF))        # (+0 / -0 or a sign change across zero)
    frac = (d > 0).float().mean().item()
    print("stem outputs differing in the last place: %.4f %%, max distance %d" % (100 * frac, int(d.max())))
    # at most one unit in the last place -- or, where a sum cancels to something small, the absolute rounding noise of the sums themselves: both
    # kernels carry a few 2^-24 of the magnitude of their 28 products (the fp32 kernels in their accumulation, the split kernel in the dropped
    # wl xl term and the rounding of xl), so results near zero differ by that much, which is many fp16 units down there
    scale = a.float().abs().max().item()
    noise = 2.0 ** -22 * scale
    close = (d <= 1) | ((a.float() - b.float()).abs() <= noise)
    print("largest output %.3g, noise floor allowed %.3g, largest difference outside one unit %.3g" %
          (scale, noise, (a.float() - b.float()).abs()[d > 1].max().item() if bool((d > 1).any()) else 0.0))
    assert bool(close.all()) and frac < 0.005, (int(d.max()), frac)
    # borders: first / last rows and columns are as close as the interior
    for sl in (close[:, 0], close[:, -1], close[:, :, 0], close[:, :, -1]):
        assert bool(sl.all())
    dl = (res["0"][0] - res["1"][0]).abs().max().item()
    dr = (res["0"][1] - res["1"][1]).abs().max().item()
    print("head outputs: max |d logits| %.3g  max |d regression| %.3g (logit scale %.3g)" % (dl, dr, res["0"][0].abs().max().item()))
    # a last-place change of 0.1 % of the first layer's outputs is a different draw of the rounding noise of the ~60 fp16 layers behind it: with these
    # weights (logits up to 14) single logits move by up to ~3e-2 and by ~4e-3 on average -- inside the bounds of the path's distance to the CPU
    # path (LOGIT_ATOL / LOGIT_MEAN), which the golden tests check for the path as it now runs
    lim = LOGIT_ATOL + LOGIT_RTOL * res["0"][0].abs().max().item()
    assert dl <= lim and dr <= lim
    assert (res["0"][0] - res["1"][0]).abs().mean().item() <= LOGIT_MEAN


def test_stem_falls_back_to_fp32_when_the_input_range_does_not_fit_fp16(monkeypatch):
    """dn_create checks what stem_split_kernel needs (plan.hip: finite weights, and a normalised input range below 3e4 for pixels in [0, 1]); a model
    whose statistics break that -- image_std 1e-5: inputs up to 6e4, the first conv's weights scaled down to match -- runs the fp32 stem without being
    told to: its head outputs equal the DN_STEM_SPLIT=0 run bit for bit (the split kernel's would differ in the last place of ~0.05 % of the stem's outputs,
    test_split_fp16_stem...), and they are finite."""
    name = "ssdlite320_mobilenet_v3_large"
    std = [1e-5, 1e-5, 1e-5]
    imgs = torch.stack(_images(models.ssdlite320_mobilenet_v3_large(num_classes=91).graph, [911, 912, 913]))
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DN_STEM_SPLIT", flag)
        m = getattr(models, name)(num_classes=91, image_std=std)
        sd = synth.state_dict(m.graph, 0)
        k0 = next(k for k in sd if k.endswith("features.0.0.0.weight"))
        sd[k0] = sd[k0] * np.float32(1e-5)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        m.cuda()
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
    assert torch.isfinite(res["1"][0]).all() and res["1"][0].abs().max().item() > 0.5
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])


def test_model_heads_match_golden(golden_dir):
    z = _golden(golden_dir, "ssdlite320_mobilenet_v3_large")
    m = _model("ssdlite320_mobilenet_v3_large", z)
    imgs = torch.stack(_images(m.graph, z["image_seeds"]))
    logits, reg = m.forward_heads(imgs)
    logits, reg = logits.cpu().numpy(), reg.cpu().numpy()
    ref = z["cls_logits_full_0"]
    err = np.abs(logits[0] - ref)
    print("logits max|err| %.4g  mean|err| %.4g  max|ref| %.3g" % (err.max(), err.mean(), np.abs(ref).max()))
    np.testing.assert_allclose(logits[0], ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    assert err.mean() < LOGIT_MEAN
    np.testing.assert_allclose(logits[1][::7], z["cls_logits_rows_1"], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    np.testing.assert_allclose(reg, z["bbox_regression"], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    # feature pyramid (NHWC fp16 on the device) against the reference's NCHW fp32 samples
    for lvl, tid in enumerate(m.graph.features):
        f = m.tensor(imgs.shape, tid).float().cpu().permute(0, 3, 1, 2).numpy()
        assert tuple(f.shape) == tuple(z[f"feat{lvl}_shape"])
        samp = f.reshape(f.shape[0], -1)[:, ::max(1, f[0].size // 4096)]
        np.testing.assert_allclose(samp, z[f"feat{lvl}_sample"], rtol=1e-2, atol=6e-2)


def test_model_heads_small_logits_within_1e3():
    """north_star tolerance: with reference-scale head weights (logits O(0.3)) the fp16 path is within 1e-3 of the
    fp32 CPU path on the logits."""
    name = "ssdlite320_mobilenet_v3_large"
    m = getattr(models, name)(num_classes=91)
    g = m.graph
    sd = synth.state_dict(g, 0)
    for k in sd:
        if k.startswith("head.") and k.endswith(".1.weight"):
            sd[k] = sd[k] * np.float32(0.05)
        if k.startswith("head.") and k.endswith(".1.bias"):
            sd[k] = sd[k] * np.float32(0.05)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.cuda()
    imgs = _images(g, [77])
    o = so.OracleSSD(name, sd, 91)
    raw = o.forward_raw([i.cpu() for i in imgs])
    logits, reg = m.forward_heads(torch.stack(imgs))
    err = (logits.cpu() - raw["cls_logits"]).abs().max().item()
    print("small-logit config: max|logit| %.3f max|err| %.2e" % (raw["cls_logits"].abs().max(), err))
    assert err < 1e-3


def test_model_heads_reference_init_within_1e3():
    """north_star: "outputs matching the reference PyTorch CPU path within 1e-3 on logits" with the reference's OWN head initialisation
    (_normal_init, ssd_mobilenetv3.py:57-62: every conv of the SSDLite heads -- depthwise 3x3 and 1x1 -- drawn from N(0, 0.03), biases 0;
    the heads' BatchNorm at its constructor defaults) on the calibrated synthetic backbone: logits O(0.1), max |error| < 1e-3 absolute."""
    name = "ssdlite320_mobilenet_v3_large"
    m = getattr(models, name)(num_classes=91)
    g = m.graph
    sd = synth.state_dict(g, 0)
    rng = np.random.default_rng(3)
    for k in sd:
        if not k.startswith("head."):
            continue
        if sd[k].ndim == 4:
            sd[k] = rng.normal(0.0, 0.03, sd[k].shape).astype(np.float32)
        elif k.endswith("running_var") or (k.endswith(".weight") and sd[k].ndim == 1):
            sd[k] = np.ones_like(sd[k])
        elif k.endswith("num_batches_tracked"):
            continue
        else:                                   # conv bias, BN bias, BN running_mean
            sd[k] = np.zeros_like(sd[k])
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    m.cuda()
    imgs = _images(g, [77, 78])
    o = so.OracleSSD(name, sd, 91)
    raw = o.forward_raw([i.cpu() for i in imgs])
    logits, reg = m.forward_heads(torch.stack(imgs))
    err = (logits.cpu() - raw["cls_logits"]).abs().max().item()
    rerr = (reg.cpu() - raw["bbox_regression"]).abs().max().item()
    print("reference-init heads: max|logit| %.3f max|err| %.2e  regression max|err| %.2e" % (raw["cls_logits"].abs().max(), err, rerr))
    assert raw["cls_logits"].abs().max() > 0.02 and err < 1e-3 and rerr < 1e-3


def test_end_to_end_detections_vs_golden(golden_dir):
    """fp16 network + GPU post-process vs the reference's detections. fp16 noise (1e-3 relative) reorders near-equal
    scores, so this is a set comparison: most (label, anchor) pairs must coincide, and the boxes of
    the coinciding ones must agree."""
    z = _golden(golden_dir, "ssdlite320_mobilenet_v3_large")
    m = _model("ssdlite320_mobilenet_v3_large", z)
    imgs = _images(m.graph, z["image_seeds"])
    out = m(imgs)
    assert len(out) == 2 and set(out[0].keys()) == {"boxes", "scores", "labels"}
    for i, d in enumerate(out):
        assert d["labels"].dtype == torch.int64 and d["boxes"].dtype == torch.float32
        n = d["labels"].shape[0]
        assert n == z[f"det_labels_{i}"].shape[0]
        s = d["scores"].cpu().numpy()
        assert (np.diff(s) <= 0).all()
        # match by box: every reference detection should have a same-label detection with IoU > 0.9
        rb, rl = z[f"det_boxes_{i}"], z[f"det_labels_{i}"]
        gb, gl = d["boxes"].cpu().numpy(), d["labels"].cpu().numpy()
        iou = so.box_iou_np(rb, gb)
        same = rl[:, None] == gl[None, :]
        hit = ((iou > 0.9) & same).any(1)
        print(f"image {i}: {hit.mean() * 100:.1f}% of reference detections reproduced")
        assert hit.mean() >= HIT_MIN
        np.testing.assert_allclose(np.sort(s)[::-1][:50], z[f"det_scores_{i}"][:50], rtol=3e-2, atol=1e-3)


def _synthetic_ground_truth(ref, num_classes, seed=77, min_gap=0.03, lo=4, hi=60):
    """Fixed synthetic ground truth for a set of CPU-path detections: per class, the detections above the widest relative score
    gap (>= 8 %) among ranks lo..hi of the class-wide ranking become objects (boxes jittered by up to 8 % of their size, 10 %
    marked difficult), plus one unmatched object per five (a miss for any detector). Random-weight networks produce hundreds of
    near-tied scores per class; cutting at a gap makes the score a statement about which objects are found and how well their
    boxes fit, not about the order of near-ties (measured on the CPU path alone: a 0.5 % score perturbation moves a top-k ground
    truth's mAP by 1-10 points and this one by 0). For the same reason a class only takes part if the NMS outcome of every
    detection above its cut does not hinge on a near-tie: no other anchor of the class scores within 10 % of it (or higher) while
    overlapping it by IoU > 0.35 (greedy NMS keeps whichever of two overlapping boxes scores higher). The margins follow the
    measured logit error of the fp16 path: mean 5e-3, maximum 6e-2, i.e. up to 6 % on a single score; 0.35 is far below the NMS
    threshold of 0.55."""
    rng = np.random.default_rng(seed)
    gt = [{"boxes": [], "labels": [], "difficult": []} for _ in ref]
    for c in range(1, num_classes):
        items = sorted(((float(d["scores"][j]), i, int(j)) for i, d in enumerate(ref) for j in np.nonzero(d["labels"] == c)[0]), reverse=True)
        if len(items) <= lo + 1:
            continue
        sc = np.array([t[0] for t in items])
        gaps = ((sc[:-1] - sc[1:]) / sc[:-1])[lo:hi]
        if gaps.size == 0 or gaps.max() < min_gap:
            continue
        cut = lo + int(np.argmax(gaps)) + 1
        stable = True
        for (score, i, j) in items[:cut]:
            if "softmax" in ref[i]:
                a = int(ref[i]["anchor_idx"][j])
                rivals = np.nonzero(ref[i]["softmax"][:, c] >= 0.97 * score)[0]
                rivals = rivals[rivals != a]
                if rivals.size and float(so.box_iou_np(ref[i]["decoded"][a:a + 1], ref[i]["decoded"][rivals]).max()) > 0.35:
                    stable = False
        if not stable:
            continue            # a detection above the cut sits next to a near-tied overlapping anchor: the class is left out
        for q, (score, i, j) in enumerate(items[:cut]):
            b = ref[i]["boxes"][j].astype(np.float64)
            wh = np.array([b[2] - b[0], b[3] - b[1]] * 2)
            gt[i]["boxes"].append(b + rng.uniform(-0.08, 0.08, 4) * wh)
            gt[i]["labels"].append(c)
            gt[i]["difficult"].append(bool(rng.random() < 0.1))
            if q % 5 == 4:                       # an object nobody detects: a 3 x 3 pixel box in a corner
                gt[i]["boxes"].append(np.array([0.0, 0.0, 3.0, 3.0]))
                gt[i]["labels"].append(c)
                gt[i]["difficult"].append(False)
    for g in gt:
        g["boxes"] = np.array(g["boxes"], dtype=np.float64).reshape(-1, 4)
        g["labels"] = np.array(g["labels"], dtype=np.int64)
        g["difficult"] = np.array(g["difficult"], dtype=bool)
    return gt


def _separated_heads(sd, num_classes, bg=12.0):
    """A 'trained-like' score distribution on top of the synthetic weights: the background logit of every anchor gets a prior of +bg
    (trained SSDs learn a strongly positive background bias: most anchors are background, few (anchor, class) pairs score high, and
    the high scores are spread over decades instead of being hundreds of near-ties per class). Same network, same kernels."""
    import re
    out = {k: v.copy() for k, v in sd.items()}
    for k in out:
        if re.fullmatch(r"head\.classification_head\.module_list\.\d+\.1\.bias", k):
            b = out[k].reshape(-1, num_classes)
            b[:, 0] += np.float32(bg)
            out[k] = b.reshape(-1)
    return out


def test_map_on_fixed_inputs_within_0p1_separated_scores():
    """north_star: "mAP on fixed inputs within 0.1 of the CPU reference". 64 fixed synthetic images; class heads with a background prior
    (_separated_heads: the score distribution of a trained detector -- scores of the kept detections from 0.04 to 0.99, median 0.14);
    a fixed synthetic ground truth built from the CPU path's detections (every detection scoring >= 0.25 is an object, box jittered by
    up to 8 %, 10 % difficult, one undetectable object per five: ~5 300 objects in 67 classes, CPU-path mAP 84); PASCAL VOC AP per
    class with the reference's TP/FP rules (evalrec, pinned to data/voc_eval.py:29-58,116-165 by tests/test_evalrec.py), area and
    11-point metric. Calibrated on the CPU before the first GPU run: logit noise of the fp16 path's measured size (sigma 7e-3 plus a
    4e-2 tail on 0.1 % of the logits) moves this mAP by 0.000 - 0.032. Asserted: |mAP(HIP) - mAP(CPU path)| <= 0.1 in both metrics."""
    from demonet_amd import evalrec
    name = "ssdlite320_mobilenet_v3_large"
    m = getattr(models, name)(num_classes=91)
    sd = _separated_heads(synth.state_dict(m.graph, 0), 91)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.cuda()
    n = 64
    imgs = [torch.from_numpy(synth.images(4000 + i, 1, 320, 320)[0]) for i in range(n)]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ref = so.OracleSSD(name, sd, 91)(imgs)
    rng = np.random.default_rng(77)
    gt = []
    for d in ref:
        boxes, labels, diff = [], [], []
        for q, j in enumerate(np.nonzero(d["scores"] >= 0.25)[0]):
            b = d["boxes"][j].astype(np.float64)
            wh = np.array([b[2] - b[0], b[3] - b[1]] * 2)
            boxes.append(b + rng.uniform(-0.08, 0.08, 4) * wh); labels.append(int(d["labels"][j])); diff.append(bool(rng.random() < 0.1))
            if q % 5 == 4:                       # an object nobody detects
                boxes.append(np.array([0.0, 0.0, 3.0, 3.0])); labels.append(int(d["labels"][j])); diff.append(False)
        gt.append({"boxes": np.array(boxes, dtype=np.float64).reshape(-1, 4), "labels": np.array(labels, dtype=np.int64),
                   "difficult": np.array(diff, dtype=bool)})
    boxes, scores, labels, counts = [t.cpu().numpy() for t in m.forward_batch(torch.stack(imgs).cuda())]
    hip = [{"boxes": boxes[i, :int(counts[i])].copy(), "scores": scores[i, :int(counts[i])].copy(), "labels": labels[i, :int(counts[i])].copy()}
           for i in range(n)]
    nobj = sum(len(g["labels"]) for g in gt)
    for metric07 in (False, True):
        map_ref, ap_ref = evalrec.voc_mean_ap(ref, gt, 0.5, metric07)
        map_hip, ap_hip = evalrec.voc_mean_ap(hip, gt, 0.5, metric07)
        differing = [c for c in ap_ref if abs(ap_ref[c] - ap_hip.get(c, 0.0)) > 1e-9]
        print(f"separated scores, mAP{'07' if metric07 else ''}: CPU path {map_ref:.3f}  HIP {map_hip:.3f}  |d| {abs(map_ref - map_hip):.4f}  "
              f"({len(ap_ref)} classes, {nobj} objects, {len(differing)} classes differ)")
        assert len(ap_ref) >= 40 and nobj >= 2000 and 60.0 < map_ref < 95.0
        assert abs(map_ref - map_hip) <= 0.1


def test_map_near_tie_stress_case_per_class_bounds():
    """The NEAR-TIE stress case of the mAP comparison (the north_star clause itself is test_map_on_fixed_inputs_within_0p1_separated_scores
    below): calibrated random-weight heads, whose detections are hundreds of near-tied scores per class. 64 fixed synthetic images, a fixed synthetic ground truth
    (_synthetic_ground_truth), PASCAL VOC AP per class (evalrec.voc_class_pr / voc_ap, pinned bit for bit to the reference's
    voc_eval: tests/test_evalrec.py) for the HIP detections and for the CPU path's detections, area and 11-point metric.
    MEASURED (64 / 128 / 256 images, 24-44 classes, 400-840 objects, five kernel variants): |mAP difference| 0.10 - 0.28 points on the
    0-100 scale, every time because ONE or TWO objects change rank inside their class: their fp16 scores are 1.5-6 % off the fp32
    scores (the tail of the logit error: mean 5e-3, max 6e-2) and with random-weight heads dozens of detections of other images
    sit inside any such band; a class has 8-30 objects, so one rank change moves its AP by 3-4 points. All other classes reproduce
    their AP to the last digit. Which objects flip changes with every rounding-level change of the kernels (it did between the SE
    variants and again with the bias-in-the-reduction kernels), so the assertion is per class: at most three classes differ at all,
    and a class that differs does so by rank changes of one or two of its objects -- <= 2 / (objects of the class) in the area metric,
    <= one step (100 / 11) of the 11-point metric, which is far coarser (measured: 2 classes x 9.09 points = 0.73 mAP07 points
    from the same two flips that move the area mAP by 0.33). Not the 0.1 of north_star, which presumes the separated scores of a
    trained model; DESIGN section 2 reports this as measured."""
    from demonet_amd import evalrec
    name = "ssdlite320_mobilenet_v3_large"
    m = _model(name, num_classes=91)
    sd = synth.state_dict(m.graph, 0)
    n = 64
    imgs = [torch.from_numpy(synth.images(4000 + i, 1, 320, 320)[0]) for i in range(n)]
    ref, _ = so.OracleSSD(name, sd, 91)(imgs, return_intermediates=True)
    gt = _synthetic_ground_truth(ref, 91)
    hip = []
    for i0 in range(0, n, 64):
        boxes, scores, labels, counts = [t.cpu().numpy() for t in m.forward_batch(torch.stack(imgs[i0:i0 + 64]).cuda())]
        hip += [{"boxes": boxes[i, :int(counts[i])].copy(), "scores": scores[i, :int(counts[i])].copy(), "labels": labels[i, :int(counts[i])].copy()}
                for i in range(boxes.shape[0])]
    for metric07 in (False, True):
        map_ref, ap_ref = evalrec.voc_mean_ap(ref, gt, 0.5, metric07)
        map_hip, ap_hip = evalrec.voc_mean_ap(hip, gt, 0.5, metric07)
        worst = max(abs(ap_ref[c] - ap_hip[c]) for c in ap_ref)
        print("classes whose AP differs:", {c: (round(ap_ref[c], 2), round(ap_hip[c], 2)) for c in ap_ref if abs(ap_ref[c] - ap_hip[c]) > 1e-9})
        print(f"mAP{'07' if metric07 else ''}: CPU path {map_ref:.3f}  HIP {map_hip:.3f}  |d| {abs(map_ref - map_hip):.4f}  "
              f"({len(ap_ref)} classes, {sum(len(g['labels']) for g in gt)} objects, worst class |d| {worst:.3f})")
        assert len(ap_ref) >= 8 and 40.0 < map_ref < 99.0           # a meaningful score: objects found and objects missed
        differing = [c for c in ap_ref if abs(ap_ref[c] - ap_hip[c]) > 1e-9]
        assert len(differing) <= 3                                   # all but a few classes agree exactly
        npos = {c: sum(int((g["labels"] == c).sum()) for g in gt) for c in differing}
        for c in differing:
            bound = 100.0 / 11 + 1e-6 if metric07 else 200.0 / max(npos[c], 1) + 1e-6
            assert abs(ap_ref[c] - ap_hip[c]) <= bound, (c, ap_ref[c], ap_hip[c], npos[c])
        assert abs(map_ref - map_hip) <= (1.2 if metric07 else 0.5)
    # informational: the same with the naive ground truth (top 12 detections per image), which near-tied scores dominate
    rng = np.random.default_rng(5)
    naive = []
    for d in ref:
        top = np.argsort(-d["scores"], kind="stable")[:12]
        b = d["boxes"][top].astype(np.float64)
        wh = np.stack([b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
        naive.append({"boxes": b + rng.uniform(-0.08, 0.08, b.shape) * np.concatenate([wh, wh], 1), "labels": d["labels"][top]})
    print("top-12-per-image ground truth (not asserted): CPU path %.2f, HIP %.2f" % (evalrec.voc_mean_ap(ref, naive)[0], evalrec.voc_mean_ap(hip, naive)[0]))


def test_list_api_batching_and_resize():
    """Different-size images (the reference docstring's use-case, ssd_mobilenetv3.py:172): resize + box back-mapping."""
    name = "ssdlite320_mobilenet_v3_large"
    m = _model(name, num_classes=91)
    g = m.graph
    sd = synth.state_dict(g, 0)
    a = torch.from_numpy(synth.images(5, 1, 320, 320)[0])
    b = torch.from_numpy(synth.uniform(6, "img_b", 3 * 200 * 260).reshape(3, 200, 260))
    out = m([a.cuda(), b.cuda(), a.cuda()])
    assert len(out) == 3
    assert torch.equal(out[0]["boxes"], out[2]["boxes"]) and torch.equal(out[0]["labels"], out[2]["labels"])
    o = so.OracleSSD(name, sd, 91)
    od = o([a, b])
    bb = out[1]["boxes"].cpu().numpy()
    assert bb[:, 0::2].max() <= 260 + 1e-3 and bb[:, 1::2].max() <= 200 + 1e-3      # mapped back to the original size
    iou = so.box_iou_np(od[1]["boxes"], bb)
    same = od[1]["labels"][:, None] == out[1]["labels"].cpu().numpy()[None, :]
    share = ((iou > 0.9) & same).any(1).mean()
    print(f"resized image: {share * 100:.1f}% of the CPU path's detections reproduced")
    assert share >= 0.95            # measured 98-99 %; the resize path adds the fp32 interpolation rounding of the input


@pytest.mark.parametrize("n", [8, 13, 37])
def test_xcd_grouping_is_placement_only(n):
    """Every kernel maps its workgroups to images so that the workgroups with equal (index % 8) -- one XCD on MI355X -- own one
    contiguous group of images (common.h). That is a speed matter only: the outputs must be bit-identical to the plain mapping
    (DN_XCD=0), for batch sizes that are / are not multiples of 8 and with the sub-batch branches (n >= 32)."""
    imgs = torch.from_numpy(synth.images(21, n, 320, 320)).cuda()
    res = {}
    for flag in ("0", "1"):
        os.environ["DN_XCD"] = flag
        try:
            m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
            heads = [t.clone() for t in m.forward_heads(imgs)]
            dets = [t.clone() for t in m.forward_batch(imgs, persistent_input=True)]
            torch.cuda.synchronize()
            res[flag] = heads + dets
        finally:
            del os.environ["DN_XCD"]
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)
    assert int(res["1"][5].sum()) > 0


@pytest.mark.parametrize("n", [2, 19, 40])
def test_se_fold_matches_separate_se_kernel(n):
    """The small squeeze-excitations (c <= 128, squeeze <= 32) run in the prologue of the projection that consumes them
    (pointwise.hip SEF variant, DN_SE_FOLD=1, default) or as their own launch (DN_SE_FOLD=0): same arithmetic up to the order of the
    fp32 FC sums. That last-bit difference of a scale flips a few fp16 roundings in block 4 of 15, and two fp16 runs that differ
    anywhere that early decorrelate: their logits differ by about the fp16 noise of the path itself (measured mean 2.8e-3, max
    3.2e-2 against mean 5.7e-3, max 6e-2 of either run vs the fp32 path) -- so the bound here is that noise, and both variants are
    held to the golden tolerance by the other tests. Tiles that span two images of a group and the last, ragged group are covered
    by the odd batch sizes."""
    imgs = torch.from_numpy(synth.images(66, n, 320, 320)).cuda()
    res = {}
    for flag in ("0", "1"):
        os.environ["DN_SE_FOLD"] = flag
        try:
            m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
            res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        finally:
            del os.environ["DN_SE_FOLD"]
    d = (res["0"][0] - res["1"][0]).abs()
    print(f"n={n}: SE fold vs SE kernel logits max|d| {d.max().item():.4g} mean {d.mean().item():.3g}")
    assert d.max().item() < 6e-2 and d.mean().item() < 5e-3
    assert (res["0"][1] - res["1"][1]).abs().max().item() < 4e-2


@pytest.mark.parametrize("knob", ["DN_SE_IN_DW", "DN_SE_SMALL", "DN_SE_FC8"])
@pytest.mark.parametrize("n", [3, 37])
def test_se_tail_in_depthwise_launch_matches_se_kernel(n, knob):
    """DN_SE_IN_DW=1 (opt-in, measured slower -- plan.hip): the FCs of the large squeeze-excitations run in the last workgroup of
    the pooling depthwise launch (device-scope atomic publish + ticket) instead of the se_fc launch; DN_SE_SMALL=1 does that for the
    small squeeze-excitations only (instead of the projection-prologue fold) and runs their projections on the register-direct
    kernel with the scale applied to its x fragments. Same arithmetic up to the order of the fp32 sums, i.e. the decorrelation
    noise of test_se_fold_matches_separate_se_kernel; a second forward checks that the counters were left at zero.
    DN_SE_FC8 (round 3, default on): the se_fc launch with 16-byte loads, asm-ordered requests and hand-written waits (se_fc8_kernel)
    against the 4-byte form (se_fc_kernel): the same three phases with the K slices cut differently."""
    imgs = torch.from_numpy(synth.images(67, n, 320, 320)).cuda()
    res = {}
    for flag in ("0", "1"):
        os.environ[knob] = flag
        try:
            m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
            first = [t.clone() for t in m.forward_heads(imgs)]
            again = [t.clone() for t in m.forward_heads(imgs)]
            assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])
            res[flag] = first
        finally:
            del os.environ[knob]
    d = (res["0"][0] - res["1"][0]).abs()
    print(f"n={n}: SE tail vs SE kernel logits max|d| {d.max().item():.4g} mean {d.mean().item():.3g}")
    assert d.max().item() < 6e-2 and d.mean().item() < 5e-3
    assert (res["0"][1] - res["1"][1]).abs().max().item() < 4e-2


@pytest.mark.parametrize("n", [64, 37, 3])
def test_head_group_fast_k_loop_bit_identical(n, monkeypatch):
    """DN_PW_FASTK on the grouped head launch (all levels' 1x1 heads, fp32 rows) and on the SE-scaled projections of the backbone: head outputs of the
    whole model equal bit for bit with the general K loop."""
    imgs = torch.from_numpy(synth.images(29, n, 320, 320)).cuda()
    res = {}
    monkeypatch.setenv("DN_HEAD_FUSE", "0")              # the grouped 1x1 head launch (since round 4 the fallback of the fused head launch)
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_PW_FASTK", flag)
        m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])


@pytest.mark.parametrize("name,ncls,kw,n", [("ssdlite320_mobilenet_v3_large", 91, {}, 64), ("ssdlite320_mobilenet_v3_large", 91, {}, 37),
                                           ("ssdlite320_mobilenet_v3_large", 91, {}, 3), ("ssdlite320_mobilenet_v3_large", 21, {}, 9),
                                           ("ssd_lite_mobilenet_v2", 91, {}, 16), ("ssd_lite_mobilenet_v2", 21, {"image_size": 300}, 9)])
def test_fused_head_launch_is_bit_identical(name, ncls, kw, n, monkeypatch):
    """Round 4 (headfuse.hip, default on): the heads of every pyramid level -- depthwise 3x3 + ReLU6 and the 1x1 behind it, class and box head -- run as ONE
    launch with the depthwise computed inside the GEMM's operand staging (`_prediction_block`, ssd_mobilenetv3.py:27-36; generalized_ssd.py:60-74).
    Same arithmetic at the same rounding points as dw_group_kernel + pw_group_kernel (bias first, taps in (ky, kx) order, fp32 accumulate, one rounding
    to fp16; one accumulator per output, K ascending, bias added last): logits and box regressions equal BIT FOR BIT -- batch 64 (XCD grouping, 8 images
    per group), 37 (two sub-batch chains, ragged groups), 3 (plain mapping, tiles spanning images on every level), K = 21 (two channel tiles per wave
    instead of five), the V2 model (other channel counts) and its 300 x 300 form (19 x 19 / 10 x 10 ... maps)."""
    import ctypes
    from demonet_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    size = kw.get("image_size", 320)
    imgs = torch.from_numpy(synth.images(29, n, size, size)).cuda()
    res, launches = {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_HEAD_FUSE", flag)
        m = _model(name, num_classes=ncls, **kw)
        before = raw.dn_debug_head_fused_launches()
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        launches[flag] = raw.dn_debug_head_fused_launches() - before
        again = m.forward_heads(imgs)                       # graph replay
        assert torch.equal(res[flag][0], again[0]) and torch.equal(res[flag][1], again[1])
    assert launches["0"] == 0 and launches["1"] >= 1, launches
    assert torch.isfinite(res["1"][0]).all() and torch.isfinite(res["1"][1]).all()
    assert torch.equal(res["0"][0], res["1"][0]), (res["0"][0] - res["1"][0]).abs().max().item()
    assert torch.equal(res["0"][1], res["1"][1]), (res["0"][1] - res["1"][1]).abs().max().item()


@pytest.mark.parametrize("name,ncls,kw,n", [("ssdlite320_mobilenet_v3_large", 91, {}, 5), ("ssdlite320_mobilenet_v3_large", 91, {}, 37),
                                           ("ssd_lite_mobilenet_v2", 21, {"image_size": 300}, 3), ("ssd_lite_mobilenet_v2", 21, {}, 9)])
def test_first_block_depthwise_in_the_projection_is_bit_identical(name, ncls, kw, n):
    """DN_PW_DW (default 1 = 16-channel blocks, 2 = 32-channel ones too): the depthwise 3x3 of the first inverted-residual block (16 / 32 channels on the 160 x 160 / 150 x 150 map;
    mobilenetv3.py:61-99, mobilenetv2.py:57-84 with expand_ratio 1) is computed straight into the B fragments of the projection behind it
    (pwdirect.hip pw_dw_direct_kernel) with the arithmetic of the stand-alone kernels: head outputs must be bit-identical to the two
    launches. 150 x 150 = 22 500 pixels per image is not a multiple of the 32-row wave tile (tiles straddle images); 37 images also run
    as two sub-batch chains with the XCD grouping."""
    size = kw.get("image_size", 320)
    imgs = torch.from_numpy(synth.images(83, n, size, size)).cuda()
    res = {}
    for flag in ("0", "2"):                 # 2: also the 32-channel block of the V2 model (1, the default: 16 channels only)
        os.environ["DN_PW_DW"] = flag
        try:
            m = _model(name, num_classes=ncls, **kw)
            res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        finally:
            del os.environ["DN_PW_DW"]
    assert torch.equal(res["0"][0], res["2"][0]) and torch.equal(res["0"][1], res["2"][1])
    assert float(res["2"][0].abs().max()) > 1.0


def test_graph_replay_equals_eager():
    m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
    imgs = torch.from_numpy(synth.images(9, 4, 320, 320)).cuda()
    m.set_graph_mode(False)
    e = [t.clone() for t in m.forward_batch(imgs, persistent_input=True)]
    m.set_graph_mode(True)
    for _ in range(3):          # first call captures, later calls replay
        r = [t.clone() for t in m.forward_batch(imgs, persistent_input=True)]
    torch.cuda.synchronize()
    for a, b in zip(e, r):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,hw", [(33, 320), (34, 300)])
def test_sub_batch_branches_equal_separate_forwards(n, hw):
    """n >= 32 images run as two parallel sub-batch launch chains (dn_batch_split). The result must be bit-identical to
    running the two sub-batches as separate single-chain forwards (DN_SPLIT=1) -- eager and replayed, odd sizes and the
    resize path included. (Against ONE unsplit forward only the tolerance holds: tile/kernel choice depends on the row count.)"""
    imgs = torch.from_numpy(synth.images(11, n, hw, hw)).cuda()
    n0 = (n + 1) // 2
    os.environ["DN_SPLIT"] = "1"
    try:
        ref_model = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
        parts = []
        for chunk in (imgs[:n0].contiguous(), imgs[n0:].contiguous()):
            parts.append([t.clone() for t in ref_model.forward_batch(chunk, persistent_input=True)])
        assert ref_model.batch_split(n) == 1
    finally:
        del os.environ["DN_SPLIT"]
    ref = [torch.cat([a, b]) for a, b in zip(*parts)]
    m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
    packed = torch.zeros(n, ref[1].shape[1] + 1, 6, device="cuda")
    m.set_graph_mode(False)
    eager = [t.clone() for t in m.forward_batch(imgs, persistent_input=True, packed=packed)]
    assert m.batch_split(n) == 2 and m.batch_split(16) == 1
    m.set_graph_mode(True)
    for _ in range(3):
        replay = [t.clone() for t in m.forward_batch(imgs, persistent_input=True, packed=packed)]
    torch.cuda.synchronize()
    for a, b, c in zip(ref, eager, replay):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(packed[:, :-1, :4], ref[0]) and torch.equal(packed[:, :-1, 4], ref[1])
    assert torch.equal(packed[:, -1, 0].to(torch.int32), ref[3])
    assert int(ref[3].sum()) > 0
    # and against one unsplit forward: same detections up to the fp16 tolerance of the path
    os.environ["DN_SPLIT"] = "1"
    try:
        whole = [t.clone() for t in _model("ssdlite320_mobilenet_v3_large", num_classes=91).forward_batch(imgs, persistent_input=True)]
    finally:
        del os.environ["DN_SPLIT"]
    assert torch.equal(whole[3], ref[3])
    # (ranks may swap between near-equal scores; the sorted score lists must agree)
    d = (whole[1] - ref[1]).abs()          # fp16 activations: logits agree to ~2e-2 between kernel variants
    assert d.max().item() < 2e-2 and d.mean().item() < 2e-3


@pytest.mark.parametrize("n,h,w", [(3, 320, 320), (2, 500, 400), (33, 240, 320)])
def test_uint8_hwc_input_equals_float_path(n, h, w):
    """decoder output (uint8 HWC) -> /255 -> bilinear resize -> planar, on the device ahead of the stem.
    Must equal the float path fed to_tensor(images) bit for bit (same resize arithmetic), incl. the box back-mapping
    (transform.py:278-292) and the sub-batch branches (n >= 32)."""
    m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
    g = torch.Generator().manual_seed(n * 1000 + h)
    u8 = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g).cuda()
    # ToTensor divides on the CPU (true division); torch's GPU kernel multiplies by 1/255 instead, which differs in the last ulp
    ref_in = (u8.cpu().permute(0, 3, 1, 2).float() / 255).contiguous().cuda()
    ref = [t.clone() for t in m.forward_batch(ref_in, persistent_input=True)]
    for _ in range(2):          # eager capture pass, then graph replay
        got = [t.clone() for t in m.forward_uint8(u8)]
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert int(ref[3].sum()) > 0
    with pytest.raises(ValueError):
        m.forward_uint8(u8.permute(0, 3, 1, 2))          # not HWC


def test_uint8_input_vs_cpu_path():
    """The uint8 HWC input path against the CPU path itself (not only against the float path of this library): a decoder-style image of
    another size goes through /255 + bilinear resize + planar conversion on the device; the CPU path gets to_tensor(image)."""
    name = "ssdlite320_mobilenet_v3_large"
    m = _model(name, num_classes=91)
    sd = synth.state_dict(m.graph, 0)
    g = torch.Generator().manual_seed(4242)
    low = torch.rand(2, 3, 60, 50, generator=g)
    u8 = (torch.nn.functional.interpolate(low, size=(480, 400), mode="bilinear", align_corners=False) * 255).round().clamp(0, 255).to(torch.uint8)
    u8 = u8.permute(0, 2, 3, 1).contiguous()                       # [N, H, W, 3] as a decoder hands it over
    ref = so.OracleSSD(name, sd, 91)([(u8[i].permute(2, 0, 1).float() / 255) for i in range(2)])
    boxes, scores, labels, counts = [t.cpu().numpy() for t in m.forward_uint8(u8.cuda())]
    for i, d in enumerate(ref):
        c = int(counts[i])
        assert abs(c - d["labels"].shape[0]) <= 3
        assert boxes[i, :c, 0::2].max() <= 400 + 1e-3 and boxes[i, :c, 1::2].max() <= 480 + 1e-3     # mapped back to the original size
        iou = so.box_iou_np(d["boxes"], boxes[i, :c])
        share = ((iou > 0.9) & (d["labels"][:, None] == labels[i, :c][None, :])).any(1).mean()
        print(f"uint8 image {i}: {share * 100:.1f}% of the CPU path's detections reproduced")
        assert share >= 0.95


def test_image_mean_std_kwargs_reach_the_kernels():
    """image_mean / image_std given to the factory (ssd_mobilenetv3.py:207-218 {**defaults, **kwargs}) are what the stem normalises
    with: head outputs against the CPU path run with the same non-default statistics."""
    name = "ssdlite320_mobilenet_v3_large"
    mean, std = [0.40, 0.55, 0.62], [0.30, 0.45, 0.60]
    m = getattr(models, name)(num_classes=91, image_mean=mean, image_std=std)
    sd = synth.state_dict(m.graph, 0)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.cuda()
    imgs = _images(m.graph, [901, 902])
    o = so.OracleSSD(name, sd, 91)
    o.mean, o.std = mean, std
    raw = o.forward_raw([i.cpu() for i in imgs])
    logits, reg = (t.cpu() for t in m.forward_heads(torch.stack(imgs)))
    err = (logits - raw["cls_logits"]).abs()
    print(f"non-default mean/std: logits max|err| {err.max().item():.4f} mean {err.mean().item():.5f}")
    assert bool((err <= LOGIT_ATOL + LOGIT_RTOL * raw["cls_logits"].abs()).all()) and err.mean().item() < LOGIT_MEAN
    # and they matter: the default statistics give different logits
    d = _model(name, num_classes=91)
    l0, _ = d.forward_heads(torch.stack(imgs))
    assert (l0.cpu() - raw["cls_logits"]).abs().max().item() > 0.5


@pytest.mark.parametrize("name,kw,n,size", [
    ("ssd_lite_mobilenet_v2", dict(image_size=300, num_classes=21, score_thresh=0.02), 128, 300),     # BASELINE config C3 at full size
    ("ssd512_vgg16", dict(num_classes=91), 32, 512),                                                  # BASELINE config C5 at full size
])
def test_full_size_batches_properties(name, kw, n, size):
    """BASELINE configs C3 (batch 128) and C5 (batch 32) at their full sizes, through properties that do not need the CPU path:
    outputs finite, counts <= detections_per_img, scores sorted, padding zero, and the first / last images and the two around the
    sub-batch boundary reproduce their own batch-1 results (same detections up to the fp16 tolerance between kernel variants)."""
    m = models.load_synthetic(getattr(models, name)(**kw), 0).cuda()
    imgs = torch.from_numpy(synth.images(31, n, size, size)).cuda()
    boxes, scores, labels, counts = [t.clone() for t in m.forward_batch(imgs, persistent_input=True)]
    torch.cuda.synchronize()
    D = m.detections_per_img
    assert bool(torch.isfinite(boxes).all()) and bool(torch.isfinite(scores).all())
    assert int(counts.max()) <= D and int(counts.min()) > 0
    idx = torch.arange(D, device="cuda")[None, :]
    valid = idx < counts[:, None]
    assert bool((scores[:, :-1] >= scores[:, 1:])[valid[:, 1:]].all())              # sorted by score
    assert bool((scores[~valid] == 0).all()) and bool((labels[~valid] == 0).all())  # padding
    assert bool((labels[valid] >= 1).all()) and bool((labels[valid] < m.graph.num_classes).all())
    for i in (0, n // 2 - 1, n // 2, n - 1):
        b1, s1, l1, c1 = [t.clone() for t in m.forward_batch(imgs[i:i + 1].contiguous(), persistent_input=True)]
        c = int(c1[0])
        assert abs(c - int(counts[i])) <= max(2, c // 50)
        iou = so.box_iou_np(b1[0, :c].cpu().numpy(), boxes[i, :int(counts[i])].cpu().numpy())
        same = l1[0, :c].cpu().numpy()[:, None] == labels[i, :int(counts[i])].cpu().numpy()[None, :]
        share = ((iou > 0.9) & same).any(1).mean()
        assert share >= 0.95, (i, share)


def test_error_behaviour_matches_reference():
    m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
    with pytest.raises(ValueError):
        m([torch.zeros(1, 3, 8, 8, device="cuda")[0:1]])                 # 4-d element: transform.py:110-112
    with pytest.raises(TypeError):
        m([torch.zeros(3, 8, 8, dtype=torch.uint8, device="cuda")])      # transform.py:130-134
    m.train()
    with pytest.raises(ValueError):
        m([torch.zeros(3, 8, 8, device="cuda")])                         # generalized_ssd.py:273-274
    m.eval()
    with pytest.raises(RuntimeError):
        m([torch.zeros(3, 320, 320)])                                    # CPU tensors: no fallback path


OTHER_MODELS = [
    # golden name, factory kwargs, logit tolerance (atol, rtol): the measured table at the top of this file
    ("ssd_lite_mobilenet_v2", dict(score_thresh=0.02), (8e-2, 1.5e-2)),
    ("ssd300_vgg16", dict(), (2e-2, 5e-3)),
    ("ssd512_vgg16", dict(), (3e-2, 5e-3)),
]


@pytest.mark.parametrize("name,kw,tol", OTHER_MODELS)
def test_other_model_families_match_golden(golden_dir, name, kw, tol):
    """Config C3 (MobileNetV2 SSDLite remnants) and C5 (VGG16 SSD, dense 3x3 implicit-GEMM convs, max-pool, L2-norm):
    head outputs against the real reference's, then detections as a set."""
    z = _golden(golden_dir, name)
    ncls = int(z["num_classes"])
    if name == "ssd_lite_mobilenet_v2":
        m = models.ssd_lite_mobilenet_v2(num_classes=ncls, **kw)
    else:
        m = getattr(models, name)(num_classes=ncls, **kw)
    models.load_synthetic(m, int(z["weight_seed"]))
    m.cuda()
    st, nt, dpi, topk = z["post"]
    assert abs(m.score_thresh - st) < 1e-9 and m.detections_per_img == int(dpi) and m.topk_candidates == int(topk)
    imgs = _images(m.graph, z["image_seeds"])
    logits, reg = m.forward_heads(torch.stack(imgs))
    logits, reg = logits.cpu().numpy(), reg.cpu().numpy()
    atol, rtol = tol
    ref_rows = z["cls_logits_rows_0"]
    err = np.abs(logits[0][::7] - ref_rows)
    print(f"{name}: logits max|err| {err.max():.4g} mean|err| {err.mean():.4g} max|ref| {np.abs(ref_rows).max():.3g}")
    np.testing.assert_allclose(logits[0][::7], ref_rows, rtol=rtol, atol=atol)
    np.testing.assert_allclose(reg, z["bbox_regression"], rtol=rtol, atol=atol)
    s = z["cls_logits_sum_0"]
    assert abs(np.abs(logits[0]).astype(np.float64).sum() - s[1]) <= 5e-3 * s[1]
    out = m(imgs)
    d = out[0]
    rb, rl = z["det_boxes_0"], z["det_labels_0"]
    gb, gl = d["boxes"].cpu().numpy(), d["labels"].cpu().numpy()
    assert abs(gl.shape[0] - rl.shape[0]) <= max(2, rl.shape[0] // 50)
    iou = so.box_iou_np(rb, gb)
    hit = ((iou > 0.9) & (rl[:, None] == gl[None, :])).any(1)
    print(f"{name}: {hit.mean() * 100:.1f}% of reference detections reproduced")
    assert hit.mean() >= HIT_MIN


def test_packed_gather_payload_matches_outputs():
    """dn_set_packed_output: the [N, D+1, 6] payload written by the merge kernel equals the regular outputs."""
    m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
    imgs = torch.from_numpy(synth.images(11, 3, 320, 320)).cuda()
    D = m.detections_per_img
    packed = torch.full((3, D + 1, 6), -1.0, device="cuda")
    b, s, l, c = [t.clone() for t in m.forward_batch(imgs, persistent_input=True, packed=packed)]
    torch.cuda.synchronize()
    assert torch.equal(packed[:, :D, :4], b) and torch.equal(packed[:, :D, 4], s)
    assert torch.equal(packed[:, :D, 5].to(torch.int64), l) and torch.equal(packed[:, D, 0].to(torch.int32), c)
    b2 = m.forward_batch(imgs, persistent_input=True)[0]            # and it can be switched off again
    assert torch.equal(b2, b)


@pytest.mark.parametrize("ncls,post", [
    (2, {}),                                                                                   # one foreground class: N = 12 / 6 head channels
    (21, dict(score_thresh=0.2, nms_thresh=0.3, detections_per_img=10, topk_candidates=50)),   # kwargs override the defaults (ssd_mobilenetv3.py:217)
    (81, dict(detections_per_img=400, topk_candidates=400)),                                   # more detections than the default buffers
])
def test_num_classes_and_postprocess_kwargs(ncls, post):
    """Factory kwargs of the reference (num_classes, score_thresh, nms_thresh, detections_per_img, topk_candidates): the head
    logits follow the fp32 CPU path within the logit tolerance, and the detections equal the oracle's post-process applied to
    the device's own head outputs (labels exact, scores / boxes to float tolerance)."""
    name = "ssdlite320_mobilenet_v3_large"
    m = getattr(models, name)(num_classes=ncls, **post)
    g = m.graph
    for k, v in post.items():
        assert g.post[k] == v
    sd = synth.state_dict(g, 0)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.cuda()
    imgs = torch.stack(_images(g, [501, 502, 503]))
    o = so.OracleSSD(name, sd, ncls, **post)
    raw = o.forward_raw([i.cpu() for i in imgs])
    logits, reg = m.forward_heads(imgs)
    logits, reg = logits.cpu(), reg.cpu()
    assert logits.shape == raw["cls_logits"].shape and reg.shape == raw["bbox_regression"].shape
    tol = LOGIT_ATOL + LOGIT_RTOL * raw["cls_logits"].abs()
    assert bool(((logits - raw["cls_logits"]).abs() <= tol).all())
    boxes, scores, labels, counts = [t.cpu().numpy() for t in m.forward_batch(imgs)]
    p = g.post
    ref = so.postprocess_detections(logits, reg, raw["anchors"], (g.size[1], g.size[0]), p["score_thresh"], p["nms_thresh"], p["detections_per_img"],
                                    p["topk_candidates"], return_intermediates=True)
    assert boxes.shape[1] == p["detections_per_img"]
    for i, d in enumerate(ref):
        cnt = int(counts[i])
        assert cnt == d["labels"].shape[0] <= p["detections_per_img"]
        mg = so.selection_margins(d["softmax"], d["decoded"], p["score_thresh"], p["nms_thresh"], p["topk_candidates"], p["detections_per_img"])
        risky = min(mg["thresh_gap"], mg["iou_gap"]) < 1e-5 or any(0 < mg[k] < 1e-5 for k in ("topk_gap", "final_gap", "order_gap"))
        if not risky:
            assert np.array_equal(labels[i, :cnt], d["labels"])
            np.testing.assert_allclose(boxes[i, :cnt], d["boxes"], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(np.sort(scores[i, :cnt])[::-1], np.sort(d["scores"])[::-1], rtol=1e-5, atol=1e-7)


def test_vgg_batched_kernels_match_single_image_path():
    """ssd300_vgg16 at batch 70 runs on the 256x256-tile conv kernels in two sub-batch chains; one image at a time runs on the
    small tiles. Same weights, same images: the head logits agree within the fp16 tolerance for images at the start, around the
    sub-batch boundary and at the end of the batch."""
    m = models.load_synthetic(models.ssd300_vgg16(num_classes=91), 0).cuda()
    imgs = torch.from_numpy(synth.images(77, 70, 300, 300)).cuda()
    picks = (0, 1, 34, 35, 69)
    single = [tuple(t[0].cpu().clone() for t in m.forward_heads(imgs[i:i + 1])) for i in picks]
    batched, batched_reg = (t.cpu() for t in m.forward_heads(imgs))
    for (ref, ref_reg), i in zip(single, picks):
        tol = LOGIT_ATOL + LOGIT_RTOL * ref.abs()
        assert bool(((batched[i] - ref).abs() <= tol).all()), i
        # the box heads of the large levels ride in the class heads' launch (channels beyond cout of the last channel tile)
        tol_reg = LOGIT_ATOL + LOGIT_RTOL * ref_reg.abs()
        assert bool(((batched_reg[i] - ref_reg).abs() <= tol_reg).all()), i
    boxes, scores, labels, counts = m.forward_batch(imgs)
    assert bool(torch.isfinite(scores).all()) and int(counts.min()) > 0


def test_v2_at_300_matches_oracle():
    """BASELINE config C3: ssd_lite_mobilenet_v2 at 300 x 300 (odd maps 150 -> 75 -> 38 -> 19 -> 10 -> 5 -> 3 -> 2 -> 1, 3000 anchors):
    head outputs against the fp32 CPU path, detections finite."""
    name = "ssd_lite_mobilenet_v2"
    m = models.ssd_lite_mobilenet_v2(image_size=300, num_classes=21, score_thresh=0.02)
    g = m.graph
    sd = synth.state_dict(g, 0)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.cuda()
    imgs = torch.stack(_images(g, [611, 612]))
    o = so.OracleSSD(name, sd, 21, size=(300, 300), score_thresh=0.02)
    raw = o.forward_raw([i.cpu() for i in imgs])
    logits, reg = (t.cpu() for t in m.forward_heads(imgs))
    assert tuple(logits.shape) == (2, 3000, 21) == tuple(raw["cls_logits"].shape)
    err = (logits - raw["cls_logits"]).abs()
    tol = 1e-1 + 2e-2 * raw["cls_logits"].abs()              # measured at this size: max|err| 0.109, mean 8.5e-3 (table at the top)
    print(f"V2 at 300: logits max|err| {err.max().item():.4f} mean {err.mean().item():.5f}, worst err / tol {(err / tol).max().item():.3f}")
    assert bool((err <= tol).all()) and err.mean().item() < 1.4e-2
    assert bool(((reg - raw["bbox_regression"]).abs() <= 8e-2 + 1.5e-2 * raw["bbox_regression"].abs()).all())
    boxes, scores, labels, counts = m.forward_batch(imgs)
    assert bool(torch.isfinite(scores).all()) and int(counts.max()) <= g.post["detections_per_img"]


def test_hub_model_legacy_two_argument_call():
    """The hub entry's call form (reference hubconf.py / box_head.py:323-381 PostProcess.forward): `model(x[N, 3, S, S], image_shapes)` with a stacked
    4-d batch instead of a list of images; every result dict then lists its keys in the order of box_head.py:379 -- scores, labels, boxes.
    Same detections as the list form on the same images."""
    m = _model("ssd_lite_mobilenet_v2", num_classes=21)
    W, H = m.graph.size
    x = torch.from_numpy(synth.images(31, 3, H, W)).cuda()
    legacy = m(x, [(H, W)] * 3)
    listed = m(list(x.unbind(0)))
    assert len(legacy) == 3
    for a, b in zip(legacy, listed):
        assert list(a.keys()) == ["scores", "labels", "boxes"] and list(b.keys()) == ["boxes", "scores", "labels"]
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(a[k], b[k])
        assert a["boxes"].shape[0] > 0 and a["boxes"].shape[1] == 4


@pytest.mark.parametrize("name,n,size", [("ssd300_vgg16", 64, 300), ("ssd512_vgg16", 16, 512)])
def test_vgg_21_class_heads_on_the_big_tile_path(name, n, size, monkeypatch):
    """VOC-sized heads (21 classes: 126 / 84 class channels per level): the class head passes the narrow-tile test alone (one 128-channel tile)
    and fails it with its box head riding along (126 + 24 = 150 channels). The rider is only attached when the launch still has a tile with it
    (round-4 advice: dn_forward returned DN_E_UNSUPPORTED). Outputs must equal the path without the big head tiles (DN_CONV_HEAD_BIG=0: grouped
    launches) within the fp16 tolerance -- the reduction order of the two tilings differs."""
    imgs = torch.from_numpy(synth.images(53, n, size, size)).cuda()
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DN_CONV_HEAD_BIG", flag)
        m = _model(name, num_classes=21)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
        boxes, scores, labels, counts = m.forward_batch(imgs)
        assert bool(torch.isfinite(scores).all()) and int(counts.min()) > 0 and int(labels.max()) <= 20
    for q in (0, 1):
        tol = LOGIT_ATOL + LOGIT_RTOL * res["0"][q].abs()
        assert bool(((res["1"][q] - res["0"][q]).abs() <= tol).all()), q


@pytest.mark.parametrize("name,n,size", [("ssd300_vgg16", 5, 300), ("ssd512_vgg16", 3, 512)])
def test_vgg_without_workspace_reuse(name, n, size, monkeypatch):
    """DN_WS_REUSE=0 (every tensor its own block: what tools/layer_errors.py runs with) gives every conv output an address, also the ones in
    front of a fused max-pool: the conv + pool launch must pick its kernel by geometry, not by whether an output pointer exists (round-4 advice:
    the forward failed on every VGG model). Same kernels, same arithmetic: bit-identical head outputs."""
    imgs = torch.from_numpy(synth.images(59, n, size, size)).cuda()
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DN_WS_REUSE", flag)
        m = _model(name, num_classes=91)
        res[flag] = [t.clone() for t in m.forward_heads(imgs)]
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])


@pytest.mark.parametrize("name,ncls,n,post", [("ssdlite320_mobilenet_v3_large", 91, 64, {}), ("ssdlite320_mobilenet_v3_large", 91, 37, {}),
                                               ("ssdlite320_mobilenet_v3_large", 91, 3, {}), ("ssdlite320_mobilenet_v3_large", 91, 1, {}),
                                               ("ssdlite320_mobilenet_v3_large", 21, 9, {}),
                                               ("ssdlite320_mobilenet_v3_large", 91, 5, {"score_thresh": 0.05, "detections_per_img": 100, "topk_candidates": 200}),
                                               ("ssdlite320_mobilenet_v3_large", 91, 8, {"score_thresh": 1e-6}),
                                               ("ssd_lite_mobilenet_v2", 21, 40, {"image_size": 300}), ("ssd_lite_mobilenet_v2", 91, 5, {})])
def test_softmax_and_decode_in_the_head_launch_are_bit_identical(name, ncls, n, post, monkeypatch):
    """Round 5 (headfuse.hip SM = true, DN_HEAD_SOFTMAX default 1): softmax over the classes, decode_single + clip and the score-histogram rows
    (generalized_ssd.py:354,362-363; _utils.py:187-224) run in the epilogue of the fused head launch -- the logits never reach memory and
    softmax_decode_kernel's launch is gone. Same arithmetic from the same accumulators (post_math.h is shared by both kernels), so the
    detections -- boxes, scores, labels, counts -- equal the logit-writing path + softmax_decode_kernel BIT FOR BIT, and so does the cut-off
    the histogram rows produce (the heaviest-first class order and tau only steer the work, but a wrong row table would show up as fallbacks
    or missing candidates). Only the levels with >= 32 pixels per image take the epilogue; the small ones keep their logits and get their
    softmax from a few tiles of softmax_decode_kernel (one score array, one histogram-row table for both). Batch 64 (XCD grouping: 8 images per
    group, tiles spanning two images), 37 (two chains, ragged groups), 3 and 1 (plain mapping), K = 21 (126 channels per pixel: other row
    geometry), non-default thresholds (a clamped / shifted histogram range), the V2 model at 300 x 300 (19 x 19 / 10 x 10 maps; its last level
    is a plain 1x1 conv outside the fused launch) and at 320."""
    import ctypes
    from demonet_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    size = post.get("image_size", 320)
    imgs = torch.from_numpy(synth.images(83, n, size, size)).cuda()
    res, launches = {}, {}
    monkeypatch.setenv("DN_HEAD_SOFTMAX_MINN", "1")          # (by default the epilogue is used from 32 images per chain up)
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_HEAD_SOFTMAX", flag)
        m = _model(name, num_classes=ncls, **post)
        before = raw.dn_debug_head_softmax_launches()
        res[flag] = [t.clone() for t in m.forward_batch(imgs)]
        launches[flag] = raw.dn_debug_head_softmax_launches() - before
        again = m.forward_batch(imgs)                       # graph replay
        for x, y in zip(res[flag], again):
            assert torch.equal(x, y)
    assert launches["0"] == 0 and launches["1"] >= 1, launches
    assert int(res["1"][3].min()) > 0
    for q, what in enumerate(("boxes", "scores", "labels", "counts")):
        assert torch.equal(res["0"][q], res["1"][q]), what


def test_softmax_in_the_head_launch_feeds_the_same_cut_off(monkeypatch):
    """The histogram rows of the fused epilogue (one per 32-pixel half tile and image, HistRows) must add up to the same per-image histogram and
    per-class counts as softmax_decode_kernel's rows: tau and the class order are read back through the kept anchors' scores -- a forward whose
    cut-off is too high flags images for the fallback, too low costs time only; so compare the kept-anchor lists of the two paths with the
    fallback forced off-limits: DN_PP_FAST=1 results must equal DN_PP_FAST=0 (no cut-off at all) on both paths."""
    imgs = torch.from_numpy(synth.images(89, 16, 320, 320)).cuda()
    res = {}
    monkeypatch.setenv("DN_HEAD_SOFTMAX_MINN", "1")
    for sm in ("0", "1"):
        for fast in ("0", "1"):
            monkeypatch.setenv("DN_HEAD_SOFTMAX", sm)
            monkeypatch.setenv("DN_PP_FAST", fast)
            m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
            res[sm, fast] = [t.clone() for t in m.forward_batch(imgs)]
    for key in (("0", "1"), ("1", "0"), ("1", "1")):
        for q in range(4):
            assert torch.equal(res["0", "0"][q], res[key][q]), (key, q)


@pytest.mark.parametrize("n", [64, 33, 5])
def test_cut_off_in_the_last_softmax_tile_equals_the_cut_off_launch(n, monkeypatch):
    """Round 5: with the large levels' scores from the head launch, the cut-off (tau, class order, fallback flags) runs in whichever of an image's
    small-level softmax tiles finishes last (ticket per image, postprocess.hip tau_body) instead of a launch of its own. Integer sums of the
    same rows: detections bit-identical to the separate launch (DN_PP_FOLD_TAU=0), on the first forward, on graph replays (the tickets must be
    back at zero), and with no cut-off at all (DN_PP_FAST=0)."""
    imgs = torch.from_numpy(synth.images(101, n, 320, 320)).cuda()
    monkeypatch.setenv("DN_HEAD_SOFTMAX_MINN", "1")
    res = {}
    for fold, fast in (("0", "1"), ("1", "1"), ("1", "0")):
        monkeypatch.setenv("DN_PP_FOLD_TAU", fold)
        monkeypatch.setenv("DN_PP_FAST", fast)
        m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
        res[fold, fast] = [t.clone() for t in m.forward_batch(imgs)]
        for _ in range(3):
            again = m.forward_batch(imgs)
            for q in range(4):
                assert torch.equal(res[fold, fast][q], again[q]), (fold, fast, q)
    for key in (("1", "1"), ("1", "0")):
        for q in range(4):
            assert torch.equal(res["0", "1"][q], res[key][q]), (key, q)
    assert int(res["1", "1"][3].sum()) > 0


def test_c2_full_batch_against_the_cpu_path():
    """BASELINE config C2 at its full size with default knobs, tied to the CPU path directly (generalized_ssd.py:271-349): until round 6 every
    test that ran the V3 model at 64 images compared the HIP path with itself, and the 64-image launch forms -- softmax / decode in the fused
    head launch's epilogue, one chain of 64 through ForwardPipeline, two sub-batch chains through forward_batch, XCD groups of 8 images --
    reached the oracle only through bit-identity with a path checked at 2 images. Images 0, 31, 32, 63 (first, last, and both sides of the
    sub-batch boundary): head logits within the logit tolerance of the fp32 CPU path, and the detections of BOTH 64-image launch forms
    reproduce >= 97 % of the CPU path's detections with matching counts."""
    from demonet_amd.pipeline import ForwardPipeline
    name, n, picks = "ssdlite320_mobilenet_v3_large", 64, (0, 31, 32, 63)
    m = _model(name, num_classes=91)
    sd = synth.state_dict(m.graph, 0)
    imgs = torch.from_numpy(synth.images(2024, n, 320, 320)).cuda()
    o = so.OracleSSD(name, sd, 91)
    ref, raw = o([imgs[i].cpu() for i in picks], return_intermediates=True)
    logits, reg = (t.cpu() for t in m.forward_heads(imgs))
    assert tuple(logits.shape) == (n, m.graph.num_anchors(), 91)
    err = (logits[list(picks)] - raw["cls_logits"]).abs()
    tol = LOGIT_ATOL + LOGIT_RTOL * raw["cls_logits"].abs()
    print(f"C2 batch 64, images {picks}: logits max|err| {err.max().item():.4f} mean {err.mean().item():.5f}, worst err / tol {(err / tol).max().item():.3f}")
    assert bool((err <= tol).all()) and err.mean().item() < LOGIT_MEAN
    assert bool(((reg[list(picks)] - raw["bbox_regression"]).abs() <= 6e-2 + 1e-2 * raw["bbox_regression"].abs()).all())
    two_chains = [t.clone() for t in m.forward_batch(imgs, persistent_input=True)]
    assert m.batch_split(n) == 2
    pipe = ForwardPipeline(m, n, depth=3, chains=1)
    try:
        t = pipe.submit(imgs, persistent_input=True)
        one_chain = [x.clone() for x in pipe.result(t)]
    finally:
        pipe.close()
    torch.cuda.synchronize()
    for form, (boxes, scores, labels, counts) in (("forward_batch, two chains of 32", two_chains), ("ForwardPipeline, one chain of 64", one_chain)):
        boxes, scores, labels, counts = boxes.cpu().numpy(), scores.cpu().numpy(), labels.cpu().numpy(), counts.cpu().numpy()
        for j, i in enumerate(picks):
            d = ref[j]
            c = int(counts[i])
            assert c == d["labels"].shape[0], (form, i, c, d["labels"].shape[0])
            iou = so.box_iou_np(d["boxes"], boxes[i, :c])
            share = ((iou > 0.9) & (d["labels"][:, None] == labels[i, :c][None, :])).any(1).mean()
            print(f"{form}: image {i}: {share * 100:.1f}% of the CPU path's detections reproduced")
            assert share >= HIT_MIN, (form, i, share)


def test_head_outputs_are_refused_after_a_full_forward():
    """dn_head_outputs is valid behind dn_forward_heads only (include/demonet_hip.h): dn_forward may finish the large levels inside the head
    launch and never write their logits. Round-5 advice: the call must say so instead of handing out pointers to stale rows."""
    from demonet_amd import _lib
    m = _model("ssdlite320_mobilenet_v3_large", num_classes=91)
    imgs = torch.from_numpy(synth.images(5, 40, 320, 320)).cuda()
    logits, _ = m.forward_heads(imgs)                       # dn_forward_heads + dn_head_outputs: fine
    assert bool(torch.isfinite(logits).all())
    m.forward_batch(imgs)
    torch.cuda.synchronize()
    b = m._buffers_for(40, 320, 320, imgs.device)
    pl, pr = C.c_void_p(), C.c_void_p()
    rc = _lib.lib().dn_head_outputs(C.c_void_p(m._handle), C.c_void_p(b["ws"].data_ptr()), 40, C.byref(pl), C.byref(pr))
    assert rc < 0 and b"dn_forward_heads" in _lib.lib().dn_last_error()
    logits2, _ = m.forward_heads(imgs)                      # ... and valid again behind the next heads-only forward
    assert torch.equal(logits, logits2)
