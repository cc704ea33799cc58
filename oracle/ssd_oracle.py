"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (PyTorch fp32 / numpy float32) restatement of the reference's SSD inference forward pass
(zhiqwang/demonet, `SSD.forward` in eval mode). Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file; the product (demonet_amd/) never does and
fails loudly if its HIP library is missing.

Pinning status:
  * network, anchors, softmax, box decode, per-class threshold/top-k: PINNED against the real
    reference run in the authoring container (oracle/run_reference.py; goldens in tests/golden/,
    test in tests/test_oracle.py).
  * clip_boxes_to_image / nms / batched_nms live in torchvision (third-party, unpinned version,
    absent from /root/reference and from this image): restated from the published algorithm
    (per-class greedy NMS, IoU = inter/(a+b-inter), suppress when IoU > thr, kept indices in
    global score-descending order). PARITY UNPINNED for that step; fixtures assert tie-freeness
    and an IoU margin (3e-6 .. 2e-5), so any torchvision that runs NMS as a per-class loop would agree;
    the coordinate-offset form of batched_nms (offsets up to ~29 000 px cost ~2e-3 px of float32
    resolution) is outside those margins.

All `reference:` citations are relative to /root/reference/demonet/models/.
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def make_divisible(v, divisor=8, min_value=None):
    """reference: mobilenetv2.py:16-29"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _act(x, act):
    if act == "HS":
        return F.hardswish(x)
    if act == "RE":
        return F.relu(x)
    if act == "R6":
        return F.relu6(x)
    assert act == "ID"
    return x


def conv_bn_act(x, sd, p, k, stride, groups, act, eps, dil=1):
    """ConvBNActivation: conv(bias=False, pad=(k-1)//2*dil) -> BN(eval) -> act.  reference: mobilenetv2.py:32-55"""
    pad = (k - 1) // 2 * dil
    x = F.conv2d(x, sd[p + ".0.weight"], None, stride, pad, dil, groups)
    x = F.batch_norm(x, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                     False, 0.0, eps)
    return _act(x, act)


def squeeze_excitation(x, sd, p):
    """reference: mobilenetv3.py:22-40"""
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.conv2d(s, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    s = F.relu(s)
    s = F.conv2d(s, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])
    s = F.hardsigmoid(s)
    return s * x


# reference: mobilenetv3.py:198-214, reduce_divider = 2 (reduced tail; ssd_mobilenetv3.py:193)
V3_LARGE_REDUCED = [
    # cin, k, exp, cout, se, act, stride
    (16, 3, 16, 16, False, "RE", 1),
    (16, 3, 64, 24, False, "RE", 2),
    (24, 3, 72, 24, False, "RE", 1),
    (24, 5, 72, 40, True, "RE", 2),
    (40, 5, 120, 40, True, "RE", 1),
    (40, 5, 120, 40, True, "RE", 1),
    (40, 3, 240, 80, False, "HS", 2),
    (80, 3, 200, 80, False, "HS", 1),
    (80, 3, 184, 80, False, "HS", 1),
    (80, 3, 184, 80, False, "HS", 1),
    (80, 3, 480, 112, True, "HS", 1),
    (112, 3, 672, 112, True, "HS", 1),
    (112, 5, 672, 80, True, "HS", 2),
    (80, 5, 480, 80, True, "HS", 1),
    (80, 5, 480, 80, True, "HS", 1),
]


def v3_inverted_residual(x, sd, p, cfg, eps):
    """reference: mobilenetv3.py:61-99 (p = key prefix of `.block`)"""
    cin, k, exp, cout, use_se, act, stride = cfg
    j = 0
    y = x
    if exp != cin:
        y = conv_bn_act(y, sd, f"{p}.{j}", 1, 1, 1, act, eps)
        j += 1
    y = conv_bn_act(y, sd, f"{p}.{j}", k, stride, exp, act, eps)
    j += 1
    if use_se:
        y = squeeze_excitation(y, sd, f"{p}.{j}")
        j += 1
    y = conv_bn_act(y, sd, f"{p}.{j}", 1, 1, 1, "ID", eps)
    if stride == 1 and cin == cout:
        y = y + x
    return y


def ssdlite_v3_features(sd, x, eps=1e-3) -> List[torch.Tensor]:
    """SSDLiteFeatureExtractorMobileNet.forward.  reference: ssd_mobilenetv3.py:98-132"""
    f0 = "backbone.features.0"
    x = conv_bn_act(x, sd, f"{f0}.0", 3, 2, 1, "HS", eps)                 # mobilenetv3.py:141
    for i, cfg in enumerate(V3_LARGE_REDUCED[:12], start=1):
        x = v3_inverted_residual(x, sd, f"{f0}.{i}.block", cfg, eps)
    cin, k, exp, cout, use_se, act, stride = V3_LARGE_REDUCED[12]
    x = conv_bn_act(x, sd, f"{f0}.13", 1, 1, 1, act, eps)                 # C4 expansion (ssd_mobilenetv3.py:106)
    feats = [x]
    f1 = "backbone.features.1"
    x = conv_bn_act(x, sd, f"{f1}.0.1", k, stride, exp, act, eps)         # C4 dw/SE/project (:107)
    x = squeeze_excitation(x, sd, f"{f1}.0.2")
    x = conv_bn_act(x, sd, f"{f1}.0.3", 1, 1, 1, "ID", eps)
    for i, cfg in enumerate(V3_LARGE_REDUCED[13:], start=1):
        x = v3_inverted_residual(x, sd, f"{f1}.{i}.block", cfg, eps)
    x = conv_bn_act(x, sd, f"{f1}.3", 1, 1, 1, "HS", eps)                 # last 1x1: 80 -> 480
    feats.append(x)
    for i in range(4):                                                   # _extra_block (:39-54)
        p = f"backbone.extra.{i}"
        mid = sd[f"{p}.0.0.weight"].shape[0]
        x = conv_bn_act(x, sd, f"{p}.0", 1, 1, 1, "R6", eps)
        x = conv_bn_act(x, sd, f"{p}.1", 3, 2, mid, "R6", eps)
        x = conv_bn_act(x, sd, f"{p}.2", 1, 1, 1, "R6", eps)
        feats.append(x)
    return feats


def scoring_head_permute(results: torch.Tensor, num_columns: int) -> torch.Tensor:
    """(N, A*K, H, W) -> (N, HWA, K).  reference: generalized_ssd.py:66-71"""
    N, _, H, W = results.shape
    results = results.view(N, -1, num_columns, H, W).permute(0, 3, 4, 1, 2)
    return results.reshape(N, -1, num_columns)


def ssdlite_head(sd, feats, num_classes, eps=1e-3) -> Dict[str, torch.Tensor]:
    """SSDLiteHead.  reference: ssd_mobilenetv3.py:27-36,65-95"""
    out = {}
    for name, cols, key in (("regression_head", 4, "bbox_regression"), ("classification_head", num_classes, "cls_logits")):
        res = []
        for lvl, f in enumerate(feats):
            p = f"head.{name}.module_list.{lvl}"
            y = conv_bn_act(f, sd, f"{p}.0", 3, 1, f.shape[1], "R6", eps)
            y = F.conv2d(y, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"])
            res.append(scoring_head_permute(y, cols))
        out[key] = torch.cat(res, dim=1)
    return out


# ---------------------------------------------------------------------------------------------
# MobileNetV2 legacy path (hub `ssd_lite_mobilenet_v2`)
# ---------------------------------------------------------------------------------------------
def v2_inverted_residual(x, sd, p, inp, oup, stride, hidden, eps=1e-5):
    """reference: backbone.py:81-119 / mobilenetv2.py:62-100 (p = prefix of the block, `.conv` inside)"""
    j = 0
    y = x
    if hidden != inp:
        y = conv_bn_act(y, sd, f"{p}.conv.{j}", 1, 1, 1, "R6", eps)
        j += 1
    y = conv_bn_act(y, sd, f"{p}.conv.{j}", 3, stride, hidden, "R6", eps)
    j += 1
    y = F.conv2d(y, sd[f"{p}.conv.{j}.weight"])
    b = f"{p}.conv.{j + 1}"
    y = F.batch_norm(y, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"], False, 0.0, eps)
    if stride == 1 and inp == oup:
        y = x + y
    return y


V2_SETTING = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]


def ssdlite_v2_features(sd, x, eps=1e-5) -> List[torch.Tensor]:
    """MobileNetWithExtraBlocks.  reference: backbone.py:45-78, mobilenetv2.py:138-168"""
    fb = "backbone.body"
    x = conv_bn_act(x, sd, f"{fb}.0", 3, 2, 1, "R6", eps)
    feats = []
    idx, cin = 1, 32
    for t, c, n, s in V2_SETTING:
        for i in range(n):
            x = v2_inverted_residual(x, sd, f"{fb}.{idx}", cin, c, s if i == 0 else 1, int(round(cin * t)), eps)
            cin = c
            if idx == 13:
                feats.append(x)
            idx += 1
    x = conv_bn_act(x, sd, f"{fb}.18", 1, 1, 1, "R6", eps)
    feats.append(x)
    cin = 1280
    for i, (oc, ratio) in enumerate(zip([512, 256, 256, 64], [0.2, 0.25, 0.5, 0.25])):
        x = v2_inverted_residual(x, sd, f"backbone.extra_blocks.{i}", cin, oc, 2, int(round(cin * ratio)), eps)
        feats.append(x)
        cin = oc
    return feats


def multibox_lite_head(sd, feats, num_classes, eps=1e-5) -> Dict[str, torch.Tensor]:
    """MultiBoxLiteHead + concat_box_prediction_layers.  reference: box_head.py:24-56,107-146"""
    out = {}
    for name, cols, key in (("bbox_pred", 4, "bbox_regression"), ("cls_logits", num_classes, "cls_logits")):
        res = []
        for lvl, f in enumerate(feats):
            p = f"head.{name}.{lvl}"
            if lvl < len(feats) - 1:
                y = F.conv2d(f, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], 1, 1, 1, f.shape[1])
                y = F.batch_norm(y, sd[f"{p}.1.running_mean"], sd[f"{p}.1.running_var"], sd[f"{p}.1.weight"],
                                 sd[f"{p}.1.bias"], False, 0.0, eps)
                y = F.relu6(y)
                y = F.conv2d(y, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"])
            else:
                y = F.conv2d(f, sd[f"{p}.weight"], sd[f"{p}.bias"])
            res.append(scoring_head_permute(y, cols))
        out[key] = torch.cat(res, dim=1)
    return out


# ---------------------------------------------------------------------------------------------
# VGG16 SSD path
# ---------------------------------------------------------------------------------------------
VGG_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]


def ssd_vgg_features(sd, x, highres: bool) -> List[torch.Tensor]:
    """SSDFeatureExtractorVGG.forward.  reference: ssd_vgg16.py:30-109"""
    idx, pools, key = 0, 0, "backbone.features"
    feats = []
    for v in VGG_CFG_D:
        if v == 'M':
            pools += 1
            if pools == 4:
                rescaled = sd["backbone.scale_weight"].view(1, -1, 1, 1) * F.normalize(x)   # :101
                feats.append(rescaled)
                key, idx = "backbone.extra.0", 0
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=(pools == 3))                            # :37
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd[f"{key}.{idx}.weight"], sd[f"{key}.{idx}.bias"], 1, 1))
            idx += 2
    fc = f"backbone.extra.0.{idx}"
    x = F.max_pool2d(x, 3, 1, 1)                                                            # :85
    x = F.relu(F.conv2d(x, sd[f"{fc}.1.weight"], sd[f"{fc}.1.bias"], 1, 6, 6))               # :86
    x = F.relu(F.conv2d(x, sd[f"{fc}.3.weight"], sd[f"{fc}.3.bias"]))                        # :88
    feats.append(x)
    extras = [(3, 2, 1), (3, 2, 1), (3, 1, 0), (3, 1, 0)] + ([(4, 1, 0)] if highres else [])
    for i, (k, s, p) in enumerate(extras, start=1):
        x = F.relu(F.conv2d(x, sd[f"backbone.extra.{i}.0.weight"], sd[f"backbone.extra.{i}.0.bias"]))
        x = F.relu(F.conv2d(x, sd[f"backbone.extra.{i}.2.weight"], sd[f"backbone.extra.{i}.2.bias"], s, p))
        feats.append(x)
    return feats


def ssd_dense_head(sd, feats, num_classes) -> Dict[str, torch.Tensor]:
    """SSDHead: dense 3x3 conv heads with bias.  reference: generalized_ssd.py:25-35,77-92"""
    out = {}
    for name, cols, key in (("regression_head", 4, "bbox_regression"), ("classification_head", num_classes, "cls_logits")):
        res = []
        for lvl, f in enumerate(feats):
            p = f"head.{name}.module_list.{lvl}"
            res.append(scoring_head_permute(F.conv2d(f, sd[f"{p}.weight"], sd[f"{p}.bias"], 1, 1), cols))
        out[key] = torch.cat(res, dim=1)
    return out


# ---------------------------------------------------------------------------------------------
# transform (normalize / resize / batch / resize_boxes)    reference: transform.py
# ---------------------------------------------------------------------------------------------
def transform_images(images: List[torch.Tensor], mean, std, size_wh: Tuple[int, int]):
    """GeneralizedRCNNTransform.forward with fixed_size.  reference: transform.py:89-127,129-138,27-53"""
    out, orig = [], []
    for img in images:
        if img.dim() != 3:
            raise ValueError("images is expected to be a list of 3d tensors of shape [C, H, W], got {}".format(img.shape))
        if not img.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {img.dtype} instead")
        orig.append((int(img.shape[-2]), int(img.shape[-1])))
        m = torch.as_tensor(mean, dtype=img.dtype)
        s = torch.as_tensor(std, dtype=img.dtype)
        img = (img - m[:, None, None]) / s[:, None, None]
        # size = [fixed_size[1], fixed_size[0]] (transform.py:40); bilinear, align_corners=False (:52-53)
        img = F.interpolate(img[None], size=[size_wh[1], size_wh[0]], mode="bilinear", align_corners=False)[0]
        out.append(img)
    return torch.stack(out), orig


def resize_boxes(boxes: np.ndarray, original_size, new_size) -> np.ndarray:
    """reference: transform.py:278-292 (ratios computed in float32)"""
    rh = np.float32(new_size[0]) / np.float32(original_size[0])
    rw = np.float32(new_size[1]) / np.float32(original_size[1])
    b = boxes.astype(np.float32).copy()
    b[:, 0] *= rw
    b[:, 2] *= rw
    b[:, 1] *= rh
    b[:, 3] *= rh
    return b


# ---------------------------------------------------------------------------------------------
# anchors   reference: anchor_utils.py:10-126
# ---------------------------------------------------------------------------------------------
def default_boxes(grid_sizes: List[Tuple[int, int]], image_size_hw: Tuple[int, int], aspect_ratios,
                  min_ratio=0.15, max_ratio=0.9, scales=None, steps=None, clip=True) -> torch.Tensor:
    """Returns [A,4] xyxy pixel default boxes (float32), ordered level-major, then y, x, anchor."""
    L = len(aspect_ratios)
    if scales is None:                                                  # :38-47
        if L > 1:
            rr = max_ratio - min_ratio
            scales = [min_ratio + rr * k / (L - 1.0) for k in range(L)]
            scales.append(1.0)
        else:
            scales = [min_ratio, max_ratio]
    wh_pairs = []
    for k in range(L):                                                  # :51-68
        s_k = scales[k]
        s_prime_k = math.sqrt(scales[k] * scales[k + 1])
        wh = [[s_k, s_k], [s_prime_k, s_prime_k]]
        for ar in aspect_ratios[k]:
            sq = math.sqrt(ar)
            w, h = scales[k] * sq, scales[k] / sq
            wh.extend([[w, h], [h, w]])
        wh_pairs.append(torch.as_tensor(wh, dtype=torch.float32))
    H, W = image_size_hw
    out = []
    for k, f_k in enumerate(grid_sizes):                                # :75-100
        if steps is not None:
            x_f_k, y_f_k = [s / steps[k] for s in (H, W)]               # literal quirk of :81 (H->x, W->y)
        else:
            y_f_k, x_f_k = f_k
        shifts_x = ((torch.arange(0, f_k[1]) + 0.5) / x_f_k).to(torch.float32)
        shifts_y = ((torch.arange(0, f_k[0]) + 0.5) / y_f_k).to(torch.float32)
        shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
        shift_x, shift_y = shift_x.reshape(-1), shift_y.reshape(-1)
        shifts = torch.stack((shift_x, shift_y) * len(wh_pairs[k]), dim=-1).reshape(-1, 2)
        whp = wh_pairs[k].clamp(min=0, max=1) if clip else wh_pairs[k]
        out.append(torch.cat((shifts, whp.repeat(f_k[0] * f_k[1], 1)), dim=1))
    d = torch.cat(out, dim=0)
    d = torch.cat([d[:, :2] - 0.5 * d[:, 2:], d[:, :2] + 0.5 * d[:, 2:]], -1)   # :121-124
    d[:, 0::2] *= W
    d[:, 1::2] *= H
    return d


# ---------------------------------------------------------------------------------------------
# post-process   reference: generalized_ssd.py:351-397, _utils.py:187-224
# ---------------------------------------------------------------------------------------------
BBOX_XFORM_CLIP = math.log(1000.0 / 16)       # _utils.py:135


def decode_single(rel_codes: torch.Tensor, boxes: torch.Tensor, weights=(10.0, 10.0, 5.0, 5.0)) -> torch.Tensor:
    """BoxCoder.decode_single.  reference: _utils.py:187-224"""
    boxes = boxes.to(rel_codes.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = rel_codes[:, 0] / wx
    dy = rel_codes[:, 1] / wy
    dw = torch.clamp(rel_codes[:, 2] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(rel_codes[:, 3] / wh, max=BBOX_XFORM_CLIP)
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = torch.exp(dw) * widths
    ph = torch.exp(dh) * heights
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def clip_boxes_to_image(boxes: np.ndarray, size_hw) -> np.ndarray:
    """torchvision.ops.boxes.clip_boxes_to_image (published semantics): x in [0,W], y in [0,H]."""
    h, w = size_hw
    b = np.array(boxes, dtype=np.float32, copy=True)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, np.float32(w))
    b[:, 1::2] = np.clip(b[:, 1::2], 0, np.float32(h))
    return b


def nms_single_class(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """Greedy hard NMS (torchvision.ops.nms published semantics), float32 arithmetic, strict `>`.
    Order: score descending, ties by ascending input index (stable). Returns kept input indices in that order."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, i].astype(np.float32) for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    thr32 = np.float32(thr)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr32]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes: np.ndarray, scores: np.ndarray, idxs: np.ndarray, thr: float) -> np.ndarray:
    """Per-class NMS, no coordinate-offset trick (SURVEY 7 'batched_nms flavour'); result = kept indices of all
    classes sorted by score descending, ties in ascending input order (stable)."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    keep_all = []
    for c in np.unique(idxs):
        sel = np.nonzero(idxs == c)[0]
        k = nms_single_class(boxes[sel], scores[sel], thr)
        keep_all.append(sel[k])
    keep_all = np.sort(np.concatenate(keep_all))
    order = np.argsort(-scores[keep_all], kind="stable")
    return keep_all[order]


def select_candidates(scores: np.ndarray, score_thresh: float, topk: int):
    """Per-class threshold + top-k (generalized_ssd.py:368-382) with the canonical tie-break
    (score desc, anchor index asc). scores: [A, K] float32 softmax output.
    Returns (anchor_idx [M], labels [M], cand_scores [M]) concatenated class-major (label 1..K-1)."""
    A, K = scores.shape
    thr = np.float32(score_thresh)
    a_all, l_all, s_all = [], [], []
    for label in range(1, K):
        sc = scores[:, label]
        idx = np.nonzero(sc > thr)[0]
        order = np.argsort(-sc[idx], kind="stable")[:min(topk, idx.size)]
        sel = idx[order]
        a_all.append(sel)
        l_all.append(np.full(sel.size, label, dtype=np.int64))
        s_all.append(sc[sel])
    return np.concatenate(a_all), np.concatenate(l_all), np.concatenate(s_all)


def postprocess_detections(cls_logits: torch.Tensor, bbox_regression: torch.Tensor, anchors: torch.Tensor,
                           image_size_hw, score_thresh, nms_thresh, detections_per_img, topk_candidates,
                           return_intermediates=False):
    """SSD.postprocess_detections.  reference: generalized_ssd.py:351-397"""
    pred_scores = F.softmax(cls_logits, dim=-1)                          # :354
    dets = []
    for reg, scores in zip(bbox_regression, pred_scores):
        boxes = decode_single(reg, anchors).numpy()                      # :362
        boxes = clip_boxes_to_image(boxes, image_size_hw)                # :363
        sc = scores.numpy()
        a_idx, labels, cs = select_candidates(sc, score_thresh, topk_candidates)
        cb = boxes[a_idx]
        keep = batched_nms(cb, cs, labels, nms_thresh)[:detections_per_img]   # :389-390
        d = {"boxes": cb[keep], "scores": cs[keep], "labels": labels[keep]}
        if return_intermediates:
            d.update(anchor_idx=a_idx[keep], cand_anchor=a_idx, cand_labels=labels, cand_scores=cs,
                     decoded=boxes, softmax=sc, keep=keep)
        dets.append(d)
    return dets


# ---------------------------------------------------------------------------------------------
# whole models
# ---------------------------------------------------------------------------------------------
MODEL_DEFAULTS = {
    # name: (size_wh, mean, std, post defaults, anchor spec)
    "ssdlite320_mobilenet_v3_large": dict(
        size=(320, 320), mean=[0.5] * 3, std=[0.5] * 3,
        post=dict(score_thresh=0.001, nms_thresh=0.55, detections_per_img=300, topk_candidates=300),   # ssd_mobilenetv3.py:207-216
        anchors=dict(aspect_ratios=[[2, 3]] * 6, min_ratio=0.2, max_ratio=0.95)),                      # :202
    "ssd_lite_mobilenet_v2": dict(
        size=(320, 320), mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225],
        post=dict(score_thresh=0.5, nms_thresh=0.45, detections_per_img=100, topk_candidates=400),
        anchors=dict(aspect_ratios=[[2, 3]] * 6, min_ratio=0.2, max_ratio=0.95)),
    "ssd300_vgg16": dict(
        size=(300, 300), mean=[0.48235, 0.45882, 0.40784], std=[1.0 / 255.0] * 3,                      # ssd_vgg16.py:202-203
        post=dict(score_thresh=0.01, nms_thresh=0.45, detections_per_img=200, topk_candidates=400),    # generalized_ssd.py:158-162
        anchors=dict(aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]],
                     scales=[0.07, 0.15, 0.33, 0.51, 0.69, 0.87, 1.05], steps=[8, 16, 32, 64, 100, 300])),   # :196-198
    "ssd512_vgg16": dict(
        size=(512, 512), mean=[0.48235, 0.45882, 0.40784], std=[1.0 / 255.0] * 3,
        post=dict(score_thresh=0.01, nms_thresh=0.45, detections_per_img=200, topk_candidates=400),
        anchors=dict(aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
                     scales=[0.04, 0.1, 0.26, 0.42, 0.58, 0.74, 0.9, 1.06], steps=[8, 16, 32, 64, 128, 256, 512])),
}


def _to_torch_sd(sd):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}


class OracleSSD:
    """Functional CPU model over a reference-keyed state_dict."""

    def __init__(self, name: str, state_dict, num_classes: int, size=None, **post):
        self.name = name
        cfg = MODEL_DEFAULTS[name]
        self.size = tuple(size) if size is not None else cfg["size"]
        self.mean, self.std = cfg["mean"], cfg["std"]
        self.post = {**cfg["post"], **post}
        self.anchor_spec = cfg["anchors"]
        self.sd = _to_torch_sd(state_dict)
        self.num_classes = num_classes

    def features(self, x):
        if self.name == "ssdlite320_mobilenet_v3_large":
            return ssdlite_v3_features(self.sd, x)
        if self.name == "ssd_lite_mobilenet_v2":
            return ssdlite_v2_features(self.sd, x)
        return ssd_vgg_features(self.sd, x, highres=(self.name == "ssd512_vgg16"))

    def head(self, feats):
        if self.name == "ssdlite320_mobilenet_v3_large":
            return ssdlite_head(self.sd, feats, self.num_classes)
        if self.name == "ssd_lite_mobilenet_v2":
            return multibox_lite_head(self.sd, feats, self.num_classes)
        return ssd_dense_head(self.sd, feats, self.num_classes)

    def anchors(self, feats):
        grid = [tuple(f.shape[-2:]) for f in feats]
        return default_boxes(grid, (self.size[1], self.size[0]), **self.anchor_spec)

    @torch.no_grad()
    def forward_raw(self, images: List[torch.Tensor]):
        x, orig = transform_images(images, self.mean, self.std, self.size)
        feats = self.features(x)
        head = self.head(feats)
        return dict(x=x, orig=orig, features=feats, cls_logits=head["cls_logits"],
                    bbox_regression=head["bbox_regression"], anchors=self.anchors(feats))

    @torch.no_grad()
    def __call__(self, images: List[torch.Tensor], return_intermediates=False):
        r = self.forward_raw(images)
        hw = (self.size[1], self.size[0])
        dets = postprocess_detections(r["cls_logits"], r["bbox_regression"], r["anchors"], hw,
                                      return_intermediates=return_intermediates, **self.post)
        for d, o in zip(dets, r["orig"]):                      # transform.postprocess (transform.py:228-247)
            d["boxes"] = resize_boxes(d["boxes"], hw, o)
        if return_intermediates:
            return dets, r
        return dets


# ---------------------------------------------------------------------------------------------
# fixture hygiene: how far is a post-process instance from an order/threshold flip?
# ---------------------------------------------------------------------------------------------
def box_iou_np(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """IoU matrix in float32 with the same operation order as nms_single_class."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    xx1 = np.maximum(a[:, None, 0], b[None, :, 0])
    yy1 = np.maximum(a[:, None, 1], b[None, :, 1])
    xx2 = np.minimum(a[:, None, 2], b[None, :, 2])
    yy2 = np.minimum(a[:, None, 3], b[None, :, 3])
    w = np.maximum(np.float32(0), xx2 - xx1)
    h = np.maximum(np.float32(0), yy2 - yy1)
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area_a[:, None] + area_b[None, :] - inter)


def selection_margins(softmax: np.ndarray, decoded: np.ndarray, score_thresh, nms_thresh, topk, dets_per_img):
    """Returns a dict of the smallest margins that decide the kept-index list:
      thresh_gap   min |score - score_thresh| / score_thresh over all (anchor, class>0)
      topk_gap     min relative gap between the last selected and first rejected candidate of a class
      order_gap    min relative gap between score-adjacent candidates of one class whose IoU is within
                   1e-3 of exceeding nms_thresh (their order decides who suppresses whom)
      iou_gap      min |IoU - nms_thresh| over same-class candidate pairs
      final_gap    min relative gap between score-adjacent entries of the final list (incl. first dropped)
    An implementation whose scores/boxes differ from the oracle's by less than these margins must return
    bit-identical indices."""
    A, K = softmax.shape
    out = dict(thresh_gap=np.inf, topk_gap=np.inf, order_gap=np.inf, iou_gap=np.inf, final_gap=np.inf)
    thr = np.float32(score_thresh)
    out["thresh_gap"] = float(np.min(np.abs(softmax[:, 1:] - thr)) / thr)
    kept_scores = []
    for label in range(1, K):
        sc = softmax[:, label]
        idx = np.nonzero(sc > thr)[0]
        order = idx[np.argsort(-sc[idx], kind="stable")]
        k = min(topk, order.size)
        if order.size > k:
            out["topk_gap"] = min(out["topk_gap"], float((sc[order[k - 1]] - sc[order[k]]) / sc[order[k - 1]]))
        sel = order[:k]
        if k >= 2:
            iou = box_iou_np(decoded[sel], decoded[sel])
            iu = np.triu_indices(k, 1)
            d = np.abs(iou[iu] - np.float32(nms_thresh))
            d = d[np.isfinite(d)]
            if d.size:
                out["iou_gap"] = min(out["iou_gap"], float(d.min()))
            s = sc[sel]
            rel = (s[:-1] - s[1:]) / s[:-1]
            adj_iou = iou[np.arange(k - 1), np.arange(1, k)]
            risky = np.nan_to_num(adj_iou, nan=1.0) > nms_thresh - 1e-3
            if risky.any():
                out["order_gap"] = min(out["order_gap"], float(rel[risky].min()))
        keep = nms_single_class(decoded[sel], sc[sel], nms_thresh)
        kept_scores.append(sc[sel][keep])
    allk = np.sort(np.concatenate(kept_scores))[::-1][:dets_per_img + 1]
    if allk.size >= 2:
        out["final_gap"] = float(np.min((allk[:-1] - allk[1:]) / allk[:-1]))
    return out


# ---------------------------------------------------------------------------------------------
# training loss, forward value -- restated from generalized_ssd.py:210-269,316-330 and _utils.py:100-133,264-294,348-362
# ---------------------------------------------------------------------------------------------
def box_iou_t(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """torchvision.ops.boxes.box_iou (third-party, published formula; called at generalized_ssd.py:326)."""
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def ssd_match(gt_boxes: torch.Tensor, anchors: torch.Tensor, iou_thresh: float = 0.5) -> torch.Tensor:
    """SSDMatcher on box_iou (generalized_ssd.py:318-327; _utils.py:264-294 with low == high threshold, then :348-362: every gt
    claims the anchor it overlaps most, in gt order). Explicit loops where the reference relies on index_put_ with duplicate
    indices (the CPU kernel applies them in order: the last gt wins)."""
    A = anchors.shape[0]
    if gt_boxes.numel() == 0:
        return torch.full((A,), -1, dtype=torch.int64)
    q = box_iou_t(gt_boxes, anchors)                        # [G, A]
    vals, matches = q.max(dim=0)                            # first maximum over the gts
    matches = matches.clone()
    matches[vals < iou_thresh] = -1
    best_anchor = q.max(dim=1)[1]                           # first maximum over the anchors
    for g in range(gt_boxes.shape[0]):
        matches[int(best_anchor[g])] = g
    return matches


def encode_boxes_t(gt: torch.Tensor, anchors: torch.Tensor, weights=(10.0, 10.0, 5.0, 5.0)) -> torch.Tensor:
    """_utils.py:100-133 (reference_boxes = gt, proposals = anchors)."""
    ew, eh = anchors[:, 2] - anchors[:, 0], anchors[:, 3] - anchors[:, 1]
    ecx, ecy = anchors[:, 0] + 0.5 * ew, anchors[:, 1] + 0.5 * eh
    gw, gh = gt[:, 2] - gt[:, 0], gt[:, 3] - gt[:, 1]
    gcx, gcy = gt[:, 0] + 0.5 * gw, gt[:, 1] + 0.5 * gh
    return torch.stack([weights[0] * (gcx - ecx) / ew, weights[1] * (gcy - ecy) / eh,
                        weights[2] * torch.log(gw / ew), weights[3] * torch.log(gh / eh)], dim=1)


def ssd_loss_oracle(cls_logits: torch.Tensor, bbox_regression: torch.Tensor, anchors: torch.Tensor, targets,
                    iou_thresh: float = 0.5, neg_to_pos_ratio: float = 3.0):
    """generalized_ssd.py:210-269 after the matching of :316-330. Returns ({'bbox_regression', 'classification'}, matched [N, A])."""
    import torch.nn.functional as F
    n, A, K = cls_logits.shape
    matched = torch.stack([ssd_match(t["boxes"].float(), anchors, iou_thresh) for t in targets])
    num_foreground, bbox_loss, cls_targets = 0, [], []
    for i, t in enumerate(targets):
        fg = torch.where(matched[i] >= 0)[0]
        mi = matched[i][fg]
        num_foreground += mi.numel()
        target_reg = encode_boxes_t(t["boxes"].float()[mi], anchors[fg])
        bbox_loss.append(F.smooth_l1_loss(bbox_regression[i][fg], target_reg, reduction="sum"))
        ct = torch.zeros((A,), dtype=torch.int64)
        ct[fg] = t["labels"][mi]
        cls_targets.append(ct)
    cls_targets = torch.stack(cls_targets)
    cls_loss = F.cross_entropy(cls_logits.reshape(-1, K), cls_targets.reshape(-1), reduction="none").view(n, A)
    fgm = cls_targets > 0
    num_negative = neg_to_pos_ratio * fgm.sum(1, keepdim=True)
    neg = cls_loss.clone()
    neg[fgm] = -float("inf")
    _, idx = neg.sort(dim=1, descending=True, stable=True)
    bgm = idx.sort(1)[1] < num_negative
    N = max(1, num_foreground)
    return ({"bbox_regression": torch.stack(bbox_loss).sum() / N,
             "classification": (cls_loss[fgm].sum() + cls_loss[bgm].sum()) / N}, matched)
