"""TEST / BASELINE INFRASTRUCTURE -- not part of the product path (only tests/, bench.py's cpu_baseline leg and __graft_entry__ import oracle/).

A vectorised CPU post-process: the same function as ssd_oracle.postprocess_detections (the restatement of
SSD.postprocess_detections, demonet/models/generalized_ssd.py:351-397), organised the way the reference's own loop runs on a box that
has torchvision: torch ops for softmax / decode / clip / per-class threshold + top-k (the reference calls torch.topk per class, :376),
and a COMPILED greedy NMS (torchvision's C++ `nms` behind batched_nms, :389; here oracle/nms_c.c). bench.py times it beside the
checker's numpy / Python loops so that the stated CPU baseline is not dominated by the port's interpreter overhead. The checker is
ssd_oracle; this file is pinned to it by tests/test_oracle.py::test_fast_postprocess_equals_the_checker (index-exact on tie-free inputs).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

import ssd_oracle as so

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnms_c.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "nms_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, src])
    return _SO


def _nms():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.nms_segments.restype = C.c_int
        L.nms_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        _lib = L
    return _lib


@torch.no_grad()
def postprocess_detections(cls_logits, bbox_regression, anchors, image_size_hw, score_thresh, nms_thresh, detections_per_img,
                           topk_candidates):
    L = _nms()
    pred_scores = F.softmax(cls_logits, dim=-1)
    H, W = image_size_hw
    K = pred_scores.shape[-1]
    dets = []
    for reg, scores in zip(bbox_regression, pred_scores):
        boxes = so.decode_single(reg, anchors)
        boxes = torch.stack([boxes[:, 0].clamp(0, W), boxes[:, 1].clamp(0, H), boxes[:, 2].clamp(0, W), boxes[:, 3].clamp(0, H)], 1)
        sc = scores[:, 1:].t().contiguous()                                   # [K-1, A]
        k = min(topk_candidates, sc.shape[1])
        vals, idx = torch.where(sc > score_thresh, sc, sc.new_full((), -1.0)).topk(k, dim=1)      # per class, descending (:376-377)
        valid = vals > score_thresh
        counts = valid.sum(1)
        a_idx = idx[valid]                                                    # class-major, score-descending inside a class
        cs = vals[valid]
        labels = torch.repeat_interleave(torch.arange(1, K), counts)
        cb = boxes[a_idx].contiguous().numpy()
        seg = np.zeros(K, dtype=np.int32)
        seg[1:] = np.cumsum(counts.numpy())
        keep = np.zeros(cb.shape[0], dtype=np.uint8)
        L.nms_segments(cb.ctypes.data, seg.ctypes.data, K - 1, C.c_float(nms_thresh), keep.ctypes.data)
        kept = np.nonzero(keep)[0]
        csn = cs.numpy()
        order = kept[np.argsort(-csn[kept], kind="stable")][:detections_per_img]      # global score order, stable (batched_nms)
        dets.append({"boxes": cb[order], "scores": csn[order], "labels": labels.numpy()[order], "anchor_idx": a_idx.numpy()[order]})
    return dets
