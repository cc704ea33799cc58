"""The step immediately after the hot path: turn the device's padded detections into the records the
reference's evaluators consume, and score them, without pycocotools.

* COCO result records: `CocoEvaluator.prepare_for_coco_detection` (demonet/data/coco_eval.py:76-98) with `convert_to_xywh`
  (:162-164) -- one dict per detection, bbox as [x, y, w, h].
* PASCAL VOC: per-class TP/FP marking and precision/recall as `voc_eval` does (demonet/data/voc_eval.py:116-165: sort by
  confidence descending, +1 pixel box arithmetic, a ground truth can be claimed once, "difficult" boxes are ignored) and
  `voc_ap` (:29-58), both the 11-point VOC07 metric and the area under the precision envelope.

`voc_ap` and the overlap / TP-FP loop of `voc_class_pr` are the canonical PASCAL VOC devkit routine (VOCdevkit `VOCevaldet.m`, as ported to
Python in py-faster-rcnn's `voc_eval.py`), which the reference itself carries in demonet/data/voc_eval.py: to be a drop-in they have to
produce that routine's numbers bit for bit (tests/golden/voc_eval.npz holds vectors from the reference's own functions), so the arithmetic
-- the precision envelope, the `+ 1` pixel box sizes, "a ground truth is claimed once" -- is restated step for step, not redesigned.

Host-side numpy; inputs are what `SSD.forward_batch` returns (copied to the host once per batch).
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np


def coco_detection_records(boxes, scores, labels, counts, image_ids: Sequence[int]) -> List[dict]:
    """boxes [N,D,4] xyxy, scores [N,D], labels [N,D], counts [N] (host arrays or tensors) -> COCO results list."""
    boxes, scores, labels, counts = (np.asarray(t.cpu() if hasattr(t, "cpu") else t) for t in (boxes, scores, labels, counts))
    out = []
    for i, img_id in enumerate(image_ids):
        c = int(counts[i])
        if c == 0:                                               # coco_eval.py:79-80: images without predictions are skipped
            continue
        b = boxes[i, :c]
        xywh = np.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), axis=1)      # convert_to_xywh, in float32
        bl, sl, ll = xywh.tolist(), scores[i, :c].tolist(), labels[i, :c].tolist()
        out.extend({"image_id": img_id, "category_id": int(ll[k]), "bbox": bl[k], "score": sl[k]} for k in range(c))
    return out


def voc_ap(rec: np.ndarray, prec: np.ndarray, use_07_metric: bool = False) -> float:
    """voc_eval.py:29-58 -- the PASCAL VOC devkit's AP (11-point VOC07 metric, or the area under the monotone precision envelope)."""
    rec, prec = np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0.0 if np.sum(rec >= t) == 0 else float(np.max(prec[rec >= t]))
            ap = ap + p / 11.0
        return float(ap)
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def voc_class_pr(det_image_ids: Sequence, det_scores: np.ndarray, det_boxes: np.ndarray,
                 gt: Dict[object, Tuple[np.ndarray, np.ndarray]], ovthresh: float = 0.5) -> Tuple[np.ndarray, np.ndarray]:
    """One class. det_*: all detections of the class over the image set; gt[image_id] = (boxes [G,4], difficult [G] bool).
    Returns (recall, precision) per detection in confidence order (voc_eval.py:97-161)."""
    det_scores = np.asarray(det_scores, dtype=np.float64)
    det_boxes = np.asarray(det_boxes, dtype=np.float64).reshape(-1, 4)
    npos = int(sum(int((~np.asarray(d, dtype=bool)).sum()) for _, d in gt.values()))
    claimed = {k: np.zeros(len(np.asarray(d)), dtype=bool) for k, (_, d) in gt.items()}
    order = np.argsort(-det_scores)
    nd = len(order)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d, j in enumerate(order):
        img = det_image_ids[j]
        bb = det_boxes[j]
        gboxes, gdiff = gt.get(img, (np.zeros((0, 4)), np.zeros(0, dtype=bool)))
        gboxes = np.asarray(gboxes, dtype=np.float64).reshape(-1, 4)
        ovmax, jmax = -np.inf, -1
        if gboxes.size > 0:
            ixmin = np.maximum(gboxes[:, 0], bb[0]); iymin = np.maximum(gboxes[:, 1], bb[1])
            ixmax = np.minimum(gboxes[:, 2], bb[2]); iymax = np.minimum(gboxes[:, 3], bb[3])
            iw = np.maximum(ixmax - ixmin + 1.0, 0.0); ih = np.maximum(iymax - iymin + 1.0, 0.0)
            inters = iw * ih
            unions = (bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (gboxes[:, 2] - gboxes[:, 0] + 1.0) * (gboxes[:, 3] - gboxes[:, 1] + 1.0) - inters
            overlaps = inters / unions
            ovmax, jmax = float(np.max(overlaps)), int(np.argmax(overlaps))
        if ovmax > ovthresh:
            if not bool(np.asarray(gdiff)[jmax]):
                if not claimed[img][jmax]:
                    tp[d] = 1.0
                    claimed[img][jmax] = True
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos) if npos > 0 else np.zeros_like(tp)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec


def voc_mean_ap(detections: Sequence[Dict[str, np.ndarray]], ground_truth: Sequence[Dict[str, np.ndarray]], ovthresh: float = 0.5,
                use_07_metric: bool = False) -> Tuple[float, Dict[int, float]]:
    """mAP over an image set as the reference's VOC evaluation reports it (data/voc_eval.py:186-214 `do_python_eval`: AP per
    class, then the mean over the classes; here over the classes that occur in the ground truth).
    detections[i] / ground_truth[i]: dicts of one image with `boxes` [K,4] xyxy, `labels` [K] (+ `scores` [K] for detections,
    optional `difficult` [K] for ground truth). Returns (mean AP in percent, {class: AP in percent})."""
    classes = sorted({int(c) for g in ground_truth for c in np.asarray(g["labels"]).reshape(-1)})
    aps = {}
    for c in classes:
        ids, scores, boxes = [], [], []
        for i, d in enumerate(detections):
            lab = np.asarray(d["labels"]).reshape(-1)
            sel = lab == c
            n = int(sel.sum())
            if n:
                ids.extend([i] * n)
                scores.append(np.asarray(d["scores"], dtype=np.float64).reshape(-1)[sel])
                boxes.append(np.asarray(d["boxes"], dtype=np.float64).reshape(-1, 4)[sel])
        gt = {}
        for i, g in enumerate(ground_truth):
            lab = np.asarray(g["labels"]).reshape(-1)
            sel = lab == c
            diff = np.asarray(g.get("difficult", np.zeros(lab.shape[0], dtype=bool))).reshape(-1).astype(bool)
            gt[i] = (np.asarray(g["boxes"], dtype=np.float64).reshape(-1, 4)[sel], diff[sel])
        if ids:
            rec, prec = voc_class_pr(ids, np.concatenate(scores), np.concatenate(boxes), gt, ovthresh)
            aps[c] = 100.0 * voc_ap(rec, prec, use_07_metric)
        else:
            aps[c] = 0.0
    return (float(np.mean(list(aps.values()))) if aps else 0.0), aps
