"""torchvision.ops.boxes stand-in: delegates to the oracle's restatement of the published semantics."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ssd_oracle as _o  # noqa: E402


def box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1, boxes2):
    area1 = box_area(boxes1)
    area2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def clip_boxes_to_image(boxes, size):
    return torch.from_numpy(_o.clip_boxes_to_image(boxes.detach().numpy(), size))


def nms(boxes, scores, iou_threshold):
    keep = _o.nms_single_class(boxes.detach().numpy().astype(np.float32),
                               scores.detach().numpy().astype(np.float32), float(iou_threshold))
    return torch.from_numpy(keep.astype(np.int64))


def batched_nms(boxes, scores, idxs, iou_threshold):
    keep = _o.batched_nms(boxes.detach().numpy().astype(np.float32),
                          scores.detach().numpy().astype(np.float32),
                          idxs.detach().numpy().astype(np.int64), float(iou_threshold))
    return torch.from_numpy(keep.astype(np.int64))
