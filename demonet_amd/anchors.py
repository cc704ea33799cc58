"""Default-box (anchor) generation on the host, float32 and in the reference's operation order.

reference: demonet/models/anchor_utils.py:28-126 (DefaultBoxGenerator). Input-independent, so it is computed
once per plan and uploaded, instead of once per forward call as the reference does (:111-126).
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np


def default_boxes(grid_sizes: Sequence[Tuple[int, int]], image_hw: Tuple[int, int], aspect_ratios: List[List[int]],
                  min_ratio: Optional[float] = 0.15, max_ratio: Optional[float] = 0.9,
                  scales: Optional[List[float]] = None, steps: Optional[List[int]] = None, clip: bool = True) -> np.ndarray:
    f32 = np.float32
    L = len(aspect_ratios)
    if scales is None:                                        # anchor_utils.py:38-47
        if L > 1:
            rr = max_ratio - min_ratio
            scales = [min_ratio + rr * k / (L - 1.0) for k in range(L)] + [1.0]
        else:
            scales = [min_ratio, max_ratio]
    H, W = image_hw
    out = []
    for k, (fh, fw) in enumerate(grid_sizes):
        s_k = scales[k]
        s_p = math.sqrt(scales[k] * scales[k + 1])            # :56-58
        wh = [[s_k, s_k], [s_p, s_p]]
        for ar in aspect_ratios[k]:
            sq = math.sqrt(ar)
            wh += [[s_k * sq, s_k / sq], [s_k / sq, s_k * sq]]
        wh = np.asarray(wh, dtype=f32)
        if clip:
            wh = np.clip(wh, f32(0), f32(1))                  # :93
        if steps is not None:                                 # :80-83 (H-derived value feeds x: replicated literally)
            x_f, y_f = f32(H / steps[k]), f32(W / steps[k])
        else:
            y_f, x_f = f32(fh), f32(fw)
        sx = (np.arange(fw, dtype=f32) + f32(0.5)) / x_f
        sy = (np.arange(fh, dtype=f32) + f32(0.5)) / y_f
        cy, cx = np.meshgrid(sy, sx, indexing="ij")
        cxy = np.stack([cx.reshape(-1), cy.reshape(-1)], axis=1)                  # [hw, 2]
        A = wh.shape[0]
        c = np.repeat(cxy[:, None, :], A, axis=1).reshape(-1, 2)                  # (y, x, a) order
        w = np.tile(wh[None, :, :], (fh * fw, 1, 1)).reshape(-1, 2)
        out.append(np.concatenate([c, w], axis=1))
    d = np.concatenate(out, axis=0).astype(f32)
    half = f32(0.5) * d[:, 2:]
    boxes = np.concatenate([d[:, :2] - half, d[:, :2] + half], axis=1)           # :121-122
    boxes[:, 0::2] *= f32(W)
    boxes[:, 1::2] *= f32(H)
    return np.ascontiguousarray(boxes, dtype=f32)
