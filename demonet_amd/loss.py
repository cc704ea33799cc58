"""SSD training loss, forward value only -- the mirror of `SSD.compute_loss` plus the anchor matching that precedes it in
`SSD.forward` (reference: demonet/models/generalized_ssd.py:210-269, 316-330; _utils.py:100-133, 264-294, 348-362), computed by
`dn_ssd_loss` (csrc/loss.hip). No gradients: this repo has no backward pass.

    losses, matched = ssd_loss(head_outputs, anchors, targets)         # {'bbox_regression', 'classification'}, [N, A] int64

`head_outputs` = {'cls_logits': [N, A, K], 'bbox_regression': [N, A, 4]} fp32 CUDA tensors (e.g. `SSD.forward_heads`), `anchors`
= [A, 4] (or the reference's list of N identical [A, 4] tensors), `targets` = list of {'boxes': [G, 4], 'labels': [G] int64}.
Same error behaviour as the reference where it has one: degenerate boxes raise ValueError (generalized_ssd.py:300-308).
Tie order: the reference ranks the negatives with two (unstable) sorts; only the SUM of the selected losses enters the result, which
does not depend on the order of equal values -- except in the corner where more negatives are wanted than exist (the -inf entries of
the foreground anchors enter the ranking): there csrc/loss.hip follows the stable-sort order, as torch's CPU sort happens to."""
import ctypes as C
from typing import Dict, List, Tuple

import torch
from torch import Tensor

from . import _lib


def ssd_loss(head_outputs: Dict[str, Tensor], anchors, targets: List[Dict[str, Tensor]], iou_thresh: float = 0.5,
             neg_to_pos_ratio: float = 3.0) -> Tuple[Dict[str, Tensor], Tensor]:
    logits, reg = head_outputs["cls_logits"], head_outputs["bbox_regression"]
    if not logits.is_cuda:
        raise RuntimeError("ssd_loss needs CUDA tensors: there is no CPU fallback path")
    if isinstance(anchors, (list, tuple)):
        anchors = anchors[0]
    n, A, K = logits.shape
    if n == 0:
        raise ValueError("ssd_loss: empty batch")
    if reg.shape != (n, A, 4) or anchors.shape != (A, 4) or len(targets) != n:
        raise ValueError("ssd_loss: inconsistent shapes")
    GMAX = 256                                  # csrc/loss.hip: ground-truth boxes per image the matcher keeps in LDS
    for ti, t in enumerate(targets):
        if int(t["boxes"].shape[0]) > GMAX:
            raise ValueError("ssd_loss: {} ground-truth boxes for target at index {}; the matcher kernel holds at most {} per image".format(
                int(t["boxes"].shape[0]), ti, GMAX))
    for ti, t in enumerate(targets):
        b = t["boxes"]
        if b.numel():
            bad = b[:, 2:] <= b[:, :2]
            if bool(bad.any()):
                bb = b[torch.where(bad.any(dim=1))[0][0]].tolist()
                raise ValueError("All bounding boxes should have positive height and width."
                                 " Found invalid box {} for target at index {}.".format(bb, ti))
    dev = logits.device
    gmax = max(1, max(int(t["boxes"].shape[0]) for t in targets))
    gb = torch.zeros((n, gmax, 4), dtype=torch.float32, device=dev)
    gl = torch.zeros((n, gmax), dtype=torch.int64, device=dev)
    gc = torch.zeros((n,), dtype=torch.int32, device=dev)
    for i, t in enumerate(targets):
        g = int(t["boxes"].shape[0])
        if g:
            gb[i, :g] = t["boxes"].to(dev, torch.float32)
            gl[i, :g] = t["labels"].to(dev, torch.int64)
        gc[i] = g
    L = _lib.lib()
    ws = torch.empty(int(L.dn_ssd_loss_workspace_bytes(n, A)), dtype=torch.uint8, device=dev)
    matched = torch.empty((n, A), dtype=torch.int64, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    lg, rg, an = logits.contiguous().float(), reg.contiguous().float(), anchors.to(dev, torch.float32).contiguous()
    P = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(L.dn_ssd_loss(P(lg), P(rg), P(an), P(gb), P(gl), P(gc), n, A, K, gmax, float(iou_thresh), float(neg_to_pos_ratio),
                                 P(matched), P(losses), P(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "dn_ssd_loss")
    return {"bbox_regression": losses[0], "classification": losses[1]}, matched
