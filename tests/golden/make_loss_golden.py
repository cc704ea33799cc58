"""Authoring-time generator of tests/golden/ssd_loss.npz: seeded head outputs, targets and the losses / matched indices the REAL
reference computes for them -- `SSD.compute_loss` (/root/reference/demonet/models/generalized_ssd.py:210-269) on the matching of
`SSD.forward` (:316-330: box_iou -> SSDMatcher, _utils.py:264-294,348-362) of a reference `ssdlite320_mobilenet_v3_large`
instance (its own box_coder, proposal_matcher, neg_to_pos_ratio). torchvision's box_iou comes from oracle/ref_shim (published
formula; torchvision itself is not installed here). Anchors: the reference DefaultBoxGenerator output already pinned in
tests/golden/ssdlite320_mobilenet_v3_large.npz. Run in the authoring container only; the .npz is the committed fixture."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import run_reference as rr  # noqa: E402

rr.import_reference_models()
from torchvision.ops import boxes as box_ops  # noqa: E402  (the shim)

ref = rr.build_reference_model("ssdlite320_mobilenet_v3_large", 91)
anchors = torch.from_numpy(np.load(os.path.join(HERE, "ssdlite320_mobilenet_v3_large.npz"))["anchors"]).float()
A, K = anchors.shape[0], 91
out = {"anchors": anchors.numpy(), "iou_thresh": np.float32(0.5), "neg_to_pos_ratio": np.float32(ref.neg_to_pos_ratio)}
rng = np.random.RandomState(20241003)
cases = []
for ci, (n, gcounts) in enumerate([(3, [4, 1, 9]), (2, [0, 6]), (4, [2, 2, 30, 1]), (1, [120])]):
    g = torch.Generator().manual_seed(500 + ci)
    logits = torch.randn(n, A, K, generator=g) * 2.0
    reg = torch.randn(n, A, 4, generator=g)
    targets = []
    for i, gc in enumerate(gcounts):
        if ci == 0 and i == 0:
            # two ground-truth boxes whose best anchor is the same one (SSDMatcher: the later gt keeps it) + an exact copy of an anchor
            a0 = anchors[1500].clone()
            b = torch.stack([a0, a0 + torch.tensor([1.0, 1.0, -1.0, -1.0]), anchors[2500] + torch.tensor([0.5, 0.0, 0.5, 0.0]),
                             torch.tensor([10.0, 20.0, 200.0, 260.0])])
        else:
            xy = torch.from_numpy(rng.uniform(0, 250, (gc, 2)).astype(np.float32))
            wh = torch.from_numpy(rng.uniform(8, 160, (gc, 2)).astype(np.float32))
            b = torch.cat([xy, torch.minimum(xy + wh, torch.tensor(320.0))], 1)
        lab = torch.from_numpy(rng.randint(1, K, (b.shape[0],)).astype(np.int64))
        targets.append({"boxes": b, "labels": lab})
    matched = []
    for t in targets:
        if t["boxes"].numel() == 0:
            matched.append(torch.full((A,), -1, dtype=torch.int64))
        else:
            matched.append(ref.proposal_matcher(box_ops.box_iou(t["boxes"], anchors)))
    losses = ref.compute_loss(targets, {"cls_logits": logits, "bbox_regression": reg}, [anchors] * n, matched)
    out[f"c{ci}_logits_seed"] = np.int64(500 + ci)
    out[f"c{ci}_n"] = np.int64(n)
    gmax = max(1, max(int(t["boxes"].shape[0]) for t in targets))
    gb = np.zeros((n, gmax, 4), np.float32)
    gl = np.zeros((n, gmax), np.int64)
    for i, t in enumerate(targets):
        k = t["boxes"].shape[0]
        gb[i, :k] = t["boxes"].numpy()
        gl[i, :k] = t["labels"].numpy()
    out[f"c{ci}_gt_boxes"], out[f"c{ci}_gt_labels"] = gb, gl
    out[f"c{ci}_gt_counts"] = np.array([t["boxes"].shape[0] for t in targets], np.int32)
    out[f"c{ci}_matched"] = torch.stack(matched).numpy()
    out[f"c{ci}_bbox_regression"] = np.float32(losses["bbox_regression"].item())
    out[f"c{ci}_classification"] = np.float32(losses["classification"].item())
    print(f"case {ci}: n={n} gts={gcounts} matched anchors {[int((m >= 0).sum()) for m in matched]} "
          f"bbox {losses['bbox_regression'].item():.6f} cls {losses['classification'].item():.6f}")
out["n_cases"] = np.int64(4)
np.savez_compressed(os.path.join(HERE, "ssd_loss.npz"), **out)
print("wrote", os.path.join(HERE, "ssd_loss.npz"))
