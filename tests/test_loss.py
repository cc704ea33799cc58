"""SSD training loss, forward value: oracle restatement vs the golden vectors of the real reference
(tests/golden/make_loss_golden.py: SSD.compute_loss + SSDMatcher of a reference model instance), and -- on the GPU -- dn_ssd_loss
through the C ABI vs both. Matched indices are index work: bit-exact. Loss values are fp32 sums taken in a different order than
torch.sum's: rtol 2e-5."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ssd_oracle as so  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "ssd_loss.npz")
LOSS_RTOL = 2e-5


def _cases():
    g = np.load(GOLD)
    anchors = torch.from_numpy(g["anchors"])
    A = anchors.shape[0]
    for ci in range(int(g["n_cases"])):
        n = int(g[f"c{ci}_n"])
        gen = torch.Generator().manual_seed(int(g[f"c{ci}_logits_seed"]))
        logits = torch.randn(n, A, 91, generator=gen) * 2.0
        reg = torch.randn(n, A, 4, generator=gen)
        cnt = g[f"c{ci}_gt_counts"]
        targets = [{"boxes": torch.from_numpy(g[f"c{ci}_gt_boxes"][i, :cnt[i]].copy()), "labels": torch.from_numpy(g[f"c{ci}_gt_labels"][i, :cnt[i]].copy())}
                   for i in range(n)]
        yield ci, g, anchors, logits, reg, targets


def test_oracle_loss_matches_reference_golden():
    for ci, g, anchors, logits, reg, targets in _cases():
        losses, matched = so.ssd_loss_oracle(logits, reg, anchors, targets, float(g["iou_thresh"]), float(g["neg_to_pos_ratio"]))
        assert np.array_equal(matched.numpy(), g[f"c{ci}_matched"]), ci
        assert abs(losses["bbox_regression"].item() - float(g[f"c{ci}_bbox_regression"])) <= 1e-6 * abs(float(g[f"c{ci}_bbox_regression"]))
        assert abs(losses["classification"].item() - float(g[f"c{ci}_classification"])) <= 1e-6 * abs(float(g[f"c{ci}_classification"]))


@pytest.mark.gpu
def test_hip_loss_matches_reference_golden():
    from demonet_amd.loss import ssd_loss
    for ci, g, anchors, logits, reg, targets in _cases():
        losses, matched = ssd_loss({"cls_logits": logits.cuda(), "bbox_regression": reg.cuda()}, anchors.cuda(), targets,
                                   float(g["iou_thresh"]), float(g["neg_to_pos_ratio"]))
        assert np.array_equal(matched.cpu().numpy(), g[f"c{ci}_matched"]), ci
        for k in ("bbox_regression", "classification"):
            ref = float(g[f"c{ci}_{k}"])
            print(f"case {ci} {k}: reference {ref:.7f} HIP {losses[k].item():.7f}")
            assert abs(losses[k].item() - ref) <= LOSS_RTOL * abs(ref), (ci, k)


@pytest.mark.gpu
@pytest.mark.parametrize("n,A,K,gmaxs,ratio", [
    (2, 3234, 91, [3, 7], 3.0),
    (3, 777, 21, [0, 1, 40], 3.0),          # an image without boxes; a ragged anchor count
    (1, 500, 5, [200], 3.0),                # more negatives wanted than exist: the -inf ranking spills into the foreground anchors
    (2, 3000, 21, [12, 5], 2.5),            # non-integer ratio: ceil(ratio * #foreground) negatives
])
def test_hip_loss_vs_oracle_random(n, A, K, gmaxs, ratio):
    from demonet_amd.loss import ssd_loss
    rng = np.random.RandomState(n * 1000 + A)
    # anchors: a grid of boxes of mixed sizes (positive extents)
    c = rng.uniform(0, 300, (A, 2)).astype(np.float32)
    wh = rng.uniform(10, 120, (A, 2)).astype(np.float32)
    anchors = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1))
    logits = torch.from_numpy(rng.randn(n, A, K).astype(np.float32) * 3)
    reg = torch.from_numpy(rng.randn(n, A, 4).astype(np.float32))
    targets = []
    for gcount in gmaxs:
        if gcount and A == 500:
            # many ground-truth boxes, each an exact anchor: > A / (1 + ratio) foreground anchors
            idx = rng.choice(A, gcount, replace=False)
            b = anchors[idx].clone()
        else:
            xy = rng.uniform(0, 250, (gcount, 2)).astype(np.float32)
            b = torch.from_numpy(np.concatenate([xy, xy + rng.uniform(8, 150, (gcount, 2)).astype(np.float32)], 1))
        targets.append({"boxes": b.reshape(-1, 4), "labels": torch.from_numpy(rng.randint(1, K, (gcount,)).astype(np.int64))})
    want, wm = so.ssd_loss_oracle(logits, reg, anchors, targets, 0.5, ratio)
    got, gm = ssd_loss({"cls_logits": logits.cuda(), "bbox_regression": reg.cuda()}, [anchors.cuda()] * n, targets, 0.5, ratio)
    assert np.array_equal(gm.cpu().numpy(), wm.numpy())
    for k in want:
        print(f"{k}: oracle {want[k].item():.7f} HIP {got[k].item():.7f}")
        assert abs(got[k].item() - want[k].item()) <= LOSS_RTOL * abs(want[k].item()) + 1e-7


@pytest.mark.gpu
def test_hip_loss_error_behaviour():
    from demonet_amd.loss import ssd_loss
    A = 64
    anchors = torch.rand(A, 4).cuda()
    anchors[:, 2:] += anchors[:, :2] + 0.1
    ho = {"cls_logits": torch.zeros(1, A, 3).cuda(), "bbox_regression": torch.zeros(1, A, 4).cuda()}
    with pytest.raises(ValueError):        # generalized_ssd.py:300-308
        ssd_loss(ho, anchors, [{"boxes": torch.tensor([[5.0, 5.0, 5.0, 9.0]]), "labels": torch.tensor([1])}])
    with pytest.raises(RuntimeError):      # no CPU fallback
        ssd_loss({k: v.cpu() for k, v in ho.items()}, anchors.cpu(), [{"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, dtype=torch.int64)}])


@pytest.mark.gpu
def test_model_loss_entry_points():
    """SSD.loss / SSD.compute_loss (the reference's method name and argument order) against the oracle on the model's own head
    outputs and default boxes; a wrong matched_idxs is rejected."""
    from demonet_amd import models, synth
    m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
    imgs = torch.from_numpy(synth.images(91, 3, 320, 320)).cuda()
    rng = np.random.RandomState(3)
    targets = []
    for gcount in (2, 0, 5):
        xy = rng.uniform(0, 220, (gcount, 2)).astype(np.float32)
        targets.append({"boxes": torch.from_numpy(np.concatenate([xy, xy + rng.uniform(20, 90, (gcount, 2)).astype(np.float32)], 1)).reshape(-1, 4),
                        "labels": torch.from_numpy(rng.randint(1, 91, (gcount,)).astype(np.int64))})
    got = m.loss(imgs, targets)
    logits, reg = m.forward_heads(imgs)
    anchors = torch.from_numpy(m._lowered.anchors)
    want, wm = so.ssd_loss_oracle(logits.cpu(), reg.cpu(), anchors, targets)
    for k in want:
        assert abs(got[k].item() - want[k].item()) <= LOSS_RTOL * abs(want[k].item()) + 1e-7
    ok = m.compute_loss(targets, {"cls_logits": logits, "bbox_regression": reg}, [anchors.cuda()] * 3, list(wm))
    assert abs(ok["classification"].item() - want["classification"].item()) <= LOSS_RTOL * abs(want["classification"].item())
    bad = wm.clone()
    bad[0, 0] = 1 if int(bad[0, 0]) != 1 else 0
    with pytest.raises(ValueError):
        m.compute_loss(targets, {"cls_logits": logits, "bbox_regression": reg}, anchors.cuda(), list(bad))
    m.train()
    with pytest.raises(NotImplementedError):
        m([imgs[0]], targets[:1])
    m.eval()
