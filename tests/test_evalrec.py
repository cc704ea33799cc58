"""evaluation records built from the device's detections (demonet_amd/evalrec.py).
voc_ap is pinned by golden vectors produced by the reference's own function (tests/golden/make_eval_golden.py);
the COCO record layout and the VOC TP/FP marking are checked against a literal per-detection restatement."""
import os

import numpy as np
import torch

from demonet_amd import evalrec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voc_eval.npz")


def test_voc_ap_matches_reference_vectors():
    z = np.load(GOLDEN)
    n = sum(1 for k in z.files if k.startswith("rec"))
    assert n >= 10
    for i in range(n):
        rec, prec = z[f"rec{i}"], z[f"prec{i}"]
        assert evalrec.voc_ap(rec, prec, True) == float(z[f"ap07_{i}"])          # data/voc_eval.py:33-41
        assert evalrec.voc_ap(rec, prec, False) == float(z[f"ap_{i}"])           # data/voc_eval.py:42-57


def test_coco_records_follow_prepare_for_coco_detection():
    g = torch.Generator().manual_seed(3)
    n, D = 3, 5
    xy = torch.rand(n, D, 2, generator=g) * 200
    wh = torch.rand(n, D, 2, generator=g) * 100 + 1
    boxes = torch.cat([xy, xy + wh], dim=2)
    scores = torch.rand(n, D, generator=g)
    labels = torch.randint(1, 91, (n, D), generator=g)
    counts = torch.tensor([5, 0, 2], dtype=torch.int32)
    recs = evalrec.coco_detection_records(boxes, scores, labels, counts, image_ids=[11, 22, 33])
    assert len(recs) == 7 and all(r["image_id"] != 22 for r in recs)            # coco_eval.py:79-80 skips empty predictions
    r = recs[5]                                                                 # first detection of image 33
    b = boxes[2, 0]
    assert r["image_id"] == 33 and r["category_id"] == int(labels[2, 0]) and r["score"] == float(scores[2, 0])
    np.testing.assert_array_equal(np.float32(r["bbox"]), torch.stack((b[0], b[1], b[2] - b[0], b[3] - b[1])).numpy())   # :162-164
    assert set(r) == {"image_id", "category_id", "bbox", "score"}


def test_coco_records_equal_reference_function_output():
    """tests/golden/coco_records.json: padded detections and the records the REAL CocoEvaluator.prepare_for_coco_detection
    (data/coco_eval.py:76-98,162-164) built from them (make_coco_records_golden.py) -- same dicts, same order, same float values."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coco_records.json")))
    recs = evalrec.coco_detection_records(np.array(g["boxes"], np.float32), np.array(g["scores"], np.float32), np.array(g["labels"], np.int64),
                                          np.array(g["counts"], np.int32), g["image_ids"])
    assert recs == g["records"]


def test_voc_matching_rules():
    """voc_eval.py:116-153: confidence order, +1 pixel IoU, one claim per ground truth, 'difficult' ignored."""
    gt = {"a": (np.array([[10, 10, 50, 50], [100, 100, 150, 150]]), np.array([False, True])),
          "b": (np.array([[0, 0, 20, 20]]), np.array([False]))}
    ids = ["a", "a", "a", "b", "b"]
    scores = np.array([0.9, 0.8, 0.7, 0.95, 0.2])
    boxes = np.array([[12, 12, 50, 50],        # TP on a[0]
                      [10, 10, 48, 48],        # duplicate of a[0] -> FP
                      [101, 99, 150, 151],     # matches the difficult box -> neither TP nor FP
                      [200, 200, 220, 220],    # no overlap -> FP
                      [1, 1, 20, 20]])         # TP on b[0]
    rec, prec = evalrec.voc_class_pr(ids, scores, boxes, gt, 0.5)
    # confidence order: b(0.95) FP, a(0.9) TP, a(0.8) FP, a(0.7) ignored, b(0.2) TP ; npos = 2
    np.testing.assert_allclose(rec, [0.0, 0.5, 0.5, 0.5, 1.0])
    np.testing.assert_allclose(prec, [0.0, 0.5, 1 / 3, 1 / 3, 0.5])
    assert abs(evalrec.voc_ap(rec, prec, False) - 0.5) < 1e-12


def test_voc_class_pr_matches_reference_voc_eval():
    """TP/FP marking + precision/recall of evalrec.voc_class_pr against vectors produced by the reference's own voc_eval
    (data/voc_eval.py:60-165) run on synthetic VOC xml / detection files (tests/golden/make_eval_golden.py): 3 classes, 14 images,
    duplicates, 'difficult' boxes, images without ground truth, two overlap thresholds."""
    z = np.load(GOLDEN)
    nimg = int(z["ve_num_images"])
    for ci in range(3):
        gt = {}
        for k in range(nimg):
            sel = z[f"ve_gt_img_{ci}"] == k
            gt[k] = (z[f"ve_gt_box_{ci}"][sel], z[f"ve_gt_diff_{ci}"][sel].astype(bool))
        ids = z[f"ve_det_img_{ci}"].tolist()
        for thr, tag in ((0.5, "t50"), (0.3, "t30")):
            rec, prec = evalrec.voc_class_pr(ids, z[f"ve_det_score_{ci}"], z[f"ve_det_box_{ci}"], gt, thr)
            np.testing.assert_array_equal(rec, z[f"ve_rec_{ci}_{tag}"])
            np.testing.assert_array_equal(prec, z[f"ve_prec_{ci}_{tag}"])
            assert evalrec.voc_ap(rec, prec, False) == float(z[f"ve_ap_{ci}_{tag}"])
            assert evalrec.voc_ap(rec, prec, True) == float(z[f"ve_ap07_{ci}_{tag}"])
        assert len(ids) > 10 and float(z[f"ve_rec_{ci}_t50"].max()) > 0
