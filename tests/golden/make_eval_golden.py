"""Authoring-time generator of tests/golden/voc_eval.npz: seeded precision/recall curves scored by the REAL reference
voc_ap, and seeded detection / ground-truth sets scored by the REAL reference voc_eval (/root/reference/demonet/data/voc_eval.py:
60-165, the TP/FP marking loop) -- the function reads PASCAL VOC xml annotations and per-class detection text files, so the
generator writes those into a temporary directory first. (voc_eval.py:92 uses the alias np.bool that numpy >= 1.24 removed: the
alias is restored for this process only.)  Run in the authoring container only; the .npz is the committed fixture."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_voc_eval", "/root/reference/demonet/data/voc_eval.py")
ref = importlib.util.module_from_spec(spec)
sys.dont_write_bytecode = True
spec.loader.exec_module(ref)

rng = np.random.RandomState(20240607)
out = {}
for i in range(12):
    n = int(rng.randint(1, 60))
    tp = (rng.rand(n) < 0.6).astype(np.float64)
    fp = 1.0 - tp
    npos = int(tp.sum() + rng.randint(0, 5)) or 1
    rec = np.cumsum(tp) / npos
    prec = np.cumsum(tp) / np.maximum(np.cumsum(tp) + np.cumsum(fp), np.finfo(np.float64).eps)
    out[f"rec{i}"], out[f"prec{i}"] = rec, prec
    out[f"ap07_{i}"] = np.float64(ref.voc_ap(rec, prec, True))
    out[f"ap_{i}"] = np.float64(ref.voc_ap(rec, prec, False))

# ---- voc_eval (TP/FP marking, voc_eval.py:116-165) on synthetic VOC-style files -----------------------------------------
import tempfile

np.bool = bool          # numpy alias removed in 1.24; the reference still uses it (voc_eval.py:92)
CLASSES = ["cat", "dog", "bus"]
with tempfile.TemporaryDirectory() as tmp:
    names = [f"img{k:03d}" for k in range(14)]
    with open(os.path.join(tmp, "set.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    gts = {}
    for nm in names:
        objs = []
        for _ in range(int(rng.randint(0, 6))):
            x1, y1 = int(rng.randint(0, 200)), int(rng.randint(0, 200))
            w, h = int(rng.randint(8, 110)), int(rng.randint(8, 110))
            objs.append((CLASSES[int(rng.randint(0, 3))], int(rng.rand() < 0.2), (x1, y1, x1 + w, y1 + h)))
        gts[nm] = objs
        with open(os.path.join(tmp, nm + ".xml"), "w") as f:
            f.write("<annotation>" + "".join(
                f"<object><name>{c}</name><difficult>{d}</difficult><bndbox><xmin>{b[0]}</xmin><ymin>{b[1]}</ymin>"
                f"<xmax>{b[2]}</xmax><ymax>{b[3]}</ymax></bndbox></object>" for c, d, b in objs) + "</annotation>")
    for ci, cls in enumerate(CLASSES):
        dets = []
        for nm in names:
            for c, d, b in gts[nm]:                      # jittered copies of the ground truth (some of them twice), plus clutter
                if c == cls:
                    for _ in range(int(rng.randint(0, 3))):
                        j = rng.randint(-12, 13, 4)
                        dets.append((nm, float(rng.rand()), [float(b[q] + j[q]) + float(rng.rand()) for q in range(4)]))
            for _ in range(int(rng.randint(0, 3))):
                x1, y1 = rng.rand(2) * 250
                dets.append((nm, float(rng.rand()), [float(x1), float(y1), float(x1 + 5 + rng.rand() * 90), float(y1 + 5 + rng.rand() * 90)]))
        with open(os.path.join(tmp, f"det_{cls}.txt"), "w") as f:
            for nm, sc, b in dets:
                f.write(f"{nm} {sc!r} {b[0]!r} {b[1]!r} {b[2]!r} {b[3]!r}\n")
        for thr, tag in ((0.5, "t50"), (0.3, "t30")):
            rec, prec, ap = ref.voc_eval(os.path.join(tmp, "det_{}.txt"), os.path.join(tmp, "{}.xml"), os.path.join(tmp, "set.txt"), cls, thr, False)
            _, _, ap07 = ref.voc_eval(os.path.join(tmp, "det_{}.txt"), os.path.join(tmp, "{}.xml"), os.path.join(tmp, "set.txt"), cls, thr, True)
            out[f"ve_rec_{ci}_{tag}"], out[f"ve_prec_{ci}_{tag}"] = rec, prec
            out[f"ve_ap_{ci}_{tag}"], out[f"ve_ap07_{ci}_{tag}"] = np.float64(ap), np.float64(ap07)
        # the inputs, as arrays: detections (image index, score, box) and ground truth (image index, difficult, box) of the class
        out[f"ve_det_img_{ci}"] = np.array([names.index(nm) for nm, _, _ in dets], dtype=np.int64)
        out[f"ve_det_score_{ci}"] = np.array([sc for _, sc, _ in dets], dtype=np.float64)
        out[f"ve_det_box_{ci}"] = np.array([b for _, _, b in dets], dtype=np.float64).reshape(-1, 4)
        g = [(names.index(nm), d, b) for nm in names for c, d, b in gts[nm] if c == cls]
        out[f"ve_gt_img_{ci}"] = np.array([t[0] for t in g], dtype=np.int64)
        out[f"ve_gt_diff_{ci}"] = np.array([t[1] for t in g], dtype=np.int64)
        out[f"ve_gt_box_{ci}"] = np.array([t[2] for t in g], dtype=np.float64).reshape(-1, 4)
out["ve_num_images"] = np.int64(14)
np.savez_compressed(os.path.join(HERE, "voc_eval.npz"), **out)
print("wrote", len(out), "arrays")
