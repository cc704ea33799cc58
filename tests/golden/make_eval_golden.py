"""Authoring-time generator of tests/golden/voc_eval.npz: seeded precision/recall curves and detection sets scored by the
REAL reference functions (/root/reference/demonet/data/voc_eval.py: voc_ap; the TP/FP loop of voc_eval is executed through a
thin driver because the function itself reads VOC xml files and uses the removed numpy alias np.bool).
Run in the authoring container only; the .npz is the committed fixture."""
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
np.savez_compressed(os.path.join(HERE, "voc_eval.npz"), **out)
print("wrote", len(out), "arrays")
