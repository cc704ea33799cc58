"""Authoring-time generator of tests/golden/coco_records.json: seeded padded detections and the result records the REAL reference
`CocoEvaluator.prepare_for_coco_detection` (/root/reference/demonet/data/coco_eval.py:76-98, with convert_to_xywh :162-164) builds
from them. The module imports pycocotools and torch._six at import time; neither is installed / exists here and neither is touched by
the function under test, so empty stand-in modules are registered for the import only. Run in the authoring container only."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
for name, attrs in (("pycocotools", {}), ("pycocotools.cocoeval", {"COCOeval": object}), ("pycocotools.coco", {"COCO": object}),
                    ("pycocotools.mask", {}), ("torch._six", {"string_classes": (str,)})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
pkg = types.ModuleType("demonet"); pkg.__path__ = ["/root/reference/demonet"]; sys.modules["demonet"] = pkg
data = types.ModuleType("demonet.data"); data.__path__ = ["/root/reference/demonet/data"]; sys.modules["demonet.data"] = data
util = types.ModuleType("demonet.util"); util.__path__ = ["/root/reference/demonet/util"]; sys.modules["demonet.util"] = util
misc = types.ModuleType("demonet.util.misc"); misc.all_gather = lambda x: [x]; sys.modules["demonet.util.misc"] = misc
spec = importlib.util.spec_from_file_location("demonet.data.coco_eval", "/root/reference/demonet/data/coco_eval.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.RandomState(77)
N, D = 5, 12
counts = np.array([3, 0, 12, 1, 7], np.int32)
boxes = np.zeros((N, D, 4), np.float32)
scores = np.zeros((N, D), np.float32)
labels = np.zeros((N, D), np.int64)
for i in range(N):
    c = counts[i]
    xy = rng.uniform(0, 300, (c, 2)).astype(np.float32)
    boxes[i, :c] = np.concatenate([xy, xy + rng.uniform(1, 200, (c, 2)).astype(np.float32)], 1)
    scores[i, :c] = np.sort(rng.uniform(0.01, 1, c).astype(np.float32))[::-1]
    labels[i, :c] = rng.randint(1, 91, c)
image_ids = [139, 285, 632, 724, 776]
pred = {iid: {"boxes": torch.from_numpy(boxes[i, :counts[i]]), "scores": torch.from_numpy(scores[i, :counts[i]]),
              "labels": torch.from_numpy(labels[i, :counts[i]])} for i, iid in enumerate(image_ids)}
# the engine skips nothing itself (engine.py builds res = {image_id: output}); prepare_for_coco_detection skips len(prediction) == 0 --
# a dict with empty tensors has len 3, so an image with zero detections contributes zero records through the empty lists
records = ref.CocoEvaluator.prepare_for_coco_detection(None, pred)
with open(os.path.join(HERE, "coco_records.json"), "w") as f:
    json.dump({"boxes": boxes.tolist(), "scores": scores.tolist(), "labels": labels.tolist(), "counts": counts.tolist(),
               "image_ids": image_ids, "records": records}, f)
print(len(records), "records ->", os.path.join(HERE, "coco_records.json"))
