"""The loop around the hot path: the reference's `evaluate` (demonet/engine.py:70-110) with the forwards kept in flight.

The reference walks the data loader one batch at a time: images to the device, synchronize, `model(images)`, every output to the
host, `{image_id: output}` into the evaluator (engine.py:84-100). Here the same loop submits each batch to a
`pipeline.ForwardPipeline` and collects a batch's detections `depth` submissions later, so the device works on the next batches
while the host converts the previous ones: same `{image_id: {"boxes", "scores", "labels"}}` records (host tensors), same order.
The COCO evaluator itself (pycocotools, `data/coco_eval.py`) is third-party and absent here; `evalrec.coco_detection_records`
turns the records into the list `CocoEvaluator.prepare_for_coco_detection` builds (data/coco_eval.py:76-98) and
`evalrec.voc_mean_ap` scores them the PASCAL VOC way.
"""
import collections
import time
from typing import Dict, Iterable, List, Tuple

import torch

from .pipeline import ForwardPipeline


def _image_id(target):
    v = target["image_id"]
    return int(v.item()) if hasattr(v, "item") else int(v)


@torch.no_grad()
def evaluate(model, data_loader: Iterable, device="cuda:0", depth: int = 3) -> Tuple[Dict[int, Dict[str, torch.Tensor]], Dict[str, float]]:
    """data_loader yields (images, targets) as the reference's loaders do (engine.py:84): images = a list of [3,H,W] float tensors
    in [0,1] (or one [N,3,H,W] tensor), targets = a list of dicts holding "image_id". Batches whose images share one size go through
    the pipeline (any size: the device resizes to the network size and maps the boxes back, transform.py:27-53,278-292); a batch of
    mixed sizes is run by `model(images)` after the pipeline has drained, so the records keep the loader's order.
    Returns ({image_id: {"boxes" [n,4], "scores" [n], "labels" [n] int64}} on the host, {"images", "seconds", "images_per_sec",
    "model_seconds" (host time spent submitting and collecting)})."""
    device = torch.device(device)
    model.eval()
    results: Dict[int, Dict[str, torch.Tensor]] = collections.OrderedDict()
    pending = collections.deque()
    pipe = None
    shape = None
    n_images = 0
    model_time = 0.0

    def collect(ticket, ids):
        boxes, scores, labels, counts = pipe.result(ticket)
        host = [t.cpu() for t in (boxes, scores, labels, counts)]           # engine.py:92: outputs to the host
        for i, image_id in enumerate(ids):
            c = int(host[3][i])
            results[image_id] = {"boxes": host[0][i, :c].clone(), "scores": host[1][i, :c].clone(), "labels": host[2][i, :c].clone()}

    def drain():
        while pending:
            collect(*pending.popleft())

    t_start = time.perf_counter()
    for images, targets in data_loader:
        ids = [_image_id(t) for t in targets]
        same = isinstance(images, torch.Tensor) or len({tuple(im.shape) for im in images}) == 1
        t0 = time.perf_counter()
        if same:
            batch = images if isinstance(images, torch.Tensor) else torch.stack(list(images))
            batch = batch.to(device, non_blocking=True)
            if pipe is None or tuple(batch.shape) != shape:
                drain()
                if pipe is not None:
                    pipe.close()
                shape = tuple(batch.shape)
                pipe = ForwardPipeline(model, shape[0], height=shape[2], width=shape[3], depth=depth, device=device)
            if len(pending) == depth:                                       # its slot is about to be reused
                collect(*pending.popleft())
            pending.append((pipe.submit(batch), ids))
        else:
            drain()
            outputs = model([im.to(device) for im in images])              # engine.py:90
            for image_id, out in zip(ids, outputs):
                results[image_id] = {k: v.cpu() for k, v in out.items()}
        model_time += time.perf_counter() - t0
        n_images += len(ids)
    t0 = time.perf_counter()
    drain()
    if pipe is not None:
        pipe.close()
    model_time += time.perf_counter() - t0
    dt = time.perf_counter() - t_start
    return results, {"images": n_images, "seconds": dt, "images_per_sec": n_images / max(dt, 1e-9), "model_seconds": model_time}


def coco_records(results: Dict[int, Dict[str, torch.Tensor]]) -> List[dict]:
    """The list `CocoEvaluator.prepare_for_coco_detection` builds from such a dict (data/coco_eval.py:76-98; xyxy -> xywh :162-164)."""
    out = []
    for image_id, pred in results.items():
        if len(pred["boxes"]) == 0:                                         # coco_eval.py:79-80
            continue
        b = pred["boxes"]
        xywh = torch.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), dim=1).tolist()
        scores, labels = pred["scores"].tolist(), pred["labels"].tolist()
        out.extend({"image_id": image_id, "category_id": labels[k], "bbox": xywh[k], "score": scores[k]} for k in range(len(xywh)))
    return out
