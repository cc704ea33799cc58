"""TEST-ONLY stand-in for the parts of torchvision the reference imports.

torchvision is NOT installed in the authoring container.
This package exists only so that `oracle/run_reference.py` can import
/root/reference/demonet/models unmodified and run it on CPU to (a) validate the
oracle restatement and (b) generate the golden vectors under tests/golden/.
It never ships in the product path and is never imported on the GPU box.

Surface = exactly what the reference touches on the inference path:
  torchvision._is_tracing                      (transform.py:31,210)
  torchvision.models.detection.image_list.ImageList   (anchor_utils.py:5, transform.py:7)
  torchvision.ops.boxes.{clip_boxes_to_image,batched_nms,box_iou}  (generalized_ssd.py:8,336,363,389)
  import-time-only stubs for everything else.
clip_boxes_to_image / nms / batched_nms are restatements of torchvision's published
semantics and delegate to oracle/ssd_oracle.py (third-party arithmetic: "parity unpinned").
"""
from . import ops, models  # noqa: F401


def _is_tracing():
    return False
