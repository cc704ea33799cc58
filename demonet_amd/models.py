"""Factory namespace mirroring `demonet.models` (reference: demonet/models/__init__.py:1-2; looked up as
`models.__dict__[name](num_classes=..., pretrained=...)` at demonet/train.py:154)."""
import warnings
from typing import Any, Optional

import torch

from . import spec
from .ssd import SSD

__all__ = ["ssdlite320_mobilenet_v3_large", "ssd300_vgg16", "ssd512_vgg16", "ssd_lite_mobilenet_v2"]

_POST_KEYS = ("score_thresh", "nms_thresh", "detections_per_img", "topk_candidates")


def _split_kwargs(kwargs):
    post = {k: kwargs.pop(k) for k in list(kwargs) if k in _POST_KEYS}
    for k in ("image_mean", "image_std", "iou_thresh", "positive_fraction"):
        kwargs.pop(k, None)
    return post


def _no_download(pretrained, pretrained_backbone):
    if pretrained or pretrained_backbone:
        raise RuntimeError("pretrained weights need a network download (ssd_mobilenetv3.py:221-226); load a checkpoint "
                           "with model.load_state_dict(torch.load(path)) instead -- keys are the reference's")


def ssdlite320_mobilenet_v3_large(pretrained: bool = False, progress: bool = True, num_classes: int = 91,
                                  pretrained_backbone: bool = False, trainable_backbone_layers: Optional[int] = None,
                                  norm_layer=None, **kwargs: Any) -> SSD:
    """reference: demonet/models/ssd_mobilenetv3.py:159-227 (same signature and defaults)."""
    if "size" in kwargs:
        warnings.warn("The size of the model is already fixed; ignoring the argument.")      # :183-184
        kwargs.pop("size")
    _no_download(pretrained, pretrained_backbone)
    post = _split_kwargs(kwargs)
    return SSD(spec.ssdlite320_mobilenet_v3_large_graph(num_classes=num_classes, **post), init="normal")


def ssd300_vgg16(pretrained: bool = False, progress: bool = True, num_classes: int = 91,
                 pretrained_backbone: bool = False, trainable_backbone_layers: Optional[int] = None, **kwargs: Any) -> SSD:
    """reference: demonet/models/ssd_vgg16.py:139-213. (pretrained_backbone defaults to False here: no network.)"""
    if "size" in kwargs:
        warnings.warn("The size of the model is already fixed; ignoring the argument.")
        kwargs.pop("size")
    _no_download(pretrained, pretrained_backbone)
    post = _split_kwargs(kwargs)
    return SSD(spec.ssd300_vgg16_graph(num_classes=num_classes, **post), init="xavier")


def ssd512_vgg16(pretrained: bool = False, progress: bool = True, num_classes: int = 91, **kwargs: Any) -> SSD:
    """Build-defined: the reference's `highres` extractor (ssd_vgg16.py:74-81) at 512x512 with SSD-paper anchors."""
    _no_download(pretrained, False)
    post = _split_kwargs(kwargs)
    return SSD(spec.ssd512_vgg16_graph(num_classes=num_classes, **post), init="xavier")


def ssd_lite_mobilenet_v2(pretrained: bool = False, image_size: int = 320, score_thresh: float = 0.5,
                          num_classes: int = 21) -> SSD:
    """reference: hubconf.py:25-44 (signature); structure from backbone.py:45-119 + box_head.py:24-56."""
    _no_download(pretrained, False)
    return SSD(spec.ssd_lite_mobilenet_v2_graph(image_size=image_size, num_classes=num_classes, score_thresh=score_thresh),
               init="xavier")


def load_synthetic(model: SSD, seed: int = 0) -> SSD:
    """Fills `model` with the build-owned synthetic weights (demonet_amd/synth.py), keyed like the reference."""
    from . import synth
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict(model.graph, seed).items()}
    model.load_state_dict(sd, strict=True)
    return model
