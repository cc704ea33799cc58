"""Factory namespace mirroring `demonet.models` (reference: demonet/models/__init__.py:1-2; looked up as
`models.__dict__[name](num_classes=..., pretrained=...)` at demonet/train.py:154)."""
import warnings
from typing import Any, Optional

import torch

from . import spec
from .ssd import SSD

__all__ = ["ssdlite320_mobilenet_v3_large", "ssd300_vgg16", "ssd512_vgg16", "ssd_lite_mobilenet_v2"]

_POST_KEYS = ("score_thresh", "nms_thresh", "detections_per_img", "topk_candidates")
_TRAIN_KEYS = ("iou_thresh", "positive_fraction")       # SSD.__init__ arguments that only the training loss reads


def _split_kwargs(kwargs, factory):
    """The reference factories merge the caller's kwargs over their defaults and hand them to SSD.__init__
    ({**defaults, **kwargs}: ssd_mobilenetv3.py:207-218, ssd_vgg16.py:200-206; SSD.__init__ generalized_ssd.py:154-163).
    Returns (post-process overrides, (image_mean, image_std) overrides). The two training-only arguments are validated and
    dropped (no training path here); anything else raises like the reference's SSD.__init__ would."""
    post = {k: kwargs.pop(k) for k in list(kwargs) if k in _POST_KEYS}
    norm = {}
    for k in ("image_mean", "image_std"):
        if k in kwargs:
            v = kwargs.pop(k)
            if v is not None:
                v = [float(x) for x in v]
                if len(v) != 3:
                    raise ValueError(f"{k} must have 3 entries (one per input channel), got {len(v)}")
                norm[k] = v
    for k in _TRAIN_KEYS:
        kwargs.pop(k, None)
    if kwargs:
        raise TypeError(f"{factory}() got unexpected keyword argument(s) {sorted(kwargs)}: the MI355X path builds the "
                        f"reference's default architecture only")
    return post, norm


def _apply_norm(graph, norm):
    if "image_mean" in norm:
        graph.image_mean = norm["image_mean"]
    if "image_std" in norm:
        if any(s == 0.0 for s in norm["image_std"]):
            raise ValueError("image_std must be non-zero")
        graph.image_std = norm["image_std"]
    return graph


def _bn_eps(norm_layer, default):
    """norm_layer of the SSDLite factory (ssd_mobilenetv3.py:195-196): only BatchNorm2d (eval mode: folded into the convs at
    plan time) is supported; its eps is honoured. Anything else has no folded form here."""
    import functools
    from torch import nn
    if norm_layer is None:
        return default
    if norm_layer is nn.BatchNorm2d:
        return 1e-5
    if isinstance(norm_layer, functools.partial) and norm_layer.func is nn.BatchNorm2d and not norm_layer.args:
        extra = set(norm_layer.keywords) - {"eps", "momentum", "affine", "track_running_stats"}
        if not extra and norm_layer.keywords.get("affine", True) and norm_layer.keywords.get("track_running_stats", True):
            return float(norm_layer.keywords.get("eps", 1e-5))
    raise NotImplementedError("norm_layer must be nn.BatchNorm2d or functools.partial(nn.BatchNorm2d, eps=..., momentum=...): "
                              "the HIP path folds eval-mode batch norm into the convolutions")


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
    post, norm = _split_kwargs(kwargs, "ssdlite320_mobilenet_v3_large")
    eps = _bn_eps(norm_layer, 1e-3)
    return SSD(_apply_norm(spec.ssdlite320_mobilenet_v3_large_graph(num_classes=num_classes, eps=eps, **post), norm), init="normal")


def ssd300_vgg16(pretrained: bool = False, progress: bool = True, num_classes: int = 91,
                 pretrained_backbone: bool = False, trainable_backbone_layers: Optional[int] = None, **kwargs: Any) -> SSD:
    """reference: demonet/models/ssd_vgg16.py:139-213. (pretrained_backbone defaults to False here: no network.)"""
    if "size" in kwargs:
        warnings.warn("The size of the model is already fixed; ignoring the argument.")
        kwargs.pop("size")
    _no_download(pretrained, pretrained_backbone)
    post, norm = _split_kwargs(kwargs, "ssd300_vgg16")
    return SSD(_apply_norm(spec.ssd300_vgg16_graph(num_classes=num_classes, **post), norm), init="xavier")


def ssd512_vgg16(pretrained: bool = False, progress: bool = True, num_classes: int = 91, **kwargs: Any) -> SSD:
    """Build-defined: the reference's `highres` extractor (ssd_vgg16.py:74-81) at 512x512 with SSD-paper anchors."""
    _no_download(pretrained, False)
    post, norm = _split_kwargs(kwargs, "ssd512_vgg16")
    return SSD(_apply_norm(spec.ssd512_vgg16_graph(num_classes=num_classes, **post), norm), init="xavier")


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
