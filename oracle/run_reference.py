"""ORACLE TOOLING -- authoring container only (needs /root/reference; never runs on the GPU box).

Imports the *real* reference (`/root/reference/demonet/models`) unmodified on CPU, using the
test-only torchvision stand-in in oracle/ref_shim/ (torchvision is not installed here; see
SURVEY.md section 8c / Appendix C). Used to validate oracle/ssd_oracle.py and to generate the
golden vectors committed under tests/golden/ (tests/golden/make_golden.py).
"""
import importlib
import os
import sys
import types
import warnings

import torch

REFERENCE_ROOT = os.environ.get("DEMONET_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "demonet", "models"))


def import_reference_models():
    """Returns the reference's `demonet.models` package."""
    if "demonet.models" in sys.modules:
        return sys.modules["demonet.models"]
    if not available():
        raise RuntimeError("reference not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True                       # /root/reference is read-only
    shim = os.path.join(_HERE, "ref_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    # skip demonet/__init__.py (it pulls demonet.data -> pycocotools, absent here)
    pkg = types.ModuleType("demonet")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "demonet")]
    sys.modules["demonet"] = pkg
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return importlib.import_module("demonet.models")


def build_reference_model(name: str, num_classes: int, state_dict=None, **kwargs):
    """Instantiate a reference model (no pretrained weights: no network) and load `state_dict`."""
    models = import_reference_models()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if name == "ssdlite320_mobilenet_v3_large":
            m = models.ssdlite320_mobilenet_v3_large(pretrained=False, pretrained_backbone=False,
                                                     num_classes=num_classes, **kwargs)
        elif name == "ssd300_vgg16":
            m = models.ssd300_vgg16(pretrained=False, pretrained_backbone=False, num_classes=num_classes, **kwargs)
        elif name == "ssd512_vgg16":
            # build-defined: the reference's own classes composed with highres=True
            from demonet.models import ssd_vgg16 as sv
            from demonet.models.generalized_ssd import SSD
            from demonet.models.anchor_utils import DefaultBoxGenerator
            backbone = sv._vgg_extractor("vgg16_features", True, False, False, 5)
            ag = DefaultBoxGenerator([[2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
                                     scales=[0.04, 0.1, 0.26, 0.42, 0.58, 0.74, 0.9, 1.06],
                                     steps=[8, 16, 32, 64, 128, 256, 512])
            defaults = {"image_mean": [0.48235, 0.45882, 0.40784], "image_std": [1.0 / 255.0] * 3}
            m = SSD(backbone, ag, (512, 512), num_classes, **{**defaults, **kwargs})
        else:
            raise ValueError(name)
    if state_dict is not None:
        sd = {k: (torch.from_numpy(v.copy()) if not isinstance(v, torch.Tensor) else v) for k, v in state_dict.items()}
        m.load_state_dict(sd, strict=True)
    m.eval()
    return m


class ReferenceV2Composite(torch.nn.Module):
    """The hub model `ssd_lite_mobilenet_v2` does not import at this commit (hubconf.py:4). This composes the
    reference's own importable pieces: mobilenetv2.features taps 13/18 ->
    backbone.ExtraBlocks -> box_head.MultiBoxLiteHead; anchors/post-process through the reference's
    DefaultBoxGenerator / SSD.postprocess_detections."""

    def __init__(self, num_classes=21, image_size=320, **post):
        super().__init__()
        models = import_reference_models()
        from demonet.models import backbone as rb, box_head as bh, mobilenetv2 as mv2
        from demonet.models.anchor_utils import DefaultBoxGenerator
        from demonet.models.transform import GeneralizedRCNNTransform
        from demonet.models import _utils as det_utils
        self.backbone = torch.nn.Module()
        self.backbone.body = mv2.mobilenet_v2().features
        self.backbone.extra_blocks = rb.ExtraBlocks(1280, [512, 256, 256, 64], [0.2, 0.25, 0.5, 0.25], [2, 2, 2, 2])
        self.head = bh.MultiBoxLiteHead([96, 1280, 512, 256, 256, 64], [6] * 6, num_classes)
        self.anchor_generator = DefaultBoxGenerator([[2, 3]] * 6, min_ratio=0.2, max_ratio=0.95)
        self.transform = GeneralizedRCNNTransform(image_size, image_size, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225],
                                                  size_divisible=1, fixed_size=(image_size, image_size))
        self.box_coder = det_utils.BoxCoder(weights=(10., 10., 5., 5.))
        p = {**dict(score_thresh=0.5, nms_thresh=0.45, detections_per_img=100, topk_candidates=400), **post}
        self.score_thresh, self.nms_thresh = p["score_thresh"], p["nms_thresh"]
        self.detections_per_img, self.topk_candidates = p["detections_per_img"], p["topk_candidates"]
        self._models = models

    def forward_raw(self, images):
        from demonet.models.generalized_ssd import SSD
        il, _ = self.transform(images, None)
        x = il.tensors
        feats = []
        for i, m in enumerate(self.backbone.body):
            x = m(x)
            if i in (13, 18):
                feats.append(x)
        for m in self.backbone.extra_blocks:
            x = m(x)
            feats.append(x)
        logits, reg = self.head(feats)
        anchors = self.anchor_generator(il, feats)
        return il, feats, {"cls_logits": logits, "bbox_regression": reg}, anchors

    def forward(self, images):
        from demonet.models.generalized_ssd import SSD
        orig = [tuple(i.shape[-2:]) for i in images]
        il, feats, head, anchors = self.forward_raw(images)
        dets = SSD.postprocess_detections(self, head, anchors, il.image_sizes)
        return self.transform.postprocess(dets, il.image_sizes, orig)
