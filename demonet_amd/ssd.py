"""Host-side mirror of the reference's detector module for the inference path.

`SSD` keeps the reference's contract (demonet/models/generalized_ssd.py:95-349): an nn.Module built by a factory,
`forward(images: List[Tensor[3,H,W]], targets=None) -> List[Dict[boxes, scores, labels]]` in eval mode, a
state_dict with the reference's key names, the same exceptions for malformed input -- but its body is one call
into the HIP library through the C ABI (include/demonet_hip.h). There is no PyTorch/CPU fallback: without the
library or without a GPU the forward raises.
"""
import ctypes as C
import warnings
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
from torch import nn, Tensor

from . import _lib
from .plan import LoweredModel
from .spec import Graph


# Structure epoch: bumped whenever a module OF THIS PACKAGE (an SSD or one of its name containers) registers a parameter, a buffer or a
# sub-module, i.e. whenever the set of tensor OBJECTS behind a model can have changed (swapped Parameters, load_state_dict(assign=True),
# add_module ...). SSD caches its tensor list per epoch, so the per-forward "did the weights change" test walks a flat list (data_ptr +
# _version of ~480 tensors: tens of microseconds) instead of the module tree (~0.4 ms per call -- a third of a synchronous 64-image
# forward). The hooks live on these two classes only: unrelated modules in the process neither pay for them nor invalidate the cache.
_STRUCT_EPOCH = [0]


class _Tracked(nn.Module):
    def register_parameter(self, name, param):
        _STRUCT_EPOCH[0] += 1
        return super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        _STRUCT_EPOCH[0] += 1
        return super().register_buffer(name, tensor, persistent)

    def add_module(self, name, module):
        _STRUCT_EPOCH[0] += 1
        return super().add_module(name, module)

    def register_module(self, name, module):
        _STRUCT_EPOCH[0] += 1
        return super().register_module(name, module)

    def __setattr__(self, name, value):
        # nn.Module.__setattr__ stores sub-modules and re-assigned buffers without going through the register_* methods
        if isinstance(value, (Tensor, nn.Module)) or name in self.__dict__.get("_parameters", ()) or name in self.__dict__.get("_buffers", ()) \
                or name in self.__dict__.get("_modules", ()):
            _STRUCT_EPOCH[0] += 1
        return super().__setattr__(name, value)

    def __delattr__(self, name):
        _STRUCT_EPOCH[0] += 1
        return super().__delattr__(name)


class _Node(_Tracked):
    """Container that only exists to reproduce the reference's dotted parameter names."""


def _attach(root: nn.Module, key: str, value: Tensor, buffer: bool):
    parts = key.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], value)
    else:
        m.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))


class SSD(_Tracked):
    def __init__(self, graph: Graph, init: str = "normal"):
        super().__init__()
        self.graph = graph
        gen = torch.Generator().manual_seed(0)
        for p in graph.params:
            if p.kind == "conv_w":
                if init == "xavier":                                   # generalized_ssd.py:17-22
                    fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                    fan_out = p.shape[0] * p.shape[2] * p.shape[3]
                    a = (6.0 / (fan_in + fan_out)) ** 0.5
                    v = (torch.rand(p.shape, generator=gen) * 2 - 1) * a
                else:                                                  # ssd_mobilenetv3.py:57-62 (std 0.03)
                    v = torch.randn(p.shape, generator=gen) * 0.03
                _attach(self, p.key, v, False)
            elif p.kind == "bias":
                _attach(self, p.key, torch.zeros(p.shape), False)
            elif p.kind in ("bn_gamma",):
                _attach(self, p.key, torch.ones(p.shape), False)
            elif p.kind == "bn_beta":
                _attach(self, p.key, torch.zeros(p.shape), False)
            elif p.kind == "bn_mean":
                _attach(self, p.key, torch.zeros(p.shape), True)
            elif p.kind == "bn_var":
                _attach(self, p.key, torch.ones(p.shape), True)
            elif p.kind == "bn_nbt":
                _attach(self, p.key, torch.zeros((), dtype=torch.long), True)
            elif p.kind == "scale20":
                _attach(self, p.key, torch.ones(p.shape) * 20, False)  # ssd_vgg16.py:40
            else:
                raise ValueError(p.kind)
        self.score_thresh = graph.post["score_thresh"]
        self.nms_thresh = graph.post["nms_thresh"]
        self.detections_per_img = graph.post["detections_per_img"]
        self.topk_candidates = graph.post["topk_candidates"]
        self._handle = None
        self._sig = None
        self._lowered = None
        self._bufs = {}
        self._call = None
        self._packed_set = -1         # address last handed to dn_set_packed_output (-1: unknown)
        self.eval()

    # ------------------------------------------------------------------------------------------------------
    def _weights_signature(self):
        """What the folded fp16 weight blob of the plan was built from: every tensor's identity, storage address and in-place
        version counter (optimizer steps, load_state_dict, .to(), load_state_dict(assign=True) and swapped Parameters all change
        one of them). Writes that bypass the version counter (`p.data.copy_()`, `p.data.mul_()`) do not: call invalidate()."""
        dev = None
        sig = []
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((id(t), t.data_ptr(), t._version))
            dev = t.device
        return (hash(tuple(sig)), str(dev), self.score_thresh, self.nms_thresh, self.detections_per_img, self.topk_candidates)

    def _weights_unchanged(self, device) -> bool:
        """The per-call form of the test above: True when the plan in hand is still what the current weights lower to. Same criteria
        (tensor identities through the structure epoch, storage addresses, in-place version counters, the post-process settings), walked
        over a cached flat list."""
        c = getattr(self, "_fast_sig", None)
        if c is None or c[0] != _STRUCT_EPOCH[0]:
            return False
        _, dev, hyper, tensors, ptrs, vers = c
        if dev != device or hyper != (self.score_thresh, self.nms_thresh, self.detections_per_img, self.topk_candidates):
            return False
        return [t._version for t in tensors] == vers and [t.data_ptr() for t in tensors] == ptrs

    def _remember_weights(self, device):
        tensors = list(self.parameters()) + list(self.buffers())
        self._fast_sig = (_STRUCT_EPOCH[0], device, (self.score_thresh, self.nms_thresh, self.detections_per_img, self.topk_candidates),
                          tensors, [t.data_ptr() for t in tensors], [t._version for t in tensors])

    def invalidate(self):
        """Drop the device plan; the next forward lowers the current parameters again. Needed only after weight edits that bypass
        autograd's version counter (`.data` writes)."""
        self.release()
        self._sig = None
        self._fast_sig = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def _plan(self, device):
        if device.type != "cuda":
            raise RuntimeError("demonet_amd runs on an MI355X only (images must be on a cuda device); "
                               "there is no CPU fallback path")
        if self._handle is not None and self._weights_unchanged(device):
            return self._handle
        sig = (self._weights_signature(), str(device))
        if self._handle is not None and sig == self._sig:
            self._remember_weights(device)
            return self._handle
        self.release()
        self.graph.post.update(score_thresh=self.score_thresh, nms_thresh=self.nms_thresh,
                               detections_per_img=self.detections_per_img, topk_candidates=self.topk_candidates)
        sd = {k: v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu() for k, v in self.state_dict().items()}
        with torch.cuda.device(device):
            self._lowered = LoweredModel(self.graph, sd)
            self._handle = self._lowered.create()
        self._sig = sig
        self._remember_weights(device)
        self._bufs = {}
        self._plan_gen = getattr(self, "_plan_gen", 0) + 1      # (a new plan may reuse the old one's address: pipelines compare this)
        self._pipe_refs = 0
        return self._handle

    def release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().dn_destroy(C.c_void_p(self._handle))
            self._handle = None
            self._bufs = {}
            self._call = None
            self._packed_set = -1

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def set_graph_mode(self, enabled: bool):
        self._graph_mode = bool(enabled)
        if self._handle is not None:
            _lib.check(_lib.lib().dn_set_graph_mode(C.c_void_p(self._handle), int(enabled)))

    def _buffers_for(self, n, h, w, device):
        key = (n, h, w, str(device))
        b = self._bufs.get(key)
        if b is None:
            L = _lib.lib()
            D = self.detections_per_img
            ws = L.dn_workspace_bytes(C.c_void_p(self._handle), n)
            # the four outputs are typed views of ONE allocation, so that the list API takes its private copy of a call's results
            # with one device copy (forward())
            out = torch.empty(self._out_bytes(n), dtype=torch.uint8, device=device)
            boxes, scores, labels, counts = self._out_views(out, n)
            b = dict(
                images=torch.empty((n, 3, h, w), dtype=torch.float32, device=device),
                ws=torch.empty(ws, dtype=torch.uint8, device=device),
                out=out, boxes=boxes, scores=scores, labels=labels, counts=counts,
            )
            if len(self._bufs) >= 4:
                self._bufs.clear()
            self._call = None
            self._bufs[key] = b
            if hasattr(self, "_graph_mode"):
                _lib.check(L.dn_set_graph_mode(C.c_void_p(self._handle), int(self._graph_mode)))
        return b

    def _out_bytes(self, n):
        D = self.detections_per_img
        return n * D * 16 + n * D * 8 + n * D * 4 + n * 4       # boxes fp32 x 4 | labels int64 | scores fp32 | counts int32 (every offset 8-byte aligned)

    def _out_views(self, out: Tensor, n: int):
        D = self.detections_per_img
        o1, o2, o3 = n * D * 16, n * D * 24, n * D * 28
        return (out[:o1].view(torch.float32).view(n, D, 4), out[o2:o3].view(torch.float32).view(n, D),
                out[o1:o2].view(torch.int64).view(n, D), out[o3:o3 + 4 * n].view(torch.int32))

    # ------------------------------------------------------------------------------------------------------
    def forward_batch(self, images: Tensor, persistent_input: bool = False, packed: Optional[Tensor] = None):
        """images: [N,3,H,W] fp32 on the GPU. Returns padded device tensors (boxes [N,D,4], scores [N,D],
        labels [N,D] int64, counts [N] int32) that stay valid until the next call with the same shape.
        persistent_input=True promises that `images` keeps its address between calls (lets the hipGraph replay).
        packed: optional [N, D+1, 6] fp32 device tensor that additionally receives the gather payload (dist.py)."""
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError("expected a [N,3,H,W] batch, got {}".format(tuple(images.shape)))
        if not images.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {images.dtype} instead")
        handle = self._plan(images.device)
        n, _, h, w = images.shape
        self._check_packed(packed, n, images.device)
        b = self._buffers_for(n, h, w, images.device)
        if persistent_input and images.dtype == torch.float32 and images.is_contiguous():
            src = images
        else:
            b["images"].copy_(images)
            src = b["images"]
        dev = images.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        # one C call per forward: the argument tuple of (plan, buffers, input address, stream, packed output) is built once and reused
        # while none of them changes -- a synchronous caller pays for every microsecond of Python in front of the graph launch
        pk = packed.data_ptr() if packed is not None else 0
        key = (handle, b["ws"].data_ptr(), src.data_ptr(), stream, pk)
        if self._packed_set != pk:                  # (plan-global state of the library; forward_uint8 and ForwardPipeline set it too)
            _lib.check(_lib.lib().dn_set_packed_output(C.c_void_p(handle), C.c_void_p(pk) if pk else None))
            self._packed_set = pk
        call = self._call
        if call is None or call[0] != key:
            L = _lib.lib()
            args = (C.c_void_p(handle), C.c_void_p(src.data_ptr()), n, h, w, C.c_void_p(b["boxes"].data_ptr()), C.c_void_p(b["scores"].data_ptr()),
                    C.c_void_p(b["labels"].data_ptr()), C.c_void_p(b["counts"].data_ptr()), C.c_void_p(b["ws"].data_ptr()), b["ws"].numel(),
                    C.c_void_p(stream))
            call = self._call = (key, L.dn_forward, args, (b["boxes"], b["scores"], b["labels"], b["counts"]))
        if torch.cuda.current_device() == dev.index:
            rc = call[1](*call[2])
        else:
            with torch.cuda.device(dev):
                rc = call[1](*call[2])
        if rc < 0:
            _lib.check(rc, "dn_forward")
        return call[3]

    def _check_packed(self, packed, n, device):
        """The merge kernel writes [n][D+1][6] fp32 rows through this pointer: anything else would be an out-of-bounds device write."""
        if packed is None:
            return
        want = (n, self.detections_per_img + 1, 6)
        if tuple(packed.shape) != want or packed.dtype != torch.float32 or not packed.is_contiguous() or packed.device != device:
            raise ValueError("packed must be a contiguous float32 tensor of shape {} on {}, got {} {} on {}".format(
                want, device, tuple(packed.shape), packed.dtype, packed.device))

    def forward_uint8(self, images: Tensor, packed: Optional[Tensor] = None):
        """images: [N,H,W,3] uint8 on the GPU -- a decoder's output (HWC, RGB). ToTensor (/255), the bilinear resize to the network
        size and the HWC -> planar conversion run on the device ahead of the stem (transform.py:27-53,129-138); results equal
        forward_batch(images.permute(0,3,1,2).float() / 255). `images` must keep its address between calls for the graph replay.
        Returns the padded device tensors of forward_batch."""
        if images.dim() != 4 or images.shape[3] != 3 or images.dtype != torch.uint8:
            raise ValueError("expected a [N,H,W,3] uint8 batch, got {} {}".format(tuple(images.shape), images.dtype))
        if not images.is_contiguous():
            raise ValueError("forward_uint8 needs a contiguous [N,H,W,3] tensor")
        handle = self._plan(images.device)
        n, h, w, _ = images.shape
        self._check_packed(packed, n, images.device)
        b = self._buffers_for(n, h, w, images.device)
        stream = torch.cuda.current_stream(images.device).cuda_stream
        _lib.check(_lib.lib().dn_set_packed_output(C.c_void_p(handle), C.c_void_p(packed.data_ptr()) if packed is not None else None))
        self._packed_set = packed.data_ptr() if packed is not None else 0
        with torch.cuda.device(images.device):
            _lib.check(_lib.lib().dn_forward_u8(C.c_void_p(handle), C.c_void_p(images.data_ptr()), n, h, w,
                                                C.c_void_p(b["boxes"].data_ptr()), C.c_void_p(b["scores"].data_ptr()),
                                                C.c_void_p(b["labels"].data_ptr()), C.c_void_p(b["counts"].data_ptr()),
                                                C.c_void_p(b["ws"].data_ptr()), b["ws"].numel(), C.c_void_p(stream)), "dn_forward_u8")
        return b["boxes"], b["scores"], b["labels"], b["counts"]

    def batch_split(self, n: int) -> int:
        """Number of parallel sub-batch launch chains a forward of n images is issued as (1 = a single chain)."""
        if self._handle is None:
            raise RuntimeError("batch_split: the plan is built on the first forward / .cuda()")
        return _lib.check(_lib.lib().dn_batch_split(C.c_void_p(self._handle), int(n)))

    def forward_heads(self, images: Tensor):
        """Backbone + heads only: returns (cls_logits [N,A,K], bbox_regression [N,A,4]) fp32 device tensors (copies)."""
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError("expected a [N,3,H,W] batch, got {}".format(tuple(images.shape)))
        if not images.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {images.dtype} instead")
        handle = self._plan(images.device)
        n, _, h, w = images.shape
        b = self._buffers_for(n, h, w, images.device)
        b["images"].copy_(images)
        L = _lib.lib()
        stream = torch.cuda.current_stream(images.device).cuda_stream
        with torch.cuda.device(images.device):
            _lib.check(L.dn_forward_heads(C.c_void_p(handle), C.c_void_p(b["images"].data_ptr()), n, h, w,
                                          C.c_void_p(b["ws"].data_ptr()), b["ws"].numel(), C.c_void_p(stream)), "dn_forward_heads")
        pl, pr = C.c_void_p(), C.c_void_p()
        _lib.check(L.dn_head_outputs(C.c_void_p(handle), C.c_void_p(b["ws"].data_ptr()), n, C.byref(pl), C.byref(pr)))
        A, K = self.graph.num_anchors(), self.graph.num_classes
        base = b["ws"].data_ptr()
        lo = (pl.value - base)
        ro = (pr.value - base)
        logits = b["ws"][lo:lo + n * A * K * 4].view(torch.float32).view(n, A, K).clone()
        reg = b["ws"][ro:ro + n * A * 16].view(torch.float32).view(n, A, 4).clone()
        return logits, reg

    def tensor(self, images_shape, tensor_id: int) -> Tensor:
        """NHWC fp16 view (copy) of an intermediate activation of the LAST forward with this batch shape."""
        n, _, h, w = images_shape
        key = next(k for k in self._bufs if k[:3] == (n, h, w))
        b = self._bufs[key]
        p, sz = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().dn_tensor_ptr(C.c_void_p(self._handle), C.c_void_p(b["ws"].data_ptr()), n, tensor_id,
                                            C.byref(p), C.byref(sz)))
        t = self.graph.t(tensor_id)
        off = p.value - b["ws"].data_ptr()
        return b["ws"][off:off + sz.value].view(torch.float16).view(n, t.h, t.w, t.c).clone()

    def compute_loss(self, targets: List[Dict[str, Tensor]], head_outputs: Dict[str, Tensor], anchors=None, matched_idxs=None,
                     iou_thresh: float = 0.5, positive_fraction: float = 0.25) -> Dict[str, Tensor]:
        """The VALUE of the training loss (generalized_ssd.py:210-269 on the matching of :316-330), computed on the GPU by
        dn_ssd_loss (demonet_amd/loss.py); no gradients -- training itself stays out of scope. `anchors` defaults to the model's
        own default boxes; `matched_idxs`, which the reference's forward computes and passes in, is recomputed here and, when
        given, checked against."""
        from .loss import ssd_loss
        dev = head_outputs["cls_logits"].device
        if anchors is None:
            self._plan(dev)
            anchors = torch.from_numpy(self._lowered.anchors).to(dev)
        losses, matched = ssd_loss(head_outputs, anchors, targets, iou_thresh, (1.0 - positive_fraction) / positive_fraction)
        if matched_idxs is not None:
            given = torch.stack(list(matched_idxs)).to(matched.device)
            if not torch.equal(given, matched):
                raise ValueError("compute_loss: matched_idxs differ from the SSDMatcher result for these targets and anchors")
        return losses

    def loss(self, images: Tensor, targets: List[Dict[str, Tensor]]) -> Dict[str, Tensor]:
        """Backbone + heads (forward_heads) -> compute_loss: the loss values a reference model in train mode would return for this
        batch with the same (eval-mode, BN folded) weights. images: [N,3,H,W] AT THE NETWORK SIZE: the reference's transform also resizes
        the target boxes with the images (transform.py:93-108), which this entry point does not do -- other sizes raise."""
        W, H = self.graph.size
        if tuple(images.shape[-2:]) != (H, W):
            raise ValueError("SSD.loss: images of {}x{} but the network size is {}x{}; resize the images AND the target boxes first "
                             "(the reference's transform does both, transform.py:93-108)".format(images.shape[-2], images.shape[-1], H, W))
        logits, reg = self.forward_heads(images)
        return self.compute_loss(targets, {"cls_logits": logits, "bbox_regression": reg})

    def forward(self, images, targets: Optional[List[Dict[str, Tensor]]] = None):
        if self.training:
            if targets is None:
                raise ValueError("In training mode, targets should be passed")     # generalized_ssd.py:273-274
            raise NotImplementedError("demonet_amd implements the inference path only (SURVEY.md section 8); the loss VALUE of a batch "
                                      "is available through SSD.loss(images, targets) / compute_loss (no gradients)")
        legacy = isinstance(images, Tensor) and images.dim() == 4                   # hub call form model(x[1,3,S,S], shapes)
        if legacy:
            images = list(images.unbind(0))
        groups: Dict = {}                                                           # shape -> indices, in order of first appearance
        for i, img in enumerate(images):
            if img.dim() != 3:
                raise ValueError("images is expected to be a list of 3d tensors "
                                 "of shape [C, H, W], got {}".format(img.shape))    # transform.py:110-112
            if not img.is_floating_point():
                raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), "
                                f"but found type {img.dtype} instead")              # transform.py:130-134
            g = groups.get(img.shape)
            if g is None:
                groups[img.shape] = [i]
            else:
                g.append(i)
        out: List[Optional[Dict[str, Tensor]]] = [None] * len(images)
        f32 = torch.float32
        for shape, idxs in groups.items():
            device = images[idxs[0]].device
            self._plan(device)
            b = self._buffers_for(len(idxs), shape[1], shape[2], device)
            torch.stack([images[i] if images[i].dtype is f32 else images[i].to(f32) for i in idxs], out=b["images"])
            self.forward_batch(b["images"], persistent_input=True)
            # The padded outputs live in buffers that the next call overwrites: ONE private copy per call (a single device copy of the
            # allocation that holds all four), and the per-image results are views into it. They therefore SHARE storage: keeping one
            # image's result alive keeps the whole batch copy (n x D x 28 bytes) alive, and an in-place edit of a full-length result
            # edits that copy -- the reference returns independent tensors per image; clone() a result to detach it.
            boxes, scores, labels, counts = self._out_views(b["out"].clone(), len(idxs))
            # The full-length views are made BEFORE the host waits for the counts: 192 tensor objects cost ~0.13 ms of host time, which
            # this way runs under the forward on the device instead of behind it. Only images with fewer than D detections need a slice
            # afterwards (2 us each).
            D = boxes.shape[1]
            ub, us, ul = boxes.unbind(0), scores.unbind(0), labels.unbind(0)
            cnt = counts.tolist()                                                   # the one device->host sync
            for j, i in enumerate(idxs):
                c = cnt[j]
                d = {"boxes": ub[j], "scores": us[j], "labels": ul[j]} if c == D else {"boxes": ub[j][:c], "scores": us[j][:c], "labels": ul[j][:c]}
                if legacy:
                    d = OrderedDict((k, d[k]) for k in ("scores", "labels", "boxes"))   # box_head.py:379 order
                out[i] = d
        return out
