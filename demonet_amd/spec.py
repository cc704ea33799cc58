"""Model graphs for the SSD inference hot path, expressed as a small op IR.

Each builder below mirrors one reference factory (file:line cited per function) but emits
*ops over NHWC tensors* instead of nn.Modules: the IR is what `plan.py` lowers into the
C-ABI op list executed by the HIP library, and it also carries the reference's state_dict
key names/shapes so checkpoints keyed like the reference load unchanged.

IR ops (all inference-only, BN is folded at plan time):
  stem   dense kxk conv reading the NCHW fp32 image, normalisation applied on load
  pw     1x1 conv == GEMM  [N*H*W, Cin] x [Cin, Cout]  (+bias/BN, act, SE scale on input, residual)
  dw     depthwise kxk conv (+BN, act, optional per-tile pooled sums for SE)
  se     squeeze-excite FCs on pooled sums -> per-(image, channel) scale
  conv   dense kxk conv (VGG path; implicit GEMM)
  maxpool, l2norm
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple
import math

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_HSWISH = 0, 1, 2, 3
ACTS = {"none": ACT_NONE, "relu": ACT_RELU, "relu6": ACT_RELU6, "hswish": ACT_HSWISH}
# second moments used ONLY to scale synthetic weights (E[act(z)^2], z~N(0,1); BN with gamma,var ~ U[0.5,1.5])
_ACT_M2 = {"none": 1.0, "relu": 0.5, "relu6": 0.5, "hswish": 0.3315}
_BN_FACTOR = 1.19
_SE_M2 = 0.50


def make_divisible(v: float, divisor: int = 8, min_value: Optional[int] = None) -> int:
    """reference: demonet/models/mobilenetv2.py:16-29"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


@dataclass
class Param:
    key: str
    shape: Tuple[int, ...]
    kind: str            # conv_w | bn_gamma | bn_beta | bn_mean | bn_var | bn_nbt | bias | scale20
    std: float = 1.0     # for conv_w / bias: synthetic std
    target: float = 0.0  # conv_w: intended output variance (pre-activation; after BN normalisation if bn)
    bn: str = ""         # conv_w: key of the BN that follows (calibration of synthetic weights only)


@dataclass
class Tensor:
    tid: int
    c: int
    h: int
    w: int
    kind: str = "act"    # act (NHWC fp16) | image (NCHW fp32) | vec (per-image fp32 vector [c]) | pool (partials)
    tiles: int = 0       # for kind == pool: number of partial-sum tiles per image
    m2: float = 1.0      # estimated second moment of the values (only used to scale SYNTHETIC weights)


@dataclass
class Node:
    op: str
    inp: int
    out: int
    cin: int = 0
    cout: int = 0
    k: int = 1
    stride: int = 1
    pad: int = 0
    dil: int = 1
    act: int = ACT_NONE
    conv_key: str = ""
    bn_key: Optional[str] = None
    bn_eps: float = 1e-5
    has_bias: bool = False
    residual: int = -1
    se: int = -1          # pw: tensor id of SE scale vec
    pool: int = -1        # dw: tensor id of pooled partials
    head: int = 0         # 0 none / 1 cls / 2 reg
    level: int = -1
    # se node
    fc1_key: str = ""
    fc2_key: str = ""
    squeeze: int = 0
    ceil_mode: bool = False
    scale_key: str = ""


@dataclass
class Graph:
    name: str
    size: Tuple[int, int]                 # (W, H) as the reference passes it (ssd_mobilenetv3.py:201)
    image_mean: List[float]
    image_std: List[float]
    num_classes: int
    tensors: List[Tensor] = field(default_factory=list)
    nodes: List[Node] = field(default_factory=list)
    params: List[Param] = field(default_factory=list)
    features: List[int] = field(default_factory=list)     # tensor ids of the pyramid levels
    anchors_per_loc: List[int] = field(default_factory=list)
    anchor_spec: Dict = field(default_factory=dict)
    post: Dict = field(default_factory=dict)              # score_thresh, nms_thresh, detections_per_img, topk_candidates

    # -- tensor / param helpers -------------------------------------------------------------
    def new_tensor(self, c, h, w, kind="act", tiles=0, m2=1.0) -> int:
        t = Tensor(len(self.tensors), c, h, w, kind, tiles, m2)
        self.tensors.append(t)
        return t.tid

    def _wstd(self, x, fan_in_taps, target, bn, se=-1):
        """Synthetic conv-weight std so the conv output has variance ~`target` after BN (if any)."""
        m2 = self.t(x).m2 * (_SE_M2 if se >= 0 else 1.0)
        return math.sqrt(target / ((_BN_FACTOR if bn else 1.0) * fan_in_taps * m2))

    def _out_m2(self, target, act, residual=-1):
        m2 = target * _ACT_M2[act]
        if residual >= 0:
            m2 += self.t(residual).m2
        return m2

    def t(self, tid) -> Tensor:
        return self.tensors[tid]

    def _conv_params(self, conv_key, cout, cin_g, k, std, bias, bias_std=0.1, target=0.0, bn=""):
        self.params.append(Param(conv_key + ".weight", (cout, cin_g, k, k), "conv_w", std, target, bn or ""))
        if bias:
            self.params.append(Param(conv_key + ".bias", (cout,), "bias", bias_std))

    def _bn_params(self, bn_key, c):
        self.params += [
            Param(bn_key + ".weight", (c,), "bn_gamma"),
            Param(bn_key + ".bias", (c,), "bn_beta"),
            Param(bn_key + ".running_mean", (c,), "bn_mean"),
            Param(bn_key + ".running_var", (c,), "bn_var"),
            Param(bn_key + ".num_batches_tracked", (), "bn_nbt"),
        ]

    # -- op emitters ----------------------------------------------------------------------------
    def input(self) -> int:
        W, H = self.size
        m2 = sum((1.0 / 12.0 + (0.5 - m) ** 2) / (sd * sd) for m, sd in zip(self.image_mean, self.image_std)) / 3.0
        return self.new_tensor(3, H, W, "image", m2=m2)

    @staticmethod
    def _out_hw(h, w, k, s, p, d=1, ceil=False):
        def o(x):
            num = x + 2 * p - d * (k - 1) - 1
            if ceil:
                r = -(-num // s) + 1
                if (r - 1) * s >= x + p:      # torch ceil_mode rule: last window must start inside input/left pad
                    r -= 1
                return r
            return num // s + 1
        return o(h), o(w)

    def stem(self, x, conv_key, bn_key, cout, k, stride, act, eps, target=1.0):
        ti = self.t(x)
        pad = (k - 1) // 2
        ho, wo = self._out_hw(ti.h, ti.w, k, stride, pad)
        out = self.new_tensor(cout, ho, wo, m2=self._out_m2(target, act))
        self._conv_params(conv_key, cout, 3, k, self._wstd(x, 3 * k * k, target, True), False, target=target, bn=bn_key)
        self._bn_params(bn_key, cout)
        self.nodes.append(Node("stem", x, out, 3, cout, k, stride, pad, 1, ACTS[act], conv_key, bn_key, eps))
        return out

    def pw(self, x, conv_key, bn_key, cout, act, eps=1e-5, bias=False, se=-1, residual=-1,
           head=0, level=-1, target=1.0, bias_std=0.1):
        ti = self.t(x)
        out = self.new_tensor(cout, ti.h, ti.w, m2=self._out_m2(target, act, residual))
        self._conv_params(conv_key, cout, ti.c, 1, self._wstd(x, ti.c, target, bool(bn_key), se), bias, bias_std,
                          target=target, bn=bn_key)
        if bn_key:
            self._bn_params(bn_key, cout)
        self.nodes.append(Node("pw", x, out, ti.c, cout, 1, 1, 0, 1, ACTS[act], conv_key, bn_key, eps,
                               bias, residual, se, -1, head, level))
        return out

    def dw(self, x, conv_key, bn_key, k, stride, act, eps=1e-5, bias=False, dil=1, target=1.0):
        ti = self.t(x)
        pad = (k - 1) // 2 * dil
        ho, wo = self._out_hw(ti.h, ti.w, k, stride, pad, dil)
        out = self.new_tensor(ti.c, ho, wo, m2=self._out_m2(target, act))
        self._conv_params(conv_key, ti.c, 1, k, self._wstd(x, k * k, target, bool(bn_key)), bias, target=target, bn=bn_key)
        if bn_key:
            self._bn_params(bn_key, ti.c)
        self.nodes.append(Node("dw", x, out, ti.c, ti.c, k, stride, pad, dil, ACTS[act], conv_key, bn_key, eps, bias))
        return out

    def se(self, x, key, squeeze):
        """SqueezeExcitation (mobilenetv3.py:22-40): avgpool -> fc1(+b) -> ReLU -> fc2(+b) -> Hardsigmoid.
        Emits the FC node; the pooled sums come from the producing dw node (fused), the scale is
        consumed by the following pw node (applied to its input rows)."""
        ti = self.t(x)
        prod = self.nodes[-1]
        assert prod.op == "dw" and prod.out == x
        pool = self.new_tensor(ti.c, 1, 1, "pool")
        prod.pool = pool
        vec = self.new_tensor(ti.c, 1, 1, "vec")
        self._conv_params(key + ".fc1", squeeze, ti.c, 1, 1.0 / math.sqrt(ti.c), True)
        self._conv_params(key + ".fc2", ti.c, squeeze, 1, 1.5 / math.sqrt(squeeze), True, bias_std=1.0)
        n = Node("se", pool, vec, ti.c, ti.c)
        n.fc1_key, n.fc2_key, n.squeeze = key + ".fc1", key + ".fc2", squeeze
        n.stride = ti.h * ti.w      # pooled pixel count (mean divisor)
        self.nodes.append(n)
        return vec

    def conv(self, x, conv_key, cout, k, stride, pad, dil, act, head=0, level=-1, target=1.0, bias_std=0.1):
        ti = self.t(x)
        ho, wo = self._out_hw(ti.h, ti.w, k, stride, pad, dil)
        out = self.new_tensor(cout, ho, wo, m2=self._out_m2(target, act))
        self._conv_params(conv_key, cout, ti.c, k, self._wstd(x, ti.c * k * k, target, False), True, bias_std, target=target)
        op = "stem" if ti.kind == "image" else "conv"
        self.nodes.append(Node(op, x, out, ti.c, cout, k, stride, pad, dil, ACTS[act], conv_key, None, 0.0, True,
                               -1, -1, -1, head, level))
        return out

    def maxpool(self, x, k, stride, pad, ceil_mode=False):
        ti = self.t(x)
        ho, wo = self._out_hw(ti.h, ti.w, k, stride, pad, 1, ceil_mode)
        out = self.new_tensor(ti.c, ho, wo, m2=ti.m2)
        n = Node("maxpool", x, out, ti.c, ti.c, k, stride, pad)
        n.ceil_mode = ceil_mode
        self.nodes.append(n)
        return out

    def l2norm(self, x, scale_key):
        ti = self.t(x)
        out = self.new_tensor(ti.c, ti.h, ti.w, m2=400.0 / ti.c)
        self.params.append(Param(scale_key, (ti.c,), "scale20"))
        n = Node("l2norm", x, out, ti.c, ti.c)
        n.scale_key = scale_key
        self.nodes.append(n)
        return out

    # -- derived ---------------------------------------------------------------------------------
    def num_anchors(self) -> int:
        return sum(a * self.t(f).h * self.t(f).w for a, f in zip(self.anchors_per_loc, self.features))


# =================================================================================================
# MobileNetV3-large (reduced tail) + SSDLite   reference: ssd_mobilenetv3.py:159-227
# =================================================================================================

# mobilenetv3.py:198-214 with reduce_divider=2 (ssd_mobilenetv3.py:193: reduce_tail = not pretrained_backbone)
#   (in, kernel, expanded, out, use_se, activation, stride)
def _v3_large_table(reduce_divider=2):
    r = reduce_divider
    return [
        (16, 3, 16, 16, False, "relu", 1),
        (16, 3, 64, 24, False, "relu", 2),
        (24, 3, 72, 24, False, "relu", 1),
        (24, 5, 72, 40, True, "relu", 2),
        (40, 5, 120, 40, True, "relu", 1),
        (40, 5, 120, 40, True, "relu", 1),
        (40, 3, 240, 80, False, "hswish", 2),
        (80, 3, 200, 80, False, "hswish", 1),
        (80, 3, 184, 80, False, "hswish", 1),
        (80, 3, 184, 80, False, "hswish", 1),
        (80, 3, 480, 112, True, "hswish", 1),
        (112, 3, 672, 112, True, "hswish", 1),
        (112, 5, 672, 160 // r, True, "hswish", 2),
        (160 // r, 5, 960 // r, 160 // r, True, "hswish", 1),
        (160 // r, 5, 960 // r, 160 // r, True, "hswish", 1),
    ]


_PROJ_VAR = 0.5             # synthetic target variance of linear projection outputs (tames residual growth)


def _v3_block(g: Graph, x, prefix, cfg, eps):
    """InvertedResidual (mobilenetv3.py:61-99). `prefix` is the key of the nn.Sequential `block`."""
    cin, k, exp, cout, use_se, act, stride = cfg
    j = 0
    y = x
    if exp != cin:
        y = g.pw(y, f"{prefix}.{j}.0", f"{prefix}.{j}.1", exp, act, eps)
        j += 1
    y = g.dw(y, f"{prefix}.{j}.0", f"{prefix}.{j}.1", k, stride, act, eps)
    j += 1
    se = -1
    if use_se:
        se = g.se(y, f"{prefix}.{j}", make_divisible(exp // 4, 8))
        j += 1
    res = x if (stride == 1 and cin == cout) else -1
    y = g.pw(y, f"{prefix}.{j}.0", f"{prefix}.{j}.1", cout, "none", eps, se=se, residual=res, target=_PROJ_VAR)
    return y


def _ssdlite_head(g: Graph, feats, num_anchors, num_classes, eps, logit_std, reg_std):
    """SSDLiteHead (ssd_mobilenetv3.py:65-95): per level dw3x3+BN+ReLU6 -> 1x1 conv with bias."""
    for kind, name, cols, std in ((2, "regression_head", 4, reg_std), (1, "classification_head", num_classes, logit_std)):
        for lvl, (f, a) in enumerate(zip(feats, num_anchors)):
            p = f"head.{name}.module_list.{lvl}"
            y = g.dw(f, f"{p}.0.0", f"{p}.0.1", 3, 1, "relu6", eps)
            g.pw(y, f"{p}.1", None, a * cols, "none", bias=True, head=kind, level=lvl,
                 target=std * std, bias_std=0.3 * std)


def ssdlite320_mobilenet_v3_large_graph(num_classes=91, logit_std=1.5, reg_std=1.0, eps=1e-3, **post) -> Graph:
    # eps: BatchNorm eps of the factory's norm_layer (default 1e-3: ssd_mobilenetv3.py:196)
    g = Graph("ssdlite320_mobilenet_v3_large", (320, 320), [0.5] * 3, [0.5] * 3, num_classes)
    tab = _v3_large_table(2)
    x = g.input()
    f0 = "backbone.features.0"
    x = g.stem(x, f"{f0}.0.0", f"{f0}.0.1", 16, 3, 2, "hswish", eps)   # mobilenetv3.py:141
    for i, cfg in enumerate(tab[:12], start=1):
        x = _v3_block(g, x, f"{f0}.{i}.block", cfg, eps)
    # C4 block split at its expansion layer (ssd_mobilenetv3.py:104-108)
    cin, k, exp, cout, use_se, act, stride = tab[12]
    x = g.pw(x, f"{f0}.13.0", f"{f0}.13.1", exp, act, eps)
    feat0 = x
    f1 = "backbone.features.1"
    # features.1.0 = backbone[13].block[1:] keeps child names "1","2","3"
    x = g.dw(x, f"{f1}.0.1.0", f"{f1}.0.1.1", k, stride, act, eps)
    se = g.se(x, f"{f1}.0.2", make_divisible(exp // 4, 8))
    x = g.pw(x, f"{f1}.0.3.0", f"{f1}.0.3.1", cout, "none", eps, se=se, target=_PROJ_VAR)
    for i, cfg in enumerate(tab[13:], start=1):
        x = _v3_block(g, x, f"{f1}.{i}.block", cfg, eps)
    last_c = 6 * tab[-1][3]                                   # mobilenetv3.py:149-151
    x = g.pw(x, f"{f1}.3.0", f"{f1}.3.1", last_c, "hswish", eps)
    feats = [feat0, x]
    # extra blocks (ssd_mobilenetv3.py:39-54,110-116)
    for i, oc in enumerate((512, 256, 256, 128)):
        p = f"backbone.extra.{i}"
        mid = oc // 2
        x = g.pw(x, f"{p}.0.0", f"{p}.0.1", mid, "relu6", eps)
        x = g.dw(x, f"{p}.1.0", f"{p}.1.1", 3, 2, "relu6", eps)
        x = g.pw(x, f"{p}.2.0", f"{p}.2.1", oc, "relu6", eps)
        feats.append(x)
    g.features = feats
    g.anchors_per_loc = [6] * 6
    g.anchor_spec = dict(aspect_ratios=[[2, 3]] * 6, min_ratio=0.2, max_ratio=0.95, scales=None, steps=None, clip=True)
    _ssdlite_head(g, feats, g.anchors_per_loc, num_classes, eps, logit_std, reg_std)
    defaults = dict(score_thresh=0.001, nms_thresh=0.55, detections_per_img=300, topk_candidates=300)  # :207-216
    g.post = {**defaults, **post}
    return g


# =================================================================================================
# MobileNetV2 + extra blocks + MultiBoxLiteHead  (legacy hub model `ssd_lite_mobilenet_v2`)
# reference: backbone.py:45-119, box_head.py:24-56, mobilenetv2.py:103-168, hubconf.py:9-44,
#            hyper-parameters test/test_model.py:26-48
# =================================================================================================

def ssd_lite_mobilenet_v2_graph(image_size=320, num_classes=21, logit_std=1.5, reg_std=1.0, **post) -> Graph:
    eps = 1e-5                                                # nn.BatchNorm2d default (backbone.py:94-95)
    g = Graph("ssd_lite_mobilenet_v2", (image_size, image_size), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225],
              num_classes)
    x = g.input()
    fb = "backbone.body"
    x = g.stem(x, f"{fb}.0.0", f"{fb}.0.1", 32, 3, 2, "relu6", eps)
    setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
    idx = 1
    cin = 32
    feats = []

    def v2_block(x, prefix, inp, oup, stride, hidden):
        j = 0
        y = x
        if hidden != inp:
            y = g.pw(y, f"{prefix}.conv.{j}.0", f"{prefix}.conv.{j}.1", hidden, "relu6", eps)
            j += 1
        y = g.dw(y, f"{prefix}.conv.{j}.0", f"{prefix}.conv.{j}.1", 3, stride, "relu6", eps)
        j += 1
        res = x if (stride == 1 and inp == oup) else -1
        return g.pw(y, f"{prefix}.conv.{j}", f"{prefix}.conv.{j + 1}", oup, "none", eps, residual=res, target=_PROJ_VAR)

    for t, c, n, s in setting:
        for i in range(n):
            x = v2_block(x, f"{fb}.{idx}", cin, c, s if i == 0 else 1, int(round(cin * t)))
            cin = c
            if idx == 13:
                feats.append(x)                               # backbone.py:52 tap "13"
            idx += 1
    x = g.pw(x, f"{fb}.18.0", f"{fb}.18.1", 1280, "relu6", eps)
    feats.append(x)                                           # tap "18"
    cin = 1280
    for i, (oc, ratio) in enumerate(zip([512, 256, 256, 64], [0.2, 0.25, 0.5, 0.25])):   # backbone.py:54-58
        x = v2_block(x, f"backbone.extra_blocks.{i}", cin, oc, 2, int(round(cin * ratio)))
        feats.append(x)
        cin = oc
    g.features = feats
    g.anchors_per_loc = [6] * 6
    # The legacy AnchorGenerator(min_sizes, max_sizes) (test_model.py:29-32) no longer exists in the reference;
    # DefaultBoxGenerator with the same 6-per-location layout is used instead (documented in DESIGN.md).
    g.anchor_spec = dict(aspect_ratios=[[2, 3]] * 6, min_ratio=0.2, max_ratio=0.95, scales=None, steps=None, clip=True)
    # MultiBoxLiteHead (box_head.py:45-56): levels 0..4 SeperableConv2d (dw WITH bias + BN + ReLU6 + 1x1 with bias),
    # last level plain 1x1.
    for kind, name, cols, std in ((2, "bbox_pred", 4, reg_std), (1, "cls_logits", num_classes, logit_std)):
        for lvl, f in enumerate(feats):
            p = f"head.{name}.{lvl}"
            if lvl < len(feats) - 1:
                y = g.dw(f, f"{p}.0", f"{p}.1", 3, 1, "relu6", eps, bias=True)
                g.pw(y, f"{p}.3", None, 6 * cols, "none", bias=True, head=kind, level=lvl, target=std * std,
                     bias_std=0.3 * std)
            else:
                g.pw(f, p, None, 6 * cols, "none", bias=True, head=kind, level=lvl, target=std * std, bias_std=0.3 * std)
    defaults = dict(score_thresh=0.5, nms_thresh=0.45, detections_per_img=100, topk_candidates=400)
    g.post = {**defaults, **post}
    return g


# =================================================================================================
# VGG16 SSD300 / SSD512(highres)   reference: ssd_vgg16.py:30-213, generalized_ssd.py:25-92
# =================================================================================================

def _ssd_vgg_graph(name, size, highres, num_classes, anchor_spec, num_anchors, logit_std, reg_std, **post) -> Graph:
    g = Graph(name, (size, size), [0.48235, 0.45882, 0.40784], [1.0 / 255.0] * 3, num_classes)
    x = g.input()
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]
    idx = 0
    pools = 0
    fkey = "backbone.features"
    feats = []
    for v in cfg:
        if v == 'M':
            pools += 1
            if pools == 4:
                # conv4_3 reached: L2 norm output is feature 0; un-normalised x continues (ssd_vgg16.py:98-107)
                feats.append(g.l2norm(x, "backbone.scale_weight"))
                fkey = "backbone.extra.0"
                idx = 0
            x = g.maxpool(x, 2, 2, 0, ceil_mode=(pools == 3))     # ssd_vgg16.py:37
            idx += 1
        else:
            x = g.conv(x, f"{fkey}.{idx}", v, 3, 1, 1, 1, "relu", target=2.0)
            idx += 2
    # fc block appended as last child of extra[0] (ssd_vgg16.py:84-96): index = 7 (maxpool4,3 convs+relus => 0..6)
    fc = f"backbone.extra.0.{idx}"
    x = g.maxpool(x, 3, 1, 1)
    x = g.conv(x, f"{fc}.1", 1024, 3, 1, 6, 6, "relu", target=2.0)
    x = g.conv(x, f"{fc}.3", 1024, 1, 1, 0, 1, "relu", target=2.0)
    feats.append(x)
    extras = [(256, 512, 3, 2, 1), (128, 256, 3, 2, 1), (128, 256, 3, 1, 0), (128, 256, 3, 1, 0)]
    if highres:
        extras.append((128, 256, 4, 1, 0))                        # ssd_vgg16.py:74-81
    for i, (mid, oc, k, s, p) in enumerate(extras, start=1):
        x = g.conv(x, f"backbone.extra.{i}.0", mid, 1, 1, 0, 1, "relu", target=2.0)
        x = g.conv(x, f"backbone.extra.{i}.2", oc, k, s, p, 1, "relu", target=2.0)
        feats.append(x)
    g.features = feats
    g.anchors_per_loc = num_anchors
    g.anchor_spec = anchor_spec
    for kind, hname, cols, std in ((2, "regression_head", 4, reg_std), (1, "classification_head", num_classes, logit_std)):
        for lvl, (f, a) in enumerate(zip(feats, num_anchors)):
            g.conv(f, f"head.{hname}.module_list.{lvl}", a * cols, 3, 1, 1, 1, "none", head=kind, level=lvl,
                   target=std * std, bias_std=0.3 * std)
    defaults = dict(score_thresh=0.01, nms_thresh=0.45, detections_per_img=200, topk_candidates=400)  # generalized_ssd.py:158-162
    g.post = {**defaults, **post}
    return g


def ssd300_vgg16_graph(num_classes=91, logit_std=1.5, reg_std=1.0, **post) -> Graph:
    spec = dict(aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]], min_ratio=None, max_ratio=None,
                scales=[0.07, 0.15, 0.33, 0.51, 0.69, 0.87, 1.05], steps=[8, 16, 32, 64, 100, 300], clip=True)
    return _ssd_vgg_graph("ssd300_vgg16", 300, False, num_classes, spec, [4, 6, 6, 6, 4, 4], logit_std, reg_std, **post)


def ssd512_vgg16_graph(num_classes=91, logit_std=1.5, reg_std=1.0, **post) -> Graph:
    """Build-defined: the reference has the `highres` extractor but no ssd512 factory/anchors."""
    spec = dict(aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]], min_ratio=None, max_ratio=None,
                scales=[0.04, 0.1, 0.26, 0.42, 0.58, 0.74, 0.9, 1.06], steps=[8, 16, 32, 64, 128, 256, 512], clip=True)
    return _ssd_vgg_graph("ssd512_vgg16", 512, True, num_classes, spec, [4, 6, 6, 6, 6, 4, 4], logit_std, reg_std, **post)


GRAPHS = {
    "ssdlite320_mobilenet_v3_large": ssdlite320_mobilenet_v3_large_graph,
    "ssd_lite_mobilenet_v2": ssd_lite_mobilenet_v2_graph,
    "ssd300_vgg16": ssd300_vgg16_graph,
    "ssd512_vgg16": ssd512_vgg16_graph,
}
