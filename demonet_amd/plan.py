"""Lowers a spec.Graph + reference-keyed state_dict into the C-ABI model description and weight blob.

BatchNorm (eval) is folded here in float64:  w' = w * gamma / sqrt(var + eps),  b' = beta - mean * gamma / sqrt(var + eps)
(+ conv bias carried through). Weights are stored fp16 in the layouts the kernels read directly (see
include/demonet_hip.h, dn_op_desc); biases stay fp32, SE FC weights are fp16 (transposed).
"""
import ctypes as C
from typing import Dict

import numpy as np

from . import _lib
from .anchors import default_boxes
from .spec import Graph


def fragment_major(w16: np.ndarray) -> np.ndarray:
    """[cout][cin] fp16 -> MFMA-fragment order [ceil(cout/32)][ceil(cin/16)][2][32][8], zero rows / columns beyond cout / cin: the layout
    of dn_op_desc.w2_off for PW ops (each wave-wide weight load of the streaming kernels is then 1 KB contiguous)."""
    cout, cin = w16.shape
    nt, ks = (cout + 31) // 32, (cin + 15) // 16
    padded = np.zeros((nt * 32, ks * 16), dtype=np.float16)
    padded[:cout, :cin] = w16
    return np.ascontiguousarray(padded.reshape(nt, 32, ks, 2, 8).transpose(0, 2, 3, 1, 4))


class _Blob:
    def __init__(self):
        self.parts = []
        self.size = 0

    def add(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        off = self.size
        b = arr.tobytes()
        pad = (-len(b)) % 256
        self.parts.append(b + b"\0" * pad)
        self.size += len(b) + pad
        return off

    def bytes(self) -> bytes:
        return b"".join(self.parts)


def _np(sd, key):
    v = sd[key]
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


def _fold(sd, node, cout):
    """Returns (scale[cout], bias[cout]) in float64 for conv -> (+bias) -> BN."""
    b = _np(sd, node.conv_key + ".bias") if node.has_bias else np.zeros(cout)
    if node.bn_key:
        g = _np(sd, node.bn_key + ".weight")
        beta = _np(sd, node.bn_key + ".bias")
        mu = _np(sd, node.bn_key + ".running_mean")
        var = _np(sd, node.bn_key + ".running_var")
        s = g / np.sqrt(var + node.bn_eps)
        return s, (b - mu) * s + beta
    return np.ones(cout), b


class LoweredModel:
    """Owns the ctypes arrays referenced by ModelDesc (they must outlive dn_create)."""

    def __init__(self, graph: Graph, state_dict: Dict):
        g = graph
        self.graph = g
        blob = _Blob()
        self.tensors = (_lib.TensorDesc * len(g.tensors))()
        for i, t in enumerate(g.tensors):
            self.tensors[i] = _lib.TensorDesc(t.c, t.h, t.w, _lib.DN_T[t.kind])
        ops = []
        head_inputs = {nd.inp for nd in g.nodes if nd.head}          # tensors a head op reads: the depthwise outputs of the SSDLite heads
        for nd in g.nodes:
            o = _lib.OpDesc()
            o.type = _lib.DN_OP[nd.op]
            o.inp, o.out, o.residual, o.se, o.pool = nd.inp, nd.out, nd.residual, nd.se, nd.pool
            o.cin, o.cout, o.k, o.stride, o.pad, o.dil, o.act = nd.cin, nd.cout, nd.k, nd.stride, nd.pad, nd.dil, nd.act
            o.head, o.level, o.squeeze, o.ceil_mode, o.pool_pixels = nd.head, nd.level, nd.squeeze, int(nd.ceil_mode), 0
            o.w_off = o.b_off = o.w2_off = o.b2_off = -1
            if nd.op == "stem":
                w = _np(state_dict, nd.conv_key + ".weight")              # [cout, 3, k, k]
                s, b = _fold(state_dict, nd, nd.cout)
                w = w * s[:, None, None, None]
                o.w_off = blob.add(w.transpose(1, 2, 3, 0).reshape(-1, nd.cout).astype(np.float32))
                o.b_off = blob.add(b.astype(np.float32))
            elif nd.op == "pw":
                w = _np(state_dict, nd.conv_key + ".weight").reshape(nd.cout, nd.cin)
                s, b = _fold(state_dict, nd, nd.cout)
                wf = (w * s[:, None]).astype(np.float16)
                o.w_off = blob.add(wf)
                o.b_off = blob.add(b.astype(np.float32))
                if nd.cin % 8 == 0:
                    # second copy in MFMA-fragment order for the kernels that stream weights straight from L2 into A fragments
                    # (tail.hip, the strip kernel, headfuse.hip -- class AND box heads): [cout tile of 32][16-deep K step][lane = (k half, channel)][8 halfs]
                    # -> each wave-wide load is 1 KB contiguous instead of 32 row pieces of 32 B
                    o.w2_off = blob.add(fragment_major(wf))
            elif nd.op == "dw":
                w = _np(state_dict, nd.conv_key + ".weight").reshape(nd.cin, nd.k * nd.k)
                s, b = _fold(state_dict, nd, nd.cin)
                wf = (w * s[:, None]).astype(np.float16)                   # [c][k*k]
                o.w_off = blob.add(np.ascontiguousarray(wf.T))             # [k*k][c]
                o.b_off = blob.add(b.astype(np.float32))
                if nd.out in head_inputs and nd.k == 3 and nd.cin % 8 == 0:
                    # second copy in GROUP-major order for the fused head launch (headfuse.hip): [c / 8][9 taps][8] -- the nine taps of an
                    # 8-channel group are 144 contiguous bytes (three scalar loads / one piece of an LDS-DMA instead of nine strided rows)
                    o.w2_off = blob.add(np.ascontiguousarray(wf.reshape(nd.cin // 8, 8, 9).transpose(0, 2, 1)))
            elif nd.op == "se":
                w1 = _np(state_dict, nd.fc1_key + ".weight").reshape(nd.squeeze, nd.cin)
                w2 = _np(state_dict, nd.fc2_key + ".weight").reshape(nd.cin, nd.squeeze)
                o.w_off = blob.add(np.ascontiguousarray(w1.T).astype(np.float16))   # fc1 transposed: [c][squeeze] fp16
                o.b_off = blob.add(_np(state_dict, nd.fc1_key + ".bias").astype(np.float32))
                o.w2_off = blob.add(np.ascontiguousarray(w2.T).astype(np.float16))  # fc2 transposed: [squeeze][c] fp16
                o.b2_off = blob.add(_np(state_dict, nd.fc2_key + ".bias").astype(np.float32))
                o.pool_pixels = nd.stride
                o.stride = 1
            elif nd.op == "conv":
                w = _np(state_dict, nd.conv_key + ".weight")               # [cout, cin, k, k]
                s, b = _fold(state_dict, nd, nd.cout)
                w = (w * s[:, None, None, None]).transpose(0, 2, 3, 1).reshape(nd.cout, -1)   # [cout][ky][kx][cin]
                o.w_off = blob.add(w.astype(np.float16))
                o.b_off = blob.add(b.astype(np.float32))
            elif nd.op == "l2norm":
                o.w_off = blob.add(_np(state_dict, nd.scale_key).astype(np.float32))
            elif nd.op == "maxpool":
                pass
            else:
                raise ValueError(nd.op)
            ops.append(o)
        # SSDLite heads, fused launch (headfuse.hip): per level ONE array of 1 KB slots, one per 32-channel chunk -- the box head's depthwise
        # weights of the chunk [4 groups][9 taps][8] fp16 (576 B), its bias [32] fp32, the class head's bias [32] fp32, zero pad -- what one
        # LDS-DMA instruction of the kernel copies per chunk. Stored as b2_off of the class head's depthwise op.
        by_out = {nd.out: i for i, nd in enumerate(g.nodes)}
        heads = {}
        for nd in g.nodes:
            if nd.op == "pw" and nd.head and nd.inp in by_out and g.nodes[by_out[nd.inp]].op == "dw":
                heads.setdefault(nd.level, {})[nd.head] = by_out[nd.inp]
        for lvl, hd in heads.items():
            if 1 not in hd or 2 not in hd:
                continue
            ic, ir = hd[1], hd[2]
            nc_, nr_ = g.nodes[ic], g.nodes[ir]
            if nc_.inp != nr_.inp or nc_.cin != nr_.cin or nc_.k != 3 or nr_.k != 3 or nc_.cin % 32 != 0 or ops[ir].w2_off < 0:
                continue
            C_ = nc_.cin
            wr = np.frombuffer(blob.bytes()[ops[ir].w2_off:ops[ir].w2_off + C_ * 18], dtype=np.float16).reshape(C_ // 32, 288)
            br = np.frombuffer(blob.bytes()[ops[ir].b_off:ops[ir].b_off + C_ * 4], dtype=np.float32).reshape(C_ // 32, 32)
            bc = np.frombuffer(blob.bytes()[ops[ic].b_off:ops[ic].b_off + C_ * 4], dtype=np.float32).reshape(C_ // 32, 32)
            slot = np.zeros((C_ // 32, 1024), dtype=np.uint8)
            slot[:, 0:576] = wr.view(np.uint8).reshape(C_ // 32, 576)
            slot[:, 576:704] = br.view(np.uint8).reshape(C_ // 32, 128)
            slot[:, 704:832] = bc.view(np.uint8).reshape(C_ // 32, 128)
            ops[ic].b2_off = blob.add(slot)
        self.ops = (_lib.OpDesc * len(ops))(*ops)
        self.blob = blob.bytes()
        grid = [(g.t(f).h, g.t(f).w) for f in g.features]
        W, H = g.size
        self.anchors = default_boxes(grid, (H, W), **g.anchor_spec)
        assert self.anchors.shape[0] == g.num_anchors()
        d = _lib.ModelDesc()
        d.abi_version = _lib.DN_ABI_VERSION
        d.n_tensors, d.n_ops = len(g.tensors), len(ops)
        d.tensors, d.ops = self.tensors, self.ops
        d.input_tensor = g.nodes[0].inp
        d.image_h, d.image_w = H, W
        d.mean = (C.c_float * 3)(*g.image_mean)
        d.std = (C.c_float * 3)(*g.image_std)
        d.num_classes = g.num_classes
        d.n_levels = len(g.features)
        d.level_tensor = (C.c_int32 * 8)(*(list(g.features) + [0] * (8 - len(g.features))))
        d.anchors_per_loc = (C.c_int32 * 8)(*(list(g.anchors_per_loc) + [0] * (8 - len(g.features))))
        d.num_anchors = self.anchors.shape[0]
        d.anchors = self.anchors.ctypes.data_as(C.POINTER(C.c_float))
        d.score_thresh = float(g.post["score_thresh"])
        d.nms_thresh = float(g.post["nms_thresh"])
        d.detections_per_img = int(g.post["detections_per_img"])
        d.topk_candidates = int(g.post["topk_candidates"])
        self.desc = d

    def create(self) -> int:
        L = _lib.lib()
        handle = C.c_void_p()
        buf = C.create_string_buffer(self.blob, len(self.blob))
        _lib.check(L.dn_create(C.byref(self.desc), buf, len(self.blob), C.byref(handle)), "dn_create")
        return handle.value
