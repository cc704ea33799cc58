"""dev tool (CPU only): where does the fp16 path's logit error come from?  Evaluates the op IR in torch with the device's rounding points
emulated -- BN folded into fp16 weights, every activation tensor rounded to fp16 where the kernels store it, the SE product rounded to
fp16 (it is an MFMA operand), fp32 accumulation -- and switches single rounding sources off:

    python tools/emulate_fp16.py [model]

    all-fp16         what the kernels do (should land near the measured device error, profiles/r0x_layer_errors.txt)
    res-fp32         the residual-stream tensors (block outputs: linear projections) kept in fp32
    w-fp32           weights not rounded to fp16
    exp-fp32         expanded / depthwise tensors kept in fp32 (everything but the residual stream)
The reference is the same IR in fp32 (pinned to the oracle by tools/layer_errors.py)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import models, synth  # noqa: E402

ACT = {0: lambda v: v, 1: F.relu, 2: F.relu6, 3: F.hardswish}
h16 = lambda t: t.half().float()


def folded(nd, sd, round_w):
    w = sd[nd.conv_key + ".weight"].double()
    b = sd[nd.conv_key + ".bias"].double() if nd.has_bias else torch.zeros(w.shape[0], dtype=torch.float64)
    if nd.bn_key:
        s = sd[nd.bn_key + ".weight"].double() / torch.sqrt(sd[nd.bn_key + ".running_var"].double() + nd.bn_eps)
        w = w * s.view(-1, 1, 1, 1)
        b = (b - sd[nd.bn_key + ".running_mean"].double()) * s + sd[nd.bn_key + ".bias"].double()
    w = w.float()
    return (h16(w) if round_w else w), b.float()


def run(name, mode, imgs, g, sd):
    round_w = mode != "w-fp32"
    is_res = set()                      # residual-stream tensors: outputs of backbone projections (pw, act none, not head)
    for nd in g.nodes:
        if nd.op == "pw" and nd.act == 0 and not nd.head:
            is_res.add(nd.out)

    def store(nd, y):
        if mode == "fp32":
            return y
        if mode == "res-fp32" and nd.out in is_res:
            return y
        if mode == "exp-fp32" and nd.out not in is_res:
            return y
        return h16(y)

    mean = torch.tensor(g.image_mean).view(1, 3, 1, 1)
    std = torch.tensor(g.image_std).view(1, 3, 1, 1)
    x0 = (imgs - mean) / std
    val, lg, rg = {}, {}, {}
    for nd in g.nodes:
        if nd.op == "stem":
            w, b = folded(nd, sd, False)            # the stem runs in fp32 on the device
            val[nd.out] = store(nd, ACT[nd.act](F.conv2d(x0, w, b, nd.stride, nd.pad)))
            continue
        x = val[nd.inp]
        if nd.op in ("pw", "conv", "dw"):
            w, b = folded(nd, sd, round_w and mode != "fp32")
            if nd.op == "pw" and nd.se >= 0:
                x = x * val[nd.se][:, :, None, None]
                if mode != "fp32":
                    x = h16(x)
            y = F.conv2d(x, w, b, nd.stride, nd.pad, nd.dil, nd.cin if nd.op == "dw" else 1)
            y = ACT[nd.act](y)
            if nd.op == "pw" and nd.residual >= 0:
                y = y + val[nd.residual]
            if nd.head and nd.op in ("pw", "conv"):
                cols = g.num_classes if nd.head == 1 else 4
                n, _, hh, ww = y.shape
                (lg if nd.head == 1 else rg)[nd.level] = y.view(n, -1, cols, hh, ww).permute(0, 3, 4, 1, 2).reshape(n, -1, cols)
                continue
            if nd.op == "dw" and nd.pool >= 0:
                val[nd.pool] = y                    # pooled in fp32 from the unrounded outputs, as the kernels do
            val[nd.out] = store(nd, y)
        elif nd.op == "se":
            s = x.mean(dim=(2, 3), keepdim=True)
            w1, w2 = sd[nd.fc1_key + ".weight"], sd[nd.fc2_key + ".weight"]
            if mode != "fp32" and round_w:
                w1, w2 = h16(w1), h16(w2)
            s = F.relu(F.conv2d(s, w1, sd[nd.fc1_key + ".bias"]))
            s = F.hardsigmoid(F.conv2d(s, w2, sd[nd.fc2_key + ".bias"]))
            val[nd.out] = s[:, :, 0, 0]
        elif nd.op == "maxpool":
            val[nd.out] = F.max_pool2d(x, nd.k, nd.stride, nd.pad, ceil_mode=nd.ceil_mode)
        elif nd.op == "l2norm":
            val[nd.out] = store(nd, sd[nd.scale_key].view(1, -1, 1, 1) * F.normalize(x))
    return torch.cat([lg[l] for l in sorted(lg)], 1), torch.cat([rg[l] for l in sorted(rg)], 1), val


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ssdlite320_mobilenet_v3_large"
    ncls = 21 if name == "ssd_lite_mobilenet_v2" else 91
    m = getattr(models, name)(num_classes=ncls)
    g = m.graph
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict(g, 0).items()}
    W, H = g.size
    imgs = torch.from_numpy(synth.images(1, 2, H, W))
    torch.set_num_threads(8)
    with torch.no_grad():
        ref_l, ref_r, ref_v = run(name, "fp32", imgs, g, sd)
        print(f"{name}: max|logit| {ref_l.abs().max():.2f}")
        for mode in ("all-fp16", "res-fp32", "w-fp32", "exp-fp32"):
            l, r, v = run(name, mode, imgs, g, sd)
            d = (l - ref_l).abs()
            dr = (r - ref_r).abs()
            last = [nd.out for nd in g.nodes if not nd.head and nd.out in v and nd.out in ref_v and v[nd.out].dim() == 4][-12]
            print(f"  {mode:9s} logits max {d.max():.4f} mean {d.mean():.5f}   regression max {dr.max():.4f} mean {dr.mean():.5f}")


if __name__ == "__main__":
    main()
