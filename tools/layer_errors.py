"""Per-layer numerical error of the HIP path against the fp32 CPU path, for every model graph.

    python tools/layer_errors.py [model ...] [--out FILE]            (needs the GPU)

For every op of the graph, on the golden input images:
  acc   = max / mean |device tensor - fp32 chain|      error accumulated from the input up to this op's output
  local = max / mean |device tensor - fp32 op(device's own fp16 input)|   what this op alone adds (rounding of its output + kernel)
The fp32 chain is a torch evaluation of the op IR (conv -> bias -> BN -> activation -> SE -> residual, unfolded, in the
reference's order); it is pinned to the oracle in this script (its logits must equal oracle/ssd_oracle.py's within 2e-4, which in
turn equals the reference bit for bit: tests/test_oracle.py). Fused launches are switched off here (DN_EXPDW=0, DN_TAIL=0) so that
every intermediate tensor exists in the workspace; their rounding points are the same as the separate kernels'.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

ACT = {0: lambda v: v, 1: F.relu, 2: F.relu6, 3: F.hardswish}


def conv_like(nd, sd, x, groups=1):
    w = sd[nd.conv_key + ".weight"]
    b = sd[nd.conv_key + ".bias"] if nd.has_bias else None
    y = F.conv2d(x, w, b, nd.stride, nd.pad, nd.dil, groups)
    if nd.bn_key:
        y = F.batch_norm(y, sd[nd.bn_key + ".running_mean"], sd[nd.bn_key + ".running_var"], sd[nd.bn_key + ".weight"],
                         sd[nd.bn_key + ".bias"], False, 0.0, nd.bn_eps)
    return ACT[nd.act](y)


def apply_op(g, nd, sd, val, images_norm):
    """fp32 value of node nd given the dict `val` of tensor id -> NCHW fp32 (vec tensors: [N, C])."""
    if nd.op == "stem":
        return conv_like(nd, sd, images_norm)
    x = val[nd.inp]
    if nd.op == "pw":
        if nd.se >= 0:
            x = x * val[nd.se][:, :, None, None]
        y = conv_like(nd, sd, x)
        if nd.residual >= 0:
            y = y + val[nd.residual]
        return y
    if nd.op == "dw":
        return conv_like(nd, sd, x, groups=nd.cin)
    if nd.op == "conv":
        return conv_like(nd, sd, x)
    if nd.op == "se":
        s = x.mean(dim=(2, 3), keepdim=True)
        s = F.relu(F.conv2d(s, sd[nd.fc1_key + ".weight"], sd[nd.fc1_key + ".bias"]))
        s = F.hardsigmoid(F.conv2d(s, sd[nd.fc2_key + ".weight"], sd[nd.fc2_key + ".bias"]))
        return s[:, :, 0, 0]
    if nd.op == "maxpool":
        return F.max_pool2d(x, nd.k, nd.stride, nd.pad, ceil_mode=nd.ceil_mode)
    if nd.op == "l2norm":
        return sd[nd.scale_key].view(1, -1, 1, 1) * F.normalize(x)
    raise ValueError(nd.op)


def head_rows(g, nd, y):
    """[N, A*cols, H, W] -> [N, H*W*A, cols] (generalized_ssd.py:66-71)"""
    cols = g.num_classes if nd.head == 1 else 4
    n, _, h, w = y.shape
    return y.view(n, -1, cols, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, cols)


def run(name, out):
    import ssd_oracle as so
    from demonet_amd import models, synth
    os.environ["DN_EXPDW"] = "0"
    os.environ["DN_TAIL"] = "0"
    os.environ["DN_WS_REUSE"] = "0"         # every intermediate tensor keeps its own block (readable after the forward)
    ncls = 21 if name == "ssd_lite_mobilenet_v2" else 91
    size = None
    if ":" in name:                         # "ssd_lite_mobilenet_v2:300": the hub model at another input size (BASELINE config C3)
        name, sz = name.split(":")
        size = int(sz)
    m = getattr(models, name)(num_classes=ncls, **({"image_size": size} if size else {}))
    g = m.graph
    sdn = synth.state_dict(g, 0)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sdn.items()}
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    W, H = g.size
    imgs = torch.from_numpy(synth.images(1, 2, H, W))
    mean = torch.tensor(g.image_mean).view(1, 3, 1, 1)
    std = torch.tensor(g.image_std).view(1, 3, 1, 1)
    xin = (imgs - mean) / std
    with torch.no_grad():
        val = {}
        logits_parts, reg_parts = {}, {}
        for nd in g.nodes:
            y = apply_op(g, nd, sd, val, xin)
            if nd.head:
                (logits_parts if nd.head == 1 else reg_parts)[nd.level] = head_rows(g, nd, y) if nd.op in ("pw", "conv") else None
                if nd.op in ("pw", "conv"):
                    continue
            val[nd.out] = y
            if nd.op == "dw" and nd.pool >= 0:
                val[nd.pool] = y            # the SE node reads the pooled partial sums of this tensor: give it the tensor itself
        ref_logits = torch.cat([logits_parts[l] for l in sorted(logits_parts)], 1)
        ref_reg = torch.cat([reg_parts[l] for l in sorted(reg_parts)], 1)
        o = so.OracleSSD(name, sdn, ncls, **({"size": (size, size)} if size else {}))
        raw = o.forward_raw(list(imgs))
        pin = (ref_logits - raw["cls_logits"]).abs().max().item()
        assert pin < 2e-4, f"fp32 chain of the op IR differs from the oracle: {pin}"
        dl, dr = m.forward_heads(imgs.cuda())
        out.write(f"## {name}{':%d' % size if size else ''}  (2 golden-style images, synthetic weights seed 0; fp32 chain vs oracle logits: max|d| {pin:.1e})\n")
        out.write(f"{'op':4s} {'key':52s} {'shape':>16s} {'max|ref|':>9s} {'acc max':>9s} {'acc mean':>9s} {'local max':>9s} {'local mean':>10s}\n")
        dev = {}
        for i, nd in enumerate(g.nodes):
            if nd.head and nd.op in ("pw", "conv"):
                continue
            t = g.t(nd.out)
            try:
                d = m.tensor(tuple(imgs.shape), nd.out)
            except Exception:
                continue
            if t.kind == "act":
                dv = d.float().cpu().permute(0, 3, 1, 2)
            else:
                continue
            dev[nd.out] = dv
            ref = val[nd.out]
            acc = (dv - ref).abs()
            # local: the same op in fp32 on the device's own (fp16) inputs
            loc_val = dict(val)
            ok = True
            for tid in (nd.inp, nd.residual):
                if tid >= 0 and g.t(tid).kind == "act":
                    if tid in dev:
                        loc_val[tid] = dev[tid]
                    elif nd.op != "stem":
                        ok = False
            if nd.op == "stem":
                loc = acc
            elif ok:
                loc = (dv - apply_op(g, nd, sd, loc_val, xin)).abs()
            else:
                loc = None
            key = nd.conv_key or nd.fc1_key or nd.scale_key or nd.op
            out.write(f"{nd.op:4s} {key[-52:]:52s} {str(tuple(ref.shape[1:])):>16s} {ref.abs().max().item():9.3g} {acc.max().item():9.2e} "
                      f"{acc.mean().item():9.2e} " + (f"{loc.max().item():9.2e} {loc.mean().item():10.2e}" if loc is not None else f"{'-':>9s} {'-':>10s}") + "\n")
        e = (dl.cpu() - ref_logits).abs()
        r = (dr.cpu() - ref_reg).abs()
        # tolerance candidates atol + rtol * |ref|: worst ratio err / tolerance over all logits (tests use <= 2x the measured envelope)
        for atol, rtol in ((1e-3, 0.0), (2e-2, 5e-3), (3e-2, 5e-3), (4e-2, 1e-2), (6e-2, 1e-2), (8e-2, 1.5e-2)):
            out.write(f"   logits: max err / ({atol:g} + {rtol:g} |ref|) = {(e / (atol + rtol * ref_logits.abs())).max().item():.3f}"
                      f"   reg: {(r / (atol + rtol * ref_reg.abs())).max().item():.3f}\n")
        lv0 = 0
        for lvl in sorted(logits_parts):
            cnt = logits_parts[lvl].shape[1]
            out.write(f"   level {lvl}: logits max|err| {e[:, lv0:lv0 + cnt].max().item():.3e} mean {e[:, lv0:lv0 + cnt].mean().item():.3e}\n")
            lv0 += cnt
        out.write(f"head cls_logits: max|ref| {ref_logits.abs().max().item():.3g}  max|err| {e.max().item():.3e}  mean|err| {e.mean().item():.3e}  "
                  f"max err/(atol-free) rel {(e / (1e-6 + ref_logits.abs())).median().item():.2e} (median)\n")
        out.write(f"head bbox_regression: max|ref| {ref_reg.abs().max().item():.3g}  max|err| {r.max().item():.3e}  mean|err| {r.mean().item():.3e}\n\n")
    m.release()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("models", nargs="*", default=["ssdlite320_mobilenet_v3_large", "ssd_lite_mobilenet_v2", "ssd300_vgg16", "ssd512_vgg16"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    f = open(a.out, "w") if a.out else sys.stdout
    for nm in a.models:
        run(nm, f)
        f.flush()
