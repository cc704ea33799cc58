"""Authoring-time tool: measure per-conv multipliers for the synthetic weights (demonet_amd/synth_calib.json).

Runs the ORACLE (CPU) once per model on two synthetic images; at every conv it rescales the synthetic weight so
that the conv output has the variance the graph builder intended (BN-normalised variance 1*target when a BN
follows, `target` otherwise), exactly the job training + BN statistics do for a real checkpoint. The result is a
small table of floats, committed as data; nothing here runs in the product path.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from demonet_amd import spec, synth  # noqa: E402
import ssd_oracle as so  # noqa: E402


def calibrate(name, num_classes, n_img=2, seed=0, **gkw):
    g = spec.GRAPHS[name](num_classes=num_classes, **gkw)
    sd = so._to_torch_sd(synth.state_dict(g, seed, calibrated=False))
    by_id = {id(sd[p.key]): p for p in g.params if p.kind == "conv_w"}
    mult = {}
    orig = F.conv2d

    def patched(x, w, b=None, *a, **k):
        y = orig(x, w, b, *a, **k)
        p = by_id.get(id(w))
        if p is None or p.target <= 0 or x.shape[-1] * x.shape[-2] * x.shape[0] < 2 and not p.bn:
            return y
        yc = orig(x, w, None, *a, **k)
        var = yc.var(dim=(0, 2, 3), unbiased=False) + yc.mean(dim=(0, 2, 3)) ** 2 * 0.0
        if x.shape[0] * y.shape[-1] * y.shape[-2] < 8:
            m2 = float((yc ** 2).mean())            # tiny maps: use second moment
            ratio = m2 / (p.target * (float(sd[p.bn + ".running_var"].mean()) if p.bn else 1.0))
        elif p.bn:
            ratio = float((var / sd[p.bn + ".running_var"]).mean()) / p.target
        else:
            ratio = float(var.mean()) / p.target
        m = float(1.0 / np.sqrt(max(ratio, 1e-12)))
        m = min(max(m, 0.05), 20.0)
        w.mul_(m)
        mult[p.key] = round(m, 5)
        return orig(x, w, b, *a, **k)

    so.F.conv2d = patched
    F.conv2d = patched
    try:
        o = so.OracleSSD(name, {}, num_classes, size=g.size)
        o.sd = sd
        imgs = synth.images(4242, n_img, g.size[1], g.size[0])
        r = o.forward_raw([torch.from_numpy(imgs[i]) for i in range(n_img)])
    finally:
        so.F.conv2d = orig
        F.conv2d = orig
    print(name, "logit std %.3f max %.2f  reg std %.3f" % (r["cls_logits"].std(), r["cls_logits"].abs().max(),
                                                            r["bbox_regression"].std()))
    for f in r["features"]:
        print("   feature", tuple(f.shape), "rms %.3f max %.2f" % (float((f ** 2).mean()) ** 0.5, float(f.abs().max())))
    return mult


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["ssdlite320_mobilenet_v3_large", "ssd_lite_mobilenet_v2", "ssd300_vgg16", "ssd512_vgg16"]
    path = os.path.join(ROOT, "demonet_amd", "synth_calib.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    ncls = {"ssd_lite_mobilenet_v2": 21}
    for name in which:
        table[name] = {}
        for seed in synth.CALIBRATED_SEEDS:
            with torch.no_grad():
                table[name][str(seed)] = calibrate(name, ncls.get(name, 91), n_img=1 if "vgg" in name else 2, seed=seed)
    json.dump(table, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path)
