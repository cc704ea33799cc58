import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demonet_amd import models, synth
from demonet_amd.pipeline import ForwardPipeline
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
n, depth = int(sys.argv[1]), int(sys.argv[2])
W, H = m.graph.size
batches = [torch.from_numpy(synth.images(40 + i, n, H, W)).cuda() for i in range(7)]
with ForwardPipeline(m, n, depth=depth, packed=True) as pipe:
    ref = [[t.clone() for t in m.forward_batch(b)] for b in batches]
    ref2 = [[t.clone() for t in m.forward_batch(b)] for b in batches]
    for k in range(7):
        print("serial vs serial", k, [bool(torch.equal(a, b)) for a, b in zip(ref[k], ref2[k])])
    for rnd in range(3):
        base = pipe.n
        ts = []
        for k, b in enumerate(batches):
            ts.append(pipe.submit(b))
            j = k - (depth - 1)
            if j >= 0:
                out = [x.clone() for x in pipe.result(base + j)]
                eq = [bool(torch.equal(a, b)) for a, b in zip(ref[j], out)]
                if not all(eq):
                    d = (ref[j][1] - out[1]).abs()
                    print("round", rnd, "batch", j, eq, "scores max|d|", float(d.max()), "nan", bool(torch.isnan(out[1]).any()), "counts", ref[j][3].tolist(), out[3].tolist(),
                          "images differing", [int(i) for i in range(n) if not torch.equal(ref[j][1][i], out[1][i])])
    # heads only: which forward differs
    hs = [[t.clone() for t in m.forward_heads(b)] for b in batches]
    hs2 = [[t.clone() for t in m.forward_heads(b)] for b in batches]
    for k in range(7):
        e = [bool(torch.equal(a, b)) for a, b in zip(hs[k], hs2[k])]
        if not all(e):
            d = (hs[k][0] - hs2[k][0]).abs()
            print("heads differ", k, e, float(d.max()), "rows", int((d.amax(2) > 0).sum()))
print("done")
