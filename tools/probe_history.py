"""dev tool: does a forward's result depend on what the workspace held before it (uninitialised / stale reads)?  One forward at a time only.
   usage: [DN_DW_ROWS=6] probe_history.py <batch>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demonet_amd import models, synth
from demonet_amd.pipeline import ForwardPipeline
n = int(sys.argv[1])
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
W, H = m.graph.size
b = [torch.from_numpy(synth.images(90 + i, n, H, W)).cuda() for i in range(4)]
def fwd(x):
    r = [t.clone() for t in m.forward_batch(x)]
    torch.cuda.synchronize()
    return r
same = lambda a, c: all(torch.equal(x, y) for x, y in zip(a, c))
for chains in (0, 1):
    ctx = ForwardPipeline(m, n, depth=2) if chains else None      # chains=1 layout while a pipeline is open
    fwd(b[0]); fwd(b[1]); r_a = fwd(b[2])                          # history: b1 before b2
    fwd(b[0]); fwd(b[3]); r_b = fwd(b[2])                          # history: b3 before b2
    r_c = fwd(b[2])                                                # history: b2 before b2
    key = next(k for k in m._bufs if k[0] == n)
    ws = m._bufs[key]["ws"]
    ws.fill_(0xFF); torch.cuda.synchronize()                       # NaN poison (fp16 and fp32)
    r_d = fwd(b[2])
    ws.zero_(); torch.cuda.synchronize()
    r_e = fwd(b[2])
    print(f"batch {n} {'single chain (pipeline layout)' if chains else 'default split'}: after b1 == after b3: {same(r_a, r_b)}; == repeated: {same(r_a, r_c)}; "
          f"== after NaN fill: {same(r_a, r_d)} (NaN in scores: {bool(torch.isnan(r_d[1]).any())}); == after zero fill: {same(r_a, r_e)}", flush=True)
    if ctx: ctx.close()
