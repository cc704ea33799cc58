"""dev tool: time every tile variant of the pointwise GEMM on every 1x1-conv shape of a model (batch B) -> table.

    python tools/tune_pw.py [model] [batch]

Each (shape, tile) is timed as the mean of 10 launches over a ring of 6 rotating input/output buffers inside a larger
ring flush, so x/out do not sit in L2 / Infinity Cache from the previous launch (as in the real forward)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, spec
from demonet_amd.plan import fragment_major
L = _lib.lib()
L.dn_debug_pw_tile.argtypes = [C.c_int]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
TILES = {1: "256x32", 2: "128x32", 3: "128x64", 4: "64x64", 5: "128x128", 6: "64x128", 7: "128x96", 8: "xs32"}
name = sys.argv[1] if len(sys.argv) > 1 else "ssdlite320_mobilenet_v3_large"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = spec.GRAPHS[name]()
shapes = []
for nd in g.nodes:
    if nd.op != "pw" or nd.head:
        continue
    to = g.t(nd.out)
    shapes.append((B * to.h * to.w, nd.cin, nd.cout, nd.residual >= 0, nd.se >= 0, to.h * to.w))
seen = set()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for (m, ci, co, res, se, hw) in shapes:
    key = (m, ci, co, res, se)
    if key in seen:
        continue
    seen.add(key)
    R = 6
    xs = [torch.randn(m, ci, device="cuda").half() for _ in range(R)]
    os_ = [torch.empty(m, co, device="cuda", dtype=torch.half) for _ in range(R)]
    rs = [torch.randn(m, co, device="cuda").half() for _ in range(R)] if res else [None] * R
    sev = torch.rand(m // hw, ci, device="cuda") if se else None
    w = torch.randn(co, ci, device="cuda").half(); b = torch.randn(co, device="cuda")
    wf = torch.from_numpy(fragment_major(w.cpu().numpy())).cuda() if ci % 16 == 0 else None
    row = []
    for t in [0] + list(TILES):
        if t == 8 and (ci % 16 or ci > 1024 or m > 60000):
            row.append(float("nan")); continue
        L.dn_debug_pw_tile(t)
        def call(i):
            _lib.check(L.dn_pointwise_conv(P(xs[i % R]), P(w), P(wf), P(b), P(rs[i % R]), P(sev), P(os_[i % R]), m, ci, co, hw, 3, 0, 0, stream))
        call(0); call(1)
        tot = 0.0
        for i in range(10):
            flush.add_(1) if m * (ci + co) * 2 < (64 << 20) else None     # push small tensors out of the caches
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); call(i); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        row.append(tot / 10 * 1e3)
    L.dn_debug_pw_tile(0)
    best = min(range(1, len(row)), key=lambda i: row[i] if row[i] == row[i] else 1e9)
    print(f"m={m:8d} {ci:4d}->{co:4d} res={int(res)} se={int(se)} auto {row[0]:6.1f} | " +
          " ".join(f"{TILES[t]}:{row[i + 1]:6.1f}" for i, t in enumerate(TILES)) + f" | best {TILES[list(TILES)[best - 1]]} {row[best]:.1f}", flush=True)
