"""dev tool: time dn_depthwise_conv on the backbone's depthwise shapes (no pooling) under the current DN_* knobs."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [(80, 72, 5, 2), (40, 120, 5, 1), (20, 480, 3, 1), (20, 672, 3, 1), (20, 672, 5, 2), (10, 960, 5, 1), (40, 240, 3, 2)]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (h, c, k, s) in SHAPES:
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // s + 1
    xs = [torch.randn(N, h, h, c, device="cuda").half() for _ in range(4)]
    outs = [torch.empty(N, ho, ho, c, device="cuda", dtype=torch.half) for _ in range(4)]
    w = (torch.randn(k * k, c, device="cuda") / k).half(); b = torch.randn(c, device="cuda")
    call = lambda i: _lib.check(L.dn_depthwise_conv(P(xs[i % 4]), P(w), P(b), P(outs[i % 4]), N, h, h, c, k, s, pad, 1, stream), "dw")
    call(0); call(1); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): call(i)
    e1.record(); torch.cuda.synchronize()
    print(f"{h:3d}x{h:<3d} c{c:4d} k{k}s{s}: {e0.elapsed_time(e1) * 50:6.1f} us/launch", flush=True)
