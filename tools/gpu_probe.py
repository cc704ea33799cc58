"""GPU micro-benchmark of single kernels through the C ABI (dev tool). usage: python tools/gpu_probe.py pw|dw ..."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def time_fn(fn, iters=20, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def pw(m, cin, cout, hw, act=0, res=False, fp32=False, nbuf=4, dbg=0):
    xs = [torch.randn(m, cin, device="cuda").half() for _ in range(nbuf)]
    w = torch.randn(cout, cin, device="cuda").half()
    b = torch.randn(cout, device="cuda")
    outs = [torch.empty(m, cout, device="cuda", dtype=torch.float32 if fp32 else torch.half) for _ in range(nbuf)]
    r = torch.randn(m, cout, device="cuda").half() if res else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fn(i):
        _lib.check(L.dn_pointwise_conv(P(xs[i % nbuf]), P(w), None, P(b), P(r), None, P(outs[i % nbuf]), m, cin, cout, hw,
                                       act | (dbg << 8), int(fp32), hw * cout, st))
    us = time_fn(fn)
    by = 2 * m * cin + (4 if fp32 else 2) * m * cout + 2 * cin * cout + (2 * m * cout if res else 0)
    fl = 2.0 * m * cin * cout
    print(f"pw m={m:8d} {cin:4d}->{cout:4d} dbg={dbg} : {us:8.1f} us  {by / us / 1e3:7.0f} GB/s  {fl / us / 1e6:7.1f} TF/s")


if __name__ == "__main__":
    shapes = [(1638400, 16, 64, 0), (409600, 24, 72, 0), (409600, 72, 24, 0), (102400, 40, 120, 0), (102400, 120, 40, 0),
              (25600, 80, 200, 3), (25600, 200, 80, 0), (25600, 112, 672, 3), (25600, 672, 112, 0), (6400, 672, 80, 0),
              (6400, 80, 480, 3), (1600, 512, 128, 2)]
    for dbg in (0, 1, 2, 3):
        for (m, ci, co, act) in shapes:
            pw(m, ci, co, m, act, dbg=dbg)
    pw(25600, 672, 546, 400, 0, fp32=True)
    pw(25600, 672, 546, 400, 0, fp32=True, dbg=1)
