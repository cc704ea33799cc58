"""dev tool: the level-0 class head as a single GEMM (m rows x 672 -> 546, fp32 rows) through dn_pointwise_conv: tile variants of the tiled kernel
(dn_debug_pw_tile) and its probe-only ablations (act >> 8: 1 = no stores, 2 = no global loads of x)."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
L.dn_debug_pw_tile.argtypes = [C.c_int]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
m, ci, co = int(sys.argv[1]) if len(sys.argv) > 1 else 25600, 672, 546
x = torch.randn(m, ci, device="cuda").half(); w = (torch.randn(co, ci, device="cuda") * 0.05).half(); b = torch.randn(co, device="cuda")
o = torch.empty(m, co, device="cuda", dtype=torch.float32)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for tile, name in [(0, "auto"), (4, "64x64"), (6, "64x128"), (5, "128x128"), (7, "128x96"), (3, "128x64")]:
    L.dn_debug_pw_tile(tile)
    for dbg, what in [(0, "as is"), (1, "no stores"), (2, "no x loads"), (3, "neither")]:
        call = lambda: _lib.check(L.dn_pointwise_conv(P(x), P(w), None, P(b), None, None, P(o), m, ci, co, 400, dbg << 8, 1, 400 * co, st))
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"m={m} tile {name:8s} {what:11s} {us:7.1f} us  {2.0 * m * ci * co / us / 1e6:6.0f} TFLOP/s", flush=True)
L.dn_debug_pw_tile(0)
