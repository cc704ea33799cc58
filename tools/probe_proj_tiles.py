"""dev tool: the SE-scaled projections of the 20 x 20 blocks (25 600 rows x 480 / 672 -> 112, fp16 out, squeeze-excitation scale on x) through
dn_pointwise_conv with every tile variant of the tiled kernel (dn_debug_pw_tile)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
L.dn_debug_pw_tile.argtypes = [C.c_int]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (hw, ci, co) in [(400, 480, 112), (400, 672, 112), (100, 672, 80), (100, 480, 80)]:
    m = n * hw
    x = torch.randn(m, ci, device="cuda").half(); w = (torch.randn(co, ci, device="cuda") * 0.05).half(); b = torch.randn(co, device="cuda")
    se = torch.rand(n, ci, device="cuda")
    o = torch.empty(m, co, device="cuda", dtype=torch.float16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tile, name in [(0, "auto"), (4, "64x64"), (6, "64x128"), (5, "128x128"), (3, "128x64"), (2, "128x32"), (1, "256x32")]:
        L.dn_debug_pw_tile(tile)
        call = lambda: _lib.check(L.dn_pointwise_conv(P(x), P(w), None, P(b), None, P(se), P(o), m, ci, co, hw, 0, 0, 0, st))
        try:
            for _ in range(3): call()
        except Exception as e:
            print(f"m={m} {ci}->{co} tile {name}: {e}"); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"m={m} {ci}->{co} tile {name:8s} {us:7.1f} us  {2.0 * m * ci * co / us / 1e6:6.0f} TFLOP/s  {(m * (ci + co) * 2) / us / 1e3:6.0f} GB/s", flush=True)
L.dn_debug_pw_tile(0)
