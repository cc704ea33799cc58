"""dev tool: time + phase stamps of the fused inverted-residual kernel on the network's block shapes."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
L.dn_debug_fused_stamps.argtypes = [C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
N = 64
#        name   h   cin cexp cout k s proj pool res
SHAPES = [("b1", 160, 16, 16, 16, 3, 1, 1, 0, 1), ("b2", 160, 16, 64, 24, 3, 2, 1, 0, 0), ("b3", 80, 24, 72, 24, 3, 1, 1, 0, 1),
          ("b4", 80, 24, 72, 0, 5, 2, 0, 1, 0), ("b5", 40, 40, 120, 0, 5, 1, 0, 1, 0), ("b7", 40, 40, 240, 80, 3, 2, 1, 0, 0),
          ("b8", 20, 80, 200, 80, 3, 1, 1, 0, 1), ("b12", 20, 112, 672, 0, 3, 1, 0, 1, 0), ("b14", 10, 80, 480, 0, 5, 1, 0, 1, 0)]
for name, h, cin, cexp, cout, k, s, proj, pool, res in SHAPES:
    exp = cexp != cin or name != "b1"
    x = torch.randn(N, h, h, cin, device="cuda").half()
    w1 = torch.randn(cexp, cin, device="cuda").half() if exp else None
    b1 = torch.randn(cexp, device="cuda") if exp else None
    wd = torch.randn(k * k, cexp, device="cuda").half(); bd = torch.randn(cexp, device="cuda")
    w3 = torch.randn(cout, cexp, device="cuda").half() if proj else None
    b3 = torch.randn(cout, device="cuda") if proj else None
    ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
    out = torch.empty(N, ho, ho, cout if proj else cexp, device="cuda", dtype=torch.half)
    tiles = L.dn_fused_tiles_per_image(ho, ho)
    pp = torch.empty(N, tiles, cexp, device="cuda") if pool else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: _lib.check(L.dn_fused_block(P(x), P(w1), P(b1), P(wd), P(bd), P(w3), P(b3), P(out), P(pp), N, h, h, cin, cexp,
                                               cout, k, s, 1, 1, 0, res, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    stm = torch.zeros(8 * N * tiles, dtype=torch.int64, device="cuda")
    L.dn_debug_fused_stamps(C.c_void_p(stm.data_ptr())); call(); torch.cuda.synchronize(); L.dn_debug_fused_stamps(None)
    sa = stm.cpu().numpy().reshape(-1, 8).astype(np.float64)
    sa = sa[sa[:, 0] > 0]
    d = lambda i, j: ((sa[:, j] - sa[:, i]) * 0.01).mean()
    last = 6 if proj else 5
    print(f"{name}: {us:7.1f} us  WGs={len(sa)} stage {d(0,1):.2f} | chunk0: expand {d(1,2) if exp else 0:.2f} dw {d(2 if exp else 1,3):.2f} "
          f"stash+sync {d(3,4):.2f} | all chunks {d(1,5):.2f} | epilogue {d(5,last):.2f} | life {d(0,last):.2f} "
          f"span {(sa[:,last].max()-sa[:,0].min())*0.01:.1f}")
