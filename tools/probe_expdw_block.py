"""dev tool: time + per-workgroup phase stamps of the WHOLE-block form of the fused kernel (expand -> depthwise -> project [+ residual]) on the 20 x 20
block shapes. Stamps (s_memrealtime, 100 MHz): 0 start, 1 region staged, 2 first chunk expanded, 3 first chunk done, 4 end."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
L.dn_debug_expdw_stamps.argtypes = [C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [(40, 40, 240, 80, 3, 2, 0), (20, 80, 200, 80, 3, 1, 1), (20, 80, 184, 80, 3, 1, 1)]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (h, cin, cexp, cout, k, s, res) in SHAPES:
    ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
    R = 4
    xs = [torch.randn(N, h, h, cin, device="cuda").half() for _ in range(R)]
    outs = [torch.empty(N, ho, ho, cout, device="cuda", dtype=torch.half) for _ in range(R)]
    w1 = (torch.randn(cexp, cin, device="cuda") / cin ** 0.5).half(); b1 = torch.randn(cexp, device="cuda")
    wd = (torch.randn(k * k, cexp, device="cuda") / k).half(); bd = torch.randn(cexp, device="cuda")
    w3 = (torch.randn(cout, cexp, device="cuda") / cexp ** 0.5).half(); b3 = torch.randn(cout, device="cuda")
    call = lambda i: _lib.check(L.dn_expand_depthwise(P(xs[i % R]), P(w1), P(b1), P(wd), P(bd), P(w3), P(b3), P(outs[i % R]), None, N, h, h, cin, cexp, cout, k, s, 3, 3, res, stream), "expdw")
    call(0); call(1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    st = torch.zeros(16 * 200000, dtype=torch.int64, device="cuda")
    L.dn_debug_expdw_stamps(C.c_void_p(st.data_ptr())); call(0); torch.cuda.synchronize(); L.dn_debug_expdw_stamps(None)
    full = st.cpu().numpy().reshape(-1, 16).astype(np.float64)
    full = full[full[:, 0] > 0]
    t = full[:, :5]
    d = np.diff(t, axis=1) * 0.01
    if full[:, 5].max() > 0:        # stamped build: the second chunk in detail
        f = full[full[:, 5] > 0][:, 5:13]
        dd = np.diff(f, axis=1).mean(0) * 0.01
        print("    2nd chunk: Wd/Bd->LDS %.2f | expand %.2f | request+barrier %.2f | depthwise %.2f | pool/zero+barrier %.2f | project %.2f | barrier %.2f  (us)" % tuple(dd))
    print(f"{h:3d}x{h:<3d} {cin:3d}->{cexp:3d}->{cout:3d} k{k}s{s}: {us:6.1f} us/launch WGs {len(t):5d} | stage-x {d[:,0].mean():5.2f} "
          f"expand(1st chunk) {d[:,1].mean():5.2f} dw+proj(1st chunk) {d[:,2].mean():5.2f} rest {d[:,3].mean():5.2f} | life {d.sum(1).mean():5.2f} "
          f"span {(t[:,4].max() - t[:,0].min()) * 0.01:6.1f}  start spread {(t[:,0].max() - t[:,0].min()) * 0.01:5.1f}", flush=True)
