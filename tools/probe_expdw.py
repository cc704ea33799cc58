"""dev tool: time + per-workgroup phase stamps (s_memrealtime, 100 MHz) of the fused expand+depthwise kernel on the
backbone's block shapes at a given batch."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
L.dn_debug_expdw_stamps.argtypes = [C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(160, 16, 64, 3, 2), (80, 24, 72, 3, 1), (80, 24, 72, 5, 2), (40, 40, 120, 5, 1), (40, 40, 240, 3, 2),
          (20, 80, 200, 3, 1), (20, 80, 480, 3, 1), (20, 112, 672, 3, 1), (10, 80, 480, 5, 1)]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (h, cin, cexp, k, s) in SHAPES:
    ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
    R = 4
    xs = [torch.randn(N, h, h, cin, device="cuda").half() for _ in range(R)]
    outs = [torch.empty(N, ho, ho, cexp, device="cuda", dtype=torch.half) for _ in range(R)]
    w1 = (torch.randn(cexp, cin, device="cuda") / cin ** 0.5).half(); b1 = torch.randn(cexp, device="cuda")
    wd = (torch.randn(k * k, cexp, device="cuda") / k).half(); bd = torch.randn(cexp, device="cuda")
    call = lambda i: _lib.check(L.dn_expand_depthwise(P(xs[i % R]), P(w1), P(b1), P(wd), P(bd), None, None, P(outs[i % R]), None, N, h, h, cin, cexp, 0, k, s, 1, 1, 0, stream))
    call(0); call(1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    st = torch.zeros(16 * 200000, dtype=torch.int64, device="cuda")
    L.dn_debug_expdw_stamps(C.c_void_p(st.data_ptr())); call(0); torch.cuda.synchronize(); L.dn_debug_expdw_stamps(None)
    t = st.cpu().numpy().reshape(-1, 16)[:, :5].astype(np.float64)
    t = t[t[:, 0] > 0]
    d = np.diff(t, axis=1) * 0.01
    mb = N * (h * h * cin + ho * ho * cexp) * 2 / 1e6
    print(f"{h:3d}x{h:<3d} {cin:3d}->{cexp:3d} k{k}s{s}: {us:6.1f} us/launch ({mb / us * 1e6 / 1e6:6.0f} GB/s ext) WGs {len(t):5d} | stage-x {d[:,0].mean():5.2f} "
          f"mfma(1st chunk) {d[:,1].mean():5.2f} dw(1st chunk) {d[:,2].mean():5.2f} rest {d[:,3].mean():5.2f} | life {d.sum(1).mean():5.2f} "
          f"span {(t[:,4].max() - t[:,0].min()) * 0.01:6.1f}", flush=True)
