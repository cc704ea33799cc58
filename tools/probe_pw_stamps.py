"""dev tool: per-workgroup phase timing of the tiled pointwise kernel (s_memrealtime, 100 MHz)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
L.dn_debug_pw_stamps.argtypes = [C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (m, ci, co, act) in [(12800, 80, 184, 3), (12800, 112, 672, 3), (51200, 40, 120, 1), (51200, 72, 40, 0), (204800, 24, 72, 1), (3200, 80, 480, 3), (12800, 80, 480, 3)]:
    x = torch.randn(m, ci, device="cuda").half(); w = torch.randn(co, ci, device="cuda").half(); b = torch.randn(co, device="cuda")
    o = torch.empty(m, co, device="cuda", dtype=torch.half)
    st = torch.zeros(8 * 40000, dtype=torch.int64, device="cuda")
    call = lambda: _lib.check(L.dn_pointwise_conv(P(x), P(w), None, P(b), None, None, P(o), m, ci, co, m, act, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for _ in range(3): call()
    torch.cuda.synchronize()
    L.dn_debug_pw_stamps(C.c_void_p(st.data_ptr())); call(); torch.cuda.synchronize(); L.dn_debug_pw_stamps(None)
    s = st.cpu().numpy().reshape(-1, 8)[:, :4].astype(np.float64)
    s = s[s[:, 0] > 0]
    d = np.diff(s, axis=1) * 0.01
    print(f"m={m} {ci}->{co}: WGs={len(s)} prologue {d[:,0].mean():.2f} kloop {d[:,1].mean():.2f} epilogue {d[:,2].mean():.2f} us;"
          f" WG life {d.sum(1).mean():.2f} (max {d.sum(1).max():.2f}); kernel span {(s[:,3].max()-s[:,0].min())*0.01:.1f} us; "
          f"start spread {(s[:,0].max()-s[:,0].min())*0.01:.1f} us")
