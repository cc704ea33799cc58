"""dev tool: stand-alone timing of one 1x1 layer through the C ABI (dn_pointwise_conv), HIP events around 200 launches.
    python tools/time_pw.py M CIN COUT ACT [HW]          (DEMONET_HIP_LIB selects a dev build)"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from demonet_amd import _lib
L = _lib.lib()
m, ci, co, act = (int(v) for v in sys.argv[1:5])
hw = int(sys.argv[5]) if len(sys.argv) > 5 else 400
x = torch.randn(m, ci, device="cuda").half(); w = torch.randn(co, ci, device="cuda").half(); b = torch.randn(co, device="cuda")
o = torch.empty(m, co, device="cuda", dtype=torch.half)
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: _lib.check(L.dn_pointwise_conv(P(x), P(w), None, P(b), None, None, P(o), m, ci, co, hw, act, 0, 0, s))
for _ in range(20): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): call()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
print(f"M={m} {ci}->{co} act={act}: {us:.2f} us per launch, {(m * (ci + co) * 2 + ci * co * 2) / us / 1e6:.2f} TB/s")
