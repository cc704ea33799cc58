"""dev tool: per-workgroup phase stamps (s_memrealtime, 100 MHz) of one fused block launch inside a real forward.
    usage: DN_EXPDW_STAMP_SEL=<10*H+stride> probe_expdw_model.py [batch]      e.g. 1602 = the 160x160 stride-2 block, 801 = the 80x80 stride-1 block
stamps: 0 start, 1 input region staged, 2 first chunk expanded, 3 first chunk depthwise + project done, 4 output written"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
raw.dn_debug_expdw_stamps.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
x = torch.from_numpy(synth.images(3, n, 320, 320)).cuda()
for _ in range(3):
    m.forward_heads(x)
torch.cuda.synchronize()
st = torch.zeros(16 * 200000, dtype=torch.int64, device="cuda")
raw.dn_debug_expdw_stamps(C.c_void_p(st.data_ptr()))
m.forward_heads(x)
torch.cuda.synchronize()
raw.dn_debug_expdw_stamps(None)
t = st.cpu().numpy().reshape(-1, 16)[:, :5].astype(np.float64)
t = t[(t[:, 0] > 0) & (t[:, 4] > 0)]
d = np.diff(t, axis=1) * 0.01
t0 = t[:, 0].min()
print(f"sel {os.environ.get('DN_EXPDW_STAMP_SEL', '0')} batch {n}: {len(t)} workgroups; stage-x {d[:,0].mean():.2f} expand {d[:,1].mean():.2f} dw+proj {d[:,2].mean():.2f} out {d[:,3].mean():.2f} | life {d.sum(1).mean():.2f} "
      f"(p10 {np.percentile(d.sum(1), 10):.2f} p90 {np.percentile(d.sum(1), 90):.2f}) span {(t[:,4].max() - t0) * 0.01:.1f} us; resident = life x wgs / span = {d.sum(1).sum() / ((t[:,4].max() - t0) * 0.01):.0f} workgroups")
