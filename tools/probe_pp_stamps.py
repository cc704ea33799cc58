"""dev tool: per-workgroup phase timing (s_memrealtime, 100 MHz) of select_nms_fast_kernel inside a real forward: stamps 0..5 =
start, column scan done, sort done, boxes staged, IoU mask done, greedy sweep done.
    usage: probe_pp_stamps.py [batch]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
raw.dn_debug_pp_stamps.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
x = torch.from_numpy(synth.images(3, n, 320, 320)).cuda()
for _ in range(3):
    m.forward_batch(x)
torch.cuda.synchronize()
nwg = 90 * ((n + 7) // 8 * 8)
st = torch.zeros(16 * nwg, dtype=torch.int64, device="cuda")
raw.dn_debug_pp_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(x)
torch.cuda.synchronize()
raw.dn_debug_pp_stamps(None)
s = st.cpu().numpy().reshape(-1, 16)[:, :6].astype(np.float64)
s = s[(s[:, 0] > 0) & (s[:, 5] > 0)]
t0 = s[:, 0].min()
d = np.diff(s, axis=1) * 0.01
life = (s[:, 5] - s[:, 0]) * 0.01
print(f"batch {n}: {len(s)} workgroups; span {(s[:,5].max()-t0)*0.01:.1f} us; start spread {(s[:,0].max()-t0)*0.01:.1f} us; life mean {life.mean():.2f} max {life.max():.2f}")
print("phase means (scan sort stage mask sweep):", " ".join(f"{v:.2f}" for v in d.mean(0)))
o = np.argsort(-life)[:12]
for i in o:
    print(f"   wg life {life[i]:6.2f} start {(s[i,0]-t0)*0.01:6.1f} end {(s[i,5]-t0)*0.01:6.1f}  phases", " ".join(f"{v:6.2f}" for v in d[i]))
ends = np.sort((s[:, 5] - t0) * 0.01)
print("end time percentiles (us):", " ".join(f"{np.percentile(ends, q):.1f}" for q in (10, 25, 50, 75, 90, 99, 100)))
starts = np.sort((s[:, 0] - t0) * 0.01)
print("start time percentiles (us):", " ".join(f"{np.percentile(starts, q):.1f}" for q in (10, 25, 50, 75, 90, 99, 100)))
