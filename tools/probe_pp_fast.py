"""dev tool: phase timing of select_nms_fast via in-kernel s_memrealtime stamps (100 MHz)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
imgs = torch.from_numpy(synth.images(1002, 64, 320, 320)).cuda()
m.forward_batch(imgs, persistent_input=True)
st = torch.zeros(64 * 90 * 16, dtype=torch.int64, device="cuda")
L.dn_debug_pp_stamps.argtypes = [C.c_void_p]
L.dn_debug_pp_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(imgs, persistent_input=True)
torch.cuda.synchronize()
L.dn_debug_pp_stamps(None)
s = st.cpu().numpy().reshape(-1, 16)[:, :6].astype(np.float64)
d = np.diff(s, axis=1) * 0.01     # us
names = ["scan", "sort", "gather", "mask", "serial"]
busy = d[:, 1] > 0
print(f"workgroups {len(s)}, with candidates {busy.sum()}")
print("per-workgroup phase times (us): mean(all) / mean(with candidates) / max")
for i, nm in enumerate(names):
    print(f"  {nm:12s} {d[:, i].mean():8.2f} {d[busy, i].mean():8.2f} {d[:, i].max():8.2f}")
life = s[:, 5] - s[:, 0]
print("  lifetime     %.2f  (max %.2f)" % (life.mean() * 0.01, life.max() * 0.01), " span of kernel %.1f us" % ((s[:, 5].max() - s[:, 0].min()) * 0.01))
t0 = s[:, 0].min()
st0 = np.sort(s[:, 0] - t0) * 0.01
print("  start times (us) percentiles 10/50/90/100:", np.percentile(st0, [10, 50, 90, 100]).round(1))
