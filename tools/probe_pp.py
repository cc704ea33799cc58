"""dev tool: phase timing of select_nms via in-kernel s_memrealtime stamps (100 MHz)."""
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
s = st.cpu().numpy().reshape(-1, 16)[:, :8].astype(np.float64)
d = np.diff(s, axis=1) * 0.01     # us
names = ["load+count", "radix", "compact", "sort", "gather", "mask", "serial"]
print("per-workgroup phase times (us): mean / max")
for i, nm in enumerate(names):
    print(f"  {nm:12s} {d[:, i].mean():8.2f} {d[:, i].max():8.2f}")
print("  total        %.2f" % d.sum(1).mean(), " span of kernel %.1f us" % ((s[:, 7].max() - s[:, 0].min()) * 0.01))
