"""dev tool: where a workgroup of trunk_kernel spends its time (in-kernel s_memrealtime stamps, 100 MHz): stamp 0 start, 1 input staged,
then one stamp per chunk, per SE FC phase and per block epilogue (the first 60 events of the run)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
imgs = torch.from_numpy(synth.images(1002, n, 320, 320)).cuda()
m.forward_batch(imgs, persistent_input=True)
st = torch.zeros(64 * 64, dtype=torch.int64, device="cuda")
L.dn_debug_trunk_stamps.argtypes = [C.c_void_p]
L.dn_debug_trunk_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(imgs, persistent_input=True)
torch.cuda.synchronize()
L.dn_debug_trunk_stamps(None)
s = st.cpu().numpy().reshape(-1, 64).astype(np.float64)
s = s[s[:, 0] > 0]
print("workgroups", len(s))
w = s[0]
last = w[63]
ev = w[:60]
ev = ev[ev > 0]
d = np.diff(ev) * 0.01
print("workgroup 0: events", len(ev), " total %.1f us (kernel end stamp)" % ((last - w[0]) * 0.01))
print("deltas (us):", " ".join(f"{x:.1f}" for x in d))
print("mean over workgroups of total: %.1f us" % (((s[:, 63] - s[:, 0]) * 0.01).mean()))
