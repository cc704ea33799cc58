"""dev tool: phase stamps of coop_kernel (coop.hip built with -DDN_DEV_STAMPS).
    python -m demonet_amd.build --stamps && DEMONET_HIP_LIB=demonet_amd/lib/libdemonet_hip_stamps.so python tools/probe_coop.py [batch]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth      # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
raw.dn_debug_coop_stamps.argtypes = [C.c_void_p]
os.environ["DN_GRAPH"] = "0"
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
imgs = torch.from_numpy(synth.images(5, n, 320, 320)).cuda()
m.forward_heads(imgs)
_lib.check(L.dn_set_chains(C.c_void_p(m._handle), 1))
for _ in range(3):
    m.forward_heads(imgs)
torch.cuda.synchronize()
st = torch.zeros(32 * 2048, dtype=torch.int64, device="cuda")
raw.dn_debug_coop_stamps(C.c_void_p(st.data_ptr()))
m.forward_heads(imgs)
torch.cuda.synchronize()
raw.dn_debug_coop_stamps(None)
s = st.cpu().numpy().reshape(-1, 32).astype(np.float64)
s = s[s[:, 0] > 0]
names = {1: "stage d0", 2: "project0", 3: "barrier", 4: "reduce", 5: "expand", 6: "depthwise", 7: "means", 8: "barrier", 9: "fc1", 10: "fc2+scale", 11: "project", 12: "barrier", 13: "reduce",
         15: "expand", 16: "depthwise", 17: "means", 18: "barrier", 19: "fc1", 20: "fc2+scale", 21: "project", 22: "barrier", 23: "reduce", 30: "last expand"}
print(f"batch {n}: {len(s)} workgroups; span {(s[:, 30].max() - s[:, 0].min()) * 0.01:.1f} us; mean life {((s[:, 30] - s[:, 0]) * 0.01).mean():.1f} us")
prev = 0
for k in sorted(names):
    d = (s[:, k] - s[:, prev]) * 0.01
    print(f"  {names[k]:12s} mean {d.mean():6.2f} us  max {d.max():6.2f}")
    prev = k
