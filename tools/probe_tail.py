"""dev tool: per-op timing inside the tail kernel (s_memrealtime stamps, 100 MHz)."""
import ctypes as C, os, sys
os.environ.setdefault("DN_SPLIT", "1")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
L.dn_debug_tail_stamps.argtypes = [C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
imgs = torch.from_numpy(synth.images(1002, B, 320, 320)).cuda()
m.forward_batch(imgs, persistent_input=True)
st = torch.zeros(B * 32, dtype=torch.int64, device="cuda")
L.dn_debug_tail_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(imgs, persistent_input=True)
torch.cuda.synchronize()
L.dn_debug_tail_stamps(None)
s = st.cpu().numpy().reshape(B, 32).astype(np.float64)
k = int((s[0] > 0).sum())
d = np.diff(s[:, :k], axis=1) * 0.01
print("tail kernel, per-workgroup us: stage+warm %.2f | ops: %s | life %.2f (max %.2f)" % (
    d[:, 0].mean(), " ".join("%.2f" % v for v in d[:, 1:].mean(0)), d.sum(1).mean(), d.sum(1).max()))
