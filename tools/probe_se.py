"""dev tool: phase timing of se_fc_kernel (s_memrealtime stamps, 100 MHz) on the SE shapes of the model, cold weights."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
L.dn_debug_se_stamps.argtypes = [C.c_void_p]
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
os.environ.setdefault("DN_SPLIT", "1")
imgs = torch.from_numpy(synth.images(1002, B, 320, 320)).cuda()
m.forward_batch(imgs, persistent_input=True)
# every SE launch overwrites the stamps: run the forward with profiling off and read after each? simpler: the LAST SE launch
# (480 @ 10x10) is what remains; to see the others we replay the forward with the hook and stop early via heads-only... keep simple:
st = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
L.dn_debug_se_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(imgs, persistent_input=True)
torch.cuda.synchronize()
L.dn_debug_se_stamps(None)
s = st.cpu().numpy().reshape(-1, 8)[:, :4].astype(np.float64)
d = np.diff(s, axis=1) * 0.01
print("last SE launch (c=480, sq=120, 10x10), per-workgroup us: reduce %.2f fc1 %.2f fc2 %.2f ; life %.2f ; span %.2f" % (
    d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d.sum(1).mean(), (s[:, 3].max() - s[:, 0].min()) * 0.01))
