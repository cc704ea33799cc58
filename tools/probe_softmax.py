"""dev tool: phase timing of softmax_decode_kernel via in-kernel s_memrealtime stamps (100 MHz)."""
import ctypes as C, os, sys
os.environ["DN_PP_STAMP_SOFTMAX"] = "1"
os.environ.setdefault("DN_SPLIT", "1")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
imgs = torch.from_numpy(synth.images(1002, B, 320, 320)).cuda()
m.forward_batch(imgs, persistent_input=True)
st = torch.zeros(B * 90 * 16, dtype=torch.int64, device="cuda")
L.dn_debug_pp_stamps.argtypes = [C.c_void_p]
L.dn_debug_pp_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(imgs, persistent_input=True)
torch.cuda.synchronize()
L.dn_debug_pp_stamps(None)
raw = st.cpu().numpy().reshape(-1, 16)[:B * 51, 8:].astype(np.float64)     # softmax stamps live in slots 8..15
s = raw[:, :5]
s = s[s[:, 0] > 0]
d = np.diff(s, axis=1) * 0.01
print(f"softmax_decode: {len(s)} workgroups; per-workgroup us: load {d[:,0].mean():.2f} softmax {d[:,1].mean():.2f} write {d[:,2].mean():.2f} "
      f"flush+decode {d[:,3].mean():.2f}; life {d.sum(1).mean():.2f} max {d.sum(1).max():.2f}; span {(s[:,4].max()-s[:,0].min())*0.01:.1f} us")
print("max per phase:", d.max(0).round(2), " 99th pct:", np.percentile(d, 99, axis=0).round(2))
slow = np.argsort(-d.sum(1))[:8]
print("slowest WGs (index, phases):", [(int(i), d[i].round(1).tolist()) for i in slow])
t0 = s[:, 0].min()
print("start times percentiles 10/50/90/100 (us):", np.percentile((s[:, 0] - t0) * 0.01, [10, 50, 90, 100]).round(1))
