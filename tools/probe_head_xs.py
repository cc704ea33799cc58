"""dev tool: per-workgroup phase stamps (s_memrealtime, 100 MHz) of head_xs_kernel inside a real forward.
stamps: 0 start, 1 strip staged, 2 split-pair tasks done (wave 0), 3 wave 0's pair done, 4 first output pass written, 5 second"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DN_PW_GROUP_STAMPS"] = "0"
from demonet_amd import _lib, models, synth
raw = C.CDLL(_lib.LIB_PATH)
raw.dn_debug_pw_stamps.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
x = torch.from_numpy(synth.images(3, n, 320, 320)).cuda()
for _ in range(3):
    m.forward_heads(x)
torch.cuda.synchronize()
st = torch.zeros(8 * 60000, dtype=torch.int64, device="cuda")
raw.dn_debug_pw_stamps(C.c_void_p(st.data_ptr()))
m.forward_heads(x)
torch.cuda.synchronize()
raw.dn_debug_pw_stamps(None)
s = st.cpu().numpy().reshape(-1, 8)[:512, :6].astype(np.float64)
s = s[(s[:, 0] > 0) & (s[:, 5] > 0)]
d = np.diff(s, axis=1) * 0.01
t0 = s[:, 0].min()
print(f"batch {n}: {len(s)} workgroups; stage {d[:,0].mean():.2f} split-pair {d[:,1].mean():.2f} pair {d[:,2].mean():.2f} out0 {d[:,3].mean():.2f} out1 {d[:,4].mean():.2f} | life {d.sum(1).mean():.2f} "
      f"(max {d.sum(1).max():.2f}) span {(s[:,5].max() - t0) * 0.01:.1f} us; start spread {(s[:,0].max() - t0) * 0.01:.1f}")
