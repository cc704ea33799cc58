"""dev tool: per-workgroup phase times of conv_patch_kernel (s_memrealtime stamps, 100 MHz) for every patch launch of one eager VGG forward.
usage (stamped build): DEMONET_HIP_LIB=demonet_amd/lib/libdemonet_hip_stamps.so python tools/probe_patch.py [model] [batch]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib, models, synth
L = _lib.lib()
L.dn_debug_patch_stamps.argtypes = [C.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else "ssd512_vgg16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = models.load_synthetic(getattr(models, name)(num_classes=91), 0).cuda()
m.set_graph_mode(False)
size = m.graph.image_size if hasattr(m.graph, "image_size") else int(name[3:6])
imgs = torch.from_numpy(synth.images(1002, B, size, size)).cuda()
m.forward_batch(imgs, persistent_input=True)
# the launches overwrite each other's slots: the LAST patch launch of the forward survives unless DN_PROBE_PATCH_OP limits the model -- so run
# the layers one by one through the stand-alone knobs: stamps of the biggest grid come first in memory, smaller grids overwrite a prefix
NW = 80 * 1024 * 8
st = torch.zeros(NW, dtype=torch.int64, device="cuda")
L.dn_debug_patch_stamps(C.c_void_p(st.data_ptr()))
m.forward_batch(imgs, persistent_input=True)
torch.cuda.synchronize()
L.dn_debug_patch_stamps(None)
s = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
s = s[s[:, 0] > 0]
# group by launch: stamps of one launch are within its span; split where start times jump by > 20 us between sorted groups
order = np.argsort(s[:, 0])
s = s[order]
gaps = np.where(np.diff(s[:, 0]) > 2000)[0]
bounds = [0] + list(gaps + 1) + [len(s)]
for a, b in zip(bounds[:-1], bounds[1:]):
    g = s[a:b]
    if len(g) < 64:
        continue
    if len(g) <= 256:             # conv_patch_resident_kernel: 0 start | 1 weights + first patch in | 2 end | 3, 4, 5 its third block: start, MFMA steps issued, behind the wait + barrier
        life = (g[:, 2] - g[:, 0]) * 0.01
        ph = np.diff(g[:, 3:6], axis=1) * 0.01
        print("resident launch, %d workgroups, span %.1f us: prologue %.2f us, life %.1f (max %.1f); third block: 36 steps (+ next patch requests + previous block's stores) %.2f | wait + barrier %.2f us"
              % (len(g), (g[:, 2].max() - g[:, 0].min()) * 0.01, ((g[:, 1] - g[:, 0]) * 0.01).mean(), life.mean(), life.max(), *ph.mean(0)))
        continue
    d = np.diff(g[:, :5], axis=1) * 0.01
    span = (g[:, 4].max() - g[:, 0].min()) * 0.01
    life = (g[:, 4] - g[:, 0]) * 0.01
    print("launch with %6d surviving workgroups, span %8.1f us: stage %.2f | taps %.2f | epilogue->LDS %.2f | store %.2f | life %.2f us (p95 %.2f); resident workgroups per CU %.2f"
          % (len(g), span, d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), life.mean(), np.percentile(life, 95), life.sum() / span / 256))
