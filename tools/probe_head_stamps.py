"""dev tool: per-workgroup phase timing (s_memrealtime, 100 MHz) of the grouped head 1x1 launch inside a real forward: the head group is
the last tiled pointwise launch of the chain, so its stamps are what is left in the buffer (workgroups 0.. = the level-0 class head)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DN_SPLIT", "1")
from demonet_amd import _lib, models, synth
L = _lib.lib()
L.dn_debug_pw_stamps.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False)
x = torch.from_numpy(synth.images(3, n, 320, 320)).cuda()
for _ in range(3):
    m.forward_heads(x)
torch.cuda.synchronize()
st = torch.zeros(8 * 60000, dtype=torch.int64, device="cuda")
L.dn_debug_pw_stamps(C.c_void_p(st.data_ptr()))
m.forward_heads(x)
torch.cuda.synchronize()
L.dn_debug_pw_stamps(None)
s = st.cpu().numpy().reshape(-1, 8)[:, :4].astype(np.float64)
nwg = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
s = s[:nwg]
s = s[s[:, 0] > 0]
d = np.diff(s, axis=1) * 0.01
t0 = s[:, 0].min()
print(f"batch {n}: {len(s)} workgroups; prologue {d[:,0].mean():.2f} kloop {d[:,1].mean():.2f} epilogue {d[:,2].mean():.2f} us; WG life {d.sum(1).mean():.2f} (max {d.sum(1).max():.2f});"
      f" span {(s[:,3].max()-t0)*0.01:.1f} us; start spread {(s[:,0].max()-t0)*0.01:.1f} us")
starts = np.sort((s[:, 0] - t0) * 0.01)
print("start time percentiles (us):", " ".join(f"{np.percentile(starts, q):.1f}" for q in (0, 10, 25, 50, 75, 90, 100)))
# timeline by workgroup index (blocks of 100): when they start and end relative to the launch
allw = st.cpu().numpy().reshape(-1, 8)[:, :4].astype(np.float64)
valid = allw[:, 0] > 0
idx = np.nonzero(valid & (np.abs(allw[:, 0] - t0) < 1e5))[0]
for lo in range(0, int(idx.max()) + 1, 100):
    sel = idx[(idx >= lo) & (idx < lo + 100)]
    if len(sel) == 0:
        continue
    b = allw[sel]
    print(f"  WGs {lo:5d}..{lo + 99:5d} ({len(sel):3d}): start {(b[:,0].min()-t0)*0.01:6.1f}..{(b[:,0].max()-t0)*0.01:6.1f}  end {(b[:,3].min()-t0)*0.01:6.1f}..{(b[:,3].max()-t0)*0.01:6.1f}  life {((b[:,3]-b[:,0]).mean())*0.01:5.1f} us")
