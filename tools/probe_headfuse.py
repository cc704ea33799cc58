"""dev tool: per-workgroup phase cycle sums of head_fused_kernel (headfuse.hip built with -DDN_DEV_STAMPS).

    python -m demonet_amd.build --stamps && DEMONET_HIP_LIB=demonet_amd/lib/libdemonet_hip_stamps.so python tools/probe_headfuse.py [batch]

Prints, per pyramid level (workgroups grouped by their K chunk count), the mean shader-clock cycles a workgroup spends in: prologue, depthwise
phases (sum over chunks), the wait + barrier behind them, matrix phases, the whole loop, the epilogue.
"""
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
raw.dn_debug_hf_stamps.argtypes = [C.c_void_p]
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
m.set_graph_mode(False) if hasattr(m, "set_graph_mode") else None
os.environ["DN_GRAPH"] = "0"
imgs = torch.from_numpy(synth.images(5, n, 320, 320)).cuda()
fwd = (lambda: m.forward_batch(imgs)) if os.environ.get('PROBE_SM') else (lambda: m.forward_heads(imgs))
fwd()
if os.environ.get("PROBE_CHAINS"):       # e.g. 1: the whole batch as ONE chain (what a forward in flight runs)
    _lib.check(L.dn_set_chains(C.c_void_p(m._handle), int(os.environ["PROBE_CHAINS"])))
for _ in range(3):
    fwd()
torch.cuda.synchronize()
st = torch.zeros(16 * 4096, dtype=torch.int64, device="cuda")
raw.dn_debug_hf_stamps(C.c_void_p(st.data_ptr()))
fwd()
torch.cuda.synchronize()
raw.dn_debug_hf_stamps(None)
s = st.cpu().numpy().reshape(-1, 16)
s = s[s[:, 6] > 0]
print(f"batch {n}: {len(s)} workgroups stamped; kernel span by s_memrealtime {(s[:, 7].max() - s[:, 7].min()) * 0.01:.1f} us (last - first END)")
for nch in sorted(set(s[:, 6]), reverse=True):
    q = s[s[:, 6] == nch].astype(np.float64)
    f = lambda k: q[:, k].mean()
    print(f"  K chunks {int(nch):3d}: {len(q):4d} WGs | prologue {f(0):8.0f} | dw {f(1):8.0f} ({f(1) / nch:6.0f}/chunk) | wait+barrier {f(2):8.0f} ({f(2) / nch:6.0f}) | "
          f"mfma {f(3):8.0f} ({f(3) / nch:6.0f}) | loop {f(4):8.0f} | epilogue {f(5):8.0f} cycles; max loop {q[:, 4].max():.0f}"
          + (f" | SM epilogue: write+tables {f(11):7.0f} softmax {f(12):7.0f} scores {f(13):7.0f}" if q[:, 13].max() > 0 else ""))

# residency: workgroups alive at the same time on one compute unit (HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13; XCC_ID bits 3:0)
cu = ((s[:, 10] & 0xf) << 8) | (((s[:, 9] >> 13) & 0x7) << 5) | (((s[:, 9] >> 12) & 1) << 4) | ((s[:, 9] >> 8) & 0xf)
peak = {}
for c in np.unique(cu):
    q = s[cu == c]
    ev = sorted([(t, 1) for t in q[:, 8]] + [(t, -1) for t in q[:, 7]])
    cur = mx = 0
    for _, d in ev:
        cur += d
        mx = max(mx, cur)
    peak[int(c)] = (mx, len(q))
pk = np.array([v[0] for v in peak.values()])
print(f"  compute units used {len(peak)}; workgroups alive at once per CU: max {pk.max()}, mean {pk.mean():.2f}; workgroups per CU: max {max(v[1] for v in peak.values())}")
print(f"  kernel span first START - last END {(s[:, 7].max() - s[:, 8].min()) * 0.01:.1f} us; mean workgroup life {((s[:, 7] - s[:, 8]) * 0.01).mean():.1f} us")

# timeline: workgroups alive per 5-us bin
t0 = s[:, 8].min()
bins = np.arange(0, (s[:, 7].max() - t0) * 0.01 + 5, 5.0)
alive = [int(((s[:, 8] - t0) * 0.01 <= b + 2.5).sum() - ((s[:, 7] - t0) * 0.01 <= b + 2.5).sum()) for b in bins]
print("  alive at t =", " ".join(f"{int(b)}:{a}" for b, a in zip(bins, alive)))
