"""dev tool: the timed loop of bench.py repeated R times inside ONE process (graph replay, device-resident input): is a slow run a
property of the process (placement of the graph's branches) or of the moment (clocks, host)?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import models, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
x = torch.from_numpy(synth.images(1002, B, 320, 320)).cuda()
for _ in range(20):
    m.forward_batch(x, persistent_input=True)
torch.cuda.synchronize()
out = []
for r in range(R):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(100):
        m.forward_batch(x, persistent_input=True)
    t_submit = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    out.append((e0.elapsed_time(e1) / 100, t_submit * 10))
print(f"batch {B}: " + "  ".join(f"{a:.4f} ms (host submit {b:.3f} ms/step)" for a, b in out))

# where the host time of one forward goes: the ctypes call of dn_forward alone (hipGraphLaunch inside) vs the Python around it
import ctypes as C
from demonet_amd import _lib
L = _lib.lib()
h = m._plan(x.device)
b = m._buffers_for(B, 320, 320, x.device)
stream = torch.cuda.current_stream(x.device).cuda_stream
args = (C.c_void_p(h), C.c_void_p(x.data_ptr()), B, 320, 320, C.c_void_p(b["boxes"].data_ptr()), C.c_void_p(b["scores"].data_ptr()),
        C.c_void_p(b["labels"].data_ptr()), C.c_void_p(b["counts"].data_ptr()), C.c_void_p(b["ws"].data_ptr()), b["ws"].numel(), C.c_void_p(stream))
for _ in range(10):
    L.dn_forward(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    L.dn_forward(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"dn_forward alone: host {1e3 * (t1 - t0) / 200:.3f} ms per call, wall incl. drain {1e3 * (t2 - t0) / 200:.3f} ms per call")
