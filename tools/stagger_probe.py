"""dev tool (round 6): do three forwards in flight run faster when their phases are STAGGERED instead of in lockstep?
The host submits the forwards back to back, so the three streams start together and stay in step (tools/step_times.py: they complete in
triples): the bandwidth-bound first launches of the three forwards overlap with each other, then their vector-bound depthwise launches.
This probe puts a one-off delay (a spin kernel) on streams 1 and 2 and measures the same 200 steps again.
    python tools/stagger_probe.py [batch] [model]"""
import sys, time
import torch
sys.path.insert(0, ".")
from demonet_amd import models, synth
from demonet_amd.pipeline import ForwardPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
name = sys.argv[2] if len(sys.argv) > 2 else "ssdlite320_mobilenet_v3_large"
dev = torch.device("cuda", 0)
m = models.load_synthetic(getattr(models, name)(num_classes=91), 0).to(dev)
W, H = m.graph.size
pipe = ForwardPipeline(m, B, depth=3, chains=1, device=dev)
batches = [torch.from_numpy(synth.images(100 + j, B, H, W)).to(dev) for j in range(3)]


def run(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        pipe.submit(batches[k % 3], persistent_input=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for k in range(30):
    pipe.submit(batches[k % 3], persistent_input=True)
# calibrate torch.cuda._sleep
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); e1.synchronize()
cyc_per_ms = 1000000 / e0.elapsed_time(e1)
print(f"lockstep: {run(300):.4f} ms per step, {run(300):.4f}")
for frac in (0.33, 0.5, 0.2):
    fwd_ms = 0.85
    for j in (1, 2):
        with torch.cuda.stream(pipe.slots[j].stream):
            torch.cuda._sleep(int(cyc_per_ms * fwd_ms * frac * j))
    a = run(300)
    b = run(300)
    print(f"stagger {frac:.2f} of a forward per slot: {a:.4f} ms per step, then {b:.4f} (does it hold?)")
