"""dev: completion time of every step of a short in-flight run (events on the forwards' own streams): where does a 20-step run lose against a 200-step run?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import models, synth
from demonet_amd.pipeline import ForwardPipeline
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, R = 64, 3
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
pipe = ForwardPipeline(m, B, depth=R, chains=1, device=torch.device("cuda:0"))
batches = [torch.from_numpy(synth.images(3002 + j, B, 320, 320)).cuda() for j in range(R)]
slot_streams = {}
for k in range(W):
    t = pipe.submit(batches[k % R], persistent_input=True)
    slot_streams[k % R] = pipe.stream_of(t)
torch.cuda.synchronize()
STAG = float(os.environ.get("STAGGER_MS", "0"))          # experiment: shift the second / third slot's stream by one / two thirds of a forward once
ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
evs = []
t0 = time.perf_counter()
sub = []
for k in range(K):
    if STAG > 0 and 1 <= k < R:
        # the slot's stream is only known after a submit: use the stream of the slot's warm-up forward (slots rotate, k % R)
        with torch.cuda.stream(slot_streams[k % R]):
            torch.cuda._sleep(int(STAG * k * 2.1e6))       # ~2.1 GHz shader clock: cycles per ms
    t = pipe.submit(batches[k % R], persistent_input=True)
    sub.append(time.perf_counter() - t0)
    e = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(pipe.stream_of(t)):
        e.record()
    evs.append(e)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
done = np.array([ev0.elapsed_time(e) for e in evs])
print("K=%d: wall %.3f ms = %.4f ms/step" % (K, dt * 1e3, dt * 1e3 / K))
print("completion (ms):", " ".join("%.2f" % v for v in done))
print("between completions (ms):", " ".join("%.2f" % v for v in np.diff(done)))
print("submit returned at (ms):", " ".join("%.2f" % (v * 1e3) for v in sub))
