"""dev tool: is the step with forwards in flight sensitive to extra HBM traffic?  Adds a device copy of <MB> megabytes (2 x MB of traffic) behind every
forward on the forward's own stream and reports the step time.    usage: traffic_sensitivity_probe.py <batch> <MB> [<MB> ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demonet_amd import models, synth
from demonet_amd.pipeline import ForwardPipeline
B = int(sys.argv[1])
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
xs = [torch.from_numpy(synth.images(5 + j, B, 320, 320)).cuda() for j in range(3)]
with ForwardPipeline(m, B, depth=3) as pipe:
    for mb in [int(v) for v in sys.argv[2:]]:
        src = [torch.empty(max(mb, 1) << 20, dtype=torch.uint8, device="cuda") for _ in range(3)]
        dst = [torch.empty(max(mb, 1) << 20, dtype=torch.uint8, device="cuda") for _ in range(3)]
        def step(k):
            t = pipe.submit(xs[k % 3], persistent_input=True)
            if mb > 0:
                with torch.cuda.stream(pipe.stream_of(t)):
                    dst[k % 3].copy_(src[k % 3], non_blocking=True)
        for k in range(30): step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(600): step(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 600
        print(f"batch {B}: +{2 * mb} MB of copy traffic per forward: {dt * 1e3:.4f} ms per step", flush=True)
