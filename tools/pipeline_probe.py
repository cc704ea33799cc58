"""Throughput with several forwards in flight: R model replicas (one plan, workspace and output set each), step k runs on replica k % R on
that replica's own stream. tools/pipeline_probe.py [--batch 64] [--replicas 1 2 3] [--model ...]"""
import argparse, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demonet_amd import models, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--replicas", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--model", default="ssdlite320_mobilenet_v3_large")
args = ap.parse_args()
dev = torch.device("cuda:0")
ncls = 21 if args.model == "ssd_lite_mobilenet_v2" else 91
reps = [models.load_synthetic(getattr(models, args.model)(num_classes=ncls), 0).to(dev) for _ in range(max(args.replicas))]
W, H = reps[0].graph.size
images = [torch.from_numpy(synth.images(1002 + i, args.batch, H, W)).to(dev) for i in range(len(reps))]
streams = [torch.cuda.Stream(dev) for _ in reps]
for R in args.replicas:
    def step(k):
        with torch.cuda.stream(streams[k % R]):
            reps[k % R].forward_batch(images[k % R], persistent_input=True)
    for k in range(20 * R):
        step(k)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print(f"{args.model} batch {args.batch} replicas {R}: {args.batch * args.steps / dt:9.1f} img/s  {dt / args.steps * 1e3:.4f} ms/step", flush=True)
