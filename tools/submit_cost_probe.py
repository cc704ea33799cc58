"""dev tool: host enqueue time per pipelined submit against the step time, with and without a live RCCL communicator.
    usage: submit_cost_probe.py plain|rccl <batch>"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from demonet_amd import models, synth
from demonet_amd.pipeline import ForwardPipeline
mode = sys.argv[1]; B = int(sys.argv[2])
if mode == "rccl":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.zeros(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
xs = [torch.from_numpy(synth.images(5 + j, B, 320, 320)).cuda() for j in range(3)]
with ForwardPipeline(m, B, depth=3) as pipe:
    for k in range(30): pipe.submit(xs[k % 3], persistent_input=True)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(600): pipe.submit(xs[k % 3], persistent_input=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mode} batch {B}: host enqueue {1e3*(t1-t0)/600:.4f} ms/step, total {1e3*(t2-t0)/600:.4f} ms/step", flush=True)
