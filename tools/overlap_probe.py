"""dev tool: does splitting the batch into two half-batches on two streams beat one full batch? (latency-bound launch chains)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import models, synth

def mk():
    return models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
imgs = torch.from_numpy(synth.images(1002, B, 320, 320)).cuda()
full = mk()
parts = [mk() for _ in range(S)]
if os.environ.get('EAGER'):
    for m in parts + [full]:
        m.set_graph_mode(False)
chunks = [c.contiguous() for c in imgs.chunk(S)]
streams = [torch.cuda.Stream() for _ in range(S)]

def run_full():
    full.forward_batch(imgs, persistent_input=True)

def run_split():
    ev = torch.cuda.Event(); ev.record()
    for m, c, s in zip(parts, chunks, streams):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            m.forward_batch(c, persistent_input=True)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)

for name, fn in (("full", run_full), ("split", run_split), ("full", run_full), ("split", run_split)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{name:6s} B={B} S={S}: {dt*1e3:.3f} ms/step  {B/dt:.0f} img/s", flush=True)

if os.environ.get('EAGER'):
    # one torch-captured graph holding both half-batch chains as parallel branches
    for fn, name in ((run_split, "graph(split branches)"), (run_full, "graph(full)")):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                fn()
        torch.cuda.synchronize()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"{name}: B={B} S={S}: {dt*1e3:.3f} ms/step  {B/dt:.0f} img/s", flush=True)
