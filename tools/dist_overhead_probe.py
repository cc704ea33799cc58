"""dev tool (run under torchrun, world 1): where does the distributed step's overhead come from?"""
import os, sys, time
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import models, synth
from demonet_amd.dist import DetectionGatherer
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).cuda()
imgs = torch.from_numpy(synth.images(1002, 64, 320, 320)).cuda()
g = DetectionGatherer(64, 300, "cuda:0")

def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def plain(): m.forward_batch(imgs, persistent_input=True)
pk = [torch.zeros(64, 301, 6, device="cuda") for _ in range(2)]
state = {"i": 0}
def packed_only():
    state["i"] ^= 1
    m.forward_batch(imgs, persistent_input=True, packed=pk[state["i"]])
def full():
    m.forward_batch(imgs, persistent_input=True, packed=g.next_buffer()); g.submit()
out = torch.zeros(64, 301, 6, device="cuda")
def packed_same():
    m.forward_batch(imgs, persistent_input=True, packed=pk[0])
imgs2 = imgs.clone()
def plain_alternating():
    state["i"] ^= 1
    m.forward_batch(imgs if state["i"] else imgs2, persistent_input=True)
def gather_same_stream():
    state["i"] ^= 1
    m.forward_batch(imgs, persistent_input=True, packed=pk[state["i"]])
    dist.all_gather_into_tensor(out, pk[state["i"]])
print("packed, one buffer %.4f | plain, two alternating inputs (two graphs) %.4f" % (timeit(packed_same), timeit(plain_alternating)))
print("plain %.4f ms | packed output only %.4f | gatherer (side stream) %.4f | all_gather on the compute stream %.4f" % (
    timeit(plain), timeit(packed_only), timeit(full), timeit(gather_same_stream)))
dist.destroy_process_group()
