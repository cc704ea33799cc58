import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demonet_amd
from demonet_amd import models
n = int(sys.argv[1]); graph = int(sys.argv[2])
m = models.ssdlite320_mobilenet_v3_large(num_classes=91).eval().cuda()
m._graph_mode = graph
x = torch.rand(n, 3, 320, 320, device="cuda")
for it in range(3):
    out = m.forward_batch(x, persistent_input=True)
    torch.cuda.synchronize()
    print("iter", it, "ok", float(out[1].sum()), flush=True)
