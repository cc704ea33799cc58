import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from demonet_amd import models, synth
from demonet_amd.dist import DetectionGatherer
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29512"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91),0).to(dev)
imgs = torch.from_numpy(synth.images(1002,64,320,320)).to(dev)
G = DetectionGatherer(64, 300, dev)
def run(mode, steps=30):
    for _ in range(5): m.forward_batch(imgs, persistent_input=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(steps):
        b,s,l,c = m.forward_batch(imgs, persistent_input=True)
        if mode>=1:
            p=G.packed[0]; p[:, :300, :4].copy_(b); p[:, :300, 4].copy_(s); p[:, :300, 5].copy_(l); p[:, 300, 0].copy_(c)
        if mode==2:
            dist.all_gather_into_tensor(G.out[0], G.packed[0])
        if mode==3:
            G.submit(b,s,l,c)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/steps*1e3
for mode in (0,1,2,3,0,1,2,3,0,1,2,3,0):
    print("mode",mode, "%.3f ms/step"%run(mode))
dist.destroy_process_group()
