"""dev tool: where the cost of the multi-GPU mode comes from at world size 1 (bench.py with DN_BENCH_FORCE_DIST=1 against the plain run).
    usage: gather_cost_probe.py <mode>      [GP_BATCH=32] [GP_EXTRA="--no-latency"]
    plain            no torch.distributed at all
    dist             bench.py's distributed mode (RCCL communicator, packed payload, windowed gather)
    nocopy / nowait / nogather / nojoin / alloff / alloff_noctx     the gatherer with one (or every) part of its work removed
    plain_packed     plain, the merge kernel also writes the packed payload
    plain_gloo / plain_initonly      process group initialised (gloo / nccl), no collective
    plain_distinit / plain_barrier   nccl group + one collective: a LIVE RCCL communicator, nothing else
    plain_destroy    the same, then destroy_process_group() before the timed loop
Round 3 (batch 32, three forwards in flight): plain 0.397 ms, dist 0.417; every gatherer part removed 0.412 - 0.415; plain_distinit 0.410 - 0.416;
plain_destroy / plain_gloo / plain_initonly 0.393 - 0.395 -> the cost is the live communicator, not the gatherer."""
import os, sys, json, io, contextlib
sys.path.insert(0, os.getcwd())
import torch
import demonet_amd.dist as d
mode = sys.argv[1]
G = d.DetectionGatherer
orig_submit, orig_gw = G.submit, G._gather_window
if mode == "nocopy":
    def submit(self, boxes=None, scores=None, labels=None, counts=None, src=None, join=None):
        slot = self.n % self.K
        if self._ring_free is not None:
            torch.cuda.current_stream(self.acc.device).wait_event(self._ring_free)
        t = self.n; self.n += 1
        if slot == self.K - 1: self._gather_window(join)
        return t
    G.submit = submit
elif mode == "nowait":
    def submit(self, boxes=None, scores=None, labels=None, counts=None, src=None, join=None):
        slot = self.n % self.K
        self.acc[slot].copy_(self.packed if src is None else src)
        t = self.n; self.n += 1
        if slot == self.K - 1: self._gather_window(join)
        return t
    G.submit = submit
elif mode == "nogather":
    def gw(self, join=None):
        w = (self.gathered // self.K) & 1
        self.filled[w] = self.n - self.gathered
        self.gathered = (self.gathered // self.K + 1) * self.K
    G._gather_window = gw
elif mode == "alloff":
    def submit(self, boxes=None, scores=None, labels=None, counts=None, src=None, join=None):
        t = self.n; self.n += 1
        if self.n % self.K == 0:
            self.gathered = self.n
        return t
    G.submit = submit
    G.flush = lambda self, join=None: None
    G.result = lambda self, t: (self.out[0][:, 0].reshape(self.world * self.B, self.D + 1, 6)[:, :self.D, :], self.out[0][:, 0].reshape(self.world * self.B, self.D + 1, 6)[:, self.D, 0].to(torch.int32))
elif mode == "nojoin":
    def gw(self, join=None):
        return orig_gw(self, None)
    G._gather_window = gw
import bench
if mode == "alloff_noctx":
    import contextlib
    bench._on = lambda stream: contextlib.nullcontext()
    mode = "alloff"
    G.submit = lambda self, **kw: (setattr(self, "n", self.n + 1), setattr(self, "gathered", self.n - self.n % self.K), self.n - 1)[2]
    G.flush = lambda self, join=None: None
    G.result = lambda self, t: (self.out[0][:, 0].reshape(self.world * self.B, self.D + 1, 6)[:, :self.D, :], self.out[0][:, 0].reshape(self.world * self.B, self.D + 1, 6)[:, self.D, 0].to(torch.int32))
if mode == "plain_packed":
    import demonet_amd.pipeline as pp
    oi = pp.ForwardPipeline.__init__
    def init(self, *a, **k):
        k["packed"] = True
        oi(self, *a, **k)
    pp.ForwardPipeline.__init__ = init
    mode = "plain"
if mode == "plain_distinit":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.zeros(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
    mode = "plain"
if mode in ("plain_gloo", "plain_destroy", "plain_initonly", "plain_barrier"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group("gloo" if mode == "plain_gloo" else "nccl", rank=0, world_size=1)
    if mode == "plain_destroy":
        x = torch.zeros(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
        dist.destroy_process_group()
    if mode == "plain_barrier":
        x = torch.zeros(4, device="cuda"); dist.all_gather_into_tensor(torch.zeros(4, device="cuda"), x); torch.cuda.synchronize()
    mode = "plain"
if mode != "plain":
    os.environ["DN_BENCH_FORCE_DIST"] = "1"
r = bench.main(["--batch", os.environ.get("GP_BATCH", "32"), "--steps", "800", "--no-cpu-baseline", "--no-roofline"] + os.environ.get("GP_EXTRA", "--no-latency").split())
