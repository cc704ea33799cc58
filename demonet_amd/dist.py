"""Multi-GPU: images shard across ranks (one process per GPU); the only collective is the gather of detections.

reference analogue: util/misc.py:75-115 `all_gather` of pickled per-image results used by coco_eval.merge
(data/coco_eval.py:167-184). Here the payload has a fixed shape -- [B_local, D, 6] fp32 (x1,y1,x2,y2,score,label)
plus [B_local] int32 counts -- so it is one all_gather_into_tensor per buffer over RCCL/xGMI (a few hundred KB).
"""
import torch
import torch.distributed as dist


def pack_detections(boxes: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """[B,D,4], [B,D], [B,D] int64 -> [B,D,6] fp32 (labels < 2^24 are exact in fp32)."""
    return torch.cat([boxes, scores.unsqueeze(-1), labels.to(torch.float32).unsqueeze(-1)], dim=-1)


def unpack_detections(packed: torch.Tensor, counts: torch.Tensor):
    out = []
    for p, c in zip(packed, counts.tolist()):
        out.append({"boxes": p[:c, :4], "scores": p[:c, 4], "labels": p[:c, 5].to(torch.int64)})
    return out


def gather_detections(packed: torch.Tensor, counts: torch.Tensor, group=None):
    """All ranks receive the detections of the global batch, rank-major (rank r holds images [r*B, (r+1)*B))."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return packed, counts
    ws = dist.get_world_size(group)
    gp = torch.empty((ws * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    gc = torch.empty((ws * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(gp, packed.contiguous(), group=group)
    dist.all_gather_into_tensor(gc, counts.contiguous(), group=group)
    return gp, gc


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous image shard of a rank (256 images -> 32 per GPU)."""
    per = (global_batch + world - 1) // world
    lo = min(global_batch, rank * per)
    return lo, min(global_batch, lo + per)


class DetectionGatherer:
    """Overlapped gather for a steady stream of batches: one packed [B, D+1, 6] fp32 buffer per step (row D carries the
    count), ONE all_gather_into_tensor per step on a side stream, double-buffered so the collective of step i runs under
    the compute of step i+1 (payload is KBs, only latency matters -> hide it)."""

    def __init__(self, batch: int, dets: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.B, self.D = batch, dets
        self.stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self.packed = [torch.zeros(batch, dets + 1, 6, dtype=torch.float32, device=device) for _ in range(2)]
        self.out = [torch.zeros(self.world * batch, dets + 1, 6, dtype=torch.float32, device=device) for _ in range(2)]
        self.work = [None, None]
        self.i = 0

    def next_buffer(self):
        """The packed buffer the NEXT forward should write into (SSD.forward_batch(packed=...): the merge kernel fills it,
        no torch-side packing -- a strided torch copy of the boxes alone costs more than the whole forward)."""
        i = self.i
        if self.work[i] is not None:
            self.work[i].wait()                       # the collective that last used this buffer pair
            self.work[i] = None
        return self.packed[i]

    def submit(self, boxes=None, scores=None, labels=None, counts=None):
        """Call on the compute stream right after the forward; returns immediately. With arguments the payload is packed
        here (CPU tests / callers without the packed output); without, packed[i] was filled by the forward."""
        i = self.i
        self.i ^= 1
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        p = self.packed[i]
        D = self.D
        if boxes is not None:
            p[:, :D, :4].copy_(boxes)
            p[:, :D, 4].copy_(scores)
            p[:, :D, 5].copy_(labels)
            p[:, D, 0].copy_(counts)
        if self.world == 1 and not dist.is_initialized():
            self.out[i].copy_(p)
            return i
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self.work[i] = dist.all_gather_into_tensor(self.out[i], p, group=self.group, async_op=True)
        else:
            self.work[i] = dist.all_gather_into_tensor(self.out[i], p, group=self.group, async_op=True)
        return i

    def result(self, i):
        """Detections of the global batch of submission i: (packed [W*B, D, 6], counts [W*B] int32)."""
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        o = self.out[i]
        return o[:, :self.D, :], o[:, self.D, 0].to(torch.int32)

    def flush(self):
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
