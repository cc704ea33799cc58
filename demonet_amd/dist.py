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
    """Gather for a steady stream of batches. Every step the forward's packed payload [B, D+1, 6] fp32 (row D carries the
    count) is copied into a slot of a device ring; every `every` steps ONE all_gather_into_tensor moves the whole window.

    Why a window: the reference gathers once, after the evaluation loop (engine.py:105 `synchronize_between_processes`);
    gathering every step is stricter than needed and costs a fixed ~45 us of collective launch per step on the compute
    stream (measured at world 1) -- more from a side stream, because a hipGraph launch does not overlap kernels of other
    streams on this runtime -- and two alternating packed buffers would mean two alternating graph executables (~35 us per
    launch). One buffer, one graph, a 0.5 MB device copy per step and one collective per window amortise all of that.
    """

    def __init__(self, batch: int, dets: int, device, group=None, every: int = 16):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.B, self.D, self.K = batch, dets, max(1, int(every))
        self.packed = torch.zeros(batch, dets + 1, 6, dtype=torch.float32, device=device)     # the forward writes here
        self.acc = torch.zeros(self.K, batch, dets + 1, 6, dtype=torch.float32, device=device)
        # gathered windows, two deep: out[w][rank][slot] = payload of step (window w, slot)
        self.out = [torch.zeros(self.world, self.K, batch, dets + 1, 6, dtype=torch.float32, device=device) for _ in range(2)]
        self.n = 0                  # next ticket (steps submitted, plus the slots a flush() of a partial window skipped)
        self.gathered = 0           # tickets covered by completed collectives (always a multiple of the window after a gather)
        self.filled = [0, 0]        # per gathered window (two deep): how many of its slots hold a submitted step
        self._ring_free = None      # event: the last collective has read the ring (submits from OTHER streams wait for it)

    def next_buffer(self):
        """The packed buffer the forward should write into (SSD.forward_batch(packed=...): the merge kernel fills it, no
        torch-side packing -- a strided torch copy of the boxes alone costs more than the whole forward)."""
        return self.packed

    def _gather_window(self, join=None):
        if join is not None:
            join()                  # forwards on several streams (pipeline.ForwardPipeline): this stream waits for all their copies
        w = (self.gathered // self.K) & 1
        self.filled[w] = self.n - self.gathered
        if self.world == 1 and not dist.is_initialized():
            self.out[w][0].copy_(self.acc)
        else:
            dist.all_gather_into_tensor(self.out[w].view(self.world, -1), self.acc.view(1, -1), group=self.group)
        self.gathered = (self.gathered // self.K + 1) * self.K
        if self.acc.is_cuda:
            self._ring_free = torch.cuda.Event()
            self._ring_free.record(torch.cuda.current_stream(self.acc.device))

    def submit(self, boxes=None, scores=None, labels=None, counts=None, src=None, join=None):
        """Call on the compute stream right after the forward; returns the step's ticket. With arguments the payload is
        packed here (CPU tests / callers without the packed output); without, `packed` was filled by the forward.
        Forwards on several streams (pipeline.ForwardPipeline): call it with the forward's stream current, src = that forward's
        packed buffer and join = pipeline.join -- the copy into the ring runs on that stream, and the collective that closes a
        window first waits for the other streams."""
        slot = self.n % self.K
        p = self.acc[slot]
        D = self.D
        if self._ring_free is not None:
            torch.cuda.current_stream(self.acc.device).wait_event(self._ring_free)
        if boxes is not None:
            p[:, :D, :4].copy_(boxes)
            p[:, :D, 4].copy_(scores)
            p[:, :D, 5].copy_(labels)
            p[:, D, 0].copy_(counts)
        else:
            p.copy_(self.packed if src is None else src)    # 0.5 MB device copy on the compute stream; frees the buffer for the next forward
        ticket = self.n
        self.n += 1
        if slot == self.K - 1:
            self._gather_window(join)
        return ticket

    def flush(self, join=None):
        """Gather the last, partial window (collective: every rank calls it after the same number of steps)."""
        if self.gathered < self.n:
            self._gather_window(join)
            # the rest of that window was never submitted: the next submit() starts a fresh window (otherwise its ticket would
            # already count as gathered and result() would hand out the stale rows of the flushed window)
            self.n = self.gathered

    def result(self, ticket: int):
        """Detections of the global batch of step `ticket`, rank-major: (packed [W*B, D, 6], counts [W*B] int32). Available
        once its window is gathered (every `every` steps or after flush()) and until two windows later."""
        if ticket >= self.gathered:
            raise RuntimeError("step %d is not gathered yet: call flush() (collectively) first" % ticket)
        if ticket < 0 or ticket < self.gathered - 2 * self.K:
            raise RuntimeError("step %d was overwritten by later windows" % ticket)
        if ticket % self.K >= self.filled[(ticket // self.K) & 1]:
            raise RuntimeError("ticket %d was never submitted (its window was flushed early)" % ticket)
        o = self.out[(ticket // self.K) & 1][:, ticket % self.K].reshape(self.world * self.B, self.D + 1, 6)
        return o[:, :self.D, :], o[:, self.D, 0].to(torch.int32)
