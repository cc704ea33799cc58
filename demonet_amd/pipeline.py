"""Several forwards of one model in flight: the serving form of `SSD.forward_batch`.

A single forward is a chain of ~50 dependent launches whose phases load the chip very differently (full-chip backbone layers,
then a post-process of one workgroup per image), and nothing of the NEXT batch can start before the last kernel of this one has
finished. `SSD.forward_batch` fights that inside one forward (two half-size sub-batch chains, plan.hip `batch_split`); a caller
with a stream of batches does better by keeping `depth` independent forwards in flight, each a single whole-batch chain on a
stream of its own with its own workspace and output buffers, all replaying graphs of the SAME plan (one copy of the weights).
Measured on MI355X, ssdlite320_mobilenet_v3_large (bench.py --inflight R, tools/pipeline_probe.py): batch 64 one forward at a time
1.04 ms (62 k img/s), three in flight 0.78 ms per forward (82 k img/s); batch 32: 0.74 -> 0.44 ms (43 k -> 73 k img/s). Depths 4 and 5
are worse than 3 (even depths pair up in lockstep), 6 - 7 level with 3.

The reference has no counterpart (engine.evaluate, engine.py:86-94, runs one synchronous forward per batch); results per batch are
those of `forward_batch` on the same images, bit for bit (tests/test_gpu_pipeline.py).
"""
import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib


class _Slot:
    __slots__ = ("stream", "images", "ws", "boxes", "scores", "labels", "counts", "packed", "done", "args", "src_ptr")


_STREAMS = {}


def _slot_stream(device, slot: int):
    """The stream of in-flight slot `slot` on `device`: ONE stream per (device, slot) for the life of the process, shared by every pipeline.
    Streams are mapped onto the GPU's few hardware queues round-robin in creation order; the first three a process creates land on three
    different queues, later ones double up with each other or with the plans' own sub-batch streams -- and two forwards that share a hardware
    queue do not overlap. Measured (round 6, one process): ssd_lite_mobilenet_v2 at 300 x 300, batch 128, as the FIRST pipeline 1.384 ms per
    step, as the third 1.498; ssd512_vgg16 6.21 / 6.47. Work of different pipelines on one stream is still ordered by the stream."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), slot)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device)
    return _STREAMS[key]


class ForwardPipeline:
    """`depth` forwards of `model` in flight on `device`, for batches of a fixed shape [batch, 3, height, width] (fp32 in [0, 1]) or,
    with uint8=True, [batch, height, width, 3] uint8 (a decoder's output, `SSD.forward_uint8`).

        pipe = ForwardPipeline(model, batch=64, depth=3)
        t = pipe.submit(images)                 # returns at once; the forward runs on the slot's own stream
        boxes, scores, labels, counts = pipe.result(t)      # waits for that forward; padded tensors as forward_batch returns them

    The tensors of ticket t are overwritten by submit number t + depth: whatever reads them must be enqueued, on the stream that
    is current at that later submit, before it (submit makes the slot's stream wait for the caller's stream). The pipeline uses
    the model's plan as it is when the pipeline is built: after a weight update build a new one (submit raises if the plan was
    re-lowered). While a pipeline is open the model's own forwards run as single chains as well (`dn_set_chains`); close() undoes it.
    """

    def __init__(self, model, batch: int, height: Optional[int] = None, width: Optional[int] = None, depth: int = 3, chains: int = 1,
                 device="cuda:0", uint8: bool = False, packed: bool = False):
        if depth < 1 or depth > 16:
            raise ValueError("depth must be in [1, 16], got {}".format(depth))
        if batch < 1:
            raise ValueError("batch must be positive")
        device = torch.device(device)
        W, H = model.graph.size
        self.model, self.device, self.depth, self.batch, self.uint8 = model, device, int(depth), int(batch), bool(uint8)
        self.h, self.w = int(height or H), int(width or W)
        self._handle = model._plan(device)          # raises without a GPU / the HIP library: there is no fallback path
        self._gen = model._plan_gen
        L = _lib.lib()
        self._L = L
        if getattr(model, "_pipe_refs", 0) > 0 and getattr(model, "_pipe_chains", chains) != chains:
            raise ValueError("another open pipeline of this model uses chains={}".format(model._pipe_chains))
        if getattr(model, "_pipe_refs", 0) == 0:    # (dn_set_chains drops the plan's graphs: only with no other pipeline replaying them)
            _lib.check(L.dn_set_chains(C.c_void_p(self._handle), int(chains)))
            model._bufs = {}                        # workspaces sized for the previous split are stale
        model._pipe_refs = getattr(model, "_pipe_refs", 0) + 1
        model._pipe_chains = chains
        D = model.detections_per_img
        self._fwd = L.dn_forward_u8 if uint8 else L.dn_forward
        self._name = "dn_forward_u8" if uint8 else "dn_forward"
        self.slots = []
        with torch.cuda.device(device):
            ws_bytes = L.dn_workspace_bytes(C.c_void_p(self._handle), batch)
            for _ in range(depth):
                s = _Slot()
                s.stream = _slot_stream(device, len(self.slots))
                s.images = (torch.empty((batch, self.h, self.w, 3), dtype=torch.uint8, device=device) if uint8
                            else torch.empty((batch, 3, self.h, self.w), dtype=torch.float32, device=device))
                s.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
                s.boxes = torch.empty((batch, D, 4), dtype=torch.float32, device=device)
                s.scores = torch.empty((batch, D), dtype=torch.float32, device=device)
                s.labels = torch.empty((batch, D), dtype=torch.int64, device=device)
                s.counts = torch.empty((batch,), dtype=torch.int32, device=device)
                s.packed = torch.zeros((batch, D + 1, 6), dtype=torch.float32, device=device) if packed else None
                s.done = torch.cuda.Event()
                s.src_ptr = None
                s.args = None
                self.slots.append(s)
        self.n = 0                                  # next ticket
        self._closed = False

    # ------------------------------------------------------------------------------------------------------
    def _args(self, s, src_ptr):
        if s.src_ptr != src_ptr:
            s.args = (C.c_void_p(self._handle), C.c_void_p(src_ptr), self.batch, self.h, self.w, C.c_void_p(s.boxes.data_ptr()),
                      C.c_void_p(s.scores.data_ptr()), C.c_void_p(s.labels.data_ptr()), C.c_void_p(s.counts.data_ptr()),
                      C.c_void_p(s.ws.data_ptr()), s.ws.numel(), C.c_void_p(s.stream.cuda_stream))
            s.src_ptr = src_ptr
        return s.args

    def submit(self, images: Tensor, persistent_input: bool = False) -> int:
        """Enqueue the forward of one batch; returns its ticket. persistent_input=True promises that this tensor keeps its
        address for every submit that lands on the same slot (ticket % depth) -- the graph replays on it directly; otherwise the
        batch is first copied into the slot's own input buffer (a device copy on the slot's stream)."""
        if self._closed:
            raise RuntimeError("the pipeline is closed")
        if self.model._handle != self._handle or self.model._plan_gen != self._gen:
            raise RuntimeError("the model's plan was rebuilt (weights changed / invalidate()): build a new ForwardPipeline")
        if not self.model._weights_unchanged(self.device):
            # in-place weight updates do not rebuild the plan by themselves (only the model's own forward does): checked on EVERY submit
            # (the flat-list test of SSD._weights_unchanged, tens of microseconds), so a pipeline never replays stale folded weights
            if (self.model._weights_signature(), str(self.device)) != self.model._sig:
                raise RuntimeError("the model's weights changed since the plan was lowered: close this pipeline and build a new one")
            self.model._remember_weights(self.device)      # only the structure epoch moved (another model was built): back to the cheap test
        s = self.slots[self.n % self.depth]
        if tuple(images.shape) != tuple(s.images.shape):
            raise ValueError("expected a batch of shape {}, got {}".format(tuple(s.images.shape), tuple(images.shape)))
        if self.uint8:
            if images.dtype != torch.uint8:
                raise ValueError("expected uint8 images, got {}".format(images.dtype))
        elif not images.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {images.dtype} instead")
        if images.device != self.device:
            raise ValueError("images are on {}, the pipeline on {}".format(images.device, self.device))
        # the slot's previous outputs may be overwritten now, and `images` must be complete: order after the caller's stream
        s.stream.wait_stream(torch.cuda.current_stream(self.device))
        direct = persistent_input and images.is_contiguous() and images.dtype == s.images.dtype
        if not direct:
            with torch.cuda.stream(s.stream):
                s.images.copy_(images, non_blocking=True)
            images.record_stream(s.stream)
        args = self._args(s, images.data_ptr() if direct else s.images.data_ptr())
        _lib.check(self._L.dn_set_packed_output(args[0], C.c_void_p(s.packed.data_ptr()) if s.packed is not None else None))
        self.model._packed_set = -1                 # (the model's own forward re-sets it)
        with torch.cuda.device(self.device):
            _lib.check(self._fwd(*args), self._name)
        s.done.record(s.stream)
        t = self.n
        self.n += 1
        return t

    def _slot_of(self, ticket: int):
        if ticket < 0 or ticket >= self.n:
            raise RuntimeError("ticket {} was never submitted".format(ticket))
        if ticket < self.n - self.depth:
            raise RuntimeError("ticket {} was overwritten: only the last {} forwards are kept".format(ticket, self.depth))
        return self.slots[ticket % self.depth]

    def wait(self, ticket: int, stream=None):
        """Make `stream` (default: the current stream) wait for forward `ticket` -- no host synchronisation."""
        s = self._slot_of(ticket)
        (stream or torch.cuda.current_stream(self.device)).wait_event(s.done)

    def result(self, ticket: int):
        """(boxes [B,D,4], scores [B,D], labels [B,D] int64, counts [B] int32) of forward `ticket`, complete on return."""
        s = self._slot_of(ticket)
        s.done.synchronize()
        return s.boxes, s.scores, s.labels, s.counts

    def detections(self, ticket: int):
        """The reference's output form for forward `ticket`: List[Dict[boxes, scores, labels]] (generalized_ssd.py:392-396)."""
        boxes, scores, labels, counts = self.result(ticket)
        return [{"boxes": boxes[i, :c], "scores": scores[i, :c], "labels": labels[i, :c]} for i, c in enumerate(counts.tolist())]

    def packed(self, ticket: int) -> Tensor:
        """The gather payload [B, D+1, 6] of forward `ticket` (packed=True); stream-ordered like the other outputs."""
        s = self._slot_of(ticket)
        if s.packed is None:
            raise RuntimeError("the pipeline was built without packed=True")
        return s.packed

    def stream_of(self, ticket: int):
        return self._slot_of(ticket).stream

    def join(self, stream=None):
        """Make `stream` (default: the current stream) wait for everything submitted so far."""
        st = stream or torch.cuda.current_stream(self.device)
        for s in self.slots:
            st.wait_stream(s.stream)

    def drain(self):
        """Host-wait for everything submitted so far."""
        for s in self.slots:
            s.stream.synchronize()

    def close(self):
        if self._closed:
            return
        self.drain()
        self._closed = True
        if self.model._handle == self._handle and self.model._plan_gen == self._gen:
            self.model._pipe_refs -= 1
            if self.model._pipe_refs == 0:          # the last pipeline of the plan: back to the automatic split
                _lib.check(self._L.dn_set_chains(C.c_void_p(self._handle), 0))
                self.model._bufs = {}
        self.slots = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
