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
