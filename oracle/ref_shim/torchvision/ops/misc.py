import torch


class FrozenBatchNorm2d(torch.nn.Module):
    """import-time stub (reference _utils.py imports the name; never instantiated on the SSD path)."""
