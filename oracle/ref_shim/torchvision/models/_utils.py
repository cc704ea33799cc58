import torch


class IntermediateLayerGetter(torch.nn.Module):
    """import-time stub (reference backbone.py:9)."""
