import torch


def run(data, params):
    """Channel-local: every output row depends on its own input row only."""
    return torch.cumsum(data, dim=1) * 2.0 + data.mean(dim=1, keepdim=True)
