"""Multi-GPU plumbing: a batch of independent images is dealt round-robin to the ranks
(frame i -> rank i mod N).  There is no exchange step, hence no collective on the data
path; torch.distributed is used only for the barrier / max-over-ranks timing of bench.py."""


def shard_indices(n_items, rank, world):
    return list(range(rank, n_items, world))


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a python float (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    if dist.get_backend() == "gloo":
        device = None   # gloo reduces host tensors
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
