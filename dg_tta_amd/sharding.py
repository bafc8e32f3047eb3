"""Sample sharding for multi-GPU TTA: units are independent (reference: tta.py:157-182), so ranks only need a static
partition and — for benchmarking — a barrier and a max-over-ranks reduction of the elapsed time.  No data-path collective."""
import os


def rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def owns(sample_index, rank, world):
    return sample_index % world == rank


def units_for_rank(num_samples, ensemble_count, rank, world):
    """(sample, ensemble) units of this rank: samples round-robin, all ensemble members of a sample on one rank (the
    sample's ensemble-averaged inference needs all of its members' files)."""
    return [(s, e) for s in range(num_samples) if owns(s, rank, world) for e in range(ensemble_count)]


def max_over_ranks(seconds, device=None):
    """Slowest rank's time (what a whole-job throughput must be quoted on)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
