"""Multi-GPU calibration over RCCL / xGMI -- one process per MI355X (torch.distributed, backend "nccl" == RCCL).

The reference has no distributed path on weight-only quantisation; this is new design, in the two
forms SURVEY.md 8(e) derives from the algorithm's structure:

  mode "layer"  (north_star): every transformer block is owned by one rank (block b -> rank b % world).  The input
                activations of block b -- computed with the FLOAT model -- are broadcast from the rank that holds
                them to the owner over xGMI, then all ranks quantise their own blocks concurrently.  Per-layer
                results equal the single-GPU ones given the same inputs; the whole-model result differs from the
                reference's sequential scheme (block b+1 calibrated on QUANTISED block b outputs, gptq.py:749-762),
                which is why this mode is opt-in (`independent_blocks=True`) at the API level.
  mode "sample" (exact): calibration samples are sharded across ranks; each rank accumulates its own running-mean
                Hessian and `allreduce_hessian` combines them:  H = sum_r (n_r / n) H_r  (H is a sample mean, so the
                combination is exact up to fp32 summation order).  One all-reduce of K*K fp32 per DISTINCT layer
                input (64 MiB for K=4096, 462 MiB for K=11008): with 7 point-to-point xGMI links per GPU a ring is
                bound by one link, so large H go as reduce-scatter + all-gather (what RCCL's all_reduce does
                internally for big messages); no activation ever crosses GPUs.

Nothing here does arithmetic beyond the collective bookkeeping; the kernels are the same single-GPU HIP kernels.
"""

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local_rank


def owner_of_block(block_idx, world):
    """Round-robin block ownership: consecutive blocks land on different GPUs so a pipeline of broadcasts overlaps."""
    return block_idx % world


def blocks_of_rank(n_blocks, rank, world):
    return [b for b in range(n_blocks) if owner_of_block(b, world) == rank]


def broadcast_calibration(acts, src, group=None, shape=None, dtype=None, device=None):
    """Broadcast one block's calibration activations ([samples, seq, hidden]) from `src` to every rank.

    Receivers pass acts=None plus (shape, dtype, device).  One large message (2 GiB for Llama-2-7B: 128x2048x4096
    bf16) rather than 128 small ones: xGMI links are per-pair, bandwidth-bound, and RCCL pipelines big messages.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return acts
    if acts is None:
        acts = torch.empty(shape, dtype=dtype, device=device)
    dist.broadcast(acts, src=src, group=group)
    return acts


def allreduce_hessian(H, nsamples, group=None):
    """Combine per-rank running-mean Hessians H_r (each = (2/n_r) sum_{own samples} X^T X) into the global one.

    Returns (H, n_total); H is updated in place.  Exact for any per-rank sample counts (also 0).
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return H, nsamples
    H.mul_(float(nsamples))  # back to (2 * sum X^T X) so that ranks can be added
    if H.is_cuda and dist.get_backend(group) == "gloo":
        # CPU-side collective (tests on a single-GPU box: two processes share one MI355X, gloo between them); the
        # production backend is "nccl" = RCCL, which reduces the HBM-resident tensor in place over xGMI
        n = torch.tensor([float(nsamples)], dtype=torch.float64)
        host = H.cpu()
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        H.copy_(host)
    else:
        n = torch.tensor([float(nsamples)], dtype=torch.float64, device=H.device)
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(H, op=dist.ReduceOp.SUM, group=group)
    total = int(round(float(n.item())))
    if total > 0:
        H.div_(float(total))
    return H, total


def shard_samples(n_samples, rank, world):
    """Contiguous, balanced split of calibration sample indices (sample ownership is fixed for the whole run, so
    block outputs never need to be exchanged: rank r forwards its own samples through every block)."""
    base, rem = divmod(n_samples, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def barrier_max_time(seconds, device=None):
    """max over ranks of a local wall-clock measurement (bench.py contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
