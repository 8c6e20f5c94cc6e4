"""Multi-GPU calibration over RCCL / xGMI -- one process per MI355X (torch.distributed, backend "nccl" == RCCL).

The reference has no distributed path on weight-only quantisation; this is new design, in the two
forms SURVEY.md 8(e) derives from the algorithm's structure:

  mode "layer"  (north_star; `prepare(model, GPTQConfig(...), independent_blocks=True)` or INC_MI355X_GPTQ_MULTI_GPU=layer; what
                `bench.py --gpus N` times by default): every transformer block is owned by one rank (block b -> rank b % world).
                Samples are sharded; per round of `world` blocks every rank forwards ITS samples through the round's FLOAT blocks,
                the inputs of block b travel to its owner over xGMI -- point-to-point by default (each rank sends its shard to the
                owner only: a balanced all-to-all over the per-pair links), INC_MI355X_GPTQ_ACT_EXCHANGE=broadcast for the
                broadcast form -- and all ranks quantise their own blocks concurrently; the packed blocks are broadcast at the
                end.  Per-layer results equal the single-GPU ones given the same inputs (bit-identical to a single process run of
                this mode, tests/test_gpu_models.py); the whole-model result differs from the reference's sequential scheme
                (block b+1 calibrated on QUANTISED block b outputs, gptq.py:749-762), which is why this mode is opt-in.
                Implementation: RAWGPTQuantizer.independent_setup / independent_round / independent_finish (gptq.py).
  mode "sample" (exact): calibration samples are sharded across ranks; each rank accumulates its own running-mean
                Hessian and `allreduce_hessian` combines them:  H = sum_r (n_r / n) H_r  (H is a sample mean, so the
                combination is exact up to fp32 summation order).  One all-reduce of K*K fp32 per DISTINCT layer
                input (64 MiB for K=4096, 462 MiB for K=11008): with 7 point-to-point xGMI links per GPU a ring is
                bound by one link, so large H go as reduce-scatter + all-gather (what RCCL's all_reduce does
                internally for big messages); no activation ever crosses GPUs.

  mode "sample+rows" (exact; `bench.py --gpus N --mgpu-mode exact`; SURVEY.md 8(e) (i)+(ii)+(iii)): ONE model quantised by N
                ranks.  Samples are sharded as in mode "sample" (block forwards and Hessian accumulation scale with N, no
                activation crosses GPUs), but the solve is distributed too instead of being repeated on every rank:
                  * the i-th DISTINCT Hessian of a block is REDUCED to rank i % N (collective C1 as a reduce), which alone
                    factorises it -- the four factorisations of a Llama block run on four GPUs at once -- and BROADCASTS
                    the inverse Cholesky factor (64 / 462 MiB; same bytes on the wire as the all-reduce it replaces);
                  * every op of the column loop is row-wise (gptq.py:1250-1304), so each rank runs the loop on its slice
                    of the (N-stacked) weight rows and the codes / Q / scales / zeros are ALL-GATHERED (collective C2,
                    8-22 MiB per Linear); every rank then holds the same quantised weights for the second forward over
                    ITS samples and packs the same model.
                With the same Hessian on every rank (`sample_sharded=False`) the result is bit-identical to the
                single-process one; with sharded samples it differs by the fp32 summation order of the Hessian only.

Nothing here does arithmetic beyond the collective bookkeeping; the kernels are the same single-GPU HIP kernels.
"""

import os

import torch
import torch.distributed as dist


# Bring-up switch: keep every multi-GPU code path LIVE on a world of ONE rank (process group initialised, CalibrationGroup built, the
# drivers' collectives issued).  On a 1-GPU box this runs RCCL's init with `device_id`, the collective wrappers, the asynchronous
# broadcast handles and the packed-block broadcasts for real (scripts/rccl_single_rank.py, `bench.py --gpus 1 --mgpu-mode ...`), so
# that the first 8-GPU run is not also RCCL's first run.  Off: a world of one rank takes the plain single-GPU path.
SINGLE_RANK_GROUP = os.environ.get("INC_MI355X_DIST_SINGLE_RANK", "0") == "1"


def live(group=None):
    """True when the multi-GPU paths should run: an initialised group of more than one rank (or of one rank under SINGLE_RANK_GROUP)."""
    return bool(dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or SINGLE_RANK_GROUP))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or SINGLE_RANK_GROUP) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # INC_MI355X_DIST_BACKEND=gloo: the collectives' host-staged test form (CalibrationGroup) -- lets the N-rank driver
            # be exercised by N processes that share one GPU (RCCL refuses two ranks on one device)
            backend = os.environ.get("INC_MI355X_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if torch.cuda.is_available() and local_rank >= torch.cuda.device_count():
        local_rank %= torch.cuda.device_count()  # more ranks than GPUs (tests only): share devices round-robin
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


def row_shard(n_rows, rank, world, align=64):
    """Rows [r0, r1) of an [n_rows, ...] solve owned by `rank`, plus the (aligned) shard size all ranks pad to.
    Shards are multiples of `align` rows (the column-loop kernels work on 16-row waves / 64-row workgroups); trailing
    ranks may own fewer rows, or none."""
    shard = -(-n_rows // world)
    shard = -(-shard // align) * align
    r0 = min(rank * shard, n_rows)
    r1 = min(r0 + shard, n_rows)
    return r0, r1, shard


class CalibrationGroup:
    """The collectives of mode "sample+rows" on one process group (default: the world).

    backend "nccl" (= RCCL) moves HBM-resident tensors over xGMI in place; under "gloo" (CPU tests, or two test
    processes sharing one GPU) device tensors are staged through the host, synchronously."""

    def __init__(self, group=None):
        assert dist.is_initialized(), "initialise torch.distributed first (distributed.init_from_env)"
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def _global(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def owner(self, index):
        """Rank that factorises the index-th distinct Hessian of a block (the expensive K=11008 one is the last of a
        Llama block and lands on its own rank for world >= 4)."""
        return index % self.world

    def _staged(self, t):
        return self.backend == "gloo" and t.is_cuda

    def all_reduce(self, t):
        if self._staged(t):
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def reduce(self, t, dst):
        """Sum onto rank `dst` (other ranks' buffers are left undefined)."""
        if self._staged(t):
            host = t.cpu()
            dist.reduce(host, dst=self._global(dst), op=dist.ReduceOp.SUM, group=self.group)
            if self.rank == dst:
                t.copy_(host)
        else:
            dist.reduce(t, dst=self._global(dst), op=dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast(self, t, src, async_op=False):
        """Returns a handle with .wait() when async_op (None when the transfer already completed)."""
        if self._staged(t):
            host = t.cpu() if self.rank == src else torch.empty(t.shape, dtype=t.dtype)
            dist.broadcast(host, src=self._global(src), group=self.group)
            if self.rank != src:
                t.copy_(host)
            return None
        work = dist.broadcast(t, src=self._global(src), group=self.group, async_op=async_op)
        return work if async_op else None

    def all_gather_rows(self, local, n_rows, shard):
        """Concatenate the ranks' row slices (`row_shard` partition) of an [n_rows, ...] tensor; `local` may be empty."""
        tail = tuple(local.shape[1:])
        padded = torch.zeros((shard,) + tail, dtype=local.dtype, device=local.device)
        if local.shape[0]:
            padded[: local.shape[0]].copy_(local)
        out = torch.empty((self.world * shard,) + tail, dtype=local.dtype, device=local.device)
        if self._staged(local):
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, padded.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, padded, group=self.group)
        if self.world * shard == n_rows:
            return out
        # ranks own [r*shard, min((r+1)*shard, n_rows)): only the last non-empty shard is short, so the valid rows are
        # exactly the first n_rows of the concatenation
        return out[:n_rows].contiguous()


def solve_cost(n_rows, n_cols):
    """Relative time of one column loop over [n_rows, n_cols] (a latency chain per column with a per-row throughput term; fitted to
    tools/kbench qlayer: 4096^2 2.2 ms, 12288 x 4096 3.8, 22016 x 4096 6.3, 4096 x 11008 8.8)."""
    return n_cols * (0.4 + n_rows / 20480.0)


def plan_solves_2d(shapes, world):
    """Mode "sample+rows", 2-D form: assign the independent solves of a block (one per DISTINCT Hessian: `shapes` = [(rows, cols), ...])
    to groups of ranks -- module x rows instead of rows alone.  With rows alone every rank walks all the column loops one after the
    other, and a loop's time hardly shrinks with fewer rows (it is a chain of `cols` dependent steps); here the loops of a block run
    side by side on disjoint rank groups and only the rows inside a group are sharded.
      world >= len(shapes): every solve gets its own contiguous rank group; the spare ranks go, one at a time, to the solve whose
                            estimated time (solve_cost on its rows / group size) is largest;
      world <  len(shapes): every rank is a group of one; solves are dealt longest-first to the least-loaded rank.
    Returns [ranks of solve 0, ranks of solve 1, ...] (lists of group-local ranks, ascending; the first is the solve's leader: it
    receives the reduced Hessian, factorises it and publishes the results).  Deterministic: every rank computes the same plan."""
    n = len(shapes)
    if n == 0:
        return []
    plan = _plan_groups(shapes, world)
    load = {}
    for (r, c), g in zip(shapes, plan):
        load[tuple(g)] = load.get(tuple(g), 0.0) + solve_cost(-(-r // len(g)), c)
    if max(load.values()) > sum(solve_cost(-(-r // world), c) for r, c in shapes):
        return [list(range(world)) for _ in shapes]  # row-dominated solves: sharding rows alone over all ranks is the better plan
    return plan


def _plan_groups(shapes, world):
    n = len(shapes)
    if world >= n:
        size = [1] * n
        for _ in range(world - n):
            t = [solve_cost(-(-r // size[i]), c) for i, (r, c) in enumerate(shapes)]
            size[max(range(n), key=lambda i: (t[i], -i))] += 1
        out, nxt = [], 0
        for sz in size:
            out.append(list(range(nxt, nxt + sz)))
            nxt += sz
        return out
    load = [0.0] * world
    out = [None] * n
    for i in sorted(range(n), key=lambda i: (-solve_cost(*shapes[i]), i)):
        r = min(range(world), key=lambda r: (load[r], r))
        out[i] = [r]
        load[r] += solve_cost(*shapes[i])
    return out


_SUBGROUPS = {}


def subgroup(ranks, parent=None):
    """CalibrationGroup over `ranks` (local ranks of `parent`, ascending).  dist.new_group is collective over the parent: every rank
    must call this with the same rank lists in the same order (the plan is deterministic, so they do); groups are created once."""
    world = dist.get_world_size(parent)
    key = (id(parent), tuple(ranks))
    if key not in _SUBGROUPS:
        if len(ranks) == world:
            _SUBGROUPS[key] = CalibrationGroup(parent)
        else:
            glob = [r if parent is None else dist.get_global_rank(parent, r) for r in ranks]
            pg = dist.new_group(ranks=glob, backend=dist.get_backend(parent))
            me = dist.get_rank(parent)
            _SUBGROUPS[key] = CalibrationGroup(pg) if me in ranks else None
    return _SUBGROUPS[key]


def shard_samples(n_samples, rank, world):
    """Contiguous, balanced split of calibration sample indices (sample ownership is fixed for the whole run, so
    block outputs never need to be exchanged: rank r forwards its own samples through every block)."""
    base, rem = divmod(n_samples, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def barrier_max_time(seconds, device=None):
    """max over ranks of a local wall-clock measurement (bench.py contract)."""
    if not live():
        return seconds
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
