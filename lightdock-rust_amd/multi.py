"""Sharding of independent swarms / pose batches over ranks (one process per GPU).

Swarms never exchange data in the reference: every swarm is its own OS process with its own
input file and output directory (lightdock-rust src/bin/lightdock-rust.rs:171-188; the task
lists of example/1czy/execution.sh:20-24).  So multi-GPU is pure sharding: rank r owns swarms
``r, r + W, r + 2W, ...``; the only cross-rank operations are a barrier / max-reduction for
timing and an optional gather of small per-swarm summaries to rank 0.  No data-path
collective exists or is needed.

The functions take a ``torch.distributed``-like module (or None for a single process) so the
same code runs under RCCL on GPUs and under gloo in the CPU tests.
"""
import time


def world(dist):
    if dist is None or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


def device_of_rank(local_rank, visible_devices, forced=None):
    """The device a rank computes on.  One process per GPU: local rank r takes device r; a rank whose launcher gave it its
    own GPU (per-rank HIP_VISIBLE_DEVICES plus LD_RANK_OWNS_DEVICE=1, or a job of one local rank) takes device 0; `forced`
    (LD_BENCH_FORCE_DEVICE in bench.py, LIGHTDOCK_DEVICE in launch.py) pins every rank to one device for dry runs of the
    N-rank path on a 1-GPU box.  More ranks than devices without `forced` is an error -- also when a job-wide
    HIP_VISIBLE_DEVICES=0 hides the others: silently sharing a GPU would report 1-GPU numbers as an N-GPU curve."""
    if forced is not None and str(forced) != "":
        d = int(forced)
        if not 0 <= d < max(1, visible_devices):
            raise ValueError("forced device %d but %d visible" % (d, visible_devices))
        return d
    if visible_devices <= 0:
        raise ValueError("no device visible")
    if visible_devices == 1:
        if local_rank > 0 and not _own_visible_device(local_rank):
            raise ValueError("local rank %d but one device visible and no per-rank HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES with LD_RANK_OWNS_DEVICE=1" % local_rank)
        return 0
    if local_rank >= visible_devices:
        raise ValueError("local rank %d but %d devices visible" % (local_rank, visible_devices))
    return local_rank


def _own_visible_device(local_rank=0):
    """One visible device is this rank's OWN only if the launcher says so (LD_RANK_OWNS_DEVICE=1 next to its per-rank
    HIP_VISIBLE_DEVICES) or the job has one local rank (and this is it: local rank 0): a visibility variable alone may be
    job-wide.  A launcher that masks devices per rank (SLURM --gpus-per-task, a wrapper script exporting
    HIP_VISIBLE_DEVICES=$LOCAL_RANK) must export LD_RANK_OWNS_DEVICE=1 too; the repository's own launchers (bench.py's
    spawn_ranks, torch.distributed.run) leave every device visible and need nothing."""
    import os
    if os.environ.get("LOCAL_WORLD_SIZE", "") == "1":
        return local_rank == 0
    masked = any(os.environ.get(k) not in (None, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    return masked and os.environ.get("LD_RANK_OWNS_DEVICE") == "1"


def shard(n_items, rank, world_size):
    """Indices owned by `rank`: strided, so every rank gets floor or ceil of n/W items."""
    return list(range(rank, n_items, world_size))


def gather_by_swarm(local_results, n_swarms, dist):
    """local_results: {swarm_id: small picklable summary}.  Returns the list ordered by swarm id
    on rank 0 (None elsewhere).  Host-side object gather; a few bytes per swarm."""
    rank, size = world(dist)
    if size == 1:
        merged = dict(local_results)
    else:
        buckets = [None] * size if rank == 0 else None
        dist.gather_object(dict(local_results), buckets, dst=0)
        if rank != 0:
            return None
        merged = {}
        for b in buckets:
            overlap = set(b) & set(merged)
            if overlap:
                raise RuntimeError("swarms evaluated twice: %s" % sorted(overlap))
            merged.update(b)
    missing = [s for s in range(n_swarms) if s not in merged]
    if missing:
        raise RuntimeError("swarms not evaluated: %s" % missing[:8])
    return [merged[s] for s in range(n_swarms)]


def timed_region(fn, dist, sync=None):
    """barrier + sync, run fn, barrier + sync; returns the MAX over ranks of the wall time."""
    rank, size = world(dist)

    def fence():
        if sync is not None:
            sync()
        if size > 1:
            dist.barrier()
        if sync is not None:
            sync()

    fence()
    t0 = time.perf_counter()
    fn()
    fence()
    elapsed = time.perf_counter() - t0
    if size > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def sum_over_ranks(value, dist):
    """Sum of one number over the ranks (timing / evaluation counts only; never pose data)."""
    rank, size = world(dist)
    if size == 1:
        return value
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def run_swarms(make_gso, n_swarms, steps, dist, summarize):
    """Run `n_swarms` independent swarms sharded over the ranks.

    make_gso(swarm_ids) -> object with .step() and whatever `summarize(gso, local_index, swarm_id)`
    reads; one batched GSO per rank over the swarms it owns."""
    rank, size = world(dist)
    mine = shard(n_swarms, rank, size)
    results = {}
    if mine:
        gso = make_gso(mine)
        for _ in range(steps):
            gso.step()
        for k, s in enumerate(mine):
            results[s] = summarize(gso, k, s)
    return gather_by_swarm(results, n_swarms, dist)
