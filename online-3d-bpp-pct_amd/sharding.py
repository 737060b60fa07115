"""Multi-GPU layout of the env batch.

Envs are independent (each reference worker process owns one env and one RNG,
wrapper/shmem_vec_env.py:119-156, envs.py:49), so the job shards by env index with NO
collective on the step path: global env g lives on rank g // per_rank as local env
g % per_rank, and its item stream / stand-in policy are keyed by the GLOBAL id g
(pct_config.env_id_base), so a sharded job is bit-identical to one big batch.

The only exchange the reference's trainer shape can need is an all-gather of the rollout
shards into storage.py-shaped tensors (storage.py:5-11); `gather_rollout` does that with
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests).
"""
import torch


def shard_envs(total_envs, rank, world_size):
    """(env_id_base, local_num_envs) of `rank`; the last rank takes the remainder (gather_rollout copes with
    unequal shards by padding)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    per = total_envs // world_size
    if per == 0:
        raise ValueError("fewer envs than ranks")
    base = rank * per
    n = per if rank < world_size - 1 else total_envs - base
    return base, n


def gather_rollout(local, group=None):
    """All-gather the ranks' shards along dim 0: [n_rank, ...] -> [sum of n_rank, ...], rank order.  Shards may
    differ in length (shard_envs gives the last rank the remainder): the lengths are exchanged first and the
    shards are padded to the longest for the one all_gather_into_tensor."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = [int(x) for x in sizes.tolist()]
    nmax = max(sizes)
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    if all(s == nmax for s in sizes):
        return out
    return torch.cat([out[r * nmax:r * nmax + sizes[r]] for r in range(world)], 0)
