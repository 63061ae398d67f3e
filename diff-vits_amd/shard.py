"""Batch sharding of the sampler across the GPUs of one node.

Utterances are independent through the whole sampler (GroupNorm/LayerNorm/attention are
per-sample), so rank r owns utterances [r*B, (r+1)*B) and there is no per-step communication.
The only collectives are one broadcast of the conditioning tensors (speaker-prompt states and
their mask) from rank 0 before a run and one all-gather of the finished mels after it —
`torch.distributed` on the "nccl" backend, i.e. RCCL over xGMI, on GPUs; gloo in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, world_size, rank):
    """Contiguous, balanced [start, stop) of the global batch for `rank`."""
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_conditioning(enc, mask, src=0, group=None):
    """In-place broadcast of the global conditioning (enc [G, L, D] float32, mask [G, L]) from
    `src` to every rank.  Bool masks travel as uint8."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return enc, mask
    dist.broadcast(enc, src=src, group=group)
    if mask is not None:
        if mask.dtype == torch.bool:
            m8 = mask.to(torch.uint8)
            dist.broadcast(m8, src=src, group=group)
            mask.copy_(m8.to(torch.bool))
        else:
            dist.broadcast(mask, src=src, group=group)
    return enc, mask


def gather_outputs(local, group=None, global_batch=None):
    """All-gather the ranks' shards [B_r, C, T] into the global [G, C, T] (rank order).  With `global_batch` the shard
    sizes are those of `shard_range` (they may differ by one when G % world != 0): every rank pads its shard to the
    largest, one `all_gather_into_tensor` moves them (RCCL has no ragged all-gather), and the padding rows are dropped."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    if global_batch is None or global_batch % world == 0:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    sizes = [b - a for a, b in (shard_range(global_batch, world, r) for r in range(world))]
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("gather_outputs: local shard has %d rows, shard_range gives %d"
                         % (local.shape[0], sizes[dist.get_rank(group)]))
    big = max(sizes)
    padded = local.new_zeros((big,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    out = torch.empty((world * big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * big:r * big + sizes[r]] for r in range(world)], dim=0)


def sharded_sample(run_local, x_shard, cond_shard, enc_global, mask_global, group=None):
    """One sharded sampler run: broadcast conditioning, run this rank's shard with
    `run_local(x, cond, enc, mask) -> mel`, all-gather the mels."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    enc_global, mask_global = broadcast_conditioning(enc_global, mask_global, 0, group)
    lo, hi = shard_range(enc_global.shape[0], world, rank)
    mel = run_local(x_shard, cond_shard, enc_global[lo:hi].contiguous(),
                    None if mask_global is None else mask_global[lo:hi].contiguous())
    return gather_outputs(mel, group, global_batch=enc_global.shape[0])
