"""Image-space sharding over one process per GPU.

Pixels are independent (scene.rs:239-242), so the frame shards with no data-path
exchange: every rank holds the whole scene, renders the 16x16 tiles
`rank, rank + world, rank + 2*world, ...` into a compact device buffer
(rt_render_shard) and the compact buffers are gathered to rank 0
(torch.distributed.gather; RCCL over xGMI with the nccl backend), where one
kernel (rt_detile) scatters them into the row-major frame.
"""
import numpy as np

TILE = 16


def n_tiles(vw, vh):
    return ((vw + TILE - 1) // TILE) * ((vh + TILE - 1) // TILE)


def shard_slots(vw, vh, shard, n_shards):
    """number of pixel slots (whole tiles) of one shard -- mirrors rt_shard_pixels"""
    t = n_tiles(vw, vh)
    mine = (t - shard + n_shards - 1) // n_shards if shard < t else 0
    return mine * TILE * TILE


def max_shard_slots(vw, vh, n_shards):
    return shard_slots(vw, vh, 0, n_shards)


def shard_pixel_indices(vw, vh, shard, n_shards):
    """linear pixel index (y*vw + x) of every slot of a shard, -1 for the padding
    slots of edge tiles -- the host-side statement of the device's slot map"""
    slots = shard_slots(vw, vh, shard, n_shards)
    k = np.arange(slots, dtype=np.int64) >> 8
    j = np.arange(slots, dtype=np.int64) & 255
    gid = k * n_shards + shard
    tiles_x = (vw + TILE - 1) // TILE
    # the 64 slots of one wave cover an 8x8 pixel block; four blocks make the 16x16 tile
    sub, w = j >> 6, j & 63
    x = (gid % tiles_x) * TILE + (sub & 1) * 8 + (w & 7)
    y = (gid // tiles_x) * TILE + (sub >> 1) * 8 + (w >> 3)
    ok = (x < vw) & (y < vh)
    return np.where(ok, y * vw + x, -1)


def detile_host(gathered, vw, vh, n_shards, slots_per_shard, elem):
    """numpy statement of rt_detile: gathered is (n_shards, slots_per_shard, elem)"""
    g = np.asarray(gathered).reshape(n_shards, slots_per_shard, elem)
    out = np.zeros((vh * vw, elem), dtype=g.dtype)
    for s in range(n_shards):
        idx = shard_pixel_indices(vw, vh, s, n_shards)
        ok = idx >= 0
        out[idx[ok]] = g[s, : len(idx)][ok]
    return out.reshape(vh, vw, elem)


def gather_frame(local, vw, vh, rank, world, detile):
    """Gathers every rank's compact shard buffer to rank 0 and assembles the frame.

    local : torch tensor (max_shard_slots * elem,) uint8 on this rank's device, padded to the
            largest shard so that all ranks contribute equal counts
    detile: callable(gathered_tensor) -> frame, run on rank 0 only
    Returns the frame on rank 0, None elsewhere.  world == 1 needs no process group.
    """
    import torch
    import torch.distributed as dist
    if world == 1:
        return detile(local)
    if rank == 0:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=parts, dst=0)
        return detile(torch.cat(parts))
    dist.gather(local, gather_list=None, dst=0)
    return None


class ShardedRenderer:
    """One rank's view of a sharded render on the GPU: rt_render_shard + gather + rt_detile."""

    def __init__(self, ctx, rank, world, device):
        import torch
        self.ctx, self.rank, self.world, self.device = ctx, rank, world, device
        self.vw, self.vh = ctx.vw, ctx.vh
        self.slots = max_shard_slots(self.vw, self.vh, world)
        self.local = torch.zeros(self.slots * 3, dtype=torch.uint8, device=device)
        self.frame = torch.zeros(self.vh * self.vw * 3, dtype=torch.uint8, device=device) if rank == 0 else None

    def render(self):
        """-> (RtStats of this rank, frame tensor on rank 0 / None)"""
        import torch
        st = self.ctx.render_shard(self.rank, self.world, self.local.data_ptr())

        def detile(gathered):
            # the gather runs on torch's stream, rt_detile on the context's own: join them here
            torch.cuda.current_stream(self.device).synchronize()
            self.ctx.detile(gathered.data_ptr(), self.world, self.slots, 3, self.frame.data_ptr())
            return self.frame

        frame = gather_frame(self.local, self.vw, self.vh, self.rank, self.world, detile)
        return st, frame
