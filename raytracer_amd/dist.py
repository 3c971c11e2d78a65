"""Image-space sharding over one process per GPU.

Pixels are independent (scene.rs:239-242), so the frame shards with no data-path
exchange: every rank holds the whole scene, renders the supertiles (4x4 tiles of 16x16 pixels)
`rank, rank + world, rank + 2*world, ...` into a compact device buffer
(rt_render_shard) and the compact buffers are gathered to rank 0
(torch.distributed.gather; RCCL over xGMI with the nccl backend), where one
kernel (rt_detile) scatters them into the row-major frame.
"""
import numpy as np

TILE = 16


def supertile():
    """tiles per supertile side (the library's value: one per process)"""
    from . import _native as N
    return int(N.gpu().rt_supertile())


def n_supertiles(vw, vh):
    side = TILE * supertile()
    return ((vw + side - 1) // side) * ((vh + side - 1) // side)


def row_rotation(supers_x, n_shards):
    """columns by which row r of supertiles is rotated, per row -- mirrors the library's shard_row_rotation"""
    import math
    if n_shards <= 1 or supers_x == 0:
        return 0
    g = max(1, int(math.floor(0.382 * n_shards + 0.5)))
    while math.gcd(g, n_shards) != 1:
        g += 1
    return (supers_x % n_shards + n_shards - g % n_shards) % n_shards


def shard_slots(vw, vh, shard, n_shards):
    """number of pixel slots (whole supertiles) of one shard -- mirrors rt_shard_pixels"""
    t = n_supertiles(vw, vh)
    mine = (t - shard + n_shards - 1) // n_shards if shard < t else 0
    return mine * (supertile() * TILE) ** 2


def max_shard_slots(vw, vh, n_shards):
    return shard_slots(vw, vh, 0, n_shards)


def shard_pixel_indices(vw, vh, shard, n_shards):
    """linear pixel index (y*vw + x) of every slot of a shard, -1 for the padding
    slots of edge supertiles -- the host-side statement of the device's slot map"""
    S = supertile()
    per_super = S * S * 256
    slots = shard_slots(vw, vh, shard, n_shards)
    slot = np.arange(slots, dtype=np.int64)
    k, t, j = slot // per_super, (slot >> 8) % (S * S), slot & 255
    sid = k * n_shards + shard
    supers_x = (vw + TILE * S - 1) // (TILE * S)
    # each row of supertiles is rotated so that the column phase a shard owns moves by a step coprime to n_shards
    # from row to row (the library's shard_row_rotation)
    sy = sid // supers_x
    sx = (sid % supers_x + sy * row_rotation(supers_x, n_shards)) % supers_x
    tx = sx * S + t % S
    ty = sy * S + t // S
    # the 64 slots of one wave cover an 8x8 pixel block; four blocks make the 16x16 tile
    sub, w = j >> 6, j & 63
    x = tx * TILE + (sub & 1) * 8 + (w & 7)
    y = ty * TILE + (sub >> 1) * 8 + (w >> 3)
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


_gather_bufs = {}


def gather_frame(local, vw, vh, rank, world, detile):
    """Gathers every rank's compact shard buffer to rank 0 and assembles the frame.

    local : torch tensor (max_shard_slots * elem,) uint8 on this rank's device, padded to the
            largest shard so that all ranks contribute equal counts
    detile: callable(gathered_tensor) -> frame, run on rank 0 only
    Returns the frame on rank 0, None elsewhere.  world == 1 needs no process group.
    """
    if world == 1:
        return detile(local)
    return gather_shards(local, rank, world, detile)


def gather_shards(local, rank, world, detile):
    """gather_frame's collective step: needs an initialised process group of `world` ranks (of any size, one included)"""
    import torch
    import torch.distributed as dist
    if local.is_cuda and dist.get_backend() == "gloo":
        # a backend that moves host memory only (rehearsals of the N > 1 path on one GPU): stage through the host
        host = local.cpu()
        if rank == 0:
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.gather(host, gather_list=parts, dst=0)
            return detile(torch.cat(parts).to(local.device))
        dist.gather(host, gather_list=None, dst=0)
        return None
    if rank == 0:
        # one buffer, gathered into through per-rank views: no allocation churn and no concatenation copy per frame
        key = (local.device, local.numel(), world)
        buf = _gather_bufs.get(key)
        if buf is None:
            buf = _gather_bufs[key] = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=list(buf.chunk(world)), dst=0)
        return detile(buf)
    dist.gather(local, gather_list=None, dst=0)
    if local.is_cuda:
        # the send is only queued when gather returns; the caller renders the next frame into `local` on another
        # stream as soon as it has the buffer back, so hand it back only once the send has left it
        torch.cuda.current_stream(local.device).synchronize()
    return None


class ShardedRenderer:
    """One rank's view of a sharded render on the GPU: rt_render_shard (or rt_render_shard_batch for `batch` frames
    at once) + gather + rt_detile."""

    def __init__(self, ctx, rank, world, device, batch=1):
        import torch
        self.ctx, self.rank, self.world, self.device, self.batch = ctx, rank, world, device, batch
        self.vw, self.vh = ctx.vw, ctx.vh
        self.slots = max_shard_slots(self.vw, self.vh, world)
        self.local = torch.zeros(batch * self.slots * 3, dtype=torch.uint8, device=device)
        self.frame = torch.zeros(batch * self.vh * self.vw * 3, dtype=torch.uint8, device=device) if rank == 0 else None

    def render(self):
        """-> (RtStats of this rank, frame tensor on rank 0 / None)"""
        st = self.ctx.render_shard(self.rank, self.world, self.local.data_ptr())
        frame = gather_frame(self.local, self.vw, self.vh, self.rank, self.world, self.detile)
        return st, frame

    def render_local(self, cameras=None):
        """this rank's shard into self.local: one frame through the scene's own camera, or one frame per camera"""
        if cameras is None:
            return self.ctx.render_shard(self.rank, self.world, self.local.data_ptr())
        assert (cameras if isinstance(cameras, int) else len(cameras)) <= self.batch
        return self.ctx.render_shard_batch(self.rank, self.world, cameras, self.local.data_ptr())

    def detile(self, gathered, n_frames=None):
        """gathered: (world, n_frames, slots, 3) bytes -- every rank's first n_frames frames of its buffer (default: all
        `batch` of them)"""
        import torch
        # the gather runs on torch's stream, rt_detile on the context's own: join them here
        if torch.device(self.device).type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        if self.batch == 1:
            self.ctx.detile(gathered.data_ptr(), self.world, self.slots, 3, self.frame.data_ptr())
        else:
            self.ctx.detile_frames(gathered.data_ptr(), self.world, n_frames or self.batch, self.slots, 3, self.frame.data_ptr())
        return self.frame


class FramePipeline:
    """Several frames of a sequence in flight on one GPU (the reference's multi-frame workload is `--sequence`,
    sdl/src/main.rs:70-95: independent frames, one after the other).

    A frame is a chain of dependent launches; near the end of each launch only a few long rays are left and most
    of the chip idles.  With `depth` contexts, each driven by its own host thread (the library calls release the
    GIL), the next frame's work fills those gaps.  Frames are handed back in order; the gather to rank 0 and the
    de-tile run on the calling thread, in frame order on every rank, so the collective sequence is the same
    everywhere.

    batch > 1: a slot renders `batch` consecutive frames of a camera path in one set of launches
    (rt_render_shard_batch) and they are gathered together.  The launches of a 1/N shard are N times smaller than a
    whole frame's and fill the chip badly; batches of N frames restore their size.
    """

    def __init__(self, make_context, rank, world, device, depth=2, batch=1):
        import queue
        import threading
        self.rank, self.world, self.device, self.depth, self.batch = rank, world, device, depth, batch
        self.ctxs = [make_context() for _ in range(depth)]
        self.renderers = [ShardedRenderer(c, rank, world, device, batch) for c in self.ctxs]
        self._go = [queue.Queue() for _ in range(depth)]
        self._done = [queue.Queue() for _ in range(depth)]
        self._threads = [threading.Thread(target=self._worker, args=(k,), daemon=True) for k in range(depth)]
        for t in self._threads:
            t.start()

    def _worker(self, k):
        r = self.renderers[k]
        while True:
            job = self._go[k].get()
            if job is None:
                return
            try:
                self._done[k].put(r.render_local(job[1]))
            except Exception as e:  # handed to the caller of run()
                self._done[k].put(e)

    def run(self, n_frames, on_frame=None, camera_of=None):
        """Renders n_frames frames; on_frame(index, stats, frame_tensor_or_None) is called in frame order (stats: the
        RtStats of the set of launches the frame was part of, with .frames = how many frames that set covered).
        camera_of(f) -> Camera of frame f; None: every frame through the scene's own camera.
        Returns the list of RtStats of this rank, one per set of launches."""
        B = self.batch
        n_jobs = (n_frames + B - 1) // B
        # sets of launches of as equal a size as possible (20 frames in sets of at most 8: 7 + 7 + 6, not 8 + 8 + 4): a
        # short last set would render alone, with nothing to fill the chip beside it
        first = [j * (n_frames // n_jobs) + min(j, n_frames % n_jobs) for j in range(n_jobs + 1)] if n_jobs else [0]

        def job(j):
            lo, hi = first[j], first[j + 1]
            if camera_of is not None:
                return (j, self.ctxs[j % self.depth].camera_array([camera_of(f) for f in range(lo, hi)]))
            return (j, None if B == 1 else hi - lo)

        stats = []
        for j in range(min(self.depth, n_jobs)):
            self._go[j % self.depth].put(job(j))
        for j in range(n_jobs):
            k = j % self.depth
            st = self._done[k].get()
            if isinstance(st, Exception):
                raise st
            r = self.renderers[k]
            lo, hi = first[j], first[j + 1]
            st.frames = hi - lo
            # only the frames this set rendered travel and are put together (a last, shorter set; a single frame)
            n = hi - lo
            part = r.local if n == B else r.local[: n * r.slots * 3]
            frames = gather_frame(part, r.vw, r.vh, self.rank, self.world, lambda g: r.detile(g, n))
            if on_frame is not None:
                per = r.vw * r.vh * 3
                for f in range(lo, hi):
                    on_frame(f, st, None if frames is None else frames[(f - lo) * per:(f - lo + 1) * per])
            stats.append(st)
            if j + self.depth < n_jobs:
                self._go[k].put(job(j + self.depth))     # the slot's buffer is free again once its frames are gathered
        return stats

    def close(self):
        for q in self._go:
            q.put(None)
        for t in self._threads:
            t.join(timeout=10)
        for c in self.ctxs:
            c.close()
