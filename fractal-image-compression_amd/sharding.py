"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).

Range blocks are independent given the pool (FractalCompression.java:125-159 carries no state
between iterations except j++), so rank g encodes a contiguous, tile-aligned span of range
blocks against its own replica of the domain pool (each rank builds the pool from the
replicated input image: cheaper than shipping the 16x-expanded pool).  The only exchange is
one gather of 24 bytes per range block (the IFS codebook rows) -- latency-bound on xGMI.
"""
import numpy as np

RECORD_WORDS = 6   # idx_local, a bits, b bits, iso, (int)(a*100), (int)b
_USE_ALL_GATHER = False   # set when the backend turns out to have no gather (see gather_records)


def shard_spans(n_ranges, ranges_per_tile, world):
    """Contiguous spans [(begin, count)] per rank, aligned to the sweep kernel's tile size so
    no tile is computed twice.  Ranks beyond the last tile get (n_ranges, 0)."""
    tiles = (n_ranges + ranges_per_tile - 1) // ranges_per_tile
    spans = []
    for r in range(world):
        t0 = (tiles * r) // world
        t1 = (tiles * (r + 1)) // world
        b = min(t0 * ranges_per_tile, n_ranges)
        e = min(t1 * ranges_per_tile, n_ranges)
        spans.append((b, e - b))
    return spans


def shard_planes(n_planes, world):
    """Config 5: whole planes per rank."""
    return [((n_planes * r) // world, (n_planes * (r + 1)) // world - (n_planes * r) // world) for r in range(world)]


def pack_records(res, begin, count, xp=None):
    """Packs the [planes, N_r] result arrays of one span into int32 [planes, count, 6] records.
    Works on numpy arrays or torch tensors (bit-casts a,b to int32)."""
    sl = slice(begin, begin + count)
    if hasattr(res["a"], "view") and not isinstance(res["a"], np.ndarray):
        import torch
        cols = [res["idx_local"][:, sl], res["a"][:, sl].contiguous().view(torch.int32),
                res["b"][:, sl].contiguous().view(torch.int32), res["iso"][:, sl],
                res["qrows"][:, sl, 1], res["qrows"][:, sl, 2]]
        return torch.stack([c.to(torch.int32) for c in cols], dim=-1).contiguous()
    cols = [res["idx_local"][:, sl], np.ascontiguousarray(res["a"][:, sl]).view(np.int32),
            np.ascontiguousarray(res["b"][:, sl]).view(np.int32), res["iso"][:, sl],
            res["qrows"][:, sl, 1], res["qrows"][:, sl, 2]]
    return np.ascontiguousarray(np.stack([c.astype(np.int32) for c in cols], axis=-1))


def unpack_records(rec):
    """int32 numpy [planes, N, 6] -> result dict (numpy)."""
    rec = np.ascontiguousarray(rec, np.int32)
    return {
        "idx_local": rec[..., 0].copy(),
        "a": np.ascontiguousarray(rec[..., 1]).view(np.float32).copy(),
        "b": np.ascontiguousarray(rec[..., 2]).view(np.float32).copy(),
        "iso": rec[..., 3].copy(),
        "qrows": np.stack([rec[..., 0], rec[..., 4], rec[..., 5]], axis=-1).astype(np.int32),
    }


def gather_records(local, spans, group=None, dst=0):
    """One collective: every rank contributes its span's records ([planes, count_r, 6] int32
    torch tensor, CUDA for nccl/RCCL or CPU for gloo), padded to the largest span; rank `dst`
    returns the concatenated [planes, N_r, 6] tensor, other ranks None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    maxc = max(c for _, c in spans)
    planes = local.shape[0]
    pad = torch.zeros((planes, maxc, RECORD_WORDS), dtype=torch.int32, device=local.device)
    pad[:, : local.shape[1]] = local
    global _USE_ALL_GATHER
    if not _USE_ALL_GATHER:
        try:
            if rank == dst:
                bufs = [torch.empty_like(pad) for _ in range(world)]
                dist.gather(pad, bufs, dst=dst, group=group)
                return torch.cat([bufs[r][:, : spans[r][1]] for r in range(world)], dim=1)
            dist.gather(pad, None, dst=dst, group=group)
            return None
        except (NotImplementedError, RuntimeError) as e:
            # a backend without gather rejects the call on every rank before communicating: switch once
            if "gather" not in str(e).lower() and "support" not in str(e).lower():
                raise
            _USE_ALL_GATHER = True
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:, : spans[r][1]] for r in range(world)], dim=1)


class ShardedEncoder:
    """Strong-scaling encode of one batch of planes across the ranks of a process group:
    rank r sweeps spans[r] of every plane; rank 0 receives the whole codebook."""

    def __init__(self, width, height, B, wK=None, n_iso=1, planes=1, device=0, group=None):
        import torch.distributed as dist
        from .host import Encoder
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.enc = Encoder(width, height, B, wK, n_iso, planes, device)
        self.spans = shard_spans(self.enc.n_ranges, self.enc.ranges_per_tile, self.world)
        self._res = None

    def set_gray(self, gray):
        self.enc.set_gray(gray)

    def encode_local(self, stream=None):
        b, c = self.spans[self.rank]
        self.enc.encode(b, c, stream)

    def gather(self):
        """Device-side pack + one RCCL gather; returns numpy result dict on rank 0, else None."""
        import torch
        b, c = self.spans[self.rank]
        if self._res is None:
            self._res = self.enc.results_device()
        self.enc.sync()
        rec = pack_records(self._res, b, c)
        if self.world == 1:
            return unpack_records(rec.cpu().numpy())
        full = gather_records(rec, self.spans, self.group, 0)
        if full is None:
            return None
        torch.cuda.synchronize()
        return unpack_records(full.cpu().numpy())

    def close(self):
        self.enc.close()
