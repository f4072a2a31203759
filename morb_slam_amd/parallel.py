"""Multi-GPU front end: frames are the independent unit.  One process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).  A time-ordered stream of
stereo frames is dealt round-robin: global frame g lives on rank g % world, local slot g // world.  Extraction
and the intra-frame stereo match are rank-local; matching a frame against its predecessor (SearchByBoW /
SearchByProjection with the last frame) needs the predecessor's keypoints + descriptors, which live on the
previous rank -> ONE all-gather of fixed-capacity slabs per batch (keypoints 28 B + descriptors 32 B + BoW node
ids 4 B per feature, plus the counts).  xGMI is point-to-point and the slabs are a few MB, so a single
all_gather_into_tensor per array (4 collectives per batch) is used instead of many small messages.

The reference has no counterpart (it is single-process, CPU-only); this layer is new."""
import numpy as np


def global_frame(rank, world, slot):
    """Global (time) index of local slot `slot` on `rank`."""
    return slot * world + rank


def owner(world, g):
    """(rank, slot) holding global frame g."""
    return g % world, g // world


def pool_index(world, slots_per_rank, g):
    """Row of global frame g in the gathered pool (rank-major: all_gather concatenates rank 0's slab first)."""
    r, s = owner(world, g)
    return r * slots_per_rank + s


def predecessor_pairs(rank, world, slots_per_rank):
    """For every local frame: (pool row of its predecessor g-1 acting as keyframe, pool row of the frame itself).
    The very first global frame has no predecessor and is paired with itself."""
    kf, fr = [], []
    for s in range(slots_per_rank):
        g = global_frame(rank, world, s)
        gp = g - 1 if g > 0 else g
        kf.append(pool_index(world, slots_per_rank, gp))
        fr.append(pool_index(world, slots_per_rank, g))
    return np.array(kf, np.int32), np.array(fr, np.int32)


class FeatureExchange:
    """All-gather of the per-frame feature slabs.  Tensors may live on any device the process group supports."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._out = {}

    def _gather(self, name, x):
        import torch
        if self.world == 1:
            return x
        shape = (self.world * x.shape[0],) + tuple(x.shape[1:])
        out = self._out.get(name)
        if out is None or out.shape != shape or out.dtype != x.dtype or out.device != x.device:
            out = torch.empty(shape, dtype=x.dtype, device=x.device)
            self._out[name] = out
        x = x.contiguous()
        if x.device.type == "cpu" or self.dist.get_backend(self.group) != "nccl":
            # gloo has no all_gather_into_tensor for every dtype: gather a list and copy
            parts = [torch.empty_like(x) for _ in range(self.world)]
            self.dist.all_gather(parts, x, group=self.group)
            out.copy_(torch.cat(parts, 0))
        else:
            self.dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    def exchange(self, kps, desc, count, node=None):
        """kps [S, cap, 28] u8, desc [S, cap, 32] u8, count [S] i32, node [S, cap] i32 -> pooled versions
        ([world*S, ...], rank-major)."""
        out = (self._gather("kps", kps), self._gather("desc", desc), self._gather("count", count),
               None if node is None else self._gather("node", node))
        return out
