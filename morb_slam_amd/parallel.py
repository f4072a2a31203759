"""Multi-GPU front end: frames are the independent unit.  One process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).  A time-ordered stream of
stereo frames is dealt round-robin: global frame g lives on rank g % world, local slot g // world.  Extraction
and the intra-frame stereo match are rank-local; matching a frame against its predecessor (SearchByBoW /
SearchByProjection with the last frame) needs the predecessor's keypoints + descriptors, which live on the
PREVIOUS rank only (frame g - 1 of rank r's slot s is rank r - 1's slot s; for rank 0 it is the last rank's slot
s - 1).  So the exchange is a ring shift, not an all-gather: every rank sends its fixed-capacity slabs (keypoints
28 B + descriptors 32 B + BoW node ids 4 B per feature, plus the counts) to rank + 1 and receives rank - 1's —
one point-to-point transfer per xGMI link and step, 1 / world of the bytes an all-gather would move
(`NeighbourExchange`).  `FeatureExchange` (the all-gather of round 1) is kept for callers that need every rank's
features (e.g. loop-closure candidates).

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

    def __init__(self, group=None, always_collective=False):
        """always_collective: run the all-gather even with one rank (tests: the RCCL call path on a one-GPU box)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._out = {}
        self.always_collective = bool(always_collective) and dist.is_initialized()

    def _gather(self, name, x):
        import torch
        if self.world == 1 and not self.always_collective:
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

    def prime(self, device):
        """Create the communicator and run one tiny all-gather NOW (NCCL / RCCL builds its communicator lazily, inside the first collective:
        seconds that would otherwise land in the first step — inside a timed region when the caller asks for no warm-up)."""
        import torch
        if self.world > 1 or self.always_collective:
            self._gather("_prime", torch.zeros((1, 4), dtype=torch.int32, device=device))
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize(device)
            self._out.pop("_prime", None)

    def pooled_bytes(self, S, cap, with_node=True):
        """HBM the pooled arrays of `exchange` take on every rank: world x S rows of keypoints (28 B), descriptors (32 B), node ids (4 B) per feature
        + counts — e.g. 8 ranks x 256 left images x 1264 features = 166 MB, against 41 MB for the ring's [own; received] pool."""
        return self.world * S * (cap * (28 + 32 + (4 if with_node else 0)) + 4)

    def exchange(self, kps, desc, count, node=None):
        """kps [S, cap, 28] u8, desc [S, cap, 32] u8, count [S] i32, node [S, cap] i32 -> pooled versions
        ([world*S, ...], rank-major)."""
        out = (self._gather("kps", kps), self._gather("desc", desc), self._gather("count", count),
               None if node is None else self._gather("node", node))
        return out


def neighbour_pairs(rank, world, slots_per_rank):
    """Pairs for NeighbourExchange's pool = [own slab (rows 0 .. S-1); previous rank's slab (rows S .. 2S-1)]:
    (pool row of the predecessor g - 1 acting as keyframe, pool row of the frame itself).  The very first global frame
    has no predecessor and is paired with itself."""
    S = slots_per_rank
    kf, fr = [], []
    for s in range(S):
        g = global_frame(rank, world, s)
        if g == 0:
            kf.append(s)
        elif world == 1:
            kf.append(s - 1)
        elif rank > 0:
            kf.append(S + s)           # rank - 1, same slot
        else:
            kf.append(S + s - 1)       # last rank, previous slot
        fr.append(s)
    return np.array(kf, np.int32), np.array(fr, np.int32)


class NeighbourExchange:
    """Ring shift of the per-frame feature slabs: send to rank + 1, receive from rank - 1.  `exchange` returns the pool
    [own; received] per array (2 S rows).  nccl (RCCL): batched isend / irecv on the current stream, device to device over
    xGMI.  gloo (tests, or ranks sharing a GPU): staged through host memory."""

    def __init__(self, group=None, matcher=None):
        """matcher: an ORBmatcher — the slabs then travel as ONE contiguous buffer per step (morb_feature_slab_pack / _unpack of the C ABI:
        one transfer per xGMI link instead of four, and the form a C++ caller ships with hipMemcpyPeerAsync); None: one transfer per array."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._pool = {}
        self.matcher = matcher
        self._slab = {}

    def prime(self, device):
        """Create the communicator and run one tiny ring step NOW (see FeatureExchange.prime)."""
        import torch
        if self.world > 1:
            a = torch.zeros((4,), dtype=torch.int32, device=device)
            self._send_recv(a, torch.empty_like(a))
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize(device)

    def _send_recv(self, send, recv):
        """One ring step for one pair of equally sized tensors: send to rank + 1, receive from rank - 1."""
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        if self.dist.get_backend(self.group) == "nccl":
            for w in self.dist.batch_isend_irecv([self.dist.P2POp(self.dist.isend, send, nxt, self.group),
                                                  self.dist.P2POp(self.dist.irecv, recv, prv, self.group)]):
                w.wait()
        else:   # gloo (tests, ranks sharing a GPU): staged through host memory
            src = send.cpu()
            dst = src.new_empty(src.shape)
            reqs = [self.dist.isend(src, nxt, group=self.group), self.dist.irecv(dst, prv, group=self.group)]
            for r in reqs:
                r.wait()
            recv.copy_(dst)

    def _exchange_slab(self, kps, desc, count, node, rows=None):
        import torch
        cap = kps.shape[1]
        if rows is None:
            kps, desc, count = kps.contiguous(), desc.contiguous(), count.contiguous()
            node = None if node is None else node.contiguous()
        S = kps.shape[0] if rows is None else int(rows.shape[0])
        key = (S, cap, kps.device, node is not None)
        st = self._slab.get(key)
        if st is None:
            nb = self.matcher.slab_bytes(S, cap)
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=kps.device)
            st = dict(send=mk((nb,), torch.uint8), recv=mk((nb,), torch.uint8), kps=mk((2 * S, cap, 28), torch.uint8), desc=mk((2 * S, cap, 32), torch.uint8),
                      count=mk((2 * S,), torch.int32), node=mk((2 * S, cap), torch.int32) if node is not None else None,
                      rows=torch.arange(S, 2 * S, dtype=torch.int32, device=kps.device), own=torch.arange(0, S, dtype=torch.int32, device=kps.device))
            self._slab[key] = st
        cs = torch.cuda.current_stream(kps.device).cuda_stream
        # rows `rows` of the caller's arrays (e.g. the left images 0, 2, 4, ...) -> the send slab; the same slab also fills the pool's own half
        self.matcher.pack_slab(kps, desc, count, node, rows=rows, out=st["send"], stream=cs)
        self.matcher.unpack_slab(st["send"], S, st["kps"], st["desc"], st["count"], st["node"], rows=st["own"], stream=cs)
        if self.world == 1:
            st["recv"].copy_(st["send"])
        else:
            self._send_recv(st["send"], st["recv"])
        self.matcher.unpack_slab(st["recv"], S, st["kps"], st["desc"], st["count"], st["node"], rows=st["rows"], stream=cs)
        return st["kps"], st["desc"], st["count"], st["node"]

    def _shift(self, named):
        import torch
        pools = []
        for name, x in named:
            S = x.shape[0]
            shape = (2 * S,) + tuple(x.shape[1:])
            pool = self._pool.get(name)
            if pool is None or pool.shape != shape or pool.dtype != x.dtype or pool.device != x.device:
                pool = torch.empty(shape, dtype=x.dtype, device=x.device)
                self._pool[name] = pool
            pool[:S].copy_(x)
            pools.append(pool)
        if self.world == 1:
            for pool in pools:
                S = pool.shape[0] // 2
                pool[S:].copy_(pool[:S])
            return pools
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        nccl = self.dist.get_backend(self.group) == "nccl"
        if nccl:
            ops = []
            for pool in pools:
                S = pool.shape[0] // 2
                ops.append(self.dist.P2POp(self.dist.isend, pool[:S], nxt, self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, pool[S:], prv, self.group))
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        else:
            for pool in pools:
                S = pool.shape[0] // 2
                src = pool[:S].cpu()
                dst = torch.empty_like(src)
                reqs = [self.dist.isend(src, nxt, group=self.group), self.dist.irecv(dst, prv, group=self.group)]
                for r in reqs:
                    r.wait()
                pool[S:].copy_(dst)
        return pools

    def exchange(self, kps, desc, count, node=None, rows=None):
        """kps [S, cap, 28] u8, desc [S, cap, 32] u8, count [S] i32, node [S, cap] i32 -> pools of 2 S rows: own slab, then
        the previous rank's.  rows (int32 device tensor, slab path only): ship these rows of larger arrays instead of all of them."""
        if self.matcher is not None and kps.is_cuda:
            return self._exchange_slab(kps, desc, count, node, rows)
        assert rows is None, "rows= needs the slab path (NeighbourExchange(matcher=...))"
        named = [("kps", kps), ("desc", desc), ("count", count)] + ([] if node is None else [("node", node)])
        out = self._shift(named)
        return (out[0], out[1], out[2], out[3] if node is not None else None)
