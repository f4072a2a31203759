"""StereoFrontEnd — the per-frame front end as Tracking drives it, batched and pipelined on one GPU:

    Frame::Frame (stereo, Frame.cc:190-226):  ORBextractor x2 (left + right), ComputeStereoMatches
    Frame::ComputeBoW (Frame.cc:822-827):     DBoW2 descent of the LEFT image's descriptors
    Tracking::TrackReferenceKeyFrame (Tracking.cc:2541-2545): ORBmatcher(0.7, true).SearchByBoW(previous frame, frame)

One `step()` = that chain over B stereo frames whose images are already in HBM.  Consecutive steps alternate between NSET
buffer sets (extractor handle with its pyramids, feature tables, matcher outputs), so step i's matchers — latency-bound
kernels — run on their own stream beside step i + 1's extraction; a set is reused only after its readers finished (events).
With a `parallel.NeighbourExchange` / `FeatureExchange` the previous frame's features come from the previous rank.

This is the object `bench.py` times and `tests/test_bench_chain_gpu.py` compares with the oracle: one definition, so the
benched configuration is the tested one.  Everything runs through the C ABI (libmorb_hip.so); there is no CPU path."""
import numpy as np

from . import parallel
from .capi import check, lib
from .extractor import ORBextractor
from .matcher import ORBmatcher


class BufferSet:
    def __init__(self, B, cap, dev):
        import torch
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.out = (e((2 * B, cap, 28), torch.uint8), e((2 * B, cap, 32), torch.uint8), e((2 * B,), torch.int32), e((2 * B,), torch.int32))
        self.st_out = (e((B, cap), torch.float32), e((B, cap), torch.float32))
        self.bow_out = (e((2 * B, cap), torch.int32), e((2 * B, cap), torch.int32))
        self.cnt_left = e((2 * B,), torch.int32)
        self.match_out = None
        self.pool = None          # (kps, desc, count, node) SearchByBoW read: the set's own arrays, or [own; received] slabs
        self.ext_done, self.stereo_done, self.bow_done = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self.exch_t0, self.exch_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # around pack -> transfer -> unpack
        self.used = False
        self.exch_timed = False
        self.ext = None


class StereoFrontEnd:
    def __init__(self, images, nfeatures, B, device=0, rank=0, world=1, nset=2, extract_streams=1, matchers="beside-pyramid",
                 stagger=False, vocab=(10, 6, 4), voc_seed=0, has_mp_seed=7, exchange=None, mbf=458.654 * 0.11, mb=0.11):
        """images: uint8 device tensor [2 B, H, W], image 2 f = left, 2 f + 1 = right of local frame f.
        vocab = (k, L, levelsup) of the synthetic complete k-ary vocabulary (ORBvoc.txt's shape is 10 / 6 / 4; the file is a missing blob).
        exchange: None (one rank: the predecessor is local), "ring" (each rank ships its left-image features to rank + 1 as one slab,
        parallel.NeighbourExchange), "allgather" (every rank's slabs pooled, parallel.FeatureExchange — north_star's wording), or an instance."""
        import torch
        from .synth import make_vocabulary
        self.torch = torch
        self.dev = dev = torch.device("cuda", device)
        self.images, self.B, self.rank, self.world = images, B, rank, world
        self.NSET = NSET = max(1, nset)
        self.exts = [ORBextractor(nfeatures, 1.2, 8, 20, 7, device=device) for _ in range(NSET)]
        self.cap = cap = self.exts[0].max_keypoints
        self.stream = torch.cuda.Stream(device=dev)                                    # extraction
        self.mstream = torch.cuda.Stream(device=dev) if NSET >= 2 else self.stream     # stereo matching
        # extract_streams 1: the sets' extractions follow each other on ONE stream and only the matchers of the previous step run
        # beside them; 2: one extraction stream per set, so two extractions also overlap each other
        self.estreams = [torch.cuda.Stream(device=dev) for _ in range(NSET)] if NSET >= 2 and extract_streams >= 2 else [self.stream] * NSET
        # the BoW chain (ComputeBoW, SearchByBoW) on its own stream beside the stereo matcher: four streams with the extraction's two (HIP
        # multiplexes streams onto 4 hardware queues).  Until round 4 the pipelined front end kept ONE matcher stream; with the extractor's
        # shorter FAST stage the two matcher chains side by side finish inside the next pyramid: 133.1 k against 131.9 k frames/s at B = 512.
        self.bstream = torch.cuda.Stream(device=dev)
        self.matcher = ORBmatcher(0.7, True, device=device)        # TrackReferenceKeyFrame: ORBmatcher(0.7, true), Tracking.cc:2541
        self.bmatcher = ORBmatcher(0.7, True, device=device)       # one workspace set per stream
        self.mbf, self.mb = mbf, mb                                # EuRoC fx * baseline, baseline (Examples/Stereo/EuRoC.yaml)
        self.VK, self.VL, self.VUP = vocab
        vd, vf = make_vocabulary(self.VK, self.VL, seed=voc_seed)
        self.voc_host = (vd, vf)
        self.vd, self.vf = torch.from_numpy(vd).to(dev), torch.from_numpy(vf).to(dev)
        # SearchByBoW pairs: left image of global frame g (as F) against left image of frame g - 1 (as the reference keyframe)
        rng = np.random.default_rng(has_mp_seed)
        if exchange == "ring":
            exchange = parallel.NeighbourExchange(matcher=self.bmatcher)
        elif exchange == "allgather":
            exchange = parallel.FeatureExchange()
        self.exch = exchange
        if exchange is None:
            kf = np.array([2 * max(f - 1, 0) for f in range(B)], np.int32)
            fr = np.array([2 * f for f in range(B)], np.int32)
        elif isinstance(exchange, parallel.FeatureExchange):
            kf, fr = parallel.predecessor_pairs(rank, world, B)
        else:
            kf, fr = parallel.neighbour_pairs(rank, world, B)
        self.kf_host, self.f_host = kf, fr
        self.kf_img, self.f_img = torch.from_numpy(kf).to(dev), torch.from_numpy(fr).to(dev)
        # 80 % of the keyframe features hold a MapPoint; rows = pool rows (2 B with one rank / the ring: [own; received])
        nrows = 2 * B if not isinstance(exchange, parallel.FeatureExchange) else world * B
        self.has_mp_host = (rng.random((nrows, cap)) < 0.8).astype(np.uint8)
        self.has_mp = torch.from_numpy(self.has_mp_host).to(dev)
        self.left_mask = torch.zeros((2 * B,), dtype=torch.int32, device=dev)
        self.left_mask[0::2] = 1
        self.left_rows = torch.arange(0, 2 * B, 2, dtype=torch.int32, device=dev)
        self.sets = [BufferSet(B, cap, dev) for _ in range(NSET)]
        self.nstep = 0
        self.lag_matchers = matchers in ("under-quadtree", "under-fast") and NSET >= 2
        # which event of the NEXT extraction releases a step's matchers: after its FAST stage (they land underneath its quadtree) or after its
        # pyramid (underneath its FAST stage: issue-bound, memory pipe idle)
        self.lag_gate = "event_after_pyramid" if matchers == "under-fast" else "event_after_fast"
        if matchers == "under-fast":
            for e in self.exts:
                e.event_after_pyramid()     # (the extractor records this event only once somebody has asked for it)
        # stagger (with one extraction stream per set): step i + 1's extraction starts when step i's FAST stage is done, so its pyramid
        # — a streaming kernel without LDS — runs beside step i's quadtree, whose few waves per CU hold the LDS and leave the rest idle
        self.stagger = bool(stagger) and self.estreams[0] is not self.estreams[-1]
        self.pending = None
        self.last = None          # the buffer set of the most recent step
        self.ablate = ""
        self.exchange_ms = None   # set to a list by the caller: HIP-event time of every COMPLETED exchange (pack -> transfer -> unpack on bstream)

    def close(self):
        for e in self.exts:
            e.close()
        self.matcher.close()
        self.bmatcher.close()

    def _wait_raw_event(self, stream, event):
        # through libmorb_hip, i.e. the HIP runtime the library is linked against (not a second dlopen of libamdhip64)
        check(lib().morb_stream_wait_event(stream.cuda_stream, event))

    def step(self, src=None, src_ready=None):
        S = self.sets[self.nstep % self.NSET]
        e = self.exts[self.nstep % self.NSET]
        S.ext = e
        stream = self.estreams[self.nstep % self.NSET]
        self.nstep += 1
        if S.used:                                # the set's previous readers (two steps ago) must be done before it is overwritten
            stream.wait_event(S.stereo_done)
            stream.wait_event(S.bow_done)
        S.used = True
        if src_ready is not None:
            stream.wait_event(src_ready)          # (H2D-inclusive variant: the upload of this step's images)
        if self.stagger and self.nstep >= 2:
            self._wait_raw_event(stream, self.exts[(self.nstep - 2) % self.NSET].event_after_fast())
        e.extract_batch(self.images if src is None else src, out=S.out, stream=stream.cuda_stream)         # Frame::ExtractORB x2
        S.ext_done.record(stream)
        self.last = S
        if self.lag_matchers:
            # this step's matchers are queued when the NEXT extraction has been queued, behind its after-FAST event
            if self.pending is not None:
                self.run_matchers(self.pending, gate=getattr(e, self.lag_gate)())
            self.pending = S
        else:
            self.run_matchers(S)
        if self.NSET == 1:                        # un-pipelined: the step ends when all streams are done
            stream.wait_stream(self.bstream)
        return S

    def run_matchers(self, S, gate=None):
        torch = self.torch
        kps, desc, cnt, _ = S.out
        e = S.ext
        mstream, bstream = self.mstream, self.bstream
        mstream.wait_event(S.ext_done)
        ablate = self.ablate            # developer experiments (tools/ablate_matchers.py): "" | "stereo" | "bow" | "all" skipped — never set by bench.py
        if ablate in ("stereo", "all"):
            S.stereo_done.record(mstream)
            if ablate == "all":
                S.bow_done.record(bstream)
                return
        if gate is not None:
            self._wait_raw_event(mstream, gate)
            if bstream is not mstream:
                self._wait_raw_event(bstream, gate)
        if ablate not in ("stereo", "all"):
            self.matcher.ComputeStereoMatches(e, kps, desc, cnt, self.mbf, self.mb, out=S.st_out, stream=mstream.cuda_stream)   # Frame.cc:217
            S.stereo_done.record(mstream)
        if ablate == "bow":
            S.bow_done.record(bstream)
            return
        bs = bstream.cuda_stream
        bstream.wait_event(S.ext_done)
        # Frame::ComputeBoW converts mDescriptors = the LEFT image's descriptors (Frame.cc:822-827): the right images take no part in BoW
        # matching, so their feature count is zeroed for the BoW kernels (which then skip them)
        with torch.cuda.stream(bstream):
            torch.mul(cnt, self.left_mask, out=S.cnt_left)
        self.bmatcher.bow_transform(desc, S.cnt_left, self.vd, self.vf, self.VK, self.VL, self.VUP, out=S.bow_out, stream=bs)   # Frame::ComputeBoW
        if self.exch is None:
            S.pool = (kps, desc, S.cnt_left, S.bow_out[1])
        else:
            with torch.cuda.stream(bstream):      # the transfer is ordered after the kernels on this stream
                if self.exchange_ms is not None and S.exch_timed and S.exch_t1.query():   # the set's previous exchange (two steps ago), if it has finished
                    self.exchange_ms.append(S.exch_t0.elapsed_time(S.exch_t1))
                S.exch_t0.record(bstream)
                if getattr(self.exch, "matcher", None) is not None:   # left images only, gathered into ONE slab by morb_feature_slab_pack
                    pk, pd, pc, pn = self.exch.exchange(kps, desc, cnt, S.bow_out[1], rows=self.left_rows)
                else:
                    pk, pd, pc, pn = self.exch.exchange(kps[0::2], desc[0::2], cnt[0::2], S.bow_out[1][0::2])
                S.exch_t1.record(bstream)
                S.exch_timed = True
            S.pool = (pk, pd, pc, pn)
        pk, pd, pc, pn = S.pool
        S.match_out = self.bmatcher.SearchByBoW(self.kf_img, self.f_img, pk, pd, pn, pc, self.has_mp, out=S.match_out, stream=bs)
        S.bow_done.record(bstream)

    def flush_matchers(self):
        """(matchers under-quadtree) the last step's matchers, which no later extraction gates"""
        if self.pending is not None:
            self.run_matchers(self.pending)
            self.pending = None

    def drain_exchange_ms(self):
        """After sync(): the exchanges still held by the buffer sets' events (the last NSET steps)."""
        if self.exchange_ms is not None:
            for S in self.sets:
                if S.exch_timed:
                    self.exchange_ms.append(S.exch_t0.elapsed_time(S.exch_t1)); S.exch_timed = False
        return self.exchange_ms

    def sync(self):
        self.flush_matchers()   # (every queued step's matchers lie inside the region the caller is closing)
        for es in self.estreams:
            es.synchronize()
        self.mstream.synchronize()
        self.bstream.synchronize()
        self.stream.synchronize()
        self.torch.cuda.synchronize(self.dev)
        for e in self.exts:
            e.check_status()      # an extraction flagged on the device must not be consumed silently
