// k_fastw: per-cell FAST-9/16 + cornerScore + 3x3 NMS with the iniThFAST -> minThFAST fallback
// (ORBextractor.cc:763-820 calling cv::FAST once or twice per 35-px cell), one WAVE per segment.
//
// Round 2's k_fast gave a 256-thread workgroup to every segment of <= 3 cells and walked it through five barrier-separated
// phases; its own phase table said the kernel took the same time without its global loads: it was bound by barrier-to-barrier
// latency and by the instructions all four waves spend around phases that one wave executes.  Here a wave owns its segment
// from the window load to the ordered candidate lists: a private LDS region, no workgroup barrier anywhere (the four waves of a
// workgroup never synchronise with each other), wave-uniform bookkeeping in scalar registers instead of LDS words and atomics.
//   load      window (<= P px wide, hCell + 6 rows) -> LDS, 16 bytes per access (global: unaligned; LDS: aligned); the strength
//             map is zeroed by the same lanes.
//   reject    one lane = 16 consecutive pixels of a window row, packed-u16 SWAR.  Every 9-arc of the 16-pixel ring contains at
//             least one pixel of each antipodal pair, so a pixel can reach strength > T only if max(min(p0, p8), min(p4, p12)) <
//             v - T (dark) or min(max(p0, p8), max(p4, p12)) > v + T (bright) — stronger than round 2's "second smallest of the
//             four compass pixels" and two packed operations shorter.  Flags outside the evaluated columns are masked before the
//             survivors are queued (a single cell evaluates 35 of its 48 loaded columns).
//   strength  runs whenever 64 survivors are queued (all lanes busy but for the last round of a job): exact max-min over the 16
//             arcs (v_min3 / v_max3 trees), only for the polarity the reject left possible.
//   nms       one corner per lane from the job's corner list; dense fallback over the strength map when the list overflows.
//   output    row-major inside each cell (cv::FAST's order): rank by broadcast, or bitmap prefix sums for > 64 keypoints.
// A cell without a keypoint after the iniThFAST pass is evaluated again with minThFAST (:795), alone.
#pragma once

#ifdef MORB_FAST_TIMING
// dynamic phase counts of k_fastw (tools/fastw_stats.py): 0 waves, 1 jobs, 2 reject rounds, 3 emit loop trips, 4 survivors, 5 strength rounds,
// 6 corners, 7 nms rounds, 8 keypoints, 9 output rank trips, 10 both-polarity rounds, 11 fallback jobs, 12 half-round splits, 13 dense nms
__device__ unsigned long long g_fwStat[16];
extern "C" int morb_fw_stats(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; MORB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fwStat), z, sizeof(z))); return 0; }
  MORB_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwStat), 16 * sizeof(unsigned long long)));
  return 0;
}
#define FW_STAT(k, v) do { if (lane == 0) atomicAdd(&g_fwStat[k], (unsigned long long)(v)); } while (0)
#else
#define FW_STAT(k, v)
#endif
#define FW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

#ifndef MORB_FW_PADLDS
#define MORB_FW_PADLDS 0   // (occupancy experiments: extra LDS bytes per wave)
#endif
constexpr int FW_QCAP = 576;   // survivor queue: < 64 left over + 512 of a reject round's flags (a fuller round is queued in several pieces)
constexpr int FW_KC = 64;      // keypoint list of a cell (typically ~6); more -> bitmap output
constexpr int FW_WAVES = 4;    // segments (waves) per workgroup
template <int P> struct FwCfg {
  static constexpr int BPR = P / 16;                               // 16-px blocks per window row
  static constexpr int BW = (P + 31) / 32;                         // keypoint-bitmap words per window row
  static constexpr int CQ = P <= 48 ? 256 : (P <= 96 ? 384 : 512); // corner list of a job (~50 per cell on the benchmark images)
};
template <int P> __host__ __device__ constexpr int fw_region_bytes(int rows) {   // LDS of one wave
  return (rows * P + 16) + rows * P + ((rows * FwCfg<P>::BW * 4 + 15) & ~15) + FW_QCAP * 2 + FwCfg<P>::CQ * 2 + 3 * FW_KC * 4 + MORB_FW_PADLDS;
}
// pixels [0, o) of a 16-px block as a mask in the reject's flag layout: pixel o -> bits f, f + 1 (dark, bright), f = o[0] << 1 | o[2] << 2 | o[3] << 3 | o[1] << 4
struct FwPixMask { unsigned m[17]; constexpr FwPixMask() : m() { unsigned a = 0; for (int o = 0; o < 16; ++o) { m[o] = a; a |= 3u << (((o & 1) << 1) | (o & 4) | (o & 8) | ((o & 2) << 3)); } m[16] = a; } };
__constant__ FwPixMask c_fwPixMask = FwPixMask();

template <int P>
__global__ __launch_bounds__(64 * FW_WAVES) void k_fastw(const morb::FastGeom fg, const morb::FastSeg* __restrict__ segTab, int nSeg,
                                                         const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                                         int* __restrict__ candCnt, int totalCells, int cellCap, int rows, int iniTh, int minTh) {
  using C = FwCfg<P>;
  extern __shared__ __align__(16) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = blockIdx.y * FW_WAVES + wv, img = blockIdx.x;
  if (seg >= nSeg) return;
  uint8_t* tile = smem + wv * fw_region_bytes<P>(rows);                    // [rows][P] pixels (+16 bytes: the last block's right neighbour)
  uint8_t* sc = tile + rows * P + 16;                                      // [rows][P] strength S of corners (S > the cell's threshold), else 0
  uint32_t* keepBm = reinterpret_cast<uint32_t*>(sc + rows * P);           // [rows][BW] keypoints (after NMS)
  uint16_t* queue = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(keepBm) + ((rows * C::BW * 4 + 15) & ~15));   // bright << 14 | y << 7 | x
  uint16_t* cornerQ = queue + FW_QCAP;                                     // y << 7 | x
  uint32_t* kept = reinterpret_cast<uint32_t*>(cornerQ + C::CQ);           // [3][FW_KC] S << 16 | y << 7 | x

  const morb::FastSeg sd = segTab[seg];
  const int l = sd.geo & 0xFF, nc = (sd.geo >> 8) & 0xFF, tw = (sd.geo >> 16) & 0xFF, th = (int)((unsigned)sd.geo >> 24);
  const int wCell = fg.wCell[l], pstride = fg.pstride[l];
  const unsigned wMagic = fg.wCellMagic[l];
  const size_t cellSlot0 = (size_t)img * totalCells + sd.cell0;
  if (tw <= 6 || th <= 6) {   // :770, :775: skipped cells, or windows cv::FAST finds nothing in
    if (lane < nc) candCnt[cellSlot0 + lane] = 0;
    return;
  }
  {
    const uint8_t* base = pyr + fg.pyrOff[l] + (size_t)img * fg.pyrImg[l] + sd.winOff;
    const int n16 = th * C::BPR;
    for (int i0 = 0; i0 < n16; i0 += 128) {
      uint4 v[2]; int off[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + lane + k * 64;
        const int rr = i / C::BPR, c16 = (i - rr * C::BPR) << 4;
        off[k] = i < n16 ? rr * P + c16 : -1;
        v[k] = make_uint4(0, 0, 0, 0);
        if (i < n16) __builtin_memcpy(&v[k], base + (unsigned)(__umul24(rr, pstride) + c16), 16);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (off[k] >= 0) {
          *reinterpret_cast<uint4*>(tile + off[k]) = v[k];
          *reinterpret_cast<uint4*>(sc + off[k]) = make_uint4(0, 0, 0, 0);
        }
    }
    for (int i = lane; i < th * C::BW; i += 64) keepBm[i] = 0;
    if (lane < 4) *reinterpret_cast<uint32_t*>(tile + rows * P + 4 * lane) = 0;
  }
  FW_SYNC();
  FW_STAT(0, 1);

  // wave-uniform state
  int keptN[3] = {0, 0, 0};        // keypoints of each cell (list length, or a bitmap count)
  bool useBm[3] = {false, false, false};   // the cell's keypoints are read from the bitmap, not from its list
  const int keyX0 = sd.key0 & 0xFFFF, keyY0 = sd.key0 >> 16;

  // keypoints of row y inside cell j (the cell's evaluated columns)
  auto row_cell_count = [&](int y, int j) -> int {
    const int a = 3 + j * wCell, b = imin(a + wCell, tw - 3);
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < C::BW; ++w) cnt += __popc(keepBm[y * C::BW + w] & range_mask(a - 32 * w, b - 32 * w));
    return cnt;
  };

  for (int job = 0; job <= nc; ++job) {
    // job 0: cv::FAST(iniThFAST) on every cell of the segment; job j >= 1: cell j - 1 again with minThFAST if it is empty (:795)
    int xa = 3, xb = tw - 3, T = iniTh;
    if (job > 0) {
      const int j = job - 1;
      if ((j == 0 ? keptN[0] : (j == 1 ? keptN[1] : keptN[2])) != 0) continue;   // (selects: a run-time index would put the array in scratch)
      xa = 3 + j * wCell; xb = imin(xa + wCell, tw - 3); T = minTh;
      if (xb <= xa) continue;
    }
    int qn = 0, cn = 0;
    FW_STAT(1, 1); FW_STAT(11, job > 0);
    {
      const int ix0 = xa >> 4, nIt = ((xb + 15) >> 4) - ix0;
      const unsigned itMagic = c_magic20.m[nIt];
      const int nItems = (th - 6) * nIt;
      // flags of pixels outside [xa, xb) in a row's first / last block are dropped before they are queued
      const unsigned mFirst = ~c_fwPixMask.m[xa & 15], mLast = c_fwPixMask.m[((xb - 1) & 15) + 1];
      const unsigned LO = 0x00FF00FFu;
      unsigned KF[8], MF[8];   // per (dword & 1, parity, polarity): the add constant and the flag bit (8 + index) in both halves
#pragma unroll
      for (int j = 0; j < 8; ++j) { KF[j] = ((1u << (8 + j)) - 1u - (unsigned)T) * 0x00010001u; MF[j] = (1u << (8 + j)) * 0x00010001u; }
      for (int i0 = 0; i0 < nItems; i0 += 64) {
        const int i = i0 + lane;
        const int iy = (int)(((unsigned)i * itMagic) >> 20);
        const int bi = i - __mul24(iy, nIt);
        const int y = iy + 3, xb0 = (bi + ix0) << 4;
        unsigned W = 0;   // bit f: f[0] polarity (0 dark, 1 bright), pixel offset in the block = f[3] f[2] f[4] f[1]
        if (i < nItems) {
          const uint8_t* rowp = tile + (__mul24(y, P) + xb0);
          const uint4 Cc = *reinterpret_cast<const uint4*>(rowp);
          const uint4 U = *reinterpret_cast<const uint4*>(rowp - 3 * P);   // ring pixel 8 (0,-3)
          const uint4 D = *reinterpret_cast<const uint4*>(rowp + 3 * P);   // ring pixel 0 (0,+3)
          const uint32_t Lw = *reinterpret_cast<const uint32_t*>(rowp - 4), Rw = *reinterpret_cast<const uint32_t*>(rowp + 16);
          const uint32_t Cw[6] = {Lw, Cc.x, Cc.y, Cc.z, Cc.w, Rw}, Uw[4] = {U.x, U.y, U.z, U.w}, Dw[4] = {D.x, D.y, D.z, D.w};
          unsigned E[6], O[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) { E[k] = Cw[k] & LO; O[k] = (Cw[k] >> 8) & LO; }
          unsigned acc[2] = {0u, 0u};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
              const unsigned Ve = par ? O[k + 1] : E[k + 1];
              const unsigned a = (par ? (Uw[k] >> 8) : Uw[k]) & LO, b = (par ? (Dw[k] >> 8) : Dw[k]) & LO;
              // ring pixels 12 (-3,0) and 4 (+3,0)
              const unsigned cc = par ? __builtin_amdgcn_alignbit(E[k + 1], E[k], 16) : O[k];
              const unsigned d = par ? E[k + 2] : __builtin_amdgcn_alignbit(O[k + 2], O[k + 1], 16);
              const unsigned mlo = pk_max(pk_min(a, b), pk_min(cc, d));   // every 9-arc holds one of {0, 8} and one of {4, 12}
              const unsigned mhi = pk_min(pk_max(a, b), pk_max(cc, d));
              const int j = (k & 1) * 4 + par * 2;
              acc[k >> 1] |= pk_add(pk_sub_sat(Ve, mlo), KF[j]) & MF[j];           // v - mlo > T
              acc[k >> 1] |= pk_add(pk_sub_sat(mhi, Ve), KF[j + 1]) & MF[j + 1];   // mhi - v > T
            }
          }
          W = ((acc[0] >> 8) & 0x00FF00FFu) | (acc[1] & 0xFF00FF00u);
          if (bi == 0) W &= mFirst;
          if (bi == nIt - 1) W &= mLast;
        }
        FW_STAT(2, 1);
        const bool lastRound = i0 + 64 >= nItems;
        // one queue entry per flag: a pixel both of whose polarities are still possible is queued twice — a darker and a brighter 9-arc
        // cannot coexist on a 16-pixel ring, so at most one of the two entries finds a strength > T and they never write the same byte
        unsigned pend = W;
        for (;;) {
          // compaction of the wave's flag words: per-lane popcount, one DPP scan, every lane emits its own entries
          unsigned take = pend;
          int cnt = __popc(take);
          int incl = cnt;
          MORB_DPP_SCAN(incl, 0, morbwave::op_add);   // inclusive prefix over the wave (all lanes active)
          int total = __builtin_amdgcn_readlane(incl, 63);
          if (total > FW_QCAP - qn) {   // (wave-uniform, rare: qn < 64 here, so more than 512 flags in one round) queue a prefix of the lanes now
            FW_STAT(12, 1);
            const bool fits = incl <= FW_QCAP - qn;   // true for at least lane 0: a lane holds at most 32 flags
            total = __builtin_amdgcn_readlane(incl, __popcll(__ballot(fits)) - 1);
            take = fits ? pend : 0u;
            cnt = fits ? cnt : 0;
          }
          pend &= ~take;
          if (total) {
#ifdef MORB_FAST_TIMING
            { const unsigned mx = ~morbwave::min_u32(~(unsigned)cnt); FW_STAT(3, mx); FW_STAT(4, total); }
#endif
            int slot = qn + incl - cnt;
            const unsigned pos0 = (unsigned)((y << 7) | xb0);
            unsigned t = take;
            while (t) {
              const unsigned f = (unsigned)__ffs(t) - 1u;
              t &= t - 1u;
              const unsigned o = ((f >> 1) & 1u) | ((f >> 3) & 2u) | (f & 12u);
              queue[slot++] = (uint16_t)(((f & 1u) << 14) | (pos0 + o));
            }
            qn += total;
            FW_SYNC();
          }
          const bool more = __ballot(pend != 0u) != 0ull;   // (only after a half round)
          const bool flush = lastRound && !more;
          // strength of the queued pixels, 64 at a time (fewer only when the job's last survivors are flushed)
          while (qn >= 64 || (flush && qn > 0)) {
            FW_STAT(5, 1);
            const int n = imin(qn, 64), q0 = qn - n;
            qn = q0;
            const bool act = lane < n;
            const unsigned e = act ? queue[q0 + lane] : 0u;
            const int x = (int)(e & 127u), yy = (int)((e >> 7) & 127u);
            const int off = __mul24(yy, P) + x;
            // every ring offset relative to the ring's top-left corner is non-negative: one address, immediates only
            constexpr int O3 = 3 * P + 3;
            int cornerOff = act ? off - O3 : 0;   // (evaluated pixels have x, y >= 3)
            asm volatile("" : "+v"(cornerOff));   // (opaque: otherwise the address is re-based on the centre and 7 offsets need their own add)
            const uint8_t* p = tile + cornerOff;
            const int v = p[O3];
            int rr[16];
            rr[0] = p[O3 + 3 * P];   rr[1] = p[O3 + 3 * P + 1];  rr[2] = p[O3 + 2 * P + 2];  rr[3] = p[O3 + P + 3];
            rr[4] = p[O3 + 3];       rr[5] = p[O3 - P + 3];      rr[6] = p[O3 - 2 * P + 2];  rr[7] = p[O3 - 3 * P + 1];
            rr[8] = p[O3 - 3 * P];   rr[9] = p[O3 - 3 * P - 1];  rr[10] = p[O3 - 2 * P - 2]; rr[11] = p[O3 - P - 3];
            rr[12] = p[O3 - 3];      rr[13] = p[O3 + P - 3];     rr[14] = p[O3 + 2 * P - 2]; rr[15] = p[O3 + 3 * P - 1];
            // d = v - p for the dark polarity (= ~p + v + 1), p - v for the bright one: (p ^ m) + c, one v_xad_u32 per ring pixel
            const bool bright = (e & 0x4000u) != 0u;
            const int xm = bright ? 0 : -1, xc = bright ? -v : v + 1;
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = (rr[k] ^ xm) + xc;
            const int S = arc9_maxmin(d);
            const bool isCorner = act && S > T;
            if (isCorner) sc[off] = (uint8_t)imin(S, 255);
            const uint64_t cm = __ballot(isCorner);
            if (cm) {   // wave-uniform
              const int idx = cn + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
              if (isCorner && idx < C::CQ) cornerQ[idx] = (uint16_t)(e & 0x3FFFu);
              cn += __popcll(cm);
            }
          }
          if (!more) break;
        }
      }
    }
    FW_SYNC();
    // NMS.  The strength map holds S for corners (S > T) and 0 elsewhere: corner score S - 1, everything else 0; keep iff
    // strictly greater than all 8 neighbours' scores, where a neighbour outside the pixel's own cell (or outside the
    // evaluated area, where the map stays 0) counts as 0.
    auto nms_keep = [&](int x, int y, int S, int* cellOut) -> bool {
      const uint8_t* c = sc + (__mul24(y, P) + x);
      const int cj = (int)(((unsigned)(x - 3) * wMagic) >> 16), cxa = 3 + __mul24(cj, wCell);
      *cellOut = cj;
      int m = imax(c[-P], c[P]);
      if (x != cxa) m = imax(m, imax(imax(c[-P - 1], c[-1]), c[P - 1]));
      if (x != cxa + wCell - 1) m = imax(m, imax(imax(c[-P + 1], c[1]), c[P + 1]));
      return S > imax(m, 1);
    };
    const bool dense = cn > C::CQ;   // corner list overflowed
    FW_STAT(6, cn); FW_STAT(13, dense);
    int newN[3] = {0, 0, 0};
    if (!dense) {   // one corner per lane
      for (int q0 = 0; q0 < cn; q0 += 64) {
        FW_STAT(7, 1);
        const int q = q0 + lane;
        const bool act = q < cn;
        const int pos = act ? cornerQ[q] : ((3 << 7) | 3), x = pos & 127, y = pos >> 7;
        const int S = sc[__mul24(y, P) + x];
        int cj;
        const bool keep = nms_keep(x, y, S, &cj) && act;
        if (keep) atomicOr(&keepBm[y * C::BW + (x >> 5)], 1u << (x & 31));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (j < nc) {   // (uniform)
            const uint64_t km = __ballot(keep && cj == j);
            if (km) {
              const int idx = newN[j] + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
              if (keep && cj == j && idx < FW_KC) kept[j * FW_KC + idx] = (uint32_t)pos | ((uint32_t)S << 16);
              newN[j] += __popcll(km);
            }
          }
        }
      }
    } else {   // one dword of the strength map per lane
      const int dpr = P / 4;
      for (int i0 = 0; i0 < (th - 6) * dpr; i0 += 64) {
        const int i = i0 + lane;
        const int iy = i / dpr, y = iy + 3, x0 = (i - iy * dpr) << 2;
        uint32_t word = i < (th - 6) * dpr ? *reinterpret_cast<const uint32_t*>(sc + y * P + x0) : 0u;
        for (int b = 0; b < 4; ++b) {
          const int S = (word >> (8 * b)) & 0xFF, x = x0 + b;
          if (S && x >= xa && x < xb) {
            int cj;
            if (nms_keep(x, y, S, &cj)) atomicOr(&keepBm[y * C::BW + (x >> 5)], 1u << (x & 31));
          }
        }
      }
    }
    FW_SYNC();
    // keypoints per cell so far (decides which cells run again)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < nc && (job == 0 || job == j + 1)) {
        if (dense || newN[j] > FW_KC) {
          int tot = 0;
          for (int y0 = 0; y0 < th; y0 += 64) tot += y0 + lane < th ? row_cell_count(y0 + lane, j) : 0;
          keptN[j] = morbwave::sum_i32(tot);
          useBm[j] = true;
        } else {
          keptN[j] = newN[j];
        }
      }
    }
    if (job == 0) {
      // a second cv::FAST call starts from nothing: forget the first pass's strengths in the cells that run again (they exist when
      // a cell's corners tied each other out in the NMS, and matter when minThFAST > iniThFAST)
      bool again = false;
#pragma unroll
      for (int j = 0; j < 3; ++j) again |= j < nc && keptN[j] == 0;
      if (again && cn > 0) {
        if (!dense) {
          for (int q = lane; q < cn; q += 64) {
            const int pos = cornerQ[q], x = pos & 127, y = pos >> 7;
            const int cj = (int)(((unsigned)(x - 3) * wMagic) >> 16);
            const int kn = cj == 0 ? keptN[0] : (cj == 1 ? keptN[1] : keptN[2]);
            if (kn == 0) sc[__mul24(y, P) + x] = 0;
          }
        } else {
          for (int i = lane; i < (th - 6) * P; i += 64) {
            const int iy = i / P, y = iy + 3, x = i - iy * P;
            if (x >= 3 && x < tw - 3) {
              const int cj = (int)(((unsigned)(x - 3) * wMagic) >> 16);
              const int kn = cj == 0 ? keptN[0] : (cj == 1 ? keptN[1] : keptN[2]);
              if (kn == 0) sc[y * P + x] = 0;
            }
          }
        }
        FW_SYNC();
      }
    }
  }
  // ordered output, row-major inside each cell
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j >= nc) break;
    const int n = keptN[j];
    FW_STAT(8, n); FW_STAT(9, useBm[j] ? 0 : n);
    uint32_t* out = cand + (cellSlot0 + j) * (size_t)cellCap;
    if (!useBm[j]) {
      // the cell's keypoints are a short unordered list: a keypoint's slot is the number of keypoints before it in row-major
      // order, counted against the list broadcast lane by lane
      const uint32_t mine = lane < n ? kept[j * FW_KC + lane] : 0xFFFFFFFFu;
      const int mpos = (int)(mine & 0xFFFFu);
      int rank = 0;
      for (int k = 0; k < n; ++k) rank += (__builtin_amdgcn_readlane(mpos, k) < mpos) ? 1 : 0;
      if (lane < n && rank < cellCap)
        out[rank] = morbqt::make_key((mpos & 127) + keyX0, (mpos >> 7) + keyY0, (int)(mine >> 16) - 1);
      if (lane == 0) candCnt[cellSlot0 + j] = imin(n, cellCap);
    } else {
      const int a = 3 + j * wCell, b = imin(a + wCell, tw - 3);
      int running = 0;
      for (int y0 = 0; y0 < th; y0 += 64) {
        const int y = y0 + lane;
        const int c = y < th ? row_cell_count(y, j) : 0;
        int inc = c;
        MORB_DPP_SCAN(inc, 0, morbwave::op_add);
        int slot = running + inc - c;
        running += __builtin_amdgcn_readlane(inc, 63);
        if (c) {
#pragma unroll
          for (int w = 0; w < C::BW; ++w) {
            uint32_t word = keepBm[y * C::BW + w] & range_mask(a - 32 * w, b - 32 * w);
            while (word) {
              const int bb = __ffs(word) - 1;
              word &= word - 1;
              const int x = w * 32 + bb;
              if (slot < cellCap) out[slot] = morbqt::make_key(x + keyX0, y + keyY0, sc[y * P + x] - 1);
              ++slot;
            }
          }
        }
      }
      if (lane == 0) candCnt[cellSlot0 + j] = imin(running, cellCap);
    }
  }
}
