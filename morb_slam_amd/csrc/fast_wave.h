// k_fastw: per-cell FAST-9/16 + cornerScore + 3x3 NMS with the iniThFAST -> minThFAST fallback
// (ORBextractor.cc:763-820 calling cv::FAST once or twice per 35-px cell), one WAVE per cell.
//
// Round 2's k_fast gave a 256-thread workgroup to every segment of <= 3 cells and walked it through five barrier-separated
// phases; its own phase table said the kernel took the same time without its global loads: it was bound by barrier-to-barrier
// latency and by the instructions all four waves spend around phases that one wave executes.  Here a wave owns its cell from
// the window load to the ordered candidate list: a private LDS region, no workgroup barrier anywhere (the four waves of a
// workgroup never synchronise with each other), wave-uniform bookkeeping in scalar registers instead of LDS words and atomics.
// The kernel answers to occupancy (measured with padded LDS: 12 / 16 / 20 waves per CU -> 1049 / 908 / 811 us per 512 images),
// so a wave's LDS is cut to ~4.6 KB (32 waves per CU): ONE window buffer, which holds the pixels while strengths are computed
// and the corners' strengths afterwards.
//   load      window (<= P px wide, hCell + 6 rows) -> LDS, 16 bytes per access (global: unaligned; LDS: aligned)
//   reject    one lane = 12 consecutive pixels of a window row (round 4; 16 before), packed-u16 SWAR.  Every 9-arc of the 16-pixel ring contains at
//             least one pixel of each antipodal pair, so a pixel can reach strength > T only if max(min(p0, p8), min(p4, p12)) <
//             v - T (dark) or min(max(p0, p8), max(p4, p12)) > v + T (bright) — stronger than round 2's "second smallest of the
//             four compass pixels" and two packed operations shorter.  Flags outside the evaluated columns are masked, the rest
//             are queued one entry per (pixel, polarity).
//   strength  runs whenever 64 survivors are queued (all lanes busy but for the last round of a pass): exact max-min over the 16
//             arcs (v_min3 / v_max3 trees) for the entry's polarity; corners (S > T) go to a list (position, S).
//   nms       the window is zeroed and the corners' strengths scattered into it; one corner per lane looks at its 8 neighbours.
//   output    row-major (cv::FAST's order): a keypoint's slot = the number of keypoints before it, counted by broadcast.
// More than 512 corners or more than 64 keypoints in one cell (noise images): the pass is redone row by row with a rolling
// four-row strength buffer (strip mode: slow, exact, bounded LDS).
// A cell without a keypoint after the iniThFAST pass is evaluated again with minThFAST (:795).
#pragma once

#ifdef MORB_FAST_CYCLES
// Per-phase shader cycles of a wave's residency (tools/fastw_cycles.py): s_memtime at the phase boundaries (the wait makes pending LDS
// traffic part of the phase that issued it): 0 load, 1 reject, 2 emit (compaction into the queue), 3 strength, 4 nms (zero + scatter + 3x3),
// 5 output, 6 strip mode, 7 lifetime.  Accumulated in registers and written once, when the wave is done, to the wave's own 64 bytes of a
// caller-provided buffer (atomics — even hashed over 64 slots — perturb the kernel: issued inside the phases they sit in front of the next
// window load in the vector-memory queue; issued at the end they keep finished waves resident).
__device__ unsigned long long* g_fwCycBuf;
extern "C" int morb_fw_cycles_buffer(unsigned long long* d_buf) {   // [images x totalCells][8]
  MORB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fwCycBuf), &d_buf, sizeof(d_buf)));
  return 0;
}
__device__ __forceinline__ unsigned long long fw_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
#define FW_CYC0() unsigned long long tc_ = fw_now(), acc_[7] = {0, 0, 0, 0, 0, 0, 0}; const unsigned long long tc0_ = tc_
#define FW_CYC(k) do { const unsigned long long now_ = fw_now(); acc_[k] += now_ - tc_; tc_ = now_; } while (0)
#define FW_CYC_END() do { const unsigned long long now_ = fw_now(); if (lane == 0 && g_fwCycBuf) { unsigned long long* g_ = g_fwCycBuf + cellSlot * 8; \
    for (int k_ = 0; k_ < 7; ++k_) g_[k_] = acc_[k_]; g_[7] = now_ - tc0_; } } while (0)
#else
#define FW_CYC0()
#define FW_CYC(k)
#define FW_CYC_END()
#endif
#ifdef MORB_FAST_TIMING
// dynamic phase counts of k_fastw (tools/fastw_stats.py): 0 waves, 1 jobs, 2 reject rounds, 3 emit loop trips, 4 survivors, 5 strength rounds,
// 6 corners, 7 nms rounds, 8 keypoints, 9 output rank trips, 10 -, 11 minThFAST passes, 12 partial queue takes, 13 strip-mode passes
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
constexpr int FW_SH = 1;       // window column c = tile column c + FW_SH
constexpr int FW_QCAP = 320;   // survivor queue: < 64 left over + 256 of a reject round's flags (a fuller round is queued in several pieces)
constexpr int FW_CQ = 512;     // corner list of a pass (~50 per cell on the benchmark images); more -> strip mode
constexpr int FW_KC = 64;      // keypoint list of a cell (~10); more -> strip mode
#ifndef MORB_FW_WAVES
#define MORB_FW_WAVES 4
#endif
constexpr int FW_WAVES = MORB_FW_WAVES;    // waves per workgroup
#ifndef MORB_FW_CPW
#define MORB_FW_CPW 1
#endif
constexpr int FW_CPW = MORB_FW_CPW;        // cells a wave works through, one after the other
// A queue entry = (tile offset of the item's first pixel) / 4 << 5 | flag index: 16 bits while the tile has fewer than 1024 dwords (P = 48, up to
// 85 rows — every cell of the usual 35-px grid); the wider tiles of unusual cell sizes take 32-bit entries.
template <int P> struct FwQueueEntry { typedef uint32_t type; };
template <> struct FwQueueEntry<48> { typedef uint16_t type; };
constexpr int FW_ROWS16 = 85;   // rows a P = 48 tile may have (85 * 48 / 4 < 1024)
template <int P> __host__ __device__ constexpr int fw_region_bytes(int rows) {   // LDS of one wave
  return (rows * P + 16) + FW_QCAP * (int)sizeof(typename FwQueueEntry<P>::type) + FW_CQ * 2 + FW_CQ + FW_KC * 4 + MORB_FW_PADLDS;
}
// pixels [0, o) of a 16-px block as a mask in the reject's flag layout: pixel o -> bits f, f + 1 (dark, bright), f = o[0] << 1 | o[2] << 2 | o[3] << 3 | o[1] << 4
struct FwPixMask { unsigned m[17]; constexpr FwPixMask() : m() { unsigned a = 0; for (int o = 0; o < 16; ++o) { m[o] = a; a |= 3u << (((o & 1) << 1) | (o & 4) | (o & 8) | ((o & 2) << 3)); } m[16] = a; } };
__constant__ FwPixMask c_fwPixMask = FwPixMask();

template <int P>
// (the wide tiles of unusual cell sizes hold fewer waves per CU by their LDS anyway: 80 registers there, no spill with the 32-bit queue entries)
__global__ __launch_bounds__(64 * FW_WAVES, P == 48 ? ((FW_WAVES * 8 + 3) / 4 > 8 ? 8 : (FW_WAVES * 8 + 3) / 4) : 6) void k_fastw(const morb::FastGeom fg, const morb::FastSeg* __restrict__ segTab, int nSeg,
                                                            const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                                            int* __restrict__ candCnt, int totalCells, int cellCap, int rows, int iniTh, int minTh) {
  constexpr int BPR = P / 16;   // 16-px blocks per window row
  constexpr unsigned BPR_MAGIC = 65536u / BPR + 1u;   // i / BPR == (i * BPR_MAGIC) >> 16 for i < 16384 (24-bit multiply: the compiler's own /3 is two quarter-rate v_mul_hi_u32)
  extern __shared__ __align__(16) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // a wave works through cells seg0, seg0 + stride, ... (the launch sizes the stride: FW_CPW cells per wave)
  const int seg0 = blockIdx.y * FW_WAVES + wv, segStride = gridDim.y * FW_WAVES, img = blockIdx.x;
  if (seg0 >= nSeg) return;
  uint8_t* tile = smem + wv * fw_region_bytes<P>(rows);                     // [rows][P] pixels, later the corners' strengths (+16 bytes: the last block's right neighbour)
  typedef typename FwQueueEntry<P>::type QE;
  QE* queue = reinterpret_cast<QE*>(tile + rows * P + 16);                  // (tile offset of the item) / 4 << 5 | flag index
  uint16_t* cornerPos = reinterpret_cast<uint16_t*>(queue + FW_QCAP);       // tile offset y * P + x
  uint8_t* cornerS = reinterpret_cast<uint8_t*>(cornerPos + FW_CQ);         // S (<= 255)
  uint32_t* kept = reinterpret_cast<uint32_t*>(cornerS + FW_CQ);            // S << 16 | y << 7 | x
  uint8_t* sbuf = reinterpret_cast<uint8_t*>(cornerPos);                    // strip mode: four rolling rows of strengths

  for (int seg = seg0; seg < nSeg; seg += segStride) {
  const morb::FastSeg sd = segTab[seg];   // (one cell per segment: the host builds k_fastw's table that way)
  const int l = sd.geo & 0xFF, tw = (sd.geo >> 16) & 0xFF, th = (int)((unsigned)sd.geo >> 24);
  const int pstride = fg.pstride[l];
  const size_t cellSlot = (size_t)img * totalCells + sd.cell0;
  if (tw <= 6 || th <= 6) {   // :770, :775: skipped cells, or windows cv::FAST finds nothing in
    if (lane == 0) candCnt[cellSlot] = 0;
    continue;
  }
  // The tile holds the window from one column to its left (FW_SH = 1): the first evaluated column, window column 3, is tile column 4 — dword
  // aligned — and the reject's items are 12 pixels (three dwords) starting there: the 35 evaluated columns of a cell are three items (36
  // px), where 16-px items on the window's own grid spent three items on 48 (round 4: 169 -> 130 vector instructions per item, the item
  // count unchanged; k_fastw is bound by vector-instruction issue, profiles/r04/README.md).
  const uint8_t* base = pyr + fg.pyrOff[l] + (size_t)img * fg.pyrImg[l] + sd.winOff - FW_SH;
  const int keyX0 = (sd.key0 & 0xFFFF) - FW_SH, keyY0 = sd.key0 >> 16;
  uint32_t* out = cand + cellSlot * (size_t)cellCap;
  const int n16 = th * BPR;
  const int xa = 3 + FW_SH, xb = tw - 3 + FW_SH;   // evaluated columns (tile coordinates); evaluated rows: [3, th - 3)

  auto load_tile = [&]() {
    for (int i0 = 0; i0 < n16; i0 += 128) {
      uint4 v[2]; int off[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + lane + k * 64;
        const int rr = (int)(__umul24((unsigned)i, BPR_MAGIC) >> 16), c16 = (i - rr * BPR) << 4;
        off[k] = i < n16 ? __mul24(rr, P) + c16 : -1;
        v[k] = make_uint4(0, 0, 0, 0);
        if (i < n16) __builtin_memcpy(&v[k], base + (unsigned)(__umul24(rr, pstride) + c16), 16);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (off[k] >= 0) *reinterpret_cast<uint4*>(tile + off[k]) = v[k];
    }
    FW_SYNC();
  };
  // the 16 ring pixels and the centre of the pixel at window offset `off` (every ring offset relative to the ring's top-left corner is
  // non-negative: one address, immediates only)
  auto ring = [&](int off, int (&rr)[16]) -> int {
    constexpr int O3 = 3 * P + 3;
    int cornerOff = off - O3;
    asm volatile("" : "+v"(cornerOff));   // (opaque: otherwise the address is re-based on the centre and 7 offsets need their own add)
    const uint8_t* p = tile + cornerOff;
    rr[0] = p[O3 + 3 * P];   rr[1] = p[O3 + 3 * P + 1];  rr[2] = p[O3 + 2 * P + 2];  rr[3] = p[O3 + P + 3];
    rr[4] = p[O3 + 3];       rr[5] = p[O3 - P + 3];      rr[6] = p[O3 - 2 * P + 2];  rr[7] = p[O3 - 3 * P + 1];
    rr[8] = p[O3 - 3 * P];   rr[9] = p[O3 - 3 * P - 1];  rr[10] = p[O3 - 2 * P - 2]; rr[11] = p[O3 - P - 3];
    rr[12] = p[O3 - 3];      rr[13] = p[O3 + P - 3];     rr[14] = p[O3 + 2 * P - 2]; rr[15] = p[O3 + 3 * P - 1];
    return p[O3];
  };
  // Strip mode: the whole pass row by row, strengths of every evaluated pixel (no reject, both polarities) into a rolling buffer of four
  // rows, NMS of the row above as soon as the row below it is known, keypoints written in order as they are found.
  auto strip_mode = [&](int T) -> int {
    FW_STAT(13, 1);
    load_tile();
    for (int i = lane; i < P; i += 64) reinterpret_cast<uint32_t*>(sbuf)[i] = 0;   // 4 rows x P bytes
    FW_SYNC();
    int running = 0;
    for (int yy = 3; yy <= th - 3; ++yy) {
      uint8_t* cur = sbuf + (yy & 3) * P;
      for (int x0 = xa; x0 < xb; x0 += 64) {
        const int x = x0 + lane;
        const bool act = x < xb && yy < th - 3;
        int rr[16], d[16];
        const int v = ring(act ? __mul24(yy, P) + x : 3 * P + 3, rr);
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = v - rr[k];
        int S = arc9_maxmin(d);
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = rr[k] - v;
        S = imax(S, arc9_maxmin(d));
        if (x < P) cur[x] = (uint8_t)((act && S > T) ? imin(S, 255) : 0);
      }
      FW_SYNC();
      if (yy > 3) {   // NMS of row yy - 1
        const uint8_t *up = sbuf + ((yy - 2) & 3) * P, *mid = sbuf + ((yy - 1) & 3) * P;
        for (int x0 = xa; x0 < xb; x0 += 64) {
          const int x = x0 + lane;
          const bool act = x < xb;
          const int xs = act ? x : 3;
          const int S = mid[xs];
          const int m = imax(imax(imax(up[xs - 1], up[xs]), imax(up[xs + 1], mid[xs - 1])), imax(imax(mid[xs + 1], cur[xs - 1]), imax(cur[xs], cur[xs + 1])));
          const bool keep = act && S > imax(m, 1);
          const uint64_t km = __ballot(keep);
          const int slot = running + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
          if (keep && slot < cellCap) out[slot] = morbqt::make_key(x + keyX0, yy - 1 + keyY0, S - 1);
          running += __popcll(km);
        }
      }
      FW_SYNC();
    }
    return running;
  };

  int n = 0;              // keypoints of the cell
  bool listed = false;    // ... are in `kept` (not yet written)
  FW_STAT(0, 1);
  FW_CYC0();
  for (int pass = 0; pass < 2 && n == 0; ++pass) {
    // pass 0: cv::FAST(iniThFAST); pass 1: again with minThFAST if the cell came back empty (:795)
    const int T = pass ? minTh : iniTh;
    listed = false;
    FW_STAT(1, 1); FW_STAT(11, pass);
    load_tile();
    FW_CYC(0);
    int qn = 0, cn = 0;
    {
      // 12-px items per evaluated row.  A row of 37 - 39 evaluated columns (a quarter of the cells: the grid's cell width is 35 - 39 px) would need a
      // fourth item for its last 1 - 3 pixels — a third reject round for the cell; there the row's LAST item takes up to 15 pixels instead (the flag
      // layout has room for 16): a fourth dword per item, two more packed pairs, and the cell stays at two rounds.
#ifndef MORB_FW_WIDE
#define MORB_FW_WIDE 1
#endif
      const bool wide = MORB_FW_WIDE && xb - xa > 36 && xb - xa <= 39;   // (wave-uniform)
      const int nIt = wide ? 3 : (xb - xa + 11) / 12;
      const unsigned itMagic = c_magic20.m[nIt];
      const int nItems = (th - 6) * nIt;
      // flags of pixels at or beyond xb in a row's last item are dropped before they are queued (the first item starts at xa)
      const unsigned mLast = c_fwPixMask.m[(xb - xa) - 12 * (nIt - 1)], mItem = c_fwPixMask.m[12];
      const unsigned LO = 0x00FF00FFu;
      unsigned KF[8], MF[8];   // per (dword & 1, parity, polarity): the add constant and the flag bit (8 + index) in both halves
#pragma unroll
      for (int j = 0; j < 8; ++j) { KF[j] = ((1u << (8 + j)) - 1u - (unsigned)T) * 0x00010001u; MF[j] = (1u << (8 + j)) * 0x00010001u; }
      for (int i0 = 0; i0 < nItems; i0 += 64) {
        const int i = i0 + lane;
        const int iy = (int)(__umul24((unsigned)i, itMagic) >> 20);   // (i < 1024, itMagic <= 2^20: a full-rate 24-bit multiply; the 32-bit one is quarter rate)
        const int bi = i - __mul24(iy, nIt);
        const int y = iy + 3;
        const int itemOff = __mul24(y, P) + xa + __mul24(bi, 12);   // tile offset of the item's first pixel: dword aligned (xa = 4)
        unsigned W = 0;   // bit f: f[0] polarity (0 dark, 1 bright), pixel offset in the item = f[3] f[2] f[4] f[1] (0 .. 11)
        if (i < nItems) {
          const uint32_t* rowp = reinterpret_cast<const uint32_t*>(tile + itemOff);
          constexpr int P4 = P / 4;
          uint32_t Cw[6] = {rowp[-1], rowp[0], rowp[1], rowp[2], rowp[3], 0u};
          uint32_t Uw[4] = {rowp[-3 * P4], rowp[-3 * P4 + 1], rowp[-3 * P4 + 2], 0u};   // ring pixel 8 (0,-3)
          uint32_t Dw[4] = {rowp[3 * P4], rowp[3 * P4 + 1], rowp[3 * P4 + 2], 0u};      // ring pixel 0 (0,+3)
          if (wide) { Cw[5] = rowp[4]; Uw[3] = rowp[-3 * P4 + 3]; Dw[3] = rowp[3 * P4 + 3]; }
          unsigned E[6], O[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) { E[k] = Cw[k] & LO; O[k] = (Cw[k] >> 8) & LO; }
          unsigned acc[2] = {0u, 0u};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k == 3 && !wide) break;
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
          W &= bi == nIt - 1 ? mLast : mItem;   // (a wide cell's other items have evaluated the next item's first pixels too)
        }
        FW_STAT(2, 1);
        FW_CYC(1);
        const bool lastRound = i0 + 64 >= nItems;
        // one queue entry per flag: a pixel both of whose polarities are still possible is queued twice — a darker and a brighter 9-arc
        // cannot coexist on a 16-pixel ring, so at most one of the two entries finds a strength > T
        unsigned pend = W;
        for (;;) {
          // compaction of the wave's flag words: per-lane popcount, one DPP scan, every lane emits its own entries
          unsigned take = pend;
          int cnt = __popc(take);
          int incl = cnt;
          MORB_DPP_SCAN(incl, 0, morbwave::op_add);   // inclusive prefix over the wave (all lanes active)
          int total = __builtin_amdgcn_readlane(incl, 63);
          if (total > FW_QCAP - qn) {   // (wave-uniform, rare: qn < 64 here, so more than 256 flags in one round) queue a prefix of the lanes now
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
            const unsigned rec0 = (unsigned)itemOff << 3;   // (offset / 4) << 5; the flag index is decoded by the strength round: once per 64 entries, not per entry
            // (two flags per trip: the loop runs for as long as ANY lane has flags left, and the fullest lane has several times the average)
            unsigned t = take;
            while (t) {
              const unsigned f = (unsigned)__ffs(t) - 1u;
              t &= t - 1u;
              queue[slot] = (QE)(rec0 | f);
              const bool two = t != 0u;
              const unsigned g = (unsigned)__ffs(t) - 1u;
              t &= t - 1u;
              if (two) queue[slot + 1] = (QE)(rec0 | (g & 31u));
              slot += two ? 2 : 1;
            }
            qn += total;
            FW_SYNC();
          }
          const bool more = __ballot(pend != 0u) != 0ull;   // (only after a partial take)
          const bool flush = lastRound && !more;
          FW_CYC(2);
          // strength of the queued pixels, 64 at a time (fewer only when the pass's last survivors are flushed)
          while (qn >= 64 || (flush && qn > 0)) {
            FW_STAT(5, 1);
            const int nq = imin(qn, 64), q0 = qn - nq;
            qn = q0;
            const bool act = lane < nq;
            const unsigned e = act ? (unsigned)queue[q0 + lane] : (unsigned)(3 * P + xa) << 3;   // (inactive lanes: pixel (xa, 3))
            // entry = (item's tile offset / 4) << 5 | f; f[0] = polarity, pixel offset in the 12-px item = f[3] f[2] f[4] f[1]
            const int off = (int)(((e >> 3) & ~3u) + (((e >> 1) & 1u) | ((e >> 3) & 2u) | (e & 12u)));
            int rr[16];
            const int v = ring(off, rr);
            // strength = max over the arcs of min over the arc of (v - p) for the dark polarity, of (p - v) for the bright one.  With
            // q = p ^ m (m = 255 for dark, 0 for bright) both are  maxmin(q) - (v ^ m):  sixteen v_xor_b32 in a row (half the cycles of the
            // v_xad_u32 per ring pixel of the (p ^ mask) + c form: tools/micro/valu_rate.hip) and one tree for both polarities
            const int m8 = (int)(((e & 1u) - 1u) & 0xFFu);
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = rr[k] ^ m8;
            const int S = arc9_maxmin(d) - (v ^ m8);
            const bool isCorner = act && S > T;
            const uint64_t cm = __ballot(isCorner);
            if (cm) {   // wave-uniform
              const int idx = cn + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
              if (isCorner && idx < FW_CQ) { cornerPos[idx] = (uint16_t)off; cornerS[idx] = (uint8_t)imin(S, 255); }
              cn += __popcll(cm);
            }
          }
          FW_CYC(3);
          if (!more) break;
        }
      }
    }
    FW_STAT(6, cn);
    if (cn > FW_CQ) {   // (wave-uniform) the corner list overflowed
      n = strip_mode(T);
      FW_CYC(6);
      continue;
    }
    // NMS.  The window becomes the strength map: S at the corners (S > T), 0 elsewhere.  Corner score S - 1, everything else 0; keep iff
    // strictly greater than all 8 neighbours' scores (a neighbour outside the evaluated area stays 0).
    FW_SYNC();
    for (int i = lane; i < n16; i += 64) {
      const int rr = (int)(__umul24((unsigned)i, BPR_MAGIC) >> 16);
      *reinterpret_cast<uint4*>(tile + (__mul24(rr, P) + ((i - rr * BPR) << 4))) = make_uint4(0, 0, 0, 0);
    }
    FW_SYNC();
    for (int q = lane; q < cn; q += 64) tile[cornerPos[q]] = cornerS[q];
    FW_SYNC();
    for (int q0 = 0; q0 < cn; q0 += 64) {
      FW_STAT(7, 1);
      const int q = q0 + lane;
      const bool act = q < cn;
      const int pos = act ? cornerPos[q] : 3 * P + 3;
      const int S = act ? cornerS[q] : 0;
      const uint8_t* c = tile + pos;
      const int m = imax(imax(imax(c[-P - 1], c[-P]), imax(c[-P + 1], c[-1])), imax(imax(c[1], c[P - 1]), imax(c[P], c[P + 1])));
      const bool keep = S > imax(m, 1);
      const uint64_t km = __ballot(keep);
      if (km) {
        const int idx = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
        if (keep && idx < FW_KC) kept[idx] = (uint32_t)pos | ((uint32_t)S << 16);
        n += __popcll(km);
      }
    }
    listed = true;
    FW_CYC(4);
    if (n > FW_KC) {   // (wave-uniform) more keypoints than the list holds
      n = strip_mode(T);
      listed = false;
      FW_CYC(6);
    }
  }
  FW_STAT(8, n); FW_STAT(9, listed ? n : 0);
  if (listed && n > 0) {
    // the cell's keypoints are a short unordered list: a keypoint's slot is the number of keypoints before it in row-major
    // order, counted against the list broadcast lane by lane
    FW_SYNC();
    const uint32_t mine = lane < n ? kept[lane] : 0xFFFFFFFFu;
    const int mpos = (int)(mine & 0xFFFFu);   // tile offset y * P + x: row-major order is offset order
    int rank = 0;
    for (int k = 0; k < n; ++k) rank += (__builtin_amdgcn_readlane(mpos, k) < mpos) ? 1 : 0;
    const int my = (int)(__umul24((unsigned)mpos >> 4, BPR_MAGIC) >> 16), mx = mpos - __mul24(my, P);
    if (lane < n && rank < cellCap)
      out[rank] = morbqt::make_key(mx + keyX0, my + keyY0, (int)(mine >> 16) - 1);
  }
  if (lane == 0) candCnt[cellSlot] = imin(n, cellCap);
  FW_CYC(5);
  FW_CYC_END();
  }
}
