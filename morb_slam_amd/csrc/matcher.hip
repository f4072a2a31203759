// Hamming matchers for MI355X (gfx950), batched over frames, behind the C ABI of include/morb_hip.h:
//   M0  ORBmatcher::DescriptorDistance                 (reference src/ORBmatcher.cc:1880-1894)
//   F1  Frame::ComputeStereoMatches                    (src/Frame.cc:889-1047)
//   F2  cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2)    (src/Frame.cc:46, :1242)
//   M3  ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ..) (src/ORBmatcher.cc:218-395, non-fisheye branch)
//   N3  DBoW2 vocabulary descent (FeatureVector node ids; Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259)
// Every "first best wins" loop of the reference is restated as a lexicographic minimum over
// (distance, position in the reference's iteration order), which is order-independent and therefore parallel;
// the greedy "skip what is already matched" dependencies are kept by running them sequentially inside one
// wave per independent unit (a BoW node, a frame).  Distances are 8 x (xor, popcount) on 32-bit words;
// reductions are wave64 shuffles.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"
#include "internal_abi.h"
#include "wave.h"
#include "extractor_internal.h"

using namespace morb;

namespace {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // ORBmatcher.cc:35-37
constexpr int EDGE_ = 19;

struct Desc { uint32_t w[8]; };

__device__ __forceinline__ Desc load_desc(const uint8_t* p) {
  Desc d;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w;
  d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
  return d;
}
__device__ __forceinline__ int hamming(const Desc& a, const Desc& b) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += __popc(a.w[i] ^ b.w[i]);
  return s;
}
// full-wave reductions on the DPP path (wave.h); every call site below is convergent over all 64 lanes
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) { return morbwave::min_u64(v); }
__device__ __forceinline__ int wave_sum(int v) { return morbwave::sum_i32(v); }
// merge two sorted pairs (a1<=a2), (b1<=b2) of u64 keys -> the two smallest
__device__ __forceinline__ void top2_merge(unsigned long long& a1, unsigned long long& a2, unsigned long long b1,
                                           unsigned long long b2) {
  const unsigned long long lo = a1 < b1 ? a1 : b1;
  const unsigned long long hi1 = a1 < b1 ? b1 : a1;
  const unsigned long long lo2 = a1 < b1 ? a2 : b2;
  a1 = lo;
  a2 = hi1 < lo2 ? hi1 : lo2;
}
// (as selects on the values: written as `if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;` the compiler stored k through a
// run-time-selected address of k1 / k2 — a private array in scratch memory, read back and written in k_knn2's inner loop)
__device__ __forceinline__ void top2_insert(unsigned long long& k1, unsigned long long& k2, unsigned long long k) {
  const bool lt1 = k < k1, lt2 = k < k2;
  k2 = lt1 ? k1 : (lt2 ? k : k2);
  k1 = lt1 ? k : k1;
}
// the two smallest of the lanes' (k1 <= k2) pairs, in every lane; keys carry the candidate index (unique apart from the ~0
// sentinel), so the runner-up is the smallest of "k2 of the winner's lane, k1 of the others".  DPP (wave.h); all lanes active.
__device__ __forceinline__ void wave_top2(unsigned long long& k1, unsigned long long& k2) {
  const unsigned long long g1 = morbwave::min_u64(k1);
  const unsigned long long g2 = morbwave::min_u64(k1 == g1 ? k2 : k1);
  k1 = g1; k2 = g2;
}

// ---------------------------------------------------------------------------------------------------
// M0: element-wise DescriptorDistance
__global__ __launch_bounds__(256) void k_hamming_pairs(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int n,
                                                       int* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = hamming(load_desc(a + (size_t)i * 32), load_desc(b + (size_t)i * 32));
}

// ---------------------------------------------------------------------------------------------------
// F2: brute-force 2-nearest neighbours.  One query per lane (descriptor in registers), train descriptors
// staged through LDS in tiles of 256 and read as wave-uniform broadcasts.
__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, const int* __restrict__ nqv, int qPitch,
                                              const uint8_t* __restrict__ t, const int* __restrict__ ntv, int tPitch,
                                              const int* __restrict__ qOffv, const int* __restrict__ tOffv,
                                              int* __restrict__ idx, int* __restrict__ dist) {
  __shared__ uint4 tile[256 * 2];
  const int prob = blockIdx.y;
  const int qOff = qOffv ? qOffv[prob] : 0, tOff = tOffv ? tOffv[prob] : 0;
  const int nq = nqv[prob] - qOff, nt = ntv[prob] - tOff;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const uint8_t* qb = q + ((size_t)prob * qPitch + qOff) * 32;
  const uint8_t* tb = t + ((size_t)prob * tPitch + tOff) * 32;
  if ((int)blockIdx.x * 256 >= nq) return;
  Desc d;
  if (i < nq) d = load_desc(qb + (size_t)i * 32);
  unsigned long long k1 = ~0ull, k2 = ~0ull;
  for (int j0 = 0; j0 < nt; j0 += 256) {
    const int j = j0 + threadIdx.x;
    if (j < nt) {
      const uint4* src = reinterpret_cast<const uint4*>(tb + (size_t)j * 32);
      tile[threadIdx.x * 2] = src[0];
      tile[threadIdx.x * 2 + 1] = src[1];
    }
    __syncthreads();
    const int m = nt - j0 < 256 ? nt - j0 : 256;
    if (i < nq) {
      for (int jj = 0; jj < m; ++jj) {
        const uint4 a = tile[jj * 2], b = tile[jj * 2 + 1];
        const int s = __popc(d.w[0] ^ a.x) + __popc(d.w[1] ^ a.y) + __popc(d.w[2] ^ a.z) + __popc(d.w[3] ^ a.w) +
                      __popc(d.w[4] ^ b.x) + __popc(d.w[5] ^ b.y) + __popc(d.w[6] ^ b.z) + __popc(d.w[7] ^ b.w);
        top2_insert(k1, k2, ((unsigned long long)s << 32) | (unsigned)(j0 + jj));
      }
    }
    __syncthreads();
  }
  if (i < nq) {
    const size_t o = ((size_t)prob * qPitch + i) * 2;
    idx[o] = k1 == ~0ull ? -1 : (int)(k1 & 0xFFFFFFFFu);
    dist[o] = k1 == ~0ull ? -1 : (int)(k1 >> 32);
    idx[o + 1] = k2 == ~0ull ? -1 : (int)(k2 & 0xFFFFFFFFu);
    dist[o + 1] = k2 == ~0ull ? -1 : (int)(k2 >> 32);
  }
}

// ---------------------------------------------------------------------------------------------------
// F1: ComputeStereoMatches (Frame.cc:889-1047), three kernels.
//   k_stereo_prep   one workgroup per frame: the right keypoints' row band / octave / x table and the per-16-row-band index (the
//                   reference's vRowIndices, coarsened), written once per frame.
//   k_stereo_match  one workgroup = SM_LK left keypoints of a frame (4 waves x SM_LK / 4 keypoints; SM_LK by batch size, sm_lk_for()):
//                   copies the frame's record into LDS, then per wave (A) the band scans -> (keypoint, candidate) pairs, (B) Hamming
//                   distances of the pairs, best = lexicographic (dist, iR) minimum = the reference's first-best over vRowIndices[row],
//                   (C) the 11 x 11 SAD search over 11 shifts on the un-blurred pyramid level, parabola refinement, disparity -> depth.
//   k_stereo_median one workgroup per frame: median of the accepted SAD distances, 1.5 * 1.4 * median cut.
// History of the middle kernel, 256 frames of 752 x 480 / 1200 features: one wave per left keypoint re-reading all right keypoints from
// global memory (round 1, bound by the number of vector-memory instructions) -> table in LDS per workgroup, candidates compacted before
// any descriptor load, patches as unaligned dwords: 391 us -> the stages run for all of a wave's keypoints together: 360 -> SAD rows by
// v_sad_u8 + 16-lane DPP sums: 314 -> table built once per frame: 271 -> per-keypoint scalar math in lane q: 224 us.
// Left keypoints per workgroup at 256 frames: 8: 357 us, 16: 251, 32: 221; fewer for a handful of frames, where a wave's keypoints are
// the latency of the call (one frame: 4 per workgroup, i.e. one per wave).
__host__ __device__ inline int sm_lk_for(int nframes) { return nframes <= 2 ? 4 : nframes <= 8 ? 8 : nframes <= 32 ? 16 : 32; }
constexpr int SM_BAND = 16;   // rows per band of the per-workgroup row index (the reference's vRowIndices, coarsened)
constexpr int SM_MAXB = 256;  // bands that fit (images up to 4096 rows); SM_LIST * cap list entries, else the full scan
// Per-level facts the stereo kernel needs, by value in the kernarg segment (scalar loads, no dependent round trip
// and no per-call host-to-device table copies).
struct StereoGeom {
  unsigned long long pyrOff[16], pyrImg[16];
  int pstride[16], w[16];
  float scale[16], invScale[16];
  int nRows;
};
struct RightKp { uint32_t band; float x; };   // band = (minr + 4096) | (maxr + 4096) << 14 | octave << 28  (vRowIndices band, Frame.cc:736-747)
constexpr int SM_LIST = 4;
constexpr int SM_KQ = 8;                      // left keypoints of one wave (SM_LK / 4 <= SM_KQ)
constexpr int SM_PCAP = 512;                  // per-wave list of (left keypoint, right candidate) pairs; flushed (descriptors compared) before it could overflow
constexpr int SM_PATCH = 11 * 12 + 11 * 24;   // bytes of one keypoint's two SAD patches
static_assert(4 * SM_PATCH <= SM_PCAP * 4, "four keypoints' patches reuse the pair list");
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// The right image's keypoints as the left keypoints' searches need them (one record per frame, built once by k_stereo_prep and copied
// into LDS by every workgroup of k_stereo_match; before, each of a frame's 38 workgroups rebuilt it from the 28-byte keypoint structs):
// the table [cap] of (row band, octave, x), the first list slot of every 16-row band, the list's length, the per-band lists.
struct StereoRec { int band, list, bytes; };   // byte offsets inside a record (16-byte aligned) and its size
__host__ __device__ inline StereoRec stereo_rec(int cap) {
  StereoRec r;
  r.band = (cap * (int)sizeof(RightKp) + 15) & ~15;
  r.list = r.band + (((SM_MAXB + 2) * 4 + 15) & ~15);
  r.bytes = r.list + ((SM_LIST * cap * 2 + 15) & ~15);
  return r;
}
// (block size: a quarter of a thousand threads per frame in a batch, 1024 for a handful of frames — one workgroup per frame walks its right keypoints)
__global__ __launch_bounds__(1024) void k_stereo_prep(const StereoGeom sg, const morb_keypoint* __restrict__ kps, const int* __restrict__ count,
                                                     int cap, uint8_t* __restrict__ rec) {
  extern __shared__ __align__(16) uint8_t smem[];
  const StereoRec ro = stereo_rec(cap);
  RightKp* tab = reinterpret_cast<RightKp*>(smem);
  int* bandStart = reinterpret_cast<int*>(smem + ro.band);
  uint16_t* list = reinterpret_cast<uint16_t*>(smem + ro.list);
  int* bandFill = reinterpret_cast<int*>(smem + ro.bytes);   // [SM_MAXB]
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
  const int imgR = 2 * f + 1, NR = count[imgR];
  const int nRows = sg.nRows;
  const int nBands = (nRows + SM_BAND - 1) / SM_BAND;
  const bool banded = nBands <= SM_MAXB;
  for (int i = tid; i < SM_MAXB + 2; i += NT) { bandStart[i] = 0; if (i < SM_MAXB) bandFill[i] = 0; }
  __syncthreads();
  for (int iR = tid; iR < NR; iR += NT) {
    const morb_keypoint kpR = kps[(size_t)imgR * cap + iR];
    const float r = 2.0f * sg.scale[kpR.octave & 15];
    const int maxr = (int)ceilf(kpR.y + r), minr = (int)floorf(kpR.y - r);   // the rows the keypoint is registered in (Frame.cc:904-912)
    // clamped to +-4095 rows: far beyond any image row, so the band test is unchanged
    const int lo = min(max(minr, -4095), 4095) + 4096, hi = min(max(maxr, -4095), 4095) + 4096;
    RightKp t;
    t.band = (uint32_t)lo | ((uint32_t)hi << 14) | ((uint32_t)kpR.octave << 28); t.x = kpR.x;
    tab[iR] = t;
    if (banded && maxr >= 0 && minr < nRows)
      for (int bnd = max(minr, 0) / SM_BAND; bnd <= min(maxr, nRows - 1) / SM_BAND; ++bnd) atomicAdd(&bandStart[bnd + 1], 1);
  }
  __syncthreads();
  if (tid < 64) {   // inclusive scan of the band counts -> bandStart[b] = first list slot of band b
    int c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = bandStart[1 + tid * 4 + k]; sum += c[k]; }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    int acc = inc - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc += c[k]; bandStart[1 + tid * 4 + k] = acc; }
    if (tid == 63) bandStart[SM_MAXB + 1] = acc;   // the list's length
  }
  __syncthreads();
  const int listTotal = bandStart[SM_MAXB + 1];
  if (banded && listTotal <= SM_LIST * cap) {
    for (int iR = tid; iR < NR; iR += NT) {
      const RightKp t = tab[iR];
      const int minr = (int)(t.band & 0x3FFF) - 4096, maxr = (int)((t.band >> 14) & 0x3FFF) - 4096;
      if (maxr >= 0 && minr < nRows)
        for (int bnd = max(minr, 0) / SM_BAND; bnd <= min(maxr, nRows - 1) / SM_BAND; ++bnd)
          list[bandStart[bnd] + atomicAdd(&bandFill[bnd], 1)] = (uint16_t)iR;   // order inside a band is irrelevant: best = min (dist, index)
    }
  }
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(smem);
  uint4* dst = reinterpret_cast<uint4*>(rec + (size_t)f * ro.bytes);
  for (int i = tid; i < (ro.bytes >> 4); i += NT) dst[i] = src[i];
}

// A wave's keypoints used to be one dependent chain each (band scan -> candidate descriptors -> best -> patches -> SAD: three global
// round trips per keypoint, eight keypoints in turn); the kernel was bound by exactly that latency.  Now the wave takes its keypoints
// through each stage together: (A) the band scans of all of them (LDS only) append (keypoint, candidate) pairs to one list; (B) the
// list is compared 64 pairs per round — full lanes instead of ~10 candidates of one keypoint — with a segmented minimum per keypoint
// (pairs of a keypoint are contiguous); (C) the patches of four keypoints are requested at once, then summed.  Three round trips per
// wave instead of three per keypoint.
__global__ __launch_bounds__(256) void k_stereo_match(const StereoGeom sg, const uint8_t* __restrict__ pyr,
                                                      const morb_keypoint* __restrict__ kps,
                                                      const uint8_t* __restrict__ desc, const int* __restrict__ count,
                                                      int cap, float mbf, float mb,
                                                      float* __restrict__ uRight, float* __restrict__ depth,
                                                      int* __restrict__ sadDist, int SM_LK, const uint8_t* __restrict__ rec) {
  extern __shared__ __align__(16) uint8_t smem[];
  uint4* dLAll = reinterpret_cast<uint4*>(smem);                                               // [4][SM_KQ][2] left descriptors
  unsigned long long* bestAll = reinterpret_cast<unsigned long long*>(dLAll + 4 * SM_KQ * 2);  // [4][SM_KQ] best (distance << 32 | right index)
  uint32_t* pairsAll = reinterpret_cast<uint32_t*>(bestAll + 4 * SM_KQ);                       // [4][SM_PCAP]; later four keypoints' SAD patches
  uint8_t* recL = reinterpret_cast<uint8_t*>(pairsAll + 4 * SM_PCAP);    // the frame's right-keypoint record, as k_stereo_prep laid it out
  const StereoRec ro = stereo_rec(cap);
  const RightKp* tab = reinterpret_cast<const RightKp*>(recL);           // [cap]
  const int* bandStart = reinterpret_cast<const int*>(recL + ro.band);   // [SM_MAXB + 1], then the list's length
  const uint16_t* list = reinterpret_cast<const uint16_t*>(recL + ro.list);   // [SM_LIST * cap] right keypoints per band
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int imgL = 2 * f, imgR = 2 * f + 1;
  const int NL = count[imgL], NR = count[imgR];
  const int iL0 = blockIdx.x * SM_LK;
  if (iL0 >= cap) return;
  const int nRows = sg.nRows;
  const int nBands = (nRows + SM_BAND - 1) / SM_BAND;
  const bool banded = nBands <= SM_MAXB;
  const uint8_t* recG = rec + (size_t)f * ro.bytes;
  const int listTotal = reinterpret_cast<const int*>(recG + ro.band)[SM_MAXB + 1];   // (wave-uniform: a scalar load)
  const bool useBands = banded && listTotal <= SM_LIST * cap;
  if (iL0 < NL) {
    // the record -> LDS: table, band starts, band lists; 16 bytes per access, all requests in flight together
    const uint4* src = reinterpret_cast<const uint4*>(recG);
    uint4* dst = reinterpret_cast<uint4*>(recL);
    const int nTab = (NR * (int)sizeof(RightKp) + 15) >> 4, nBand = ((SM_MAXB + 2) * 4 + 15) >> 4, nList = useBands ? (listTotal * 2 + 15) >> 4 : 0;
    for (int i = tid; i < nTab; i += 256) dst[i] = src[i];
    for (int i = tid; i < nBand; i += 256) dst[(ro.band >> 4) + i] = src[(ro.band >> 4) + i];
    for (int i = tid; i < nList; i += 256) dst[(ro.list >> 4) + i] = src[(ro.list >> 4) + i];
  }
  __syncthreads();
#if defined(MORB_STEREO_STOP) && MORB_STEREO_STOP == 1
  return;
#endif
  const int KQ = SM_LK / 4;
  uint32_t* pairs = pairsAll + wv * SM_PCAP;
  uint4* dLs = dLAll + wv * (SM_KQ * 2);
  unsigned long long* bestL = bestAll + wv * SM_KQ;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const float maxD = mbf / mb;
  // the wave's left keypoints (one per lane) and their descriptors (16 bytes per lane), one round trip
  float myX = 0.f, myY = 0.f;
  int myOct = 0;
  if (lane < KQ) {
    const int iL = iL0 + lane * 4 + wv;
    if (iL < NL) { const morb_keypoint k = kps[(size_t)imgL * cap + iL]; myX = k.x; myY = k.y; myOct = k.octave; }
  }
  if (lane < 2 * KQ) {
    const int iL = iL0 + (lane >> 1) * 4 + wv;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (iL < NL) v = reinterpret_cast<const uint4*>(desc + ((size_t)imgL * cap + iL) * 32)[lane & 1];
    dLs[lane] = v;
  }
  if (lane < SM_KQ) bestL[lane] = ~0ull;
  WAVE_SYNC();

  // (B) compare the n collected pairs; entries of one keypoint are contiguous and the keypoints ascend along the list
  auto flush = [&](int n) {
    WAVE_SYNC();
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int c = c0 + lane;
      const bool act = c < n;
      const uint32_t pr = pairs[act ? c : n - 1];
      const int q = (int)(pr >> 16), jR = (int)(pr & 0xFFFFu);
      unsigned long long key = ~0ull;
      if (act) {
        Desc dL;
        const uint4 a = dLs[2 * q], b = dLs[2 * q + 1];
        dL.w[0] = a.x; dL.w[1] = a.y; dL.w[2] = a.z; dL.w[3] = a.w; dL.w[4] = b.x; dL.w[5] = b.y; dL.w[6] = b.z; dL.w[7] = b.w;
        const int d = hamming(dL, load_desc(desc + ((size_t)imgR * cap + jR) * 32));
        if (d < TH_HIGH) key = ((unsigned long long)d << 32) | (unsigned)jR;
      }
      if (cap <= 65536) {
        // Segmented minimum in ONE scan: the list's keypoints ascend, so with (SM_KQ - 1 - q) in the top bits the ordinary inclusive prefix
        // minimum of (SM_KQ - 1 - q) << 24 | distance << 16 | right index is, in every lane, the minimum over the lanes of ITS OWN keypoint up
        // to there (an earlier keypoint's entries are larger in the top bits); the last lane of a keypoint's run holds the run's
        // minimum.  (Before: one 64-bit wave minimum — two DPP scans — per keypoint of the round, ~250 vector instructions a round.)
        static_assert(SM_KQ <= 256 && TH_HIGH <= 256, "the packed key holds (SM_KQ - 1 - q) in 8 bits, a distance < TH_HIGH in 8 bits and the right index in 16");
        uint32_t ck = ((uint32_t)(SM_KQ - 1 - q) << 24) | (key == ~0ull ? 0xFFFFFFu : (((uint32_t)(key >> 32) << 16) | (uint32_t)(key & 0xFFFFu)));
        MORB_DPP_SCAN(ck, 0xFFFFFFFFu, morbwave::op_umin);
        const int qn = (int)(pairs[c + 1 < n ? c + 1 : n - 1] >> 16);
        const bool runEnd = act && (c == n - 1 || lane == 63 || qn != q);
        const uint32_t m24 = ck & 0xFFFFFFu;
        if (runEnd && m24 != 0xFFFFFFu) {
          const unsigned long long m = ((unsigned long long)(m24 >> 16) << 32) | (m24 & 0xFFFFu);
          if (m < bestL[q]) bestL[q] = m;   // (one lane per keypoint: a run that continues in the next round meets its own earlier minimum here)
        }
      } else {
        const int qlo = __builtin_amdgcn_readfirstlane(q), qhi = __builtin_amdgcn_readlane(q, 63);   // (inactive lanes repeat the last entry)
        for (int qq = qlo; qq <= qhi; ++qq) {
          const unsigned long long m = wave_min_u64(q == qq ? key : ~0ull);
          if (lane == 0 && m < bestL[qq]) bestL[qq] = m;
        }
      }
    }
    WAVE_SYNC();
  };
  // (A) band scans: the reference's vRowIndices[row] candidates that pass the octave and disparity gates (Frame.cc:933-952)
  {
    int n = 0;
    for (int q = 0; q < KQ; ++q) {
      const int iL = iL0 + q * 4 + wv;
      if (iL >= NL) break;
      const int levelL = __builtin_amdgcn_readlane(myOct, q);
      const float vL = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myY), q));
      const float uL = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myX), q));
      const int row = (int)vL;
      const float minU = uL - maxD, maxU = uL - 0.f;
      if (!(row >= 0 && row < nRows && !(maxU < 0))) continue;
      const int e0 = useBands ? bandStart[row / SM_BAND] : 0, e1 = useBands ? bandStart[row / SM_BAND + 1] : NR;
      for (int j0 = e0; j0 < e1; j0 += 64) {
        const int ei = j0 + lane;
        bool ok = false;
        int iR = 0;
        if (ei < e1) {
          iR = useBands ? (int)list[ei] : ei;
          const RightKp t = tab[iR];
          const int minr = (int)(t.band & 0x3FFF) - 4096, maxr = (int)((t.band >> 14) & 0x3FFF) - 4096, oct = (int)(t.band >> 28);
          ok = row >= minr && row <= maxr && !(oct < levelL - 1 || oct > levelL + 1) && t.x >= minU && t.x <= maxU;
        }
        const uint64_t m = __ballot(ok);
        if (ok) pairs[n + __popcll(m & lt)] = ((uint32_t)q << 16) | (uint32_t)iR;
        n += __popcll(m);
        if (n > SM_PCAP - 64) { flush(n); n = 0; }
      }
    }
    if (n) flush(n);
  }
#if defined(MORB_STEREO_STOP) && MORB_STEREO_STOP == 2
  return;
#endif
  // (C) sub-pixel refinement by SAD over 11 shifts on the un-blurred level (Frame.cc:963-1031).  Everything about a keypoint that is one
  // number per keypoint — scaled coordinates, patch addresses, and afterwards the parabola and the depth — is computed by lane q for
  // keypoint q (gfx950 has no scalar float unit: as wave-uniform code it cost 64 lanes' worth of issue per keypoint); the wave only
  // fetches and sums the patches together, four keypoints' loads in flight.
  bool cDo = false;
  float cUL = 0.f, cUR0 = 0.f;
  int cLvl = 0, cPs = 0;
  unsigned long long cOffL = 0, cOffR = 0;
  if (lane < KQ) {
    const int iL = iL0 + lane * 4 + wv;
    if (iL < NL) {
      const unsigned long long best = bestL[lane];
      const int bestDist = best == ~0ull ? TH_HIGH : (int)(best >> 32);
      if (bestDist < (TH_HIGH + TH_LOW) / 2) {
        const int bestIdxR = (int)(best & 0xFFFFFFFFu);
        const int levelL = myOct;
        const float vL = myY, uL = myX;
        const float uR0 = tab[bestIdxR].x;
        const float sf = sg.invScale[levelL & 15];
        const float scaleduL = roundf(uL * sf), scaledvL = roundf(vL * sf), scaleduR0 = roundf(uR0 * sf);
        const unsigned long long pyrOff = sg.pyrOff[levelL & 15], pyrImg = sg.pyrImg[levelL & 15];
        const int pstride = sg.pstride[levelL & 15], w = sg.w[levelL & 15];
        const float iniu = scaleduR0 + 5 - 5, endu = scaleduR0 + 5 + 5 + 1;
        if (!(iniu < 0 || endu >= (float)w)) {
          cDo = true; cUL = uL; cUR0 = scaleduR0; cLvl = levelL; cPs = pstride;
          cOffL = pyrOff + (size_t)imgL * pyrImg + (size_t)(EDGE_ + (int)scaledvL) * pstride + EDGE_ + (int)scaleduL;
          cOffR = pyrOff + (size_t)imgR * pyrImg + (size_t)(EDGE_ + (int)scaledvL) * pstride + EDGE_ + (int)scaleduR0;
        }
      }
    }
  }
  // a lane's two patch dwords: 11 x 3 dwords of the left patch + 11 x 6 dwords of the right strip = 99 dword loads (the bytes past the
  // 11 / 21 used columns lie inside the level's 19-pixel pad); which dword a lane takes does not depend on the keypoint
  int pRow[2], pCol[2], pLds[2];
  bool pLeft[2], pAct[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = lane + 64 * j;
    pAct[j] = t < 99; pLeft[j] = t < 33;
    if (t < 33) {
      const int r = t / 3, c4 = (t - r * 3) * 4;
      pRow[j] = r - 5; pCol[j] = c4 - 5; pLds[j] = r * 12 + c4;
    } else {
      const int t2 = min(t, 98) - 33, r = t2 / 6, c4 = (t2 - r * 6) * 4;
      pRow[j] = r - 5; pCol[j] = c4 - 10; pLds[j] = 11 * 12 + r * 24 + c4;
    }
  }
  const uint64_t doMask = __ballot(cDo);
  const uint32_t cOffLlo = (uint32_t)cOffL, cOffLhi = (uint32_t)(cOffL >> 32), cOffRlo = (uint32_t)cOffR, cOffRhi = (uint32_t)(cOffR >> 32);
  uint8_t* patch = reinterpret_cast<uint8_t*>(pairs);
  int rS = -1, rInc = 0;        // lane q: keypoint q's best SAD and its shift
  float rD1 = 0.f, rD3 = 0.f;   // and the distances on either side of it
  const int dy = min(lane & 15, 10);
  for (int q0 = 0; q0 < KQ; q0 += 4) {
    if (((doMask >> q0) & 15ull) == 0) continue;   // wave-uniform
    uint32_t pv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = q0 + k;
      pv[k][0] = pv[k][1] = 0;
      if ((doMask >> q) & 1ull) {   // wave-uniform
        const unsigned long long oL = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cOffLhi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cOffLlo, q);
        const unsigned long long oR = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cOffRhi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)cOffRlo, q);
        const int ps = __builtin_amdgcn_readlane(cPs, q);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (pAct[j]) pv[k][j] = load_u32_unaligned(pyr + (pLeft[j] ? oL : oR) + (ptrdiff_t)pRow[j] * ps + pCol[j]);
      }
    }
    WAVE_SYNC();   // (the pair list / the previous four keypoints' patches are dead)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!((doMask >> (q0 + k)) & 1ull)) continue;   // wave-uniform
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (pAct[j]) *reinterpret_cast<uint32_t*>(patch + k * SM_PATCH + pLds[j]) = pv[k][j];
    }
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = q0 + k;
      if (!((doMask >> q) & 1ull)) continue;   // wave-uniform
      const uint8_t* Lp = patch + k * SM_PATCH;   // [11][12]: columns -5 .. 6 of the left patch rows
      const uint8_t* Rp = Lp + 11 * 12;           // [11][24]: columns -10 .. 13 of the right strip rows
      // cv::norm(IL, IR, NORM_L1) for the 11 shifts (Frame.cc:987-1003): a lane sums one patch row of one shift — three v_sad_u8 on the
      // row's 11 bytes, the right row's window cut out with v_alignbyte — and a 16-lane row of the wave adds up the 11 rows of a shift
      // by DPP: four shifts per round, three rounds (before: two pixels per lane and a 64-lane reduction per shift, 11 in a chain)
      const uint32_t l0 = *reinterpret_cast<const uint32_t*>(Lp + dy * 12), l1 = *reinterpret_cast<const uint32_t*>(Lp + dy * 12 + 4),
                     l2 = *reinterpret_cast<const uint32_t*>(Lp + dy * 12 + 8) & 0x00FFFFFFu;
      // the reference walks the shifts in order and keeps the first smallest distance (`dist < bestDist` on floats that are exact integers
      // <= 121 * 255): the minimum of (sum << 4 | shift index).  The sums are wave-uniform (v_readlane), so this runs on the scalar unit.
      uint32_t bestKey = 0xFFFFFFFFu;
      int sums[12];   // (constant indices only: a run-time index would put the array into scratch)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int sh = min(r * 4 + (lane >> 4), 10);   // shift index = inc + 5: the window starts at column sh of the 21-column strip
        const uint8_t* rp = Rp + dy * 24 + (sh & ~3);
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(rp), w1 = *reinterpret_cast<const uint32_t*>(rp + 4),
                       w2 = *reinterpret_cast<const uint32_t*>(rp + 8), w3 = *reinterpret_cast<const uint32_t*>(rp + 12);
        const uint32_t bsh = (uint32_t)(sh & 3);
        const uint32_t r0 = __builtin_amdgcn_alignbyte(w1, w0, bsh), r1 = __builtin_amdgcn_alignbyte(w2, w1, bsh),
                       r2 = __builtin_amdgcn_alignbyte(w3, w2, bsh) & 0x00FFFFFFu;
        int sv = (int)__builtin_amdgcn_sad_u8(l2, r2, __builtin_amdgcn_sad_u8(l1, r1, __builtin_amdgcn_sad_u8(l0, r0, 0u)));
        if ((lane & 15) > 10) sv = 0;
        sv += __builtin_amdgcn_update_dpp(0, sv, 0x111, 0xf, 0xf, true);   // row_shr:1, 2, 4, 8: lane 15 of a row ends with the row's total
        sv += __builtin_amdgcn_update_dpp(0, sv, 0x112, 0xf, 0xf, true);
        sv += __builtin_amdgcn_update_dpp(0, sv, 0x114, 0xf, 0xf, true);
        sv += __builtin_amdgcn_update_dpp(0, sv, 0x118, 0xf, 0xf, true);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sums[r * 4 + j] = __builtin_amdgcn_readlane(sv, 16 * j + 15);
          if (r * 4 + j < 11) {
            const uint32_t key = ((uint32_t)sums[r * 4 + j] << 4) | (uint32_t)(r * 4 + j);
            bestKey = key < bestKey ? key : bestKey;
          }
        }
      }
      const int bestS = (int)(bestKey >> 4), bi = (int)(bestKey & 15u);
      // vDists[best - 1] and [best + 1] (only used when 1 <= best <= 9), by compares against the constant indices
      int s1 = 0, s3 = 0;
#pragma unroll
      for (int i = 0; i < 11; ++i) { s1 = (i == bi - 1) ? sums[i] : s1; s3 = (i == bi + 1) ? sums[i] : s3; }
      if (lane == q) { rS = bestS; rInc = bi - 5; rD1 = (float)s1; rD3 = (float)s3; }
    }
  }
  if (lane < KQ) {
    const int iL = iL0 + lane * 4 + wv;
    if (iL < cap) {
      float outU = -1.0f, outD = -1.0f;
      int outS = -1;
      if (cDo) {
        const int bestinc = rInc, bestS = rS;
        if (!(bestinc == -5 || bestinc == 5)) {
          const float dist1 = rD1, dist2 = (float)bestS, dist3 = rD3;
          const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
          if (!(deltaR < -1 || deltaR > 1)) {
            float bestuR = sg.scale[cLvl & 15] * ((float)cUR0 + (float)bestinc + deltaR);
            float disparity = cUL - bestuR;
            if (disparity >= 0.f && disparity < maxD) {
              if (disparity <= 0) {
                disparity = 0.01f;
                bestuR = (float)((double)cUL - 0.01);
              }
              outD = mbf / disparity;
              outU = bestuR;
              outS = bestS;
            }
          }
        }
      }
      const size_t o = (size_t)f * cap + iL;
      uRight[o] = outU; depth[o] = outD; sadDist[o] = outS;
    }
  }
}

// Round 6: the frame's SAD distances are read ONCE, into registers (SMED_V per thread, 1024 threads: frames of up to 8192 features; larger ones
// re-read global memory as before).  The 15 bisection steps then cost two barriers each instead of a global-memory round trip per 256 features:
// 22.9 -> 8.8 us for one 4000-feature frame.
// SMED_WAVES: 16 for a handful of frames (latency), 4 in a batch — 512 workgroups of 16 waves meeting at 17 barriers each took 474 us where four
// waves take ~50 (profiles/r06/README.md).
constexpr int SMED_V = 8;
template <int SMED_WAVES>
__global__ __launch_bounds__(64 * SMED_WAVES) void k_stereo_median(const int* __restrict__ count, int cap, float* __restrict__ uRight,
                                                       float* __restrict__ depth, const int* __restrict__ sadDist) {
  constexpr int NT = 64 * SMED_WAVES;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int NL = count[2 * f];
  const size_t o = (size_t)f * cap;
  const bool inRegs = NL <= NT * SMED_V;
  int dv[SMED_V];
#pragma unroll
  for (int j = 0; j < SMED_V; ++j) { const int i = tid + j * NT; dv[j] = (inRegs && i < NL) ? sadDist[o + i] : -1; }
  int n = 0;
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < SMED_V; ++j) n += dv[j] >= 0 ? 1 : 0;
  } else {
    for (int i = tid; i < NL; i += NT) n += sadDist[o + i] >= 0 ? 1 : 0;
  }
  __shared__ int red[2][SMED_WAVES];
  int phase = 0;
  auto blockSum = [&](int v) -> int {   // (double-buffered: one barrier per sum)
    v = wave_sum(v);
    if ((tid & 63) == 0) red[phase][tid >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int q = 0; q < SMED_WAVES; ++q) t += red[phase][q];
    phase ^= 1;
    return t;
  };
  const int total = blockSum(n);
  if (total == 0) return;
  const int k = total / 2;  // vDistIdx[size/2].first of the ascending sort
  int lo = 0, hi = 121 * 255;  // smallest v with #(d <= v) > k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int c = 0;
    if (inRegs) {
#pragma unroll
      for (int j = 0; j < SMED_V; ++j) c += (dv[j] >= 0 && dv[j] <= mid) ? 1 : 0;
    } else {
      for (int i = tid; i < NL; i += NT) { const int d = sadDist[o + i]; c += (d >= 0 && d <= mid) ? 1 : 0; }
    }
    c = blockSum(c);
    if (c > k) hi = mid; else lo = mid + 1;
  }
  const float median = (float)lo;
  const float thDist = 1.5f * 1.4f * median;
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < SMED_V; ++j) {
      const int i = tid + j * NT;
      if (dv[j] >= 0 && !((float)dv[j] < thDist)) { uRight[o + i] = -1; depth[o + i] = -1; }
    }
  } else {
    for (int i = tid; i < NL; i += NT) {
      const int d = sadDist[o + i];
      if (d >= 0 && !((float)d < thDist)) { uRight[o + i] = -1; depth[o + i] = -1; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// N3: DBoW2 transform descent, FOUR lanes per feature (round 3; one thread per feature before).  A node's children lie one after the other,
// 32 bytes each: the four lanes of a feature read a child's descriptor as one contiguous 32-byte segment (8 bytes per lane) instead of each
// lane of the wave fetching its own feature's child as two scattered 16-byte loads — half the load instructions, a quarter of the cache lines
// per instruction, four times the waves to hide the six dependent levels behind.  A child's distance is the sum of the four lanes' popcounts
// (two DPP quad swaps); all four lanes then take the same decision.
__device__ __forceinline__ int quad_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1, 0, 3, 2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2, 3, 0, 1]
  return v;
}
__global__ __launch_bounds__(256) void k_bow_transform(const uint8_t* __restrict__ feat, const int* __restrict__ count,
                                                       int cap, const uint8_t* __restrict__ nodeDesc,
                                                       const int* __restrict__ firstChild, int k, int L, int levelsup,
                                                       int* __restrict__ wordId, int* __restrict__ nodeId,
                                                       const int* __restrict__ childCount) {   // per node, or NULL: k everywhere
  const int img = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, sub = t & 3;
  if (i >= cap) return;   // (whole quads)
  const size_t o = (size_t)img * cap + i;
  if (i >= count[img]) { if (sub == 0) { wordId[o] = -1; nodeId[o] = -1; } return; }
  const uint2 d = *reinterpret_cast<const uint2*>(feat + o * 32 + sub * 8);
  const uint8_t* nd = nodeDesc + sub * 8;
  const int nid_level = L - levelsup;
  int final_id = 0, level = 0, nid = 0;
  do {
    ++level;
    const int c0 = firstChild[final_id];
    const int nc = childCount ? childCount[final_id] : k;   // trained vocabularies have nodes with fewer than k children
    // The children are requested five at a time before the first distance is taken (one child per loop trip is one dependent memory round
    // trip per child).  Slots past the last child repeat it: a repeat never wins (the comparison is strict, its first copy came earlier).
    int best = c0, bestd = 0x7fffffff;
    constexpr int CH = 5;
    for (int cb = c0; cb < c0 + nc; cb += CH) {
      uint2 tc[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) tc[j] = *reinterpret_cast<const uint2*>(nd + (size_t)min(cb + j, c0 + nc - 1) * 32);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int dd = quad_sum(__popc(d.x ^ tc[j].x) + __popc(d.y ^ tc[j].y));
        if (dd < bestd) { bestd = dd; best = min(cb + j, c0 + nc - 1); }
      }
    }
    final_id = best;
    if (level == nid_level) nid = final_id;
  } while (firstChild[final_id] >= 0);
  if (sub == 0) { wordId[o] = final_id; nodeId[o] = nid; }
}

// Round 5: the first three levels of the tree from LDS.  Every feature walks all of them and they are small — 10 + 100 + 1000 nodes of a
// 10-ary vocabulary, 35 KB of descriptors — while each level costs a dependent round trip to L2 (a level's children can only be requested once
// the level above is decided): half of the six round trips of an ORBvoc-shaped descent.  A workgroup stages them ONCE, by walking the tree
// itself (firstChild / childCount, so any node numbering — ORBvoc.txt's is depth-first — and nodes with fewer than k children work), then works
// through many 64-feature chunks (a persistent grid: the staging is paid ~1000 times per launch, not 20 000 times).  Slot of a staged node =
// its path: level 1: a, level 2: 10 + 10 a + b, level 3: 110 + 100 a + 10 b + c.  Deeper levels come from global memory as before.
constexpr int BT_K = 10, BT_N1 = BT_K, BT_N2 = BT_K * BT_K, BT_N3 = BT_K * BT_K * BT_K, BT_SLOTS = BT_N1 + BT_N2 + BT_N3;   // 1110
template <int NLV> struct BtLds {
  static constexpr int SLOTS = NLV == 3 ? BT_SLOTS : BT_N1 + BT_N2;
  uint2 desc[SLOTS][4];      // [slot][quarter] 32-byte descriptors
  int id[SLOTS];             // node id of the slot, -1: no such node
  int fc[SLOTS];             // its firstChild (-1: a leaf)
  uint16_t nc[SLOTS];        // its child count (more than BT_K: its children are not staged, the descent continues in global memory)
};
// NLV = staged levels: 3 (47 KB of LDS per workgroup: three workgroups per CU) or 2 (5 KB: no limit on the occupancy)
template <int NLV>
__global__ __launch_bounds__(256) void k_bow_transform_lds(const uint8_t* __restrict__ feat, const int* __restrict__ count, int cap, int nimg,
                                                           const uint8_t* __restrict__ nodeDesc, const int* __restrict__ firstChild, int k,
                                                           int L, int levelsup, int* __restrict__ wordId, int* __restrict__ nodeId,
                                                           const int* __restrict__ childCount) {
  extern __shared__ __align__(16) uint8_t btRaw[];
  BtLds<NLV>& S = *reinterpret_cast<BtLds<NLV>*>(btRaw);
  const int tid = threadIdx.x;
  const int fc0 = firstChild[0], nc0 = childCount ? childCount[0] : k;
  // ---- staging: level by level (a level's ids come from the level above)
  auto stage = [&](int slot, int parentFc, int parentNc, int j) {
    int id = -1, fc = -1, nc = 0;
    if (parentFc >= 0 && j < parentNc) {
      id = parentFc + j;
      fc = firstChild[id];
      nc = fc >= 0 ? (childCount ? childCount[id] : k) : 0;
      const uint4* src = reinterpret_cast<const uint4*>(nodeDesc + (size_t)id * 32);
      const uint4 a = src[0], b = src[1];
      S.desc[slot][0] = make_uint2(a.x, a.y); S.desc[slot][1] = make_uint2(a.z, a.w);
      S.desc[slot][2] = make_uint2(b.x, b.y); S.desc[slot][3] = make_uint2(b.z, b.w);
    }
    S.id[slot] = id; S.fc[slot] = fc; S.nc[slot] = (uint16_t)min(nc, 0xFFFF);
  };
  if (tid < BT_N1) stage(tid, nc0 <= BT_K ? fc0 : -1, nc0, tid);
  __syncthreads();
  if (tid < BT_N2) { const int a = tid / BT_K; stage(BT_N1 + tid, S.nc[a] <= BT_K ? S.fc[a] : -1, S.nc[a], tid % BT_K); }
  __syncthreads();
  if (NLV == 3) for (int t = tid; t < BT_N3; t += 256) { const int ab = t / BT_K; stage(BT_N1 + BT_N2 + t, S.nc[BT_N1 + ab] <= BT_K ? S.fc[BT_N1 + ab] : -1, S.nc[BT_N1 + ab], t % BT_K); }
  __syncthreads();
  // ---- the chunks: 64 features (four lanes each) of one image per trip
  const int sub = tid & 3, nid_level = L - levelsup;
  const int chunksPerImg = (cap + 63) / 64, nChunks = chunksPerImg * nimg;
  const uint8_t* nd = nodeDesc + sub * 8;
  for (int ch = blockIdx.x; ch < nChunks; ch += gridDim.x) {
    const int img = ch / chunksPerImg, i = (ch - img * chunksPerImg) * 64 + (tid >> 2);
    if (i >= cap) continue;
    const size_t o = (size_t)img * cap + i;
    if (i >= count[img]) { if (sub == 0) { wordId[o] = -1; nodeId[o] = -1; } continue; }
    const uint2 d = *reinterpret_cast<const uint2*>(feat + o * 32 + sub * 8);
    int final_id = 0, level = 0, nid = 0;
    int slotBase = 0, nc = nc0, path = 0;   // children of the current node: slots slotBase .. slotBase + nc - 1
    bool leaf = fc0 < 0;
    // levels 1 - 3 from LDS
#pragma unroll
    for (int lv = 0; lv < NLV; ++lv) {
      if (leaf || nc > BT_K) break;   // (a node with more than ten children: not staged)
      ++level;
      int best = 0, bestd = 0x7fffffff;
      for (int j = 0; j < nc; ++j) {
        const uint2 tc = S.desc[slotBase + j][sub];
        const int dd = quad_sum(__popc(d.x ^ tc.x) + __popc(d.y ^ tc.y));
        if (dd < bestd) { bestd = dd; best = j; }
      }
      const int slot = slotBase + best;
      final_id = S.id[slot];
      if (level == nid_level) nid = final_id;
      leaf = S.fc[slot] < 0;
      nc = S.nc[slot];
      path = path * BT_K + best;
      slotBase = (lv == 0 ? BT_N1 : BT_N1 + BT_N2) + path * BT_K;
    }
    // deeper levels from global memory (the round-3 form)
    while (!leaf) {
      ++level;
      const int c0 = firstChild[final_id];
      const int ncg = childCount ? childCount[final_id] : k;
      int best = c0, bestd = 0x7fffffff;
      constexpr int CH = 5;
      for (int cb = c0; cb < c0 + ncg; cb += CH) {
        uint2 tc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) tc[j] = *reinterpret_cast<const uint2*>(nd + (size_t)min(cb + j, c0 + ncg - 1) * 32);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int dd = quad_sum(__popc(d.x ^ tc[j].x) + __popc(d.y ^ tc[j].y));
          if (dd < bestd) { bestd = dd; best = min(cb + j, c0 + ncg - 1); }
        }
      }
      final_id = best;
      if (level == nid_level) nid = final_id;
      leaf = firstChild[final_id] < 0;
    }
    if (sub == 0) { wordId[o] = final_id; nodeId[o] = nid; }
  }
}

// ---------------------------------------------------------------------------------------------------
// M3: SearchByBoW.  (1) per frame, sort (node, index) pairs: bitonic sort in LDS; features with node < 0 go
// last.  (2) one wave per BoW node present in the keyframe: merge-join against the frame's sorted list,
// then the keyframe's features of that node IN ORDER (the greedy dependency), lanes over the frame's features
// of the node with a top-2 lexicographic reduction.  (3) per pair: rotation histogram, three maxima, filter.
// Bitonic sort, one thread per compare-exchange (P / 2 threads up to 1024).  A step whose partner distance j is below 64 pairs elements of one 128-key
// block — the block of ONE wave — so only the steps with j >= 64 need the workgroup barrier (15 of the 66 steps of 2048 keys); the others order
// their LDS traffic inside the wave.  (256 threads looping over the pairs with a barrier per step: 69 us for the 40 images of the keyframe searches.)
__global__ __launch_bounds__(1024) void k_bow_sort(const int* __restrict__ node, const int* __restrict__ count, int cap,
                                                   int P, unsigned long long* __restrict__ sorted) {
  extern __shared__ unsigned long long skeys[];
  const int img = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int n = count[img];
  if (n == 0) {   // (an image that takes no part in BoW matching — the right images of a stereo batch — costs one store pass)
    for (int i = tid; i < cap; i += NT) sorted[(size_t)img * cap + i] = ~0ull;
    return;
  }
  for (int i = tid; i < P; i += NT) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const int nd = node[(size_t)img * cap + i];
      if (nd >= 0) k = ((unsigned long long)(unsigned)nd << 32) | (unsigned)i;
    }
    skeys[i] = k;
  }
  __syncthreads();
  const bool onePass = P / 2 <= NT;   // every pair has its own thread: pairs 64 w .. 64 w + 63 (keys 128 w .. 128 w + 127) belong to wave w
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int l = tid; l < P / 2; l += NT) {
        const int i = ((l & ~(j - 1)) << 1) | (l & (j - 1)), ixj = i | j;
        const unsigned long long a = skeys[i], b = skeys[ixj];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { skeys[i] = b; skeys[ixj] = a; }
      }
      // the next step's partner distance: j / 2 within this merge, k for the first step of the next merge
      const int jn = j > 1 ? j >> 1 : k;
      if (onePass && jn < 64 && j < 64) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
      else __syncthreads();
    }
  for (int i = tid; i < cap; i += NT) sorted[(size_t)img * cap + i] = i < P ? skeys[i] : ~0ull;
}

// SearchByBoW core (ORBmatcher.cc:222-404).  Features of one vocabulary node only compete with each other, so a
// node is an independent unit; inside a node the keyframe features are visited in order and each takes the best
// still-unmatched frame feature (greedy), which one wave replays sequentially.
// v2: BM_NB workgroups per pair stage the two sorted (node, index) arrays in LDS once; a wave then owns a stretch of
// keyframe positions.  For a node with <= 64 features on either side everything it needs — descriptors, angles,
// MapPoint flags — is fetched in ONE round trip into registers (frame feature s in lane s, keyframe feature q in
// lane q) and the greedy loop runs on registers with readlane broadcasts; v1 paid ~5 dependent global round trips
// per keyframe feature plus a 10-step binary search in global memory per node (0.10 VALU utilisation).
#ifdef MORB_FAST_TIMING
__device__ unsigned long long g_bowTrace[16384 * 4];   // per-wave clocks (start, staged, compacted, end) for tools/bow_phases.py
#define BOW_MARK(k) do { if (lane == 0 && bw_ < 16384) g_bowTrace[bw_ * 4 + (k)] = wall_clock64(); } while (0)
#else
#define BOW_MARK(k)
#endif
constexpr int BM_NB = 25;
constexpr int BM_FJ = 4;    // frame features of a node held in registers: up to BM_FJ per lane   // x 4 waves: one node per wave for vocabularies with ~100 nodes at the matching level
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return morbwave::min_u32(v); }
__global__ __launch_bounds__(256) void k_bow_match(const unsigned long long* __restrict__ sortedKF,
                                                   const unsigned long long* __restrict__ sortedF,
                                                   const int* __restrict__ nKFv, const int* __restrict__ nFv, int cap,
                                                   const uint8_t* __restrict__ descKF, const uint8_t* __restrict__ hasMP,
                                                   const morb_keypoint* __restrict__ kpsKF,
                                                   const uint8_t* __restrict__ descF, const morb_keypoint* __restrict__ kpsF,
                                                   const int* __restrict__ kfImg, const int* __restrict__ fImg,
                                                   float nnratio, int* __restrict__ matchF, int* __restrict__ binF,
                                                   const int* __restrict__ nLeftv,     // F.Nleft per pair or NULL (pinhole)
                                                   // SearchByBoW(pKF1, pKF2) (:702-819) when hasMP2 != NULL: the "frame" side is
                                                   // keyframe 2, only its features with a MapPoint compete, the threshold is strict,
                                                   // the table is indexed by keyframe-1 feature (match12[idx1] = idx2)
                                                   const uint8_t* __restrict__ hasMP2, const int* __restrict__ nValidv,
                                                   int* __restrict__ matched2) {
  extern __shared__ __align__(16) unsigned long long bowLds[];
#ifdef MORB_FAST_TIMING
  const int bw_ = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0 && bw_ < 16384) g_bowTrace[bw_ * 4] = wall_clock64();
#endif
  unsigned long long* sk = bowLds;         // [cap] keyframe: node << 32 | feature index, ascending, ~0 = no word
  unsigned long long* sf = bowLds + cap;   // [cap] frame
  const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ik = kfImg[pair], jf = fImg[pair];
  const int nKF = nKFv[ik], nF = nFv[jf];
  // fisheye frame (ORBmatcher.cc:262-299, :333-365): features >= nLeft are the right camera's and are ranked separately
  const bool fish = nLeftv != nullptr && nLeftv[pair] >= 0;
  const int nLeft = fish ? nLeftv[pair] : 0x7fffffff;
  const bool kfkf = hasMP2 != nullptr;
  const int nValid1 = (kfkf && nValidv) ? nValidv[ik] : 0x7fffffff, nValid2 = (kfkf && nValidv) ? nValidv[jf] : 0x7fffffff;
  const int thLow = kfkf ? TH_LOW - 1 : TH_LOW;    // bestDist1 < TH_LOW (:768) vs <= TH_LOW (:307)
  {
    // stage both tables with all of a thread's loads in flight at once (one global round trip, not one per element)
    const unsigned long long* gk = sortedKF + (size_t)ik * cap;
    const unsigned long long* gf = sortedF + (size_t)jf * cap;
    for (int i0 = 0; i0 < cap; i0 += 256 * 4) {
      unsigned long long a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 256 + tid;
        a[k] = i < nKF ? gk[i] : ~0ull;
        b[k] = i < nF ? gf[i] : ~0ull;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 256 + tid;
        if (i < nKF) sk[i] = a[k];
        if (i < nF) sf[i] = b[k];
      }
    }
  }
  __syncthreads();
  BOW_MARK(1);
  // node starts of the keyframe, compacted in position order; wave g of the pair takes nodes g, g + G, ...
  // (node sizes are very uneven — 1 .. 60 features — and a node is a sequential loop, so nodes, not position
  // ranges, are the unit that is dealt out)
  uint16_t* starts = reinterpret_cast<uint16_t*>(sf + cap);   // [cap]
  __shared__ int wcnt[4];
  const uint64_t ltm = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int nStarts = 0;
  for (int i0 = 0; i0 < nKF; i0 += 256) {
    const int i = i0 + tid;
    bool st = false;
    if (i < nKF) {
      const unsigned long long k = sk[i];
      st = k != ~0ull && (i == 0 || (unsigned)(sk[i - 1] >> 32) != (unsigned)(k >> 32));
    }
    const uint64_t m = __ballot(st);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = nStarts;
    for (int w = 0; w < wv; ++w) off += wcnt[w];
    if (st) starts[off + __popcll(m & ltm)] = (uint16_t)i;
    nStarts += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int G = gridDim.x * 4, g = blockIdx.x * 4 + wv;
  BOW_MARK(2);
  int* mF = matchF + (size_t)pair * cap;
  int* bF = binF + (size_t)pair * cap;
  const float factor = 1.0f / HISTO_LENGTH;
  for (int o = g; o < nStarts; o += G) {
    const int p = starts[o];
    const unsigned node = (unsigned)(sk[p] >> 32);
    // frame segment of this node: lower_bound(node << 32) in LDS
    int lo = 0, hi = nF;
    const unsigned long long target = (unsigned long long)node << 32;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sf[mid] < target) lo = mid + 1; else hi = mid;
    }
    const int fBeg = lo;
    // frame segment length, up to BM_FJ * 64 positions (lane l looks at positions l, l + 64, ...)
    int nFs = 0;
    bool more = false;
#pragma unroll
    for (int j = 0; j < BM_FJ; ++j) {
      const int s = fBeg + j * 64 + lane;
      const uint64_t mm = __ballot(s < nF && (unsigned)(sf[s] >> 32) == node);
      nFs += __popcll(mm);
      more = mm == ~0ull;     // after the last round: the segment may continue past BM_FJ * 64
      if (mm != ~0ull) break;
    }
    if (nFs == 0) continue;
    if (!more) {
      // ---- register path: frame feature s of the node lives in slot s / 64 of lane s % 64; the keyframe features are
      // taken 64 at a time (feature q of the chunk in lane q) and visited in order
      const int nJ = (nFs + 63) >> 6;
      int idxF[BM_FJ], myMatch[BM_FJ], myBin[BM_FJ];
      Desc dF[BM_FJ];
      float angF[BM_FJ];
#pragma unroll
      for (int j = 0; j < BM_FJ; ++j) {
        idxF[j] = 0; myMatch[j] = -1; myBin[j] = 0; angF[j] = 0.f; dF[j] = Desc{};
        if (j < nJ && j * 64 + lane < nFs) {
          idxF[j] = (int)(sf[fBeg + j * 64 + lane] & 0xFFFFFFFFu);
          dF[j] = load_desc(descF + ((size_t)jf * cap + idxF[j]) * 32);
          angF[j] = kpsF[(size_t)jf * cap + idxF[j]].angle;
          if (kfkf && (!hasMP2[(size_t)jf * cap + idxF[j]] || idxF[j] >= nValid2)) myMatch[j] = -2;   // never a candidate
        }
      }
      for (int q0 = p; q0 < nKF; q0 += 64) {
        const uint64_t mK = __ballot(q0 + lane < nKF && (unsigned)(sk[q0 + lane] >> 32) == node);
        const int nKs = mK == ~0ull ? 64 : (int)__builtin_ctzll(~mK);   // the segment is contiguous
        if (nKs == 0) break;
        int idxK = 0, has = 0;
        Desc dK = {};
        float angK = 0.f;
        if (lane < nKs) {
          idxK = (int)(sk[q0 + lane] & 0xFFFFFFFFu);
          has = hasMP[(size_t)ik * cap + idxK] && idxK < nValid1;
          dK = load_desc(descKF + ((size_t)ik * cap + idxK) * 32);
          angK = kpsKF[(size_t)ik * cap + idxK].angle;
        }
        for (int q = 0; q < nKs; ++q) {
          if (!__builtin_amdgcn_readlane(has, q)) continue;
          Desc dk;
#pragma unroll
          for (int w = 0; w < 8; ++w) dk.w[w] = (uint32_t)__builtin_amdgcn_readlane((int)dK.w[w], q);
          unsigned a1 = ~0u, a2 = ~0u;   // this lane's two best keys: dist << 16 | frame feature index (< cap < 65536)
          unsigned b1 = ~0u;             // fisheye: this lane's best right-camera key (the second best is unused: `|| true`, :336)
#pragma unroll
          for (int j = 0; j < BM_FJ; ++j)
            if (j < nJ && j * 64 + lane < nFs && myMatch[j] == -1) {
              const unsigned key = ((unsigned)hamming(dk, dF[j]) << 16) | (unsigned)idxF[j];
              if (idxF[j] < nLeft) { if (key < a1) { a2 = a1; a1 = key; } else if (key < a2) a2 = key; }
              else if (key < b1) b1 = key;
            }
          const unsigned k1 = wave_min_u32(a1);
          const unsigned k2 = wave_min_u32(a1 == k1 ? a2 : a1);
          const unsigned r1 = fish ? wave_min_u32(b1) : ~0u;
          const int bestDist1 = k1 == ~0u ? 256 : (int)(k1 >> 16);
          const int bestDist2 = k2 == ~0u ? 256 : (int)(k2 >> 16);
          const int bestDist1R = r1 == ~0u ? 256 : (int)(r1 >> 16);
          if (bestDist1 <= thLow) {
            const int realIdxKF = __builtin_amdgcn_readlane(idxK, q);
            const float aK = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angK), q));
            const bool takeL = (float)bestDist1 < nnratio * (float)bestDist2;
            const bool takeR = bestDist1R <= TH_LOW;   // nested in the LEFT test, no ratio test of its own (:333-337)
            // exactly one lane owns each winning key: keys differ in the index bits
            const int wantIdx = (takeL && a1 == k1) ? (int)(k1 & 0xFFFFu) : -1;
            const int wantIdxR = (takeR && b1 == r1 && r1 != ~0u) ? (int)(r1 & 0xFFFFu) : -1;
#pragma unroll
            for (int j = 0; j < BM_FJ; ++j)
              if (j < nJ && j * 64 + lane < nFs && myMatch[j] == -1 && (idxF[j] == wantIdx || idxF[j] == wantIdxR)) {
                float rot = aK - angF[j];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                myMatch[j] = realIdxKF; myBin[j] = bin;
              }
          }
        }
        if (nKs < 64) break;
      }
#pragma unroll
      for (int j = 0; j < BM_FJ; ++j)
        if (j < nJ && j * 64 + lane < nFs && myMatch[j] >= 0) {
          if (kfkf) { mF[myMatch[j]] = idxF[j]; bF[myMatch[j]] = myBin[j]; }     // indexed by the keyframe-1 feature
          else { mF[idxF[j]] = myMatch[j]; bF[idxF[j]] = myBin[j]; }
        }
      continue;
    }
    // ---- general path (a node with more than BM_FJ * 64 frame features): v1 logic on the LDS tables
    int fEnd = fBeg;
    while (fEnd < nF && (unsigned)(sf[fEnd] >> 32) == node) ++fEnd;
    for (int q = p; q < nKF; ++q) {
      const unsigned long long kq = sk[q];
      if (kq == ~0ull || (unsigned)(kq >> 32) != node) break;
      const int realIdxKF = (int)(kq & 0xFFFFFFFFu);
      if (!hasMP[(size_t)ik * cap + realIdxKF] || realIdxKF >= nValid1) continue;
      const Desc dKF = load_desc(descKF + ((size_t)ik * cap + realIdxKF) * 32);
      unsigned long long k1 = ~0ull, k2 = ~0ull, r1 = ~0ull, r2 = ~0ull;
      for (int s0 = fBeg; s0 < fEnd; s0 += 64) {
        const int s = s0 + lane;
        if (s < fEnd) {
          const int realIdxF = (int)(sf[s] & 0xFFFFFFFFu);
          const bool cand = kfkf ? (hasMP2[(size_t)jf * cap + realIdxF] && realIdxF < nValid2 && !matched2[(size_t)pair * cap + realIdxF])
                                 : (mF[realIdxF] < 0);
          if (cand) {
            const int d = hamming(dKF, load_desc(descF + ((size_t)jf * cap + realIdxF) * 32));
            // (left / right camera ranking by value selects, not by a run-time choice of which pair to update: that is a private array in scratch)
            const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)realIdxF;
            const bool left = realIdxF < nLeft;
            unsigned long long t1 = left ? k1 : r1, t2 = left ? k2 : r2;
            top2_insert(t1, t2, key);
            k1 = left ? t1 : k1; k2 = left ? t2 : k2; r1 = left ? r1 : t1; r2 = left ? r2 : t2;
          }
        }
      }
      wave_top2(k1, k2);
      if (fish) wave_top2(r1, r2);
      const int bestDist1 = k1 == ~0ull ? 256 : (int)(k1 >> 32);
      const int bestDist2 = k2 == ~0ull ? 256 : (int)(k2 >> 32);
      const int bestDist1R = r1 == ~0ull ? 256 : (int)(r1 >> 32);
      if (bestDist1 <= thLow) {
        const float aK = kpsKF[(size_t)ik * cap + realIdxKF].angle;
        const bool takeL = (float)bestDist1 < nnratio * (float)bestDist2, takeR = bestDist1R <= TH_LOW;
        if (lane == 0) {
          if (takeL) {
            const int bestIdxF = (int)(k1 & 0xFFFFFFFFu);
            float rot = aK - kpsF[(size_t)jf * cap + bestIdxF].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (kfkf) { mF[realIdxKF] = bestIdxF; bF[realIdxKF] = bin; matched2[(size_t)pair * cap + bestIdxF] = 1; }
            else { mF[bestIdxF] = realIdxKF; bF[bestIdxF] = bin; }
          }
          if (takeR) {
            const int bestIdxFR = (int)(r1 & 0xFFFFFFFFu);
            float rot = aK - kpsF[(size_t)jf * cap + bestIdxFR].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            mF[bestIdxFR] = realIdxKF; bF[bestIdxFR] = bin;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
#ifdef MORB_FAST_TIMING
  BOW_MARK(3);
#endif
}

// ---------------------------------------------------------------------------------------------------
// N4: MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:367-435): one wave per map point, one observed descriptor
// per lane (64 at a time).  The median of a row is found without storing the N x N matrix: the smallest v with
// #(distances <= v) > floor(0.5 (N - 1)) by bisection over v in [0, 256], recomputing the row's distances per step
// (the descriptors of the point are read through uniform addresses).  Best = first row with the smallest median.
__global__ __launch_bounds__(64) void k_distinctive(const int* __restrict__ start, const uint8_t* __restrict__ desc,
                                                    int* __restrict__ bestIdx) {
  const int mp = blockIdx.x, lane = threadIdx.x;
  const int s0 = start[mp], N = start[mp + 1] - s0;
  if (N <= 0) { if (lane == 0) bestIdx[mp] = -1; return; }
  const uint8_t* D = desc + (size_t)s0 * 32;
  const int k = (N - 1) >> 1;                       // (size_t)(0.5 * (N - 1))
  unsigned best = ~0u;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    unsigned key = ~0u;
    if (i < N) {
      const Desc di = load_desc(D + (size_t)i * 32);
      int lo = 0, hi = 256;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int c = 0;
        for (int j = 0; j < N; ++j) c += hamming(di, load_desc(D + (size_t)j * 32)) <= mid ? 1 : 0;
        if (c > k) hi = mid; else lo = mid + 1;
      }
      key = ((unsigned)lo << 16) | (unsigned)i;     // N < 65536
    }
    best = key < best ? key : best;
  }
  best = wave_min_u32(best);
  if (lane == 0) bestIdx[mp] = (int)(best & 0xFFFFu);
}

__device__ void three_maxima(const int* cnt, int L, int& ind1, int& ind2, int& ind3) {  // ORBmatcher.cc:1844-1876
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = cnt[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

__global__ __launch_bounds__(256) void k_rot_filter(const int* __restrict__ nFv, const int* __restrict__ fImg, int cap,
                                                    int checkOri, int* __restrict__ matchF, const int* __restrict__ binF,
                                                    int* __restrict__ nmatches) {
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep[3];
  __shared__ int total;
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int nF = nFv[fImg[pair]];
  int* mF = matchF + (size_t)pair * cap;
  const int* bF = binF + (size_t)pair * cap;
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int i = tid; i < nF; i += 256)
    if (mF[i] >= 0) { atomicAdd(&total, 1); if (checkOri) atomicAdd(&hist[bF[i]], 1); }
  __syncthreads();
  if (checkOri) {
    if (tid == 0) three_maxima(hist, HISTO_LENGTH, keep[0], keep[1], keep[2]);
    __syncthreads();
    for (int i = tid; i < nF; i += 256)
      if (mF[i] >= 0) {
        const int b = bF[i];
        if (b != keep[0] && b != keep[1] && b != keep[2]) { mF[i] = -1; atomicSub(&total, 1); }
      }
    __syncthreads();
  }
  if (tid == 0) nmatches[pair] = total;
}

__global__ void k_fill_i32(int* p, size_t n, int v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

// =====================================================================================================
struct morb_matcher {
  int device = 0;
  hipStream_t stream = nullptr;
  // workspace (grown on demand)
  unsigned long long *d_sortA = nullptr, *d_sortB = nullptr;
  int* d_bin = nullptr;
  int* d_sad = nullptr;
  uint8_t* d_stereoRec = nullptr;   // k_stereo_prep's per-frame records
  float *d_scale = nullptr, *d_invScale = nullptr;
  int* d_idx = nullptr;
  size_t sortElems = 0, binElems = 0, sadElems = 0, idxElems = 0, stereoRecBytes = 0;
  void* ws[8] = {nullptr};   // generic workspaces for projection.hip
  size_t wsBytes[8] = {0};
  std::vector<void*> retired;   // outgrown workspaces: kernels queued on a caller's stream may still read them, so they are freed with the handle
  // small constant tables (PredictScale thresholds, camera parameters): device copy + the host bytes it was made from, so that a
  // call with the same table neither uploads nor waits (projection.hip: morb_matcher_const)
  // Four most-recently-used copies per slot (a KB8 rig alternates the left and the right camera's table on one slot); a live copy is never
  // rewritten — a kernel of an earlier call, queued on another caller's stream, may still be reading it.
  struct ConstCopy { void* d = nullptr; std::vector<uint8_t> host; unsigned long long used = 0; };
  struct ConstSlot { ConstCopy way[4]; unsigned long long clock = 0; };
  ConstSlot consts[4];
};

namespace {
template <typename T>
int grow(T*& p, size_t& have, size_t need) {
  if (have >= need) return MORB_OK;
  if (p) (void)hipFree(p);
  p = nullptr;
  have = 0;
  MORB_HIP_CHECK(hipMalloc(&p, sizeof(T) * need));
  have = need;
  return MORB_OK;
}
}  // namespace

extern "C" {

int morb_matcher_create(morb_matcher** out, int device) {
  MORB_REQUIRE(out, MORB_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  MORB_HIP_CHECK(hipGetDeviceCount(&ndev));
  MORB_REQUIRE(device >= 0 && device < ndev, MORB_ERR_INVALID, "no such HIP device");
  MORB_HIP_CHECK(hipSetDevice(device));
  morb_matcher* m = new morb_matcher();
  m->device = device;
  // the handle's own stream (used when the caller passes stream = NULL) is a BLOCKING stream: it is implicitly ordered with the
  // legacy default stream, so callers that prepare inputs / consume outputs there (torch's default stream) need no events
  if (hipStreamCreateWithFlags(&m->stream, hipStreamDefault) != hipSuccess) {
    delete m;
    set_error("cannot create stream");
    return MORB_ERR_HIP;
  }
  if (hipMalloc(&m->d_scale, sizeof(float) * 32) != hipSuccess || hipMalloc(&m->d_invScale, sizeof(float) * 32) != hipSuccess) {
    set_error("hipMalloc failed");
    delete m;
    return MORB_ERR_HIP;
  }
  *out = m;
  return MORB_OK;
}

int morb_matcher_sync(morb_matcher* m) {
  MORB_REQUIRE(m, MORB_ERR_INVALID, "NULL matcher");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  MORB_HIP_CHECK(hipStreamSynchronize(m->stream));
  return MORB_OK;
}

void morb_matcher_destroy(morb_matcher* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipStreamSynchronize(m->stream);
  auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  F(m->d_sortA); F(m->d_sortB); F(m->d_bin); F(m->d_sad); F(m->d_stereoRec); F(m->d_scale); F(m->d_invScale); F(m->d_idx);
  for (auto& w : m->ws) F(w);
  for (auto& w : m->retired) F(w);
  for (auto& c : m->consts) for (auto& k : c.way) F(k.d);
  (void)hipStreamDestroy(m->stream);
  delete m;
}

int morb_matcher_device(const morb_matcher* m) { return m->device; }
void* morb_matcher_stream(const morb_matcher* m) { return (void*)m->stream; }
int morb_matcher_workspace(morb_matcher* m, int which, size_t bytes, void** out) {
  MORB_REQUIRE(m && out && which >= 0 && which < 8, MORB_ERR_INVALID, "bad workspace request");
  if (m->wsBytes[which] < bytes) {
    // no device-wide wait and no hipFree here (hipFree waits for the whole device: the Tracking thread would stall behind a
    // LocalBundleAdjustment running on another stream): the outgrown buffer is retired and the new one is half as large again
    // as asked, so the retired bytes stay below twice the final size
    void* fresh = nullptr;
    const size_t want = bytes + bytes / 2;
    if (hipMalloc(&fresh, want) != hipSuccess) {
      (void)hipGetLastError();
      MORB_HIP_CHECK(hipMalloc(&fresh, bytes));
      m->wsBytes[which] = bytes;
    } else {
      m->wsBytes[which] = want;
    }
    if (m->ws[which]) m->retired.push_back(m->ws[which]);
    m->ws[which] = fresh;
  }
  *out = m->ws[which];
  return MORB_OK;
}
// Device copy of a small host table that rarely changes (level thresholds, camera parameters).  Same bytes as the last call on
// this slot: no upload, no wait.  Otherwise the table is uploaded in stream order and the call waits for the copy.
int morb_matcher_const(morb_matcher* m, int slot, const void* host, size_t bytes, void** d_out, void* stream) {
  MORB_REQUIRE(m && host && d_out && slot >= 0 && slot < 4 && bytes > 0, MORB_ERR_INVALID, "bad constant-table request");
  morb_matcher::ConstSlot& c = m->consts[slot];
  int victim = 0;
  for (int w = 0; w < 4; ++w) {
    morb_matcher::ConstCopy& k = c.way[w];
    if (k.d && k.host.size() == bytes && memcmp(k.host.data(), host, bytes) == 0) { k.used = ++c.clock; *d_out = k.d; return MORB_OK; }
    if (c.way[w].used < c.way[victim].used) victim = w;   // (an empty way has used == 0)
  }
  morb_matcher::ConstCopy& k = c.way[victim];
  void* fresh = nullptr;
  MORB_HIP_CHECK(hipMalloc(&fresh, bytes));
  if (k.d) m->retired.push_back(k.d);   // freed with the handle: never overwritten while a kernel may read it
  k.d = fresh;
  k.host.assign((const uint8_t*)host, (const uint8_t*)host + bytes);
  k.used = ++c.clock;
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  MORB_HIP_CHECK(hipMemcpyAsync(k.d, k.host.data(), bytes, hipMemcpyHostToDevice, st));
  MORB_HIP_CHECK(hipStreamSynchronize(st));   // (k.host may be reassigned by the next miss on this way)
  *d_out = k.d;
  return MORB_OK;
}
int morb_bow_sort_images(morb_matcher* m, int nimg, const int* d_node, const int* d_count, int cap, unsigned long long** d_sorted,
                         void* stream) {
  int P = 1;
  while (P < cap) P <<= 1;
  MORB_REQUIRE((size_t)P * 8 <= 160 * 1024, MORB_ERR_UNSUPPORTED, "too many features per frame for the LDS sort");
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  int rc = grow(m->d_sortA, m->sortElems, (size_t)nimg * cap);
  if (rc != MORB_OK) return rc;
  MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_sort), hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  hipLaunchKernelGGL(k_bow_sort, dim3(nimg), dim3(P / 2 < 64 ? 64 : P / 2 > 1024 ? 1024 : P / 2), (size_t)P * 8, st, d_node, d_count, cap, P, m->d_sortA);
  *d_sorted = m->d_sortA;
  return MORB_OK;
}

// ---- feature slabs: what one GPU ships to another so that a frame can be matched against its predecessor ----
// slab of S frames, capacity cap: [S][cap] keypoint records (28 B) | [S][cap][32] descriptors | [S][cap] BoW node ids (int32) | [S] counts (int32)
size_t morb_feature_slab_bytes(int S, int cap) {
  if (S <= 0 || cap <= 0) return 0;
  return (size_t)S * cap * (sizeof(morb_keypoint) + 32 + sizeof(int)) + (size_t)S * sizeof(int);
}
namespace {
// one workgroup per (frame, array): 32-bit words, coalesced; dir 0 = pool rows -> slab, 1 = slab -> pool rows
__global__ __launch_bounds__(256) void k_slab_copy(int S, int cap, const int* __restrict__ rowIdx, uint32_t* kps, uint32_t* desc, uint32_t* node, int* count,
                                                   uint32_t* slab, int dir) {
  const int f = blockIdx.x, which = blockIdx.y;
  const int row = rowIdx ? rowIdx[f] : f;
  const size_t wK = (size_t)cap * 7, wD = (size_t)cap * 8, wN = (size_t)cap;
  uint32_t* sK = slab; uint32_t* sD = sK + (size_t)S * wK; uint32_t* sN = sD + (size_t)S * wD; uint32_t* sC = sN + (size_t)S * wN;
  uint32_t *a, *b; size_t n;
  if (which == 0) { a = kps + (size_t)row * wK; b = sK + (size_t)f * wK; n = wK; }
  else if (which == 1) { a = desc + (size_t)row * wD; b = sD + (size_t)f * wD; n = wD; }
  else { a = node ? node + (size_t)row * wN : nullptr; b = sN + (size_t)f * wN; n = wN; }
  if (which == 2 && threadIdx.x == 0) { if (dir == 0) sC[f] = (uint32_t)min(max(count[row], 0), cap); else count[row] = (int)min(sC[f], (uint32_t)cap); }   // (a count never exceeds the row pitch)
  if (!a) { if (dir == 0) for (size_t i = threadIdx.x; i < n; i += 256) b[i] = 0xFFFFFFFFu; return; }
  if (dir == 0) for (size_t i = threadIdx.x; i < n; i += 256) b[i] = a[i];
  else for (size_t i = threadIdx.x; i < n; i += 256) a[i] = b[i];
}
}  // namespace
int morb_feature_slab_pack(morb_matcher* m, int S, int cap, const int* d_rows, const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node,
                           const int* d_count, void* d_slab, void* stream) {
  MORB_REQUIRE(m && S > 0 && cap > 0 && d_kps && d_desc && d_count && d_slab, MORB_ERR_INVALID, "bad argument");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  hipLaunchKernelGGL(k_slab_copy, dim3(S, 3), dim3(256), 0, st, S, cap, d_rows, (uint32_t*)d_kps, (uint32_t*)d_desc, (uint32_t*)d_node, (int*)d_count,
                     (uint32_t*)d_slab, 0);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}
int morb_feature_slab_unpack(morb_matcher* m, int S, int cap, const void* d_slab, const int* d_rows, morb_keypoint* d_kps, uint8_t* d_desc, int* d_node,
                             int* d_count, void* stream) {
  MORB_REQUIRE(m && S > 0 && cap > 0 && d_kps && d_desc && d_count && d_slab, MORB_ERR_INVALID, "bad argument");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  hipLaunchKernelGGL(k_slab_copy, dim3(S, 3), dim3(256), 0, st, S, cap, d_rows, (uint32_t*)d_kps, (uint32_t*)d_desc, (uint32_t*)d_node, d_count,
                     (uint32_t*)const_cast<void*>(d_slab), 1);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_hamming_pairs(morb_matcher* m, const uint8_t* d_a, const uint8_t* d_b, int n, int* d_out, void* stream) {
  MORB_REQUIRE(m && d_a && d_b && d_out && n >= 0, MORB_ERR_INVALID, "bad argument");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  if (n) hipLaunchKernelGGL(k_hamming_pairs, dim3(div_up(n, 256)), dim3(256), 0, st, d_a, d_b, n, d_out);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_hamming_knn2_batch(morb_matcher* m, int nprob, const uint8_t* d_query, const int* d_nq, int qPitch,
                            const int* d_qOff, const uint8_t* d_train, const int* d_nt, int tPitch, const int* d_tOff,
                            int* d_idx, int* d_dist, void* stream) {
  MORB_REQUIRE(m && d_query && d_train && d_nq && d_nt && d_idx && d_dist, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nprob > 0 && qPitch > 0 && tPitch > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  hipLaunchKernelGGL(k_knn2, dim3(div_up(qPitch, 256), nprob), dim3(256), 0, st, d_query, d_nq, qPitch, d_train, d_nt,
                     tPitch, d_qOff, d_tOff, d_idx, d_dist);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_stereo_match_batch(morb_matcher* m, const morb_extractor* e, int nframes, const morb_keypoint* d_kps,
                            const uint8_t* d_desc, const int* d_count, int cap, float mbf, float mb, float* d_uRight,
                            float* d_depth, void* stream) {
  MORB_REQUIRE(m && e && d_kps && d_desc && d_count && d_uRight && d_depth, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && mb > 0.f, MORB_ERR_INVALID, "bad sizes");
  MORB_REQUIRE(e->W > 0 && e->nimgLast >= 2 * nframes, MORB_ERR_INVALID,
               "the extractor must have processed the 2*nframes images (left = 2f, right = 2f+1) of this batch");
  MORB_REQUIRE(e->device == m->device, MORB_ERR_INVALID, "extractor and matcher live on different devices");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  int rc = grow(m->d_sad, m->sadElems, (size_t)nframes * cap);
  if (rc != MORB_OK) return rc;
  StereoGeom sg = {};
  for (int l = 0; l < 16; ++l) {
    const LevelGeom& g = e->geom[l < e->nlevels ? l : e->nlevels - 1];
    sg.pyrOff[l] = g.pyrOff; sg.pyrImg[l] = g.pyrImg; sg.pstride[l] = g.pstride; sg.w[l] = g.w;
    sg.scale[l] = e->scale[l < e->nlevels ? l : e->nlevels - 1]; sg.invScale[l] = e->invScale[l < e->nlevels ? l : e->nlevels - 1];
  }
  sg.nRows = e->geom[0].h;
  const StereoRec ro = stereo_rec(cap);
  rc = grow(m->d_stereoRec, m->stereoRecBytes, (size_t)nframes * ro.bytes);
  if (rc != MORB_OK) return rc;
  const size_t prepSmem = (size_t)ro.bytes + SM_MAXB * sizeof(int);
  const size_t stereoSmem = (size_t)ro.bytes + 4 * SM_KQ * (32 + 8) + 4 * SM_PCAP * sizeof(uint32_t);
  MORB_REQUIRE(prepSmem <= 160 * 1024 && stereoSmem <= 160 * 1024 && cap < 65536 && e->nlevels <= 16, MORB_ERR_UNSUPPORTED, "too many keypoints per image for the LDS-resident right-keypoint table");
  MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stereo_prep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)prepSmem));
  MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stereo_match), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stereoSmem));
  const int lk = sm_lk_for(nframes);
  hipLaunchKernelGGL(k_stereo_prep, dim3(nframes), dim3(nframes <= 16 ? 1024 : 256), prepSmem, st, sg, d_kps, d_count, cap, m->d_stereoRec);
  hipLaunchKernelGGL(k_stereo_match, dim3(div_up(cap, lk), nframes), dim3(256), stereoSmem, st, sg, e->d_pyr, d_kps, d_desc,
                     d_count, cap, mbf, mb, d_uRight, d_depth, m->d_sad, lk, m->d_stereoRec);
  if (nframes <= 16) hipLaunchKernelGGL(k_stereo_median<16>, dim3(nframes), dim3(64 * 16), 0, st, d_count, cap, d_uRight, d_depth, m->d_sad);
  else hipLaunchKernelGGL(k_stereo_median<4>, dim3(nframes), dim3(64 * 4), 0, st, d_count, cap, d_uRight, d_depth, m->d_sad);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

namespace {
// the LDS-staged descent when the tree is at most 10-ary (DBoW2's ORB vocabulary: k = 10) and the batch is large enough to pay for the staging
int launch_bow_transform(hipStream_t st, int nimg, const uint8_t* d_desc, const int* d_count, int cap, const uint8_t* d_nodeDesc,
                         const int* d_firstChild, int k, int kmax, int L, int levelsup, int* d_wordId, int* d_nodeId, const int* d_childCount) {
  const int nChunks = div_up(cap, 64) * nimg;
  // Measured (profiles/r05/README.md): alone on the chip the global-memory descent takes 0.112 ms per 512 frames, the staged forms 0.108 - 0.141
  // (three staged levels cost the occupancy 47 KB of LDS per workgroup); inside the bench step all of them land within +-0.3 % of each other —
  // the descent's six round trips hit L2 and are hidden by the kernel's 20 000 workgroups.  The staged kernels are kept for vocabularies that do
  // not fit L2's share (MORB_BOW_STAGE = 2 | 3 selects them); the default stays the global-memory descent.
  const char* sv = getenv("MORB_BOW_STAGE");
  if (sv && kmax <= BT_K && nChunks >= 64) {
    const int nlv = atoi(sv);
    const char* gv = getenv("MORB_BOW_GRID");
    const int grid = std::min(nChunks, gv ? atoi(gv) : 1024);   // persistent workgroups: the launch runs beside the extractor's kernels
    if (nlv == 3) {
      MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_transform_lds<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BtLds<3>)));
      hipLaunchKernelGGL(k_bow_transform_lds<3>, dim3(grid), dim3(256), sizeof(BtLds<3>), st, d_desc, d_count, cap, nimg, d_nodeDesc, d_firstChild, k, L,
                         levelsup, d_wordId, d_nodeId, d_childCount);
    } else {
      hipLaunchKernelGGL(k_bow_transform_lds<2>, dim3(grid), dim3(256), sizeof(BtLds<2>), st, d_desc, d_count, cap, nimg, d_nodeDesc, d_firstChild, k, L,
                         levelsup, d_wordId, d_nodeId, d_childCount);
    }
  } else {
    hipLaunchKernelGGL(k_bow_transform, dim3(div_up(4 * cap, 256), nimg), dim3(256), 0, st, d_desc, d_count, cap, d_nodeDesc, d_firstChild, k, L,
                       levelsup, d_wordId, d_nodeId, d_childCount);
  }
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}
}  // namespace

int morb_bow_transform_batch(morb_matcher* m, int nimg, const uint8_t* d_desc, const int* d_count, int cap,
                             const uint8_t* d_nodeDesc, const int* d_firstChild, int k, int L, int levelsup,
                             int* d_wordId, int* d_nodeId, void* stream) {
  MORB_REQUIRE(m && d_desc && d_count && d_nodeDesc && d_firstChild && d_wordId && d_nodeId, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nimg > 0 && cap > 0 && k > 0 && L > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  return launch_bow_transform(st, nimg, d_desc, d_count, cap, d_nodeDesc, d_firstChild, k, k, L, levelsup, d_wordId, d_nodeId, nullptr);
}

int morb_bow_transform_tree_batch(morb_matcher* m, int nimg, const uint8_t* d_desc, const int* d_count, int cap,
                                  const uint8_t* d_nodeDesc, const int* d_firstChild, const int* d_childCount, int L, int levelsup,
                                  int* d_wordId, int* d_nodeId, void* stream) {
  MORB_REQUIRE(m && d_desc && d_count && d_nodeDesc && d_firstChild && d_childCount && d_wordId && d_nodeId, MORB_ERR_INVALID,
               "NULL argument");
  MORB_REQUIRE(nimg > 0 && cap > 0 && L > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  return launch_bow_transform(st, nimg, d_desc, d_count, cap, d_nodeDesc, d_firstChild, 0, BT_K, L, levelsup, d_wordId, d_nodeId, d_childCount);   // (nodes with more than ten children are handled inside)
}

// ---- DBoW2 text vocabulary (TemplatedVocabulary::loadFromTextFile, TemplatedVocabulary.h:1338-1420) ------------------
struct morb_vocabulary {
  int k = 0, L = 0, scoring = 0, weighting = 0;
  std::vector<uint8_t> desc;       // [nNodes][32]
  std::vector<int> firstChild;     // [nNodes] first child id or -1 (leaf)
  std::vector<int> childCount;     // [nNodes]
  std::vector<int> parent;         // [nNodes]
  std::vector<int> wordId;         // [nNodes] word id of a leaf (order of appearance, :1403-1408) or -1
  std::vector<double> weight;      // [nNodes] WordValue is double in DBoW2
};

int morb_vocabulary_load_text(const char* path, morb_vocabulary** out) {
  MORB_REQUIRE(path && out, MORB_ERR_INVALID, "NULL argument");
  *out = nullptr;
  FILE* f = fopen(path, "r");
  MORB_REQUIRE(f, MORB_ERR_INVALID, "cannot open the vocabulary file");
  std::unique_ptr<morb_vocabulary> v(new morb_vocabulary());
  int n1 = 0, n2 = 0;
  if (fscanf(f, "%d %d %d %d", &v->k, &v->L, &n1, &n2) != 4 || v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 ||
      n2 < 0 || n2 > 3) {   // :1356-1360
    fclose(f);
    set_error("Vocabulary loading failure: This is not a correct text file!");
    return MORB_ERR_INVALID;
  }
  v->scoring = n1; v->weighting = n2;
  // node 0 = root
  v->desc.assign(32, 0); v->firstChild.assign(1, -1); v->childCount.assign(1, 0); v->parent.assign(1, -1); v->wordId.assign(1, -1);
  v->weight.assign(1, 0.0);
  int nWords = 0;
  for (;;) {
    int pid = 0, isLeaf = 0;
    if (fscanf(f, "%d %d", &pid, &isLeaf) != 2) break;
    const int nid = (int)v->parent.size();
    int d[32];
    bool ok = pid >= 0 && pid < nid;
    for (int i = 0; i < 32 && ok; ++i) ok = fscanf(f, "%d", &d[i]) == 1;
    double w = 0;
    if (ok) ok = fscanf(f, "%lf", &w) == 1;
    if (!ok) { fclose(f); set_error("malformed vocabulary node %d", nid); return MORB_ERR_INVALID; }
    v->parent.push_back(pid);
    for (int i = 0; i < 32; ++i) v->desc.push_back((uint8_t)d[i]);
    v->weight.push_back((double)w);
    v->firstChild.push_back(-1); v->childCount.push_back(0);
    v->wordId.push_back(isLeaf > 0 ? nWords++ : -1);
    // the descent kernels address children as [firstChild, firstChild + childCount): DBoW2 creates the children of a node
    // together (HKmeansStep), so they are contiguous in every vocabulary it writes; anything else is refused
    if (v->childCount[pid] == 0) v->firstChild[pid] = nid;
    else if (v->firstChild[pid] + v->childCount[pid] != nid) {
      fclose(f);
      set_error("children of vocabulary node %d are not contiguous", pid);
      return MORB_ERR_UNSUPPORTED;
    }
    v->childCount[pid]++;
  }
  fclose(f);
  *out = v.release();
  return MORB_OK;
}
void morb_vocabulary_destroy(morb_vocabulary* v) { delete v; }
int morb_vocabulary_info(const morb_vocabulary* v, int* k, int* L, int* nNodes, int* nWords) {
  MORB_REQUIRE(v, MORB_ERR_INVALID, "NULL vocabulary");
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (nNodes) *nNodes = (int)v->parent.size();
  if (nWords) { int n = 0; for (int w : v->wordId) n += w >= 0; *nWords = n; }
  return MORB_OK;
}
int morb_vocabulary_arrays(const morb_vocabulary* v, uint8_t* nodeDesc, int* firstChild, int* childCount, int* wordId, float* weight) {
  MORB_REQUIRE(v, MORB_ERR_INVALID, "NULL vocabulary");
  const size_t n = v->parent.size();
  if (nodeDesc) memcpy(nodeDesc, v->desc.data(), n * 32);
  if (firstChild) memcpy(firstChild, v->firstChild.data(), n * sizeof(int));
  if (childCount) memcpy(childCount, v->childCount.data(), n * sizeof(int));
  if (wordId) memcpy(wordId, v->wordId.data(), n * sizeof(int));
  if (weight) for (size_t i = 0; i < n; ++i) weight[i] = (float)v->weight[i];
  return MORB_OK;
}

int morb_vocabulary_weights(const morb_vocabulary* v, double* weight, int* scoring, int* weighting) {
  MORB_REQUIRE(v, MORB_ERR_INVALID, "NULL vocabulary");
  if (weight) memcpy(weight, v->weight.data(), v->weight.size() * sizeof(double));
  if (scoring) *scoring = v->scoring;
  if (weighting) *weighting = v->weighting;
  return MORB_OK;
}

static int search_by_bow_impl(morb_matcher* m, int npairs, const int* d_kfImg, const int* d_fImg, int nimg,
                              const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node, const int* d_count,
                              const uint8_t* d_hasMP, int cap, float nnratio, int checkOri, int* d_matchF,
                              int* d_nmatches, const int* d_nLeft, void* stream, const uint8_t* d_hasMP2 = nullptr,
                              const int* d_nValid = nullptr) {
  MORB_REQUIRE(m && d_kfImg && d_fImg && d_kps && d_desc && d_node && d_count && d_hasMP && d_matchF && d_nmatches,
               MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(npairs > 0 && nimg > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  int P = 1;
  while (P < cap) P <<= 1;
  MORB_REQUIRE((size_t)P * 8 <= 160 * 1024, MORB_ERR_UNSUPPORTED, "too many features per frame for the LDS sort");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  int rc = grow(m->d_sortA, m->sortElems, (size_t)nimg * cap);
  if (rc != MORB_OK) return rc;
  rc = grow(m->d_bin, m->binElems, (size_t)npairs * cap);
  if (rc != MORB_OK) return rc;
  MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_sort), hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  hipLaunchKernelGGL(k_bow_sort, dim3(nimg), dim3(P / 2 < 64 ? 64 : P / 2 > 1024 ? 1024 : P / 2), (size_t)P * 8, st, d_node, d_count, cap, P, m->d_sortA);
  const size_t nm = (size_t)npairs * cap;
  hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, d_matchF, nm, -1);
  int* matched2 = nullptr;
  if (d_hasMP2) {   // vbMatched2 of the general path (nodes with very many features)
    void* w = nullptr;
    rc = morb_matcher_workspace(m, 7, sizeof(int) * nm, &w);
    if (rc != MORB_OK) return rc;
    matched2 = (int*)w;
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, matched2, nm, 0);
  }
  MORB_REQUIRE(cap < 65536 && (size_t)cap * 18 <= 160 * 1024, MORB_ERR_UNSUPPORTED, "too many features per frame for the LDS-resident node tables");
  MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_match), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)cap * 18)));
  hipLaunchKernelGGL(k_bow_match, dim3(BM_NB, npairs), dim3(256), (size_t)cap * 18, st, m->d_sortA, m->d_sortA, d_count, d_count,
                     cap, d_desc, d_hasMP, d_kps, d_desc, d_kps, d_kfImg, d_fImg, nnratio, d_matchF, m->d_bin, d_nLeft, d_hasMP2,
                     d_nValid, matched2);
  // the table is indexed by frame feature (M3) or by keyframe-1 feature (SearchByBoW(KF, KF))
  hipLaunchKernelGGL(k_rot_filter, dim3(npairs), dim3(256), 0, st, d_count, d_hasMP2 ? d_kfImg : d_fImg, cap, checkOri, d_matchF,
                     m->d_bin, d_nmatches);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_search_by_bow_batch(morb_matcher* m, int npairs, const int* d_kfImg, const int* d_fImg, int nimg,
                             const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node, const int* d_count,
                             const uint8_t* d_hasMP, int cap, float nnratio, int checkOri, int* d_matchF,
                             int* d_nmatches, void* stream) {
  return search_by_bow_impl(m, npairs, d_kfImg, d_fImg, nimg, d_kps, d_desc, d_node, d_count, d_hasMP, cap, nnratio, checkOri,
                            d_matchF, d_nmatches, nullptr, stream);
}

int morb_distinctive_descriptors_batch(morb_matcher* m, int nMP, const int* d_start, const uint8_t* d_desc, int* d_bestIdx,
                                       void* stream) {
  MORB_REQUIRE(m && d_start && d_desc && d_bestIdx && nMP >= 0, MORB_ERR_INVALID, "bad argument");
  MORB_HIP_CHECK(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  if (nMP) hipLaunchKernelGGL(k_distinctive, dim3(nMP), dim3(64), 0, st, d_start, d_desc, d_bestIdx);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_search_by_bow_kfkf_batch(morb_matcher* m, int npairs, const int* d_kf1Img, const int* d_kf2Img, const int* d_nValid, int nimg,
                                  const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node, const int* d_count,
                                  const uint8_t* d_hasMP, int cap, float nnratio, int checkOri, int* d_match12, int* d_nmatches,
                                  void* stream) {
  return search_by_bow_impl(m, npairs, d_kf1Img, d_kf2Img, nimg, d_kps, d_desc, d_node, d_count, d_hasMP, cap, nnratio, checkOri,
                            d_match12, d_nmatches, nullptr, stream, d_hasMP, d_nValid);
}

int morb_search_by_bow_fisheye_batch(morb_matcher* m, int npairs, const int* d_kfImg, const int* d_fImg, const int* d_nLeft,
                                     int nimg, const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node,
                                     const int* d_count, const uint8_t* d_hasMP, int cap, float nnratio, int checkOri,
                                     int* d_matchF, int* d_nmatches, void* stream) {
  MORB_REQUIRE(d_nLeft, MORB_ERR_INVALID, "NULL argument");
  return search_by_bow_impl(m, npairs, d_kfImg, d_fImg, nimg, d_kps, d_desc, d_node, d_count, d_hasMP, cap, nnratio, checkOri,
                            d_matchF, d_nmatches, d_nLeft, stream);
}

#ifdef MORB_FAST_TIMING
int morb_bow_timing(unsigned long long* out, int reset) {
  (void)reset;
  MORB_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bowTrace), sizeof(unsigned long long) * 16384 * 4));
  return 0;
}
#endif

}  // extern "C"
