// ORB extractor for MI355X (gfx950): batched, device-resident implementation of
// ORB_SLAM3::ORBextractor::operator() (reference src/ORBextractor.cc:1006-1086) behind the C ABI of
// include/morb_hip.h.  Written for CDNA4: wave64 ballots/popcounts for ordered compaction, LDS tiles for the
// stencils, one wave per (image, level) for the order-exact quadtree, integer arithmetic bit-identical to the
// CPU oracle (oracle/).  Built with -ffp-contract=off: the few float expressions (fastAtan2, the rBRIEF
// rotation, pt *= scale) must round exactly like the reference's separate mul/add.
//
// HBM layout (level-major slabs; one batch = nimg images of one size):
//   pyramid : for level l, image i : (h_l + 38) rows x pstride_l bytes, 19-px BORDER_REFLECT_101 pad included
//             (= mvImagePyramid[l] with its pad, ORBextractor.cc:1088-1112); pstride_l is a multiple of 64.
//   blur    : for level l, image i : h_l rows x bstride_l bytes (GaussianBlur 7x7 s=2 of the level, :1049-1050)
//   cand    : per image, per FAST cell (all levels, flat cell index): cellCap packed keys + a count
//   sel     : per image, per level: keys chosen by DistributeOctTree, in the reference's list order
// Kernels per batch: level0 pad-copy, 7 resizes, blur, FAST(+score+NMS+cell fallback), distribute, layout,
// describe (IC_Angle + rBRIEF + final keypoint records).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "libm_f32.h"
#include "wave.h"
#include "extractor_internal.h"
#include "quadtree.h"

namespace morb {
std::string& last_error() {
  static thread_local std::string e;
  return e;
}
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
}
}  // namespace morb

using namespace morb;

namespace {

#ifndef MORB_TEAM_MAX_IMAGES
#define MORB_TEAM_MAX_IMAGES 64   // (round 6, teams of 16 waves + the full sweeps at once: 16 -> 64 images.  tools/r06_team_max.sh: 752 x 480, 32 / 64 images per call +6.6 / +6 - 8 %
#endif                            //  end to end; 1920 x 1080, 32 / 64 images +67 / +29 % (12.8 -> 21.4 k frames/s at 16 stereo frames per step); at 128 images the teams lose 11 %)
constexpr int kTeamMaxImages = MORB_TEAM_MAX_IMAGES;
#ifndef MORB_PYR_CHUNK_IMAGES
#define MORB_PYR_CHUNK_IMAGES (1 << 30)
#endif
constexpr int kPyrChunkImages = MORB_PYR_CHUNK_IMAGES;   // images per group of pyramid launches (morb_extract_batch)
#ifndef MORB_TEAM_MIN_PIXELS
#define MORB_TEAM_MIN_PIXELS 160000   // k_distribute: levels of at least this many pixels are worked by a team of waves in the team packing
#endif
#ifndef MORB_QT_KEYF
#define MORB_QT_KEYF 200   // LDS key capacity of level 0, in percent of (pixels / 233); 135 / 160 measured: no change (the big bin still takes a CU alone)
#endif   // k_distribute: calls with at most this many images use the team packing of the big levels

// bit_pattern_31_ (ORBextractor.cc:147-404) as floats: k_describe multiplies the coordinates by cos / sin (the conversion of 16 integers per wave was 16 of its ~575
// vector instructions, and the kernel — like the whole step — is bound by vector-instruction issue)
__constant__ __align__(16) float c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[16];


// ---------------------------------------------------------------------------------------------------
// K1a: level 0 = copyMakeBorder(image, BORDER_REFLECT_101)  (ORBextractor.cc:1108)
__device__ __forceinline__ int reflect101(int p, int len) {
  // one reflection is enough for |pad| = 19 < len
  if (p < 0) p = -p;
  if (p >= len) p = 2 * (len - 1) - p;
  return p;
}

#ifndef MORB_PY_ROWS
#define MORB_PY_ROWS 8
#endif
constexpr int PY_ROWS = MORB_PY_ROWS;  // rows per thread in k_resize_gather (block = 64 x 4 threads -> 256 px x 32 rows)
#ifndef MORB_PR
#define MORB_PR 8
#endif
constexpr int PR = MORB_PR;    // rows per item of k_resize
constexpr int P0R = 4;   // rows per item of the level-0 copy
// Workgroup -> tile mapping of the pyramid kernels.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with
// its own L2); with the plain (x, y, image) order the tiles of one image are spread over all of them and the source rows that
// vertically adjacent tiles share are fetched once per XCD.  Remapped, XCD k works through images k, k + 8, ... tile by tile.
struct PyTile { int bx, by, img; };
__device__ __forceinline__ PyTile py_tile() {
  if ((gridDim.z & 7) == 0) {   // (measured at 512 images: pyramid 557 -> 540 us; neutral at 128)
    const unsigned n = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), tiles = gridDim.x * gridDim.y;
    const unsigned xcd = n & 7u, slot = n >> 3, g = slot / tiles, t = slot - g * tiles;
    PyTile r; r.img = (int)(xcd + 8u * g); r.by = (int)(t / gridDim.x); r.bx = (int)(t - (unsigned)r.by * gridDim.x);
    // (the divisions run on the vector ALU; the results are wave-uniform and belong in SGPRs)
    r.img = __builtin_amdgcn_readfirstlane(r.img); r.by = __builtin_amdgcn_readfirstlane(r.by); r.bx = __builtin_amdgcn_readfirstlane(r.bx);
    return r;
  }
  PyTile r; r.bx = blockIdx.x; r.by = blockIdx.y; r.img = blockIdx.z;
  return r;
}
// K1b: level l = resize(level l-1, INTER_LINEAR) + copyMakeBorder(BORDER_REFLECT_101|ISOLATED)
// (ORBextractor.cc:1101-1104).  Every padded pixel is computed directly from level l-1 through tables that
// already fold the reflection, so one launch writes interior and pad.
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// General form (any scale factor): four byte gathers per output pixel.  Used only when a level's four-pixel chunks do not fit the
// 8-byte source windows of k_resize (scale factors above ~1.75).
__global__ __launch_bounds__(256) void k_resize_gather(uint8_t* __restrict__ pyr, LevelGeom gs, LevelGeom gd,
                                                       const ResizeTab* __restrict__ xtab, const ResizeTab* __restrict__ ytab, int img0) {
  PyTile pt = py_tile();
  pt.img += img0;   // (the launch covers images img0 .. img0 + gridDim.z - 1)
  const uint8_t* sbase = pyr + gs.pyrOff + (size_t)pt.img * gs.pyrImg + (size_t)EDGE * gs.pstride + EDGE;
  const int px = (pt.bx * 64 + (threadIdx.x & 63)) * 4;
  const int py0 = (pt.by * 4 + (threadIdx.x >> 6)) * PY_ROWS;
  const int H = gd.h + 2 * EDGE;
  if (px >= gd.pstride || py0 >= H) return;
  ResizeTab tx[4];
  const int pxc = px < gd.w + 2 * EDGE ? px : gd.w + 2 * EDGE - 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) tx[k] = xtab[pxc + k];
  uint8_t* dbase = pyr + gd.pyrOff + (size_t)pt.img * gd.pyrImg + px;
  for (int r = 0; r < PY_ROWS; ++r) {
    if (py0 + r >= H) break;
    const ResizeTab ty = ytab[py0 + r];
    const uint8_t* r0 = sbase + (size_t)ty.s0 * gs.pstride;
    const uint8_t* r1 = sbase + (size_t)ty.s1 * gs.pstride;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int h0 = r0[tx[k].s0] * tx[k].c0 + r0[tx[k].s1] * tx[k].c1;
      const int h1 = r1[tx[k].s0] * tx[k].c0 + r1[tx[k].s1] * tx[k].c1;
      const int o = ((((int)ty.c0 * (h0 >> 4)) >> 16) + (((int)ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
      v |= (uint32_t)(o & 0xFF) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dbase + (size_t)(py0 + r) * gd.pstride) = v;
  }
}

// The resize proper.  Round 3 found the stage bound by VALU issue and by idle lanes, not by memory (27 lane-cycles per pixel against
// 16.6 VALU instructions per pixel; a 256-px-wide wave tile leaves 15 - 45 % of the lanes of levels 2 - 6 outside the image).
//  * work item = one destination dword (4 px) x PR rows; the items of a level are numbered linearly (row group major) and dealt to
//    lanes 256 per workgroup, so every lane of every wave but the level's last works;
//  * per column chunk the host precomputes (PyrCol) the 8-byte source window's offset, a v_perm_b32 selector per pixel that drops the
//    pixel's left and right source bytes into the two halves of a dword, and the coefficient pair in the same layout: the horizontal
//    pass of a pixel is v_perm_b32 + v_dot2_u32_u16;
//  * the vertical term (c * (h >> 4)) >> 16 is v_mul_hi_u32_u24(c << 12, h & ~15): the row table (PyrRow) carries c << 12;
//  * rows are no longer wave-uniform (a wave's items can span two or three row groups), so the workgroup stages its slice of the row
//    table in LDS and every lane reads its eight entries from there (ds_read_b128, off the vector-memory path);
//  * (c0 + c1 + 2) >> 2 and the byte packing run two pixels per instruction (v_pk_lshrrev_b16, one v_perm_b32 per dword).
// Arithmetic identical to cv::resize's fixed-point path (the oracle's resize_linear_u8): 11-bit coefficients, horizontal sums >> 4,
// vertical products >> 16, + 2 >> 2.
struct ResizeArgs {
  const PyrCol* col; const PyrRow* row;
  int nC, nItems, H, dpstride;
  uint32_t magicC;
  unsigned long long dOff, dImg;
};
typedef unsigned short pyr_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b) {   // v_mul_hi_u32_u24 (the masks tell the compiler so; they cost nothing)
  return (uint32_t)(((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) >> 32);
}
__device__ __forceinline__ void resize_items(const uint8_t* __restrict__ sbase, int sstride, uint8_t* __restrict__ dimg,
                                             const ResizeArgs& a, int tile, PyrRow* __restrict__ sRow) {
  const int id0 = tile * 256;
  if (id0 >= a.nItems) return;   // (whole workgroup)
  const int gLo = (int)__umulhi((uint32_t)id0, a.magicC), gHi = (int)__umulhi((uint32_t)min(id0 + 255, a.nItems - 1), a.magicC);
  for (int t = threadIdx.x; t < (gHi - gLo + 1) * PR; t += 256) sRow[t] = a.row[gLo * PR + t];
  __syncthreads();
  const int id = id0 + (int)threadIdx.x;
  if (id >= a.nItems) return;
  const int g = (int)__umulhi((uint32_t)id, a.magicC), c = id - g * a.nC;
  const PyrCol pc = a.col[c];
  const PyrRow* rr = sRow + (g - gLo) * PR;
  // every source dword of the item's rows is requested before the first one is used (no branch inside the row loops: a per-row exit
  // makes the compiler sink each row's loads behind it, one dependent round trip per row)
  // The memory pipe charges a wave's load 16 cycles per dword unless every lane's address is dword aligned, then 16 cycles flat up to
  // 16 bytes per lane (tools/micro/ta_rate.hip: an 8-byte load at a byte-granular address costs 32, an aligned 12-byte one 16): the
  // 8-byte window is fetched as the three aligned dwords around it and cut out with two v_alignbyte_b32.
  uint32_t A0[PR], A1[PR], A2[PR], B0[PR], B1[PR], B2[PR];
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    const int2 sr = *reinterpret_cast<const int2*>(&rr[r].s0);
    struct W3 { uint32_t x, y, z; } wa, wb;
    __builtin_memcpy(&wa, __builtin_assume_aligned(sbase + (uint32_t)(__umul24(sr.x, sstride) + pc.base), 4), 12);
    __builtin_memcpy(&wb, __builtin_assume_aligned(sbase + (uint32_t)(__umul24(sr.y, sstride) + pc.base), 4), 12);
    A0[r] = wa.x; A1[r] = wa.y; A2[r] = wa.z; B0[r] = wb.x; B1[r] = wb.y; B2[r] = wb.z;
  }
  const int row0 = g * PR;
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    const uint2 cy = *reinterpret_cast<const uint2*>(&rr[r].c0s);
    const uint32_t ax = __builtin_amdgcn_alignbyte(A1[r], A0[r], pc.sh), ay = __builtin_amdgcn_alignbyte(A2[r], A1[r], pc.sh);
    const uint32_t bx = __builtin_amdgcn_alignbyte(B1[r], B0[r], pc.sh), by = __builtin_amdgcn_alignbyte(B2[r], B1[r], pc.sh);
    uint32_t sum[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t pa = __builtin_amdgcn_perm(ay, ax, pc.sel[k]), pb = __builtin_amdgcn_perm(by, bx, pc.sel[k]);
      const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, pa), __builtin_bit_cast(pyr_u16x2, pc.coef[k]), 0u, false);
      const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, pb), __builtin_bit_cast(pyr_u16x2, pc.coef[k]), 0u, false);
      sum[k] = mul_hi_u24(cy.x, h0 & 0xFFFFF0u) + mul_hi_u24(cy.y, h1 & 0xFFFFF0u) + 2u;
    }
    pyr_u16x2 lo = __builtin_bit_cast(pyr_u16x2, sum[0] | (sum[1] << 16)), hi = __builtin_bit_cast(pyr_u16x2, sum[2] | (sum[3] << 16));
    lo >>= 2; hi >>= 2;
    const uint32_t v = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
    // rows past the bottom of the last group: the row table repeats its last entry, the store rewrites the last row with the same bytes
#if MORB_EXP == 4
    if (v == 0x12345678u)
#endif
    *reinterpret_cast<uint32_t*>(dimg + (uint32_t)(__umul24(min(row0 + r, a.H - 1), a.dpstride) + c * 4)) = v;
  }
}

__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, unsigned long long sOff, unsigned long long sImg, int sstride, ResizeArgs a) {
  extern __shared__ __align__(16) uint8_t pyr_smem[];
  const PyTile pt = py_tile();
  resize_items(pyr + sOff + (size_t)pt.img * sImg, sstride, pyr + a.dOff + (size_t)pt.img * a.dImg, a, pt.bx, reinterpret_cast<PyrRow*>(pyr_smem));
}

// Level 0 = copyMakeBorder(image, 19, BORDER_REFLECT_101), as items of two kinds: "interior" = one aligned 16-byte destination chunk
// whose sixteen source pixels are consecutive (one unaligned 16-byte load — the caller's image has whatever alignment it has —, one 16-byte store) x P0R rows, and "edge" = one destination
// dword of the reflected pad (or of the ragged ends of the interior) x P0R rows, a dword load + v_perm_b32 with a host-made selector.
struct Level0Args {
  const PyrEdge* edge;
  int nInt, j0, nEdge, nIntItems, nItems, H, h, dpstride;
  uint32_t magicInt, magicEdge;
  unsigned long long dOff, dImg;
};
__device__ __forceinline__ void level0_items(const uint8_t* __restrict__ src, int stride, uint8_t* __restrict__ dimg, const Level0Args& a, int tile, int nTiles) {
  for (int id = tile * 256 + (int)threadIdx.x; id < a.nItems; id += nTiles * 256) {
    if (id < a.nIntItems) {
      const int g = (int)__umulhi((uint32_t)id, a.magicInt), j = id - g * a.nInt + a.j0;
      uint4 v[P0R];
#pragma unroll
      for (int r = 0; r < P0R; ++r) {
        const int py = min(g * P0R + r, a.H - 1);
        __builtin_memcpy(&v[r], src + (uint32_t)(__umul24(reflect101(py - EDGE, a.h), stride) + 16 * j - EDGE), 16);
      }
#pragma unroll
      for (int r = 0; r < P0R; ++r)
        *reinterpret_cast<uint4*>(dimg + (uint32_t)(__umul24(min(g * P0R + r, a.H - 1), a.dpstride) + 16 * j)) = v[r];
    } else {
      const int e = id - a.nIntItems;
      const int g = (int)__umulhi((uint32_t)e, a.magicEdge);
      const PyrEdge pe = a.edge[e - g * a.nEdge];
      uint32_t w[P0R];
#pragma unroll
      for (int r = 0; r < P0R; ++r) {
        const int py = min(g * P0R + r, a.H - 1);
        __builtin_memcpy(&w[r], src + (uint32_t)(__umul24(reflect101(py - EDGE, a.h), stride) + pe.base), 4);
      }
#pragma unroll
      for (int r = 0; r < P0R; ++r)
        *reinterpret_cast<uint32_t*>(dimg + (uint32_t)(__umul24(min(g * P0R + r, a.H - 1), a.dpstride) + pe.dword * 4)) = __builtin_amdgcn_perm(0u, w[r], pe.sel);
    }
  }
}
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ src, int stride, size_t pitch, uint8_t* __restrict__ pyr, Level0Args a0) {
  const PyTile pt = py_tile();
  level0_items(src + (size_t)pt.img * pitch, stride, pyr + a0.dOff + (size_t)pt.img * a0.dImg, a0, pt.bx, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------
// K5: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), OpenCV fixed-point path: 8.8 kernel
// {18,34,48,56,48,34,18}, row pass exact, column pass rounded (+0.5) to u8.  The level's own 19-px
// reflect-101 pad is exactly the border the blur needs, so the tile loads straight from the padded pyramid.
// v2: no LDS.  A thread owns a 4-pixel-wide column strip and walks down BT_ROWS rows keeping the last seven
// row-pass results (4 x u16, packed in two dwords) in registers; every output dword (4 pixels) needs three
// aligned dword loads of the source row (bytes x-3 .. x+8; the interior starts 19 bytes into the padded row and
// 19 - 3 = 16, so x % 4 == 0 makes the window 4-byte aligned).  HBM traffic = read P (+halo rows) + write P.
#ifndef MORB_BT_ROWS
#define MORB_BT_ROWS 24   // rows per strip (round 4, B = 512, blur underneath the quadtree: 16 / 24 / 32 rows -> 131.9 / 132.9 / 132.0 k frames/s: fewer halo rows against emptier tiles)
#endif
constexpr int BT_W = 256, BT_ROWS = MORB_BT_ROWS, BT_TY = 8;   // workgroup: 32 lanes x 8 px wide, 8 half-waves x BT_ROWS rows tall -> 256 px x 192 rows per tile
constexpr int BT_H = BT_ROWS * BT_TY;
#ifndef MORB_BT_AHEAD
#define MORB_BT_AHEAD 4   // (1 / 2 / 3 / 4 / 6 rows ahead: 440 / 372 / 363 / 359 / 374 us per 512 images on one box)
#endif
constexpr int BT_AHEAD = MORB_BT_AHEAD;   // source rows in flight ahead of the row being filtered
typedef unsigned short blur_u16x2 __attribute__((ext_vector_type(2)));
#ifndef MORB_BLUR_SHIFTED_WEIGHTS
#define MORB_BLUR_SHIFTED_WEIGHTS 1
#endif
// Horizontal 7-tap of eight neighbouring pixels: v_dot4_u32_u8 against the packed kernel weights (18 34 48 56 | 48 34 18 0),
// the byte windows cut out of the four loaded dwords with v_alignbyte.  Results are exact integers <= 255 * 256.
__device__ __forceinline__ uint4 blur_load16(const uint8_t* __restrict__ row) {
  // row points at byte (x - 3) of the padded row (8-byte aligned): 16 bytes = pixels x-3 .. x+12, ONE vector-memory instruction
  uint4 w;
  __builtin_memcpy(&w, __builtin_assume_aligned(row, 8), 16);
  return w;
}
__device__ __forceinline__ void blur_h8(const uint4 w, uint32_t h[8]) {
#if MORB_BLUR_SHIFTED_WEIGHTS
  // Round 4: the seven taps of pixel s of a dword group sit on bytes s .. s + 6 of the loaded dwords; instead of cutting that window out with
  // v_alignbyte the WEIGHTS are shifted (constants): pixel 0 and 1 are two v_dot4_u32_u8, pixel 2 and 3 three — 20 instructions per eight
  // pixels instead of 16 + 9.
  constexpr uint32_t K0 = 18u, K1 = 34u, K2 = 48u, K3 = 56u;
  constexpr uint32_t A0 = K0 | (K1 << 8) | (K2 << 16) | (K3 << 24), B0 = K2 | (K1 << 8) | (K0 << 16);                   // s = 0: w0, w1
  constexpr uint32_t A1 = (K0 << 8) | (K1 << 16) | (K2 << 24), B1 = K3 | (K2 << 8) | (K1 << 16) | (K0 << 24);             // s = 1: w0, w1
  constexpr uint32_t A2 = (K0 << 16) | (K1 << 24), B2 = K2 | (K3 << 8) | (K2 << 16) | (K1 << 24), C2 = K0;               // s = 2: w0, w1, w2
  constexpr uint32_t A3 = (K0 << 24), B3 = K1 | (K2 << 8) | (K3 << 16) | (K2 << 24), C3 = K1 | (K0 << 8);                // s = 3: w0, w1, w2
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const uint32_t w0 = ws[g], w1 = ws[g + 1], w2 = ws[g + 2];
    h[4 * g + 0] = __builtin_amdgcn_udot4(w0, A0, __builtin_amdgcn_udot4(w1, B0, 0u, false), false);
    h[4 * g + 1] = __builtin_amdgcn_udot4(w0, A1, __builtin_amdgcn_udot4(w1, B1, 0u, false), false);
    h[4 * g + 2] = __builtin_amdgcn_udot4(w0, A2, __builtin_amdgcn_udot4(w1, B2, __builtin_amdgcn_udot4(w2, C2, 0u, false), false), false);
    h[4 * g + 3] = __builtin_amdgcn_udot4(w0, A3, __builtin_amdgcn_udot4(w1, B3, __builtin_amdgcn_udot4(w2, C3, 0u, false), false), false);
  }
#else
  constexpr uint32_t WA = 18u | (34u << 8) | (48u << 16) | (56u << 24), WB = 48u | (34u << 8) | (18u << 16);
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const uint32_t w0 = ws[g], w1 = ws[g + 1], w2 = ws[g + 2];
    h[4 * g + 0] = __builtin_amdgcn_udot4(w0, WA, __builtin_amdgcn_udot4(w1, WB, 0u, false), false);
    h[4 * g + 1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), WA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), WB, 0u, false), false);
    h[4 * g + 2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), WA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), WB, 0u, false), false);
    h[4 * g + 3] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), WA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), WB, 0u, false), false);
  }
#endif
}
__device__ __forceinline__ uint32_t blur_dot2(uint32_t pair, uint32_t w, uint32_t acc) {   // v_dot2_u32_u16
  return __builtin_amdgcn_udot2(__builtin_bit_cast(blur_u16x2, pair), __builtin_bit_cast(blur_u16x2, w), acc, false);
}
// Vertical pass: consecutive rows' horizontal sums are kept as 16-bit pairs (row r | row r+1 << 16), so the 7 taps are three
// v_dot2_u32_u16 and one multiply-add.  Round 2: 8 pixels per thread — one 16-byte load and one 8-byte store per row instead of
// 3 + 1 dword accesses per 4 pixels (the texture path is priced per instruction).
#ifndef MORB_BLUR_MIN_WAVES
#define MORB_BLUR_MIN_WAVES 1
#endif
__global__ __launch_bounds__(256, MORB_BLUR_MIN_WAVES) void k_blur(const LevelGeom* __restrict__ geom, int nlevels,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
  // Workgroup -> (image, tile).  Consecutive workgroup ids go round-robin to the 8 XCDs: in the plain (tile, image) order the 256-px-wide
  // neighbours of a tile row — whose 264-byte row segments start 16 bytes into a 128-byte line and so share a line with each neighbour —
  // and the tiles above / below (6 halo rows) sit on different L2s, and the stage fetched 1.9 x the pyramid.  Remapped (image counts that
  // are multiples of 8), XCD k works through images k, k + 8, ... tile by tile.
  int tileId = blockIdx.x, img = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const unsigned n = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = n & 7u, slot = n >> 3, grp = slot / gridDim.x;
    tileId = __builtin_amdgcn_readfirstlane((int)(slot - grp * gridDim.x)); img = __builtin_amdgcn_readfirstlane((int)(xcd + 8u * grp));
  }
  int l = 0;
  while (l + 1 < nlevels && tileId >= geom[l + 1].blurTileBase) ++l;
  const LevelGeom g = geom[l];
  const int t = tileId - g.blurTileBase;
  const int tx = t % g.blurTilesX, ty = t / g.blurTilesX;
  const int x = tx * BT_W + (threadIdx.x & 31) * 8;
  const int y0 = ty * BT_H + (threadIdx.x >> 5) * BT_ROWS;
  if (x >= g.w || y0 >= g.h) return;
  // padded-row origin of this strip: row (y + 19 - 3), byte (19 + x - 3) = 16 + x
  const uint8_t* src = pyr + g.pyrOff + (size_t)img * g.pyrImg + (size_t)(EDGE + y0 - 3) * g.pstride + (EDGE - 3) + x;
  uint8_t* dst = blur + g.blurOff + (size_t)img * g.blurImg + (size_t)y0 * g.bstride + x;
  uint32_t P[5][8], hl[8];   // P[j] = rows (y + j, y + j + 1) of the horizontal sums, hl = row y + 5
  {
    uint32_t h[6][8];
#pragma unroll
    for (int k = 0; k < 6; ++k) blur_h8(blur_load16(src + (size_t)k * g.pstride), h[k]);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) P[j][k] = h[j][k] | (h[j + 1][k] << 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) hl[k] = h[5][k];
  }
  const int rows = (g.h - y0) < BT_ROWS ? (g.h - y0) : BT_ROWS;
  constexpr uint32_t W01 = 18u | (34u << 16), W23 = 48u | (56u << 16), W45 = 48u | (34u << 16);
  // Rows past the bottom of the level (the last strip of a tile column) are computed like the others — from the last padded row, so no
  // load leaves the level — and only their store is skipped: with the whole row body under `if (r < rows)` (per-lane: the two halves of a
  // wave work different strips) the compiler merged the rotating row window through 688 v_mov per wave, a third of the kernel's VALU work.
  const int lastRow = g.h + 2 * EDGE - 1 - (EDGE + y0 - 3);   // last padded row, relative to src
  // The source rows are requested BT_AHEAD rows before they are used: with the load at the top of the row that needs it the wave sat
  // through a full memory round trip per row (load, s_waitcnt vmcnt(0), 75 VALU instructions, store — sixteen times), at five waves per SIMD.
  uint4 wq[BT_AHEAD];
#pragma unroll
  for (int d = 0; d < BT_AHEAD; ++d) wq[d] = blur_load16(src + (uint32_t)__umul24(min(6 + d, lastRow), g.pstride));
#pragma unroll
  for (int r = 0; r < BT_ROWS; ++r) {
    {   // (fully unrolled: the row window rotates by renaming)
    uint32_t hn[8], acc[8];
    const uint4 wcur = wq[r % BT_AHEAD];
    if (r + BT_AHEAD < BT_ROWS) wq[r % BT_AHEAD] = blur_load16(src + (uint32_t)__umul24(min(r + 6 + BT_AHEAD, lastRow), g.pstride));
    blur_h8(wcur, hn);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      acc[k] = blur_dot2(P[0][k], W01, blur_dot2(P[2][k], W23, blur_dot2(P[4][k], W45, __umul24(18u, hn[k]) + 32768u)));   // (hn < 2^16: v_mad_u32_u24, full rate)
    // acc < 2^24: the rounded output is byte 2 of each accumulator
    uint2 o;
    o.x = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
    o.y = __builtin_amdgcn_perm(acc[5], acc[4], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[7], acc[6], 0x06020c0cu);
    if (r < rows) *reinterpret_cast<uint2*>(dst + (size_t)r * g.bstride) = o;   // columns >= w land in the row's alignment slack
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      P[0][k] = P[1][k]; P[1][k] = P[2][k]; P[2][k] = P[3][k]; P[3][k] = P[4][k];
      P[4][k] = hl[k] | (hn[k] << 16);
      hl[k] = hn[k];
    }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K2: per-cell FAST-9/16 + cornerScore + 3x3 NMS with the iniThFAST -> minThFAST fallback
// (ORBextractor.cc:763-820 calling cv::FAST once or twice per 35-px cell): k_fastw, fast_wave.h.
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {   // v_pk_min_u16
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {   // v_pk_max_u16
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {   // v_pk_sub_u16 clamp
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {   // v_pk_add_u16
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}

// i / d == (i * ceil(2^20 / d)) >> 20 for i < 2^15, d < 40 (both factors fit v_mul_u32_u24)
struct Magic20 { unsigned m[40]; constexpr Magic20() : m() { for (int d = 1; d < 40; ++d) m[d] = 0xFFFFFu / (unsigned)d + 1u; } };
__constant__ Magic20 c_magic20 = Magic20();

// max over the 16 circular 9-arcs of the arc's minimum: 16 + 16 v_min3_i32 and 8 v_max3_i32.  The three-operand instructions are written out:
// left to itself the compiler shares two-operand minima between neighbouring arcs and ends up with 47 instructions instead of 40 — and every
// vector instruction of k_fastw is 4 cycles of a SIMD that has nothing else to wait for (profiles/r04/valu_rate_saturated.txt).
__device__ __forceinline__ int vmin3(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int vmax3(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int arc9_maxmin(const int (&d)[16]) {
  int mn3[16], mn9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) mn3[k] = vmin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; ++k) mn9[k] = vmin3(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
  // (a tree, not a chain: eight dependent instructions of one wave would each wait out the previous one's latency)
  const int a0 = vmax3(mn9[0], mn9[1], mn9[2]), a1 = vmax3(mn9[3], mn9[4], mn9[5]), a2 = vmax3(mn9[6], mn9[7], mn9[8]);
  const int a3 = vmax3(mn9[9], mn9[10], mn9[11]), a4 = vmax3(mn9[12], mn9[13], mn9[14]);
  return vmax3(vmax3(a0, a1, a2), vmax3(a3, a4, mn9[15]), -256);
}

#ifdef MORB_FAST_TIMING
extern "C" int morb_fast_timing(unsigned long long* out, int reset) {   // phase clocks of k_distribute (tools/fast_phases.py)
  if (reset) { unsigned long long z[32] = {0}; MORB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fastPhase), z, sizeof(z))); return 0; }
  MORB_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastPhase), 32 * sizeof(unsigned long long)));
  return 0;
}
#endif
#include "fast_wave.h"

// ---------------------------------------------------------------------------------------------------
#ifndef QT_GATHER
#define QT_GATHER 16   // candidate loads in flight per lane while the cells' lists are gathered (4: 457 us, 8: 444, 16: 438 per 512 images under the blur)
#endif
// K3: DistributeOctTree, one wave per (level, image).  See quadtree.h.
// A wave's LDS need (node arrays by the level's quota, key arrays by its area) falls by ~25 % per level, and a launch has ONE LDS size:
// sized for level 0, three one-wave workgroups fill a CU's LDS and every small level holds as much as level 0.  So the levels of an image
// are packed (host: first fit, decreasing) into a few workgroups of up to QT_MAX_WAVES waves whose needs add up to what level 0 takes
// (752 x 480 / 1200: {0} {1, 5} {2, 3} {4, 6, 7}); the waves of a workgroup never synchronise with each other.
__global__ __launch_bounds__(64 * QT_TEAM_WAVES) void k_distribute(const LevelGeom* __restrict__ geom, const uint32_t* __restrict__ cand,
                                                   const int* __restrict__ candCnt, int totalCells, int cellCap,
                                                   uint32_t* __restrict__ qtScratch, uint32_t* __restrict__ sel,
                                                   int* __restrict__ selCnt, int selPerImg, int nlevels, int groupBase, int* __restrict__ status) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int img = blockIdx.x, lane = threadIdx.x & 63;
  __shared__ int teamSh[QT_TEAM_SH];
  int lvl = -1;
  const int wvIdx = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const int grp = blockIdx.y + groupBase;
    // a workgroup holds up to QT_MAX_WAVES levels, a wave each — or ONE level worked by all its waves as a team (quadtree.h)
    for (int l = 0; l < nlevels; ++l) if (geom[l].distGroup == grp && (geom[l].distTeam || geom[l].distWave == wvIdx)) lvl = l;
  }
  if (lvl < 0) return;   // (wave-uniform: this workgroup packs fewer levels than the launch has waves)
  morbqt::Team tm;
  tm.nw = geom[lvl].distTeam ? (int)(blockDim.x >> 6) : 1; tm.tw = geom[lvl].distTeam ? wvIdx : 0; tm.sh = teamSh;
#ifndef QT_PRIO
#define QT_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(QT_PRIO);   // a long dependent chain: win instruction arbitration against the blur waves sharing the SIMD (measured 0 / 1 / 3: within 0.6 % of each other)
  const LevelGeom g = geom[lvl];
  const int nodeCap = g.nodeCap, keyCap = g.distKeyCap;
  const int ncell = g.nRows * g.nCols;
  // the level's LDS region (the host sizes it with the same arithmetic: dist_lds_bytes)
  uint8_t* sp = smem + g.distLdsOff;
  uint64_t* vA = reinterpret_cast<uint64_t*>(sp); sp += (size_t)nodeCap * 8;
  uint64_t* vB = reinterpret_cast<uint64_t*>(sp); sp += (size_t)nodeCap * 8;
  uint64_t* bcnt = vB;   // the batch split's child counts: vB is only live between the sort and the order[] it feeds
  morbqt::Node* nodes = reinterpret_cast<morbqt::Node*>(sp); sp += (size_t)nodeCap * sizeof(morbqt::Node);
  uint32_t* ldsKeys = reinterpret_cast<uint32_t*>(sp); sp += (size_t)keyCap * 4;
  uint32_t* ldsTmp = reinterpret_cast<uint32_t*>(sp); sp += (size_t)keyCap * 4;
  // the cell offsets are dead once the candidates are gathered and the split ranks are first written inside qt_distribute: one region
  // serves both when the offsets fit it (the host sizes the allocation the same way)
  const bool aliasCells = ncell + 1 <= nodeCap;
  int* cellOff = reinterpret_cast<int*>(sp); if (!aliasCells) sp += (size_t)(ncell + 1) * 4;
  uint32_t* brank = reinterpret_cast<uint32_t*>(sp); sp += (size_t)nodeCap * 4;
  uint16_t* freeIds = reinterpret_cast<uint16_t*>(sp); sp += (size_t)nodeCap * 2;
  uint16_t* order = reinterpret_cast<uint16_t*>(sp); sp += (size_t)nodeCap * 2;
  uint16_t* list = reinterpret_cast<uint16_t*>(sp);

#ifdef MORB_FAST_TIMING
  unsigned long long d0_ = wall_clock64();
#define DMARK(k) do { if (lvl == 0 && lane == 0 && tm.tw == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_fastPhase[k], now_ - d0_); d0_ = now_; } } while (0)
#else
#define DMARK(k)
#endif
  const int* counts = candCnt + (size_t)img * totalCells + g.cellBase;
  // first slot of every cell's candidate list: exclusive prefix of the cells' counts, a slice of the cells per wave of the team
  int T = 0;
  {
    const int per = ((ncell + tm.nw * 64 - 1) / (tm.nw * 64)) * 64;
    const int lo = per * tm.tw < ncell ? per * tm.tw : ncell, hi = lo + per < ncell ? lo + per : ncell;
    int running = 0;
    for (int c0 = lo; c0 < hi; c0 += 64) {
      const int c = c0 + lane;
      const int n = c < hi ? counts[c] : 0;
      int incl = n;
      MORB_DPP_SCAN(incl, 0, morbwave::op_add);   // inclusive prefix over the wave
      if (c < hi) cellOff[c] = running + incl - n;
      running += __builtin_amdgcn_readlane(incl, 63);
    }
    if (tm.nw > 1) {
      if (lane == 0) teamSh[16 + tm.tw] = running;
      __syncthreads();
      const int v = lane < tm.nw ? teamSh[16 + lane] : 0;
      T = morbwave::sum_i32(v);
      const int before = morbwave::sum_i32(lane < tm.tw ? v : 0);
      if (before) for (int c = lo + lane; c < hi; c += 64) cellOff[c] += before;
    } else {
      T = running;
    }
    if (tm.tw == 0 && lane == 0) cellOff[ncell] = T;
  }
  QT_TEAM_SYNC(tm);
  // (any number of candidates: the child counts of Work::bcnt saturate and the few nodes of more than 65535 keys are counted again, quadtree.h)
  DMARK(8);

  const uint32_t* cbase = cand + ((size_t)img * totalCells + g.cellBase) * (size_t)cellCap;
  uint32_t* out = sel + (size_t)img * selPerImg + g.selBase;
  int n = 0;
  // The key arrays are LDS (the usual case) or the global scratch.  The two cases are two inlined copies of the whole distribution: with ONE
  // copy and a run-time choice of the pointers every access to the keys was a FLAT instruction (PMC: 828 vector-memory instructions per wave,
  // nearly all of them LDS traffic taking the slow way), with two the compiler proves the address space of each.
  auto run = [&](uint32_t* keys, uint32_t* tmp) {
    // gather the cells' candidate lists into one array, cell-major: a lane per cell copies its list, QT_GATHER loads in flight per lane
    // (a lane per candidate had to binary-search its cell first: 8 dependent LDS reads in front of every global load, 34 of level 0's 152 us)
    int cFrom = tm.tw * 64;
    if (tm.nw > 1) {
      // a team (latency): two cells per lane and step, the first QT_GATHER candidates of both requested before any is stored — level 0 of a 1920 x 1080
      // image has 1705 cells for 1024 lanes, i.e. the whole gather is about one global-memory round trip instead of four
      for (; cFrom < ncell; cFrom += 2 * tm.nw * 64) {
        int cc[2], off[2], nc[2];
        uint32_t v[2][QT_GATHER];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          cc[h] = cFrom + h * tm.nw * 64 + lane;
          const bool in = cc[h] < ncell;
          off[h] = in ? cellOff[cc[h]] : 0;
          nc[h] = in ? cellOff[cc[h] + 1] - off[h] : 0;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // 16-byte loads (the host keeps cellCap a multiple of 4): QT_GATHER / 4 requests per cell
          const uint4* src4 = reinterpret_cast<const uint4*>(cbase + (size_t)(cc[h] < ncell ? cc[h] : 0) * cellCap);
#pragma unroll
          for (int k4 = 0; k4 < QT_GATHER / 4; ++k4) {
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (4 * k4 < nc[h]) q = src4[k4];
            v[h][4 * k4] = q.x; v[h][4 * k4 + 1] = q.y; v[h][4 * k4 + 2] = q.z; v[h][4 * k4 + 3] = q.w;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int k = 0; k < QT_GATHER; ++k) if (k < nc[h]) keys[off[h] + k] = v[h][k];
          if (nc[h] > QT_GATHER) {   // (rare: a cell with more candidates)
            const uint32_t* src = cbase + (size_t)cc[h] * cellCap;
            for (int k = QT_GATHER; k < nc[h]; ++k) keys[off[h] + k] = src[k];
          }
        }
      }
    }
    for (int c0 = cFrom; c0 < ncell; c0 += tm.nw * 64) {
      const int c = c0 + lane;
      const int off = c < ncell ? cellOff[c] : 0;
      const int nc = c < ncell ? cellOff[c + 1] - off : 0;
      const int nmax = (int)~morbwave::min_u32(~(uint32_t)nc);   // wave maximum
      const uint32_t* src = cbase + (size_t)(c < ncell ? c : 0) * cellCap;
      for (int i0 = 0; i0 < nmax; i0 += QT_GATHER) {
        uint32_t v[QT_GATHER];
#pragma unroll
        for (int k = 0; k < QT_GATHER; ++k) v[k] = i0 + k < nc ? src[i0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < QT_GATHER; ++k) if (i0 + k < nc) keys[off + i0 + k] = v[k];
      }
    }
    QT_TEAM_SYNC(tm);
    DMARK(9);
    morbqt::Work w;
    w.keys = keys; w.tmp = tmp; w.nodes = nodes; w.freeIds = freeIds; w.list = list; w.vA = vA; w.vB = vB;
    w.order = order; w.bcnt = bcnt; w.brank = brank;
    w.nodeCap = g.nodeCap; w.listCap = g.listCap;
    n = morbqt::qt_distribute(w, (uint32_t)T, g.maxBorderX - MINB, g.maxBorderY - MINB, g.quota, out, g.selCap, tm);
  };
  if (T <= keyCap) {
    run(ldsKeys, ldsTmp);
  } else {
    uint32_t* gk = qtScratch + g.qtOff + (size_t)img * g.qtImg;
    run(gk, gk + g.qtImg / 2);
  }
  if (lane == 0 && tm.tw == 0) selCnt[img * nlevels + lvl] = n < g.selCap ? n : g.selCap;
  DMARK(10);
#ifdef MORB_FAST_TIMING
  if (lvl == 0 && lane == 0 && tm.tw == 0) { atomicAdd(&g_fastPhase[11], (unsigned long long)T); atomicAdd(&g_fastPhase[12], (unsigned long long)n); }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Layout: final slot of every keypoint (operator() tail, ORBextractor.cc:1041-1085): levels in order, keypoints
// in list order; x in [lap0, lap1] (after pt *= scale) fills from the back, the rest from the front.
// LAYOUT_WAVES: the loop below is one global round trip and two barriers per 64 * waves keypoints.  16 waves for a handful of images (22 -> 9.6 us for one
// 4000-feature image), 4 in a batch (1024 workgroups of 16 waves: 13 -> 162 us per launch — most of their lanes have no keypoint).
template <int LAYOUT_WAVES>
__global__ __launch_bounds__(64 * LAYOUT_WAVES) void k_layout(const LevelGeom* __restrict__ geom, int nlevels,
                                                const uint32_t* __restrict__ sel, const int* __restrict__ selCnt,
                                                int selPerImg, const int* __restrict__ lap, int2* __restrict__ kref,
                                                int* __restrict__ nkp, int* __restrict__ mono, int cap) {
  constexpr int NT = 64 * LAYOUT_WAVES;
  __shared__ int wm[LAYOUT_WAVES], ws[LAYOUT_WAVES];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int base[kMaxLevels + 1];
  int total = 0;
  for (int l = 0; l < nlevels; ++l) { base[l] = total; total += selCnt[img * nlevels + l]; }
  base[nlevels] = total;
  const float lap0 = (float)lap[img * 2], lap1 = (float)lap[img * 2 + 1];
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int monoRun = 0, stereoRun = 0;
  for (int g0 = 0; g0 < total; g0 += NT) {
    const int gi = g0 + tid;
    const bool valid = gi < total;
    bool isLap = false;
    int l = 0;
    uint32_t key = 0;
    if (valid) {
      while (gi >= base[l + 1]) ++l;
      key = sel[(size_t)img * selPerImg + geom[l].selBase + (gi - base[l])];
      float x = (float)(morbqt::key_x(key) + MINB);
      if (l != 0) x = x * geom[l].scale;
      isLap = (x >= lap0) && (x <= lap1);
    }
    const uint64_t mS = __ballot(valid && isLap), mM = __ballot(valid && !isLap);
    if (lane == 0) { ws[wv] = __popcll(mS); wm[wv] = __popcll(mM); }
    __syncthreads();
    int offS = stereoRun, offM = monoRun, totS = 0, totM = 0;
#pragma unroll
    for (int q = 0; q < LAYOUT_WAVES; ++q) { const int a = ws[q], b = wm[q]; if (q < wv) { offS += a; offM += b; } totS += a; totM += b; }
    if (valid) {
      const int slot = isLap ? (total - 1 - (offS + __popcll(mS & lt))) : (offM + __popcll(mM & lt));
      // everything k_describe needs about keypoint gi in ONE record: output slot | level << 24 (or -1), packed key
      kref[(size_t)img * selPerImg + gi] = make_int2(slot < cap ? (slot | (l << 24)) : -1, (int)key);
    }
    stereoRun += totS;
    monoRun += totM;
    __syncthreads();
  }
  for (int gi = total + tid; gi < selPerImg; gi += NT) kref[(size_t)img * selPerImg + gi] = make_int2(-1, 0);
  if (tid == 0) { nkp[img] = total < cap ? total : cap; mono[img] = monoRun; }
}

// ---------------------------------------------------------------------------------------------------
// Describe: IC_Angle (:75-99) + computeOrbDescriptor (:102-145) + keypoint record, one wave per keypoint.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2, f32 throughout
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// cos / sin of the keypoint angle: glibc 2.35's sincosf restated (libm_f32.h), bit-equal to the CPU libm the reference runs on
using morbm::sincosf_glibc;

// The wave's lifetime is a chain of dependent global-memory round trips, so the chain is kept short: one record
// per keypoint from k_layout (slot, level, key), level geometry from the kernarg segment, and the (keypoint-
// independent) rBRIEF pattern rows of the lane requested before anything else.
#ifndef MORB_DESC_STAGED
#define MORB_DESC_STAGED 1
#endif
#ifndef MORB_DESC_KPW
#define MORB_DESC_KPW 2   // (round 3, with the window in LDS: 2 / 3 / 4 / 8 keypoints per wave = 469 / 473 / 486 / 628 us per 512 images; 1: 519)
#endif
constexpr int DESC_KPW = MORB_DESC_KPW;   // keypoints per wave in k_describe
constexpr int DESC_R = 18;    // the rotated rBRIEF pattern stays within +-18 px of the keypoint (|(+-13, +-13)| = 18.4, then cvRound)
constexpr int DESC_WIN = 2 * DESC_R + 1, DESC_WP = 48;   // window rows / LDS pitch (three 16-byte segments)
// One wave describes DESC_KPW keypoints *in lock step*: the kernel is bound by dependent global-memory round trips per
// keypoint (record -> patch -> angle -> pattern gathers), so each stage is issued for all DESC_KPW keypoints before the
// next stage waits on it — three round trips per wave instead of three per keypoint.  Slots without a keypoint repeat
// the wave's first valid one (no divergent control flow around the loads) and skip the stores.
#ifndef MORB_DESC_WAVES
#define MORB_DESC_WAVES 4
#endif
#ifndef MORB_DESC_ANGLE_WG
#define MORB_DESC_ANGLE_WG 1   // fastAtan2 + sincosf of the workgroup's keypoints in one pass of wave 0 (0: each wave for its own two)
#endif
constexpr int DESC_WAVES = MORB_DESC_WAVES;   // waves (of DESC_KPW keypoints each) per workgroup
__global__ __launch_bounds__(64 * DESC_WAVES) void k_describe(const morb::DescGeom dg, const uint8_t* __restrict__ pyr,
                                                  const uint8_t* __restrict__ blur, const int2* __restrict__ kref,
                                                  int selPerImg, morb_keypoint* __restrict__ kps,
                                                  uint8_t* __restrict__ desc, int cap, int imgRev) {
#if MORB_DESC_STAGED
  __shared__ __align__(16) uint8_t s_win[DESC_WAVES * DESC_KPW * DESC_WIN * DESC_WP];
#endif
  // Workgroup -> (image, keypoint chunk).  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2: with the
  // plain (chunk, image) order the chunks of one image — whose patches cover the image's whole pyramid and blur between them — are spread
  // over all eight L2s and every XCD fetches (nearly) every line (PMC r02: 2.2 x the gather bytes fetched, 4.4 x with the FETCH_SIZE
  // correction).  Remapped (image counts that are multiples of 8), XCD k works through images k, k + 8, ... chunk by chunk.
  int imgIdx = blockIdx.y, chunk = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const unsigned n = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = n & 7u, slot = n >> 3, grp = slot / gridDim.x;
    chunk = (int)(slot - grp * gridDim.x); imgIdx = (int)(xcd + 8u * grp);
    chunk = __builtin_amdgcn_readfirstlane(chunk); imgIdx = __builtin_amdgcn_readfirstlane(imgIdx);
  }
  const int img = imgRev ? gridDim.y - 1 - imgIdx : imgIdx, lane = threadIdx.x & 63;
  const int gi0 = (chunk * DESC_WAVES + (threadIdx.x >> 6)) * DESC_KPW;
#if MORB_DESC_ANGLE_WG
  // (round 5) the workgroup's waves meet at two barriers (the angles of all its keypoints are computed in ONE pass of wave 0): a wave without
  // work goes through them and leaves
  __shared__ int s_mom[DESC_WAVES * DESC_KPW][2];
  __shared__ float s_trig[DESC_WAVES * DESC_KPW][3];
  // The trig pass belongs to threads 0 .. DESC_WAVES * DESC_KPW - 1, i.e. to wave 0 — which therefore runs it even when it has no keypoint of its own
  // (today k_layout writes the valid records first, so wave 0 is never the idle one; the kernel no longer depends on that).
  auto trig_pass = [&]() {
    if (threadIdx.x < DESC_WAVES * DESC_KPW) {
      const float ang = fast_atan2_deg((float)s_mom[threadIdx.x][1], (float)s_mom[threadIdx.x][0]);
      const float factorPI = (float)(3.14159265358979323846 / 180.f);
      float sn, cs;
      sincosf_glibc(ang * factorPI, &sn, &cs);
      s_trig[threadIdx.x][0] = ang; s_trig[threadIdx.x][1] = cs; s_trig[threadIdx.x][2] = sn;
    }
  };
  auto leave = [&]() {   // a wave without work: through the two barriers (its own s_mom slots are never read by anybody), wave 0 still doing the pass
    if (lane < DESC_KPW) { s_mom[(threadIdx.x >> 6) * DESC_KPW + lane][0] = 0; s_mom[(threadIdx.x >> 6) * DESC_KPW + lane][1] = 0; }
    __syncthreads();
    trig_pass();
    __syncthreads();
  };
  if (gi0 >= selPerImg) { leave(); return; }
#else
  if (gi0 >= selPerImg) return;
#endif
  // (the pattern fetched once per workgroup into LDS instead — one 16-byte load per thread, a barrier, four ds_read_b128 per lane later —
  // measured no faster: 2016 - 2058 against 2002 - 2015 us per 512 images for the whole extraction, three runs each on one box)
  float4 pat[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pat[q] = reinterpret_cast<const float4*>(c_pattern)[lane * 4 + q];
  int2 ref[DESC_KPW];
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) ref[kk] = kref[(size_t)img * selPerImg + min(gi0 + kk, selPerImg - 1)];
  bool ok[DESC_KPW];
  bool any = false;
  int2 firstRef = ref[DESC_KPW - 1];   // the first valid record, by selects: `ref[first]` with a run-time index put the array into LDS / scratch
#pragma unroll
  for (int kk = DESC_KPW - 1; kk >= 0; --kk) {
    ok[kk] = gi0 + kk < selPerImg && ref[kk].x >= 0;
    if (ok[kk]) { firstRef = ref[kk]; any = true; }
  }
#if MORB_DESC_ANGLE_WG
  if (!any) { leave(); return; }   // wave-uniform
#else
  if (!any) return;   // wave-uniform
#endif
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) {
    if (!ok[kk]) ref[kk] = firstRef;
    // wave-uniform by construction: in SGPRs the per-keypoint geometry and base addresses are scalar arithmetic
    ref[kk].x = __builtin_amdgcn_readfirstlane(ref[kk].x);
    ref[kk].y = __builtin_amdgcn_readfirstlane(ref[kk].y);
  }

  int lvl[DESC_KPW], cx[DESC_KPW], cy[DESC_KPW], pstride[DESC_KPW], bstride[DESC_KPW];
  const uint8_t* ctr[DESC_KPW];
  const uint8_t* center[DESC_KPW];
  int wsh[DESC_KPW] = {};
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) {
    const int l = ref[kk].x >> 24;
    const uint32_t key = (uint32_t)ref[kk].y;
    lvl[kk] = l; pstride[kk] = dg.pstride[l]; bstride[kk] = dg.bstride[l];
    cx[kk] = morbqt::key_x(key) + MINB; cy[kk] = morbqt::key_y(key) + MINB;
    // top-left corners of the 31 x 31 moment patch and of the (2 * EDGE + 1)^2 window the rotated pattern stays inside:
    // the lanes add unsigned 32-bit offsets (scalar base + vector offset addressing)
    ctr[kk] = pyr + dg.pyrOff[l] + (size_t)img * dg.pyrImg[l] + (size_t)(EDGE + cy[kk] - HALF_PATCH) * pstride[kk] + EDGE + cx[kk] - HALF_PATCH;
    center[kk] = blur + dg.blurOff[l] + (size_t)img * dg.blurImg[l] + (ptrdiff_t)(cy[kk] - DESC_R) * bstride[kk] + cx[kk] - DESC_R;
#if MORB_DESC_STAGED
    // the window is fetched from the dword boundary below its left edge (blurred rows are 64-byte aligned, a keypoint sits at x >= 19):
    // a wave's 16-byte load costs the memory pipe 64 cycles at a byte-granular address and 16 at a dword-aligned one
    // (tools/micro/ta_rate.hip); the 37 columns + the shift still fit the 48-byte rows, the gathers add the shift
    wsh[kk] = (cx[kk] - DESC_R) & 3;
    center[kk] -= wsh[kk];
#endif
  }
  // IC_Angle on the un-blurred level.  The texture path handles a byte load of a wave no faster than a dword load, so the
  // 31 x 31 patch is read as 31 rows x 8 unaligned dwords = 248 dword loads, four per lane.
  uint32_t wv[DESC_KPW][4];
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = lane + 64 * j;              // row = t / 8, dword = t % 8
      const int tt = t < 248 ? t : 247;
      wv[kk][j] = load_u32_unaligned(ctr[kk] + (uint32_t)(__umul24(tt >> 3, pstride[kk]) + (tt & 7) * 4));
    }
  // The blurred window the rotated pattern can reach — 37 x 37 around the keypoint, |coordinate| <= 18 = round(13 * sqrt 2) — does not
  // depend on the angle: it is requested in the same round trip as the patch, 37 rows x three 16-byte loads (the 11 bytes past the window
  // stay inside the row's slack or the next row), and goes to LDS, where the 512 byte gathers of the tests cost a fraction of what they
  // cost the vector cache: PMC showed the kernel at 1.13 cache-line accesses per cycle per CU (TCP_TOTAL_CACHE_ACCESSES: 692 per keypoint,
  // 512 of them the byte gathers), i.e. bound by the L1's one tag lookup per clock; the window's row loads are ~60 accesses.
#if MORB_DESC_STAGED
  uint4 ww[DESC_KPW][2];
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = min(lane + 64 * j, DESC_WIN * 3 - 1);   // row = t / 3, 16-byte segment = t % 3
      const int row = (int)(((unsigned)t * 21846u) >> 16), seg = t - row * 3;
      __builtin_memcpy(&ww[kk][j], __builtin_assume_aligned(center[kk] + (uint32_t)(__umul24(row, bstride[kk]) + seg * 16), 4), 16);
    }
#endif
  // The circular mask and the column weights of a lane's four dwords do not depend on the keypoint: byte masks and the
  // biased weights (u + 15, so that v_dot4_u32_u8 applies) are built once, and per keypoint a dword costs one AND, two
  // dot products and a multiply-add: m10 = sum (u + 15) I - 15 sum I, m01 = sum v (row sum of I).
  // (round 6: the same constants from a compile-time table, two 16-byte loads per lane instead of ~75 vector instructions per wave: 478 -> 501 us per 512 images —
  // the kernel answers to its vector-memory instructions before it answers to its VALU count; the masks as byte RANGES, ~40 instructions fewer: no difference outside the noise)
  uint32_t pmask[4], pw[4];
  int pv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = lane + 64 * j;
    const int v = (t >> 3) - HALF_PATCH, u0 = (t & 7) * 4 - HALF_PATCH;
    const int av = v < 0 ? -v : v;
    const int d = (int)((0x3689ABCDDEEEFFFFull >> (4 * av)) & 15);   // umax[|v|] (== c_umax, checked on the host)
    uint32_t m = 0, w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = u0 + k;
      if (t < 248 && u >= -d && u <= d) m |= 0xFFu << (8 * k);
      w |= (uint32_t)(u + HALF_PATCH) << (8 * k);
    }
    pmask[j] = m; pw[j] = w; pv[j] = v;
  }
  float angle[DESC_KPW];
  int t0v[DESC_KPW][4], t1v[DESC_KPW][4];
  int m10k[DESC_KPW], m01k[DESC_KPW];
#if MORB_DESC_STAGED
  uint8_t* win = s_win + (threadIdx.x >> 6) * (DESC_KPW * DESC_WIN * DESC_WP);   // this wave's windows: [DESC_KPW][37 rows][48 bytes]
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = lane + 64 * j;
      if (t < DESC_WIN * 3) *reinterpret_cast<uint4*>(win + kk * (DESC_WIN * DESC_WP) + t * 16) = ww[kk][j];   // row * 48 + seg * 16 == t * 16
    }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) {
    uint32_t usumB = 0, sumAll = 0;
    int m01 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t px = wv[kk][j] & pmask[j];
      const uint32_t sum = __builtin_amdgcn_udot4(px, 0x01010101u, 0u, false);
      usumB = __builtin_amdgcn_udot4(px, pw[j], usumB, false);
      sumAll += sum;
      m01 += __mul24(pv[j], (int)sum);
    }
    int m10 = (int)usumB - __mul24(HALF_PATCH, (int)sumAll);
    m10k[kk] = morbwave::sum_i32(m10);   // DPP reductions (wave.h): all 64 lanes are active here
    m01k[kk] = morbwave::sum_i32(m01);
  }
  // fastAtan2 and sincosf are one number per keypoint: lane kk computes keypoint kk's (as wave-uniform code the four of them cost four
  // serial passes of ~70 instructions, the FP64 polynomial of glibc's sincosf included, on all 64 lanes), and every keypoint's gathers
  // can be issued as soon as that one pass is through
#if MORB_DESC_ANGLE_WG
  // Round 5: ONE pass for the whole workgroup.  fastAtan2 + glibc's sincosf (~90 instructions, FP64 polynomial included) occupy all 64 lanes of a
  // wave for the two numbers its two keypoints need; the kernel is bound by vector-instruction issue (634 VALU per wave, 5.3 SIMD cycles each,
  // profiles/r05/pmc_issue_table_b512.txt), so the four waves' passes become one: moments to LDS, wave 0 computes lane k = keypoint k of the
  // workgroup, everybody picks its two (cos, sin, angle) up again.  The same instructions on the same inputs: identical bits.
  {
    const int w0 = (threadIdx.x >> 6) * DESC_KPW;
    if (lane < DESC_KPW) {
      int m10 = m10k[DESC_KPW - 1], m01 = m01k[DESC_KPW - 1];
#pragma unroll
      for (int kk = DESC_KPW - 2; kk >= 0; --kk) { m10 = lane == kk ? m10k[kk] : m10; m01 = lane == kk ? m01k[kk] : m01; }
      s_mom[w0 + lane][0] = m10; s_mom[w0 + lane][1] = m01;
    }
    __syncthreads();
    trig_pass();
    __syncthreads();
  }
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) {
    const int ws = (threadIdx.x >> 6) * DESC_KPW + kk;
    angle[kk] = s_trig[ws][0];
    const float a = s_trig[ws][1];
    const float bsin = s_trig[ws][2];
#else
  float angL, aL, bL;
  {
    int m10 = m10k[DESC_KPW - 1], m01 = m01k[DESC_KPW - 1];
#pragma unroll
    for (int kk = DESC_KPW - 2; kk >= 0; --kk) { m10 = lane == kk ? m10k[kk] : m10; m01 = lane == kk ? m01k[kk] : m01; }
    angL = fast_atan2_deg((float)m01, (float)m10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    sincosf_glibc(angL * factorPI, &bL, &aL);
  }
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) {
    angle[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, angL), kk));
    // rBRIEF on the blurred level: lane k evaluates tests 4k..4k+3
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, aL), kk));
    const float bsin = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bL), kk));
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = pat[q].x, y0 = pat[q].y, x1 = pat[q].z, y1 = pat[q].w;
      const int r0 = __float2int_rn(x0 * bsin + y0 * a), c0 = __float2int_rn(x0 * a - y0 * bsin);
      const int r1 = __float2int_rn(x1 * bsin + y1 * a), c1 = __float2int_rn(x1 * a - y1 * bsin);
#if MORB_DESC_STAGED
      // (24-bit multiplies: the plain `* 48` compiled to the quarter-rate v_mad_u64_u32, sixteen of them per pair of keypoints)
      t0v[kk][q] = win[kk * (DESC_WIN * DESC_WP) + __mul24(r0 + DESC_R, DESC_WP) + c0 + DESC_R + wsh[kk]];
      t1v[kk][q] = win[kk * (DESC_WIN * DESC_WP) + __mul24(r1 + DESC_R, DESC_WP) + c1 + DESC_R + wsh[kk]];
#else
      t0v[kk][q] = center[kk][(uint32_t)((r0 + DESC_R) * bstride[kk] + c0 + DESC_R)];   // (unstaged build: center is the window's own corner)
      t1v[kk][q] = center[kk][(uint32_t)((r1 + DESC_R) * bstride[kk] + c1 + DESC_R)];
#endif
    }
  }
#pragma unroll
  for (int kk = 0; kk < DESC_KPW; ++kk) {
    uint32_t nib = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) nib |= (uint32_t)(t0v[kk][q] < t1v[kk][q]) << q;
    const uint32_t hi = __shfl_down(nib, 1, 64);
    if (!ok[kk]) continue;   // wave-uniform
    const int slot = ref[kk].x & 0xFFFFFF, l = lvl[kk];
    if ((lane & 1) == 0) desc[((size_t)img * cap + slot) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
    if (lane < 7) {   // the 28-byte cv::KeyPoint record, one dword per lane: a single vector store
      float x = (float)cx[kk], y = (float)cy[kk];
      if (l != 0) { x = x * dg.scale[l]; y = y * dg.scale[l]; }
      uint32_t w;
      switch (lane) {
        case 0: w = __float_as_uint(x); break;
        case 1: w = __float_as_uint(y); break;
        case 2: w = __float_as_uint(dg.kpSize[l]); break;
        case 3: w = __float_as_uint(angle[kk]); break;
        case 4: w = __float_as_uint((float)morbqt::key_r((uint32_t)ref[kk].y)); break;
        case 5: w = (uint32_t)l; break;
        default: w = 0xFFFFFFFFu; break;   // class_id = -1
      }
      reinterpret_cast<uint32_t*>(kps + (size_t)img * cap + slot)[lane] = w;
    }
  }
}

}  // namespace

// =====================================================================================================
// Host side
// =====================================================================================================
namespace {

static int cvRoundF(float v) { return (int)lrintf(v); }

void free_buffers(morb_extractor* e) {
  auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  F(e->d_geom); F(e->d_geomTeam); F(e->d_tabs); F(e->d_pcol); F(e->d_prow); F(e->d_pedge); F(e->d_segTab); F(e->d_pyr); F(e->d_blur); F(e->d_cand); F(e->d_qt); F(e->d_sel);
  F(e->d_candCnt); F(e->d_selCnt); F(e->d_kref); F(e->d_lap);
  e->W = e->H = e->nimgCap = 0;
  e->lapLast.clear();
}
void free_staging(morb_extractor* e) {
  auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  F(e->d_img); F(e->d_out1); e->d_kps1 = nullptr; e->d_desc1 = nullptr; e->d_cnt1 = nullptr; e->d_mono1 = nullptr;
  e->imgBytes = 0;
  if (e->h_io1) { (void)hipHostFree(e->h_io1); e->h_io1 = nullptr; e->ioBytes1 = 0; }
}

// Build geometry + tables for (W, H) and allocate for nimg images.
int configure(morb_extractor* e, int W, int H, int nimg) {
  if (e->W == W && e->H == H && e->nimgCap >= nimg) return MORB_OK;
  MORB_HIP_CHECK(hipSetDevice(e->device));
  MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
  free_buffers(e);
  const int L = e->nlevels;
  std::vector<ResizeTab> tabs;
  std::vector<PyrCol> pcols; std::vector<PyrRow> prows; std::vector<PyrEdge> pedges;
  bool pyrPacked = true;
  size_t pyrOff = 0, blurOff = 0, qtOff = 0;
  int cellBase = 0, selBase = 0, blurTileBase = 0;
  e->cellCap = 0; e->maxCells = 0; e->maxNodeCap = 0; e->maxListCap = 0;
  std::vector<FastSeg> segs;
  LevelGeom teamGeom[kMaxLevels];
  memset(teamGeom, 0, sizeof teamGeom);
  for (int l = 0; l < L; ++l) {
    LevelGeom& g = e->geom[l];
    memset(&g, 0, sizeof g);
    const float s = e->invScale[l];
    g.w = cvRoundF((float)W * s);  // ORBextractor.cc:1091-1092
    g.h = cvRoundF((float)H * s);
    MORB_REQUIRE(g.w >= 2 * EDGE + 35 + 3 && g.h >= 2 * EDGE + 35 + 3 && g.w <= 4095 && g.h <= 4095, MORB_ERR_UNSUPPORTED,
                 "image size unsupported: every pyramid level must be between 76 and 4095 px in each dimension");
    g.pstride = (int)align_up((size_t)g.w + 2 * EDGE, 64);
    g.bstride = (int)align_up((size_t)g.w, 64);
    g.pyrOff = pyrOff; g.pyrImg = (size_t)(g.h + 2 * EDGE) * g.pstride; pyrOff += g.pyrImg * nimg;
    g.blurOff = blurOff; g.blurImg = (size_t)g.h * g.bstride; blurOff += g.blurImg * nimg;
    g.maxBorderX = g.w - EDGE + 3; g.maxBorderY = g.h - EDGE + 3;  // :748-749
    const float width = (float)(g.maxBorderX - MINB), height = (float)(g.maxBorderY - MINB);
    g.nCols = (int)(width / 35.f); g.nRows = (int)(height / 35.f);                    // :755-758
    g.wCell = (int)std::ceil(width / g.nCols); g.hCell = (int)std::ceil(height / g.nRows);  // :760-761
    g.cellBase = cellBase; cellBase += g.nCols * g.nRows;
    e->maxCells = std::max(e->maxCells, g.nCols * g.nRows);
    e->cellCap = std::max(e->cellCap, ((g.wCell + 1) / 2) * ((g.hCell + 1) / 2));
    // k_fastw works cell by cell: one descriptor per cell (a "segment" of one cell)
    MORB_REQUIRE(g.wCell + 7 <= 80 && g.hCell + 6 < 128, MORB_ERR_UNSUPPORTED, "FAST cell too large for the LDS window");
    for (int ci = 0; ci < g.nRows; ++ci)
      for (int c0 = 0; c0 < g.nCols; ++c0) {
        const int X0 = MINB + c0 * g.wCell, iniY = MINB + ci * g.hCell;
        int tw = std::min(X0 + g.wCell + 6, g.maxBorderX) - X0, th = std::min(iniY + g.hCell + 6, g.maxBorderY) - iniY;
        if (iniY >= g.maxBorderY - 3 || tw <= 6 || th <= 6) tw = th = 0;   // ORBextractor.cc:770, :775
        FastSeg sd;
        sd.winOff = (unsigned)((size_t)(EDGE + iniY) * g.pstride + EDGE + X0);
        sd.cell0 = g.cellBase + ci * g.nCols + c0;
        sd.geo = l | (1 << 8) | (tw << 16) | (int)((unsigned)th << 24);
        sd.key0 = (c0 * g.wCell) | ((ci * g.hCell) << 16);
        segs.push_back(sd);
      }
    g.quota = e->quota[l];
    g.nIni = (int)std::round(width / height);  // :545
    MORB_REQUIRE(g.nIni >= 1 && g.nIni <= 4, MORB_ERR_UNSUPPORTED, "aspect ratio unsupported (need 0.5 <= w/h < 4.5)");
    g.nodeCap = morbqt::qt_node_cap(g.quota, g.nIni);
    g.listCap = morbqt::qt_list_cap(g.nodeCap);
    MORB_REQUIRE(g.listCap < 65535, MORB_ERR_UNSUPPORTED, "nfeatures too large for 16-bit node lists");
    e->maxNodeCap = std::max(e->maxNodeCap, g.nodeCap);
    e->maxListCap = std::max(e->maxListCap, g.listCap);
    g.selCap = std::max(g.quota + 3, 4 * g.nIni) + 1;
    g.selBase = selBase; selBase += g.selCap;
    g.scale = e->scale[l];
    g.kpSize = (float)(int)(PATCH * e->scale[l]);
    g.blurTilesX = div_up(g.w, BT_W); g.blurTilesY = div_up(g.h, BT_H);
    g.blurTileBase = blurTileBase; blurTileBase += g.blurTilesX * g.blurTilesY;
  }
  e->cellCap = (std::max(e->cellCap, QT_GATHER) + 3) / 4 * 4;   // (k_distribute's team gather reads a cell's list as 16-byte words)
  // global key scratch: 2 x (cells x cellCap) per (level, image), used only when a level has > kLdsKeys candidates
  for (int l = 0; l < L; ++l) {
    LevelGeom& g = e->geom[l];
    g.qtImg = 2ull * g.nCols * g.nRows * e->cellCap;
    g.qtOff = qtOff; qtOff += g.qtImg * nimg;
  }
  // resize tables in padded destination coordinates (imgproc/resize.cpp coefficient set-up, see oracle)
  for (int l = 1; l < L; ++l) {
    LevelGeom& g = e->geom[l];
    const LevelGeom& gs = e->geom[l - 1];
    const int ONE = 2048;
    auto build = [&](int dn, int sn, bool horizontal, std::vector<ResizeTab>& out) {
      const double inv = (double)dn / sn, sc = 1. / inv;
      std::vector<ResizeTab> interior(dn);
      for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * sc - 0.5);
        int si = (int)std::floor(f);
        f -= si;
        if (horizontal) {
          if (si < 0) { f = 0; si = 0; }
          if (si >= sn - 1) { f = 0; si = sn - 1; }
        }
        ResizeTab t;
        int s0 = si, s1 = si + 1;
        s0 = s0 < 0 ? 0 : (s0 < sn ? s0 : sn - 1);
        s1 = s1 < 0 ? 0 : (s1 < sn ? s1 : sn - 1);
        t.s0 = (short)s0; t.s1 = (short)s1;
        t.c0 = (short)std::min(std::max(cvRoundF((1.f - f) * ONE), -32768), 32767);
        t.c1 = (short)std::min(std::max(cvRoundF(f * ONE), -32768), 32767);
        interior[d] = t;
      }
      for (int p = 0; p < dn + 2 * EDGE + PY_ROWS; ++p) {   // (PY_ROWS repeats of the last entry: k_resize reads whole groups unclamped)
        int q = std::min(p, dn + 2 * EDGE - 1) - EDGE;
        if (q < 0) q = -q;
        if (q >= dn) q = 2 * (dn - 1) - q;
        out.push_back(interior[q]);
      }
    };
    g.xtabOff = (int)tabs.size(); build(g.w, gs.w, true, tabs);
    g.ytabOff = (int)tabs.size(); build(g.h, gs.h, false, tabs);
    // k_resize's item tables: one PyrCol per destination dword, one PyrRow per destination row (padded to whole groups of PR)
    PyrLevel& pl = e->pyrLv[l];
    pl.nC = div_up(g.w + 2 * EDGE, 4); pl.nG = div_up(g.h + 2 * EDGE, PR); pl.nItems = pl.nC * pl.nG;
    pl.magicC = (uint32_t)((1ull << 32) / (unsigned)pl.nC + 1);
    MORB_REQUIRE((unsigned long long)pl.nItems * (unsigned)pl.nC < (1ull << 32), MORB_ERR_UNSUPPORTED, "pyramid level too large for the item index arithmetic");
    pl.ldsRows = (255 / pl.nC + 2) * PR;
    pl.colOff = (int)pcols.size(); pl.rowOff = (int)prows.size();
    const ResizeTab* xt = tabs.data() + g.xtabOff; const ResizeTab* yt = tabs.data() + g.ytabOff;
    for (int c = 0; c < pl.nC; ++c) {
      ResizeTab t4[4];
      for (int k = 0; k < 4; ++k) t4[k] = xt[std::min(4 * c + k, g.w + 2 * EDGE - 1)];
      int base = t4[0].s0, top = t4[0].s0;
      for (int k = 0; k < 4; ++k) { base = std::min(base, (int)std::min(t4[k].s0, t4[k].s1)); top = std::max(top, (int)std::max(t4[k].s0, t4[k].s1)); }
      if (top - base >= 8) pyrPacked = false;
      // the window as three aligned dwords of the source's padded row (the interior starts EDGE bytes into it; rows are 64-byte aligned)
      PyrCol pc; pc.base = (EDGE + base) & ~3; pc.sh = (EDGE + base) & 3; pc.pad_[0] = pc.pad_[1] = 0;
      for (int k = 0; k < 4; ++k) {
        pc.sel[k] = (uint32_t)((t4[k].s0 - base) & 7) | 0x0C00u | (uint32_t)((t4[k].s1 - base) & 7) << 16 | 0x0C000000u;
        pc.coef[k] = (uint32_t)(uint16_t)t4[k].c0 | (uint32_t)(uint16_t)t4[k].c1 << 16;
        if (t4[k].c0 < 0 || t4[k].c1 < 0) pyrPacked = false;
      }
      pcols.push_back(pc);
    }
    for (int r = 0; r < pl.nG * PR; ++r) {
      const ResizeTab t = yt[std::min(r, g.h + 2 * EDGE - 1)];
      PyrRow pr; pr.s0 = t.s0; pr.s1 = t.s1; pr.c0s = (uint32_t)t.c0 << 12; pr.c1s = (uint32_t)t.c1 << 12;
      if (t.c0 < 0 || t.c1 < 0) pyrPacked = false;
      prows.push_back(pr);
    }
  }
  {
    // level 0 = the image plus its reflected pad: interior 16-byte chunks and edge dwords (k_level0)
    const LevelGeom& g0 = e->geom[0];
    PyrLevel0& p0 = e->pyrL0;
    const int nC0 = div_up(g0.w + 2 * EDGE, 4);               // destination dwords per row
    p0.j0 = div_up(EDGE, 16);                                  // first 16-byte chunk that lies wholly inside the image
    p0.nInt = std::max(0, (EDGE + g0.w) / 16 - p0.j0);
    p0.nG = div_up(g0.h + 2 * EDGE, P0R);
    for (int d = 0; d < nC0; ++d) {
      if (d >= 4 * p0.j0 && d < 4 * (p0.j0 + p0.nInt)) continue;
      int xs[4], base = 1 << 30;
      for (int k = 0; k < 4; ++k) {
        int q = 4 * d + k - EDGE;
        if (q < 0) q = -q;
        if (q >= g0.w) q = 2 * (g0.w - 1) - q;
        xs[k] = q; base = std::min(base, q);
      }
      base = std::min(base, g0.w - 4);
      PyrEdge pe; pe.dword = d; pe.base = base; pe.sel = 0; pe.pad_ = 0;
      for (int k = 0; k < 4; ++k) pe.sel |= (uint32_t)(xs[k] - base) << (8 * k);
      pedges.push_back(pe);
    }
    p0.nEdge = (int)pedges.size();
    p0.nIntItems = p0.nInt * p0.nG; p0.nItems = p0.nIntItems + p0.nEdge * p0.nG;
    p0.magicInt = (uint32_t)((1ull << 32) / (unsigned)std::max(p0.nInt, 1) + 1);
    p0.magicEdge = (uint32_t)((1ull << 32) / (unsigned)std::max(p0.nEdge, 1) + 1);
    MORB_REQUIRE((unsigned long long)p0.nItems * (unsigned)std::max(p0.nInt, p0.nEdge) < (1ull << 32), MORB_ERR_UNSUPPORTED, "image too large for the item index arithmetic");
  }
  e->pyrPacked = pyrPacked;
  e->totalCells = cellBase;
  {
    // Two launches of k_fastw: the LDS window is sized by the tallest cell, and the few large cells of the small top levels would
    // cost every workgroup of the big levels its occupancy.  Group 0 = segments of levels whose cells are at most two rows taller
    // than level 0's, group 1 = the rest (possibly empty).
    const int rowsA = e->geom[0].hCell + 6 + 2;
    std::vector<FastSeg> a, b;
    int ra = 0, rb = 0;
    for (const FastSeg& sd : segs) {
      const int rows = e->geom[sd.geo & 0xFF].hCell + 6;
      if (rows <= rowsA) { a.push_back(sd); ra = std::max(ra, rows); } else { b.push_back(sd); rb = std::max(rb, rows); }
    }
    // cells are dispatched from the top level down — the reverse of the order in which the pyramid stage wrote the levels, so the most
    // recently written (still cached) levels are read first (measured on round 2's kernel at 512 images: 997 -> 969 us)
    std::reverse(a.begin(), a.end()); std::reverse(b.begin(), b.end());
    e->fastSegs[0] = (int)a.size(); e->fastSegs[1] = (int)b.size();
    e->fastRows[0] = ra; e->fastRows[1] = rb;
    segs = a;
    segs.insert(segs.end(), b.begin(), b.end());
  }

  e->selPerImg = selBase;
  e->blurTiles = blurTileBase;
  e->pyrBytes = pyrOff; e->blurBytes = blurOff; e->qtElems = qtOff;
  e->outCap = selBase;
  // Candidate keys of a level live in LDS (two arrays) when they fit, else in the global scratch (much slower: every partition pass goes
  // through the L2).  Level 0's capacity scales with its area — twice the ~1 candidate per 233 px the benchmark images give, i.e. the
  // kLdsKeys = 3072 of a 752 x 480 image — and is cut to what the LDS leaves beside the node arrays (1920 x 1080 / 4000 features: ~8 k keys);
  // the other levels' capacities follow their width (below).
  {
    constexpr size_t kLdsBudget = 155 * 1024;   // largest dynamic allocation of one workgroup (160 KB minus the kernel's static LDS — 4.4 KB since round 6 — with room to spare)
    constexpr size_t kLdsCu = 160 * 1024, kLdsStatic = 4608;   // k workgroups share a CU when each takes <= 160 KB / k, static part included
    auto lds_bytes = [&](const LevelGeom& g, int keyCap) -> size_t {   // == the carve-up in k_distribute
      const int ncell = g.nCols * g.nRows;
      const size_t b = (size_t)g.nodeCap * (8 + 8 + sizeof(morbqt::Node) + 4 + 2 + 2) + (ncell + 1 <= g.nodeCap ? 0 : (size_t)(ncell + 1) * 4) +
                       (size_t)g.listCap * 2 + (size_t)keyCap * 8;
      return (b + 63) / 64 * 64;
    };
    const size_t fixed0 = lds_bytes(e->geom[0], 0);
    MORB_REQUIRE(fixed0 + 1024 * 8 <= kLdsBudget, MORB_ERR_UNSUPPORTED, "nfeatures too large for the LDS-resident quadtree");
    const double area0 = (double)e->geom[0].w * e->geom[0].h;
    const long long want = std::max<long long>((long long)kLdsKeys * MORB_QT_KEYF / 200, (long long)(area0 * MORB_QT_KEYF / 23300));
    const long long fit = (long long)((kLdsBudget - fixed0) / 8);
    e->distKeyCap = (int)std::min(want, fit) / 64 * 64;
    size_t need[kMaxLevels];
    size_t needMax = 0;
    for (int l = 0; l < L; ++l) {
      LevelGeom& g = e->geom[l];
      // candidates per level fall like the level's width, not its area (measured on the benchmark images: 1400, 1255, 1018, 839, 767, 652,
      // 570, 440 at 752 x 480 — FAST answers to the same structures at every scale)
      const long long k = (long long)std::ceil(e->distKeyCap * ((double)g.w / e->geom[0].w));
      g.distKeyCap = (int)std::min<long long>(e->distKeyCap, std::max<long long>(256, (k + 63) / 64 * 64));
      need[l] = lds_bytes(g, g.distKeyCap);
      needMax = std::max(needMax, need[l]);
    }
    // First fit, decreasing need, into bins of 1 / k of a CU's LDS.  Measured at 752 x 480 / 1200 features, 512 images (us alone | under the
    // blur | end-to-end frames/s): one level per workgroup 470 | 495 | 85.3 k; k = 3: 498 | 521 | 84.2 k; k = 2: 398 | 424 | 85.9 k;
    // k = 1 (two workgroups of four waves per image, one per CU): 433 | 465 | 86.1 k — what is built.
    int order[kMaxLevels];
    for (int l = 0; l < L; ++l) order[l] = l;
    std::stable_sort(order, order + L, [&](int a, int b) { return need[a] > need[b]; });
#ifndef MORB_QT_PACK
#define MORB_QT_PACK 2   // (round 4, B = 512: bins of half a CU — levels {0, 4}, {1, 3}, {2, 5, 6}, {7}: four workgroups per 752 x 480 image, two of the big ones per CU — 131.6 k against 130.0 k frames/s with whole-CU bins, 126.4 k with thirds)
#endif
    constexpr int packEnv = MORB_QT_PACK;   // workgroups per CU the bins are sized for (measured in round 2: 0 = one level per workgroup, 2, 3: no better end to end; again at the end of round 3 with the faster blur: 1 / 2 / 3 -> quadtree 427 / 398 / 485 us per 512 images but the blur beside it 364 / 404 / 377 and the join 13 / 23 / 13: the slot stays ~430)
    int kMax = 1;
    while (std::min(kLdsBudget, kLdsCu / (kMax + 1) - kLdsStatic) >= needMax) ++kMax;
    const int bestK = std::max(1, std::min(packEnv, kMax));
    const size_t binCap = std::min(kLdsBudget, kLdsCu / bestK - kLdsStatic);
    size_t binFill[kMaxLevels] = {0};
    int binWaves[kMaxLevels] = {0};
    int nb = 0;
    for (int i = 0; i < L; ++i) {
      const int l = order[i];
      int b = 0;
      while (b < nb && !(packEnv > 0 && binWaves[b] < QT_MAX_WAVES && binFill[b] + need[l] <= binCap)) ++b;
      if (b == nb) ++nb;
      e->geom[l].distGroup = b; e->geom[l].distWave = binWaves[b]++; e->geom[l].distLdsOff = (int)binFill[b];
      binFill[b] += need[l];
    }
    e->distGroups = nb; e->distWaves = 1; e->distSmem = 0;
    for (int b = 0; b < nb; ++b) { e->distWaves = std::max(e->distWaves, binWaves[b]); e->distSmem = std::max(e->distSmem, binFill[b]); }
    // The packing for calls with few images (one frame at a time: latency, or BASELINE configs[3]'s one 1920 x 1080 frame per GPU): a level
    // of >= 160 k pixels gets a workgroup of its own whose QT_TEAM_WAVES waves work it as a team (quadtree.h), the smaller levels are
    // packed as above.  A second copy of the geometry carries these assignments.
    for (int l = 0; l < L; ++l) { teamGeom[l] = e->geom[l]; teamGeom[l].distTeam = 0; }
    int tb = 0;
    size_t tFill[kMaxLevels] = {0};
    int tWaves[kMaxLevels] = {0};
    e->distSmemTeam = 0;
    for (int l = 0; l < L; ++l)
      if ((long long)e->geom[l].w * e->geom[l].h >= MORB_TEAM_MIN_PIXELS) {
        teamGeom[l].distTeam = 1; teamGeom[l].distGroup = tb; teamGeom[l].distWave = 0; teamGeom[l].distLdsOff = 0;
        tFill[tb] = need[l]; tWaves[tb] = QT_MAX_WAVES; ++tb;
      }
    const int firstPacked = tb;
    for (int i = 0; i < L; ++i) {
      const int l = order[i];
      if (teamGeom[l].distTeam) continue;
      int b = firstPacked;
      while (b < tb && !(tWaves[b] < QT_MAX_WAVES && tFill[b] + need[l] <= binCap)) ++b;
      if (b == tb) ++tb;
      teamGeom[l].distGroup = b; teamGeom[l].distWave = tWaves[b]++; teamGeom[l].distLdsOff = (int)tFill[b];
      tFill[b] += need[l];
    }
    e->distGroupsTeam = firstPacked > 0 ? tb : 0;   // (no big level: the team launch is not used)
    for (int b = 0; b < tb; ++b) e->distSmemTeam = std::max(e->distSmemTeam, tFill[b]);
  }

  MORB_HIP_CHECK(hipMalloc(&e->d_geom, sizeof(LevelGeom) * kMaxLevels));
  MORB_HIP_CHECK(hipMemcpy(e->d_geom, e->geom, sizeof(LevelGeom) * kMaxLevels, hipMemcpyHostToDevice));
  MORB_HIP_CHECK(hipMalloc(&e->d_geomTeam, sizeof(LevelGeom) * kMaxLevels));
  MORB_HIP_CHECK(hipMemcpy(e->d_geomTeam, teamGeom, sizeof(LevelGeom) * kMaxLevels, hipMemcpyHostToDevice));
  for (int l = 0; l < kMaxLevels; ++l) {
    const LevelGeom& g = e->geom[l < L ? l : L - 1];
    e->fastGeom.cellBase[l] = l < L ? g.cellBase : 0x7fffffff;
    e->fastGeom.nCols[l] = g.nCols; e->fastGeom.wCell[l] = g.wCell; e->fastGeom.hCell[l] = g.hCell;
    e->fastGeom.pstride[l] = g.pstride; e->fastGeom.maxBorderX[l] = g.maxBorderX; e->fastGeom.maxBorderY[l] = g.maxBorderY;
    e->fastGeom.pyrOff[l] = g.pyrOff; e->fastGeom.pyrImg[l] = g.pyrImg;
    e->descGeom.pyrOff[l] = g.pyrOff; e->descGeom.pyrImg[l] = g.pyrImg; e->descGeom.blurOff[l] = g.blurOff;
    e->descGeom.blurImg[l] = g.blurImg; e->descGeom.pstride[l] = g.pstride; e->descGeom.bstride[l] = g.bstride;
    e->descGeom.scale[l] = g.scale; e->descGeom.kpSize[l] = g.kpSize;
  }
  MORB_HIP_CHECK(hipMalloc(&e->d_segTab, sizeof(FastSeg) * segs.size()));
  MORB_HIP_CHECK(hipMemcpy(e->d_segTab, segs.data(), sizeof(FastSeg) * segs.size(), hipMemcpyHostToDevice));
  MORB_HIP_CHECK(hipMalloc(&e->d_tabs, sizeof(ResizeTab) * std::max<size_t>(tabs.size(), 1)));
  if (!tabs.empty()) MORB_HIP_CHECK(hipMemcpy(e->d_tabs, tabs.data(), sizeof(ResizeTab) * tabs.size(), hipMemcpyHostToDevice));
  MORB_HIP_CHECK(hipMalloc(&e->d_pcol, sizeof(PyrCol) * std::max<size_t>(pcols.size(), 1)));
  if (!pcols.empty()) MORB_HIP_CHECK(hipMemcpy(e->d_pcol, pcols.data(), sizeof(PyrCol) * pcols.size(), hipMemcpyHostToDevice));
  MORB_HIP_CHECK(hipMalloc(&e->d_prow, sizeof(PyrRow) * std::max<size_t>(prows.size(), 1)));
  if (!prows.empty()) MORB_HIP_CHECK(hipMemcpy(e->d_prow, prows.data(), sizeof(PyrRow) * prows.size(), hipMemcpyHostToDevice));
  MORB_HIP_CHECK(hipMalloc(&e->d_pedge, sizeof(PyrEdge) * std::max<size_t>(pedges.size(), 1)));
  if (!pedges.empty()) MORB_HIP_CHECK(hipMemcpy(e->d_pedge, pedges.data(), sizeof(PyrEdge) * pedges.size(), hipMemcpyHostToDevice));
  MORB_HIP_CHECK(hipMalloc(&e->d_pyr, e->pyrBytes + 256));
  MORB_HIP_CHECK(hipMemset(e->d_pyr, 0, e->pyrBytes + 256));   // (the alignment slack behind each row is never written; other kernels' wide loads may touch it)
  MORB_HIP_CHECK(hipMalloc(&e->d_blur, e->blurBytes + 256));
  MORB_HIP_CHECK(hipMalloc(&e->d_cand, sizeof(uint32_t) * (size_t)nimg * e->totalCells * e->cellCap));
  MORB_HIP_CHECK(hipMalloc(&e->d_candCnt, sizeof(int) * (size_t)nimg * e->totalCells));
  MORB_HIP_CHECK(hipMalloc(&e->d_qt, sizeof(uint32_t) * std::max<size_t>(e->qtElems, 1)));
  MORB_HIP_CHECK(hipMalloc(&e->d_sel, sizeof(uint32_t) * (size_t)nimg * e->selPerImg));
  MORB_HIP_CHECK(hipMalloc(&e->d_selCnt, sizeof(int) * (size_t)nimg * L));
  MORB_HIP_CHECK(hipMalloc(&e->d_kref, sizeof(int2) * (size_t)nimg * e->selPerImg));
  MORB_HIP_CHECK(hipMalloc(&e->d_lap, sizeof(int) * (size_t)nimg * 2));
  MORB_HIP_CHECK(hipMemset(e->d_selCnt, 0, sizeof(int) * (size_t)nimg * L));
  MORB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), e->umax, sizeof(int) * 16));
  MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_distribute),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(e->distSmem, e->distSmemTeam)));
  {
    // k_fastw: the LDS pitch of a wave's window = the widest segment window, rounded up to whole 16-px blocks
    int twMax = 0;
    for (const FastSeg& sd : segs) twMax = std::max(twMax, (sd.geo >> 16) & 0xFF);
    twMax += FW_SH;   // (the tile holds the window from one column to its left, fast_wave.h)
    e->fastP = twMax <= 48 ? 48 : (twMax <= 64 ? 64 : 80);   // (wCell <= 73: a cell's window is at most 79 px wide)
    if (e->fastP == 48 && std::max(e->fastRows[0], e->fastRows[1]) > FW_ROWS16) e->fastP = 64;   // (16-bit queue entries address 1024 dwords of tile)
    for (int k = 0; k < 2; ++k) {
      const int r = e->fastRows[k];
      const int region = e->fastP == 48 ? fw_region_bytes<48>(r) : e->fastP == 64 ? fw_region_bytes<64>(r) : fw_region_bytes<80>(r);
      e->fastSmem[k] = (size_t)FW_WAVES * region;
    }
    const void* fn = e->fastP == 48 ? reinterpret_cast<const void*>(k_fastw<48>) : e->fastP == 64 ? reinterpret_cast<const void*>(k_fastw<64>)
                   : reinterpret_cast<const void*>(k_fastw<80>);
    MORB_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  }
  e->W = W; e->H = H; e->nimgCap = nimg;
  return MORB_OK;
}

}  // namespace

extern "C" {

const char* morb_last_error(void) { return last_error().c_str(); }

int morb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int morb_extractor_create(morb_extractor** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                          int minThFAST, int device) {
  MORB_REQUIRE(out, MORB_ERR_INVALID, "out is NULL");
  *out = nullptr;
  MORB_REQUIRE(nfeatures > 0 && nlevels >= 1 && nlevels <= kMaxLevels && scaleFactor > 1.0f, MORB_ERR_INVALID,
               "bad extractor parameters");
  MORB_REQUIRE(iniThFAST >= 0 && iniThFAST <= 255 && minThFAST >= 0 && minThFAST <= 255, MORB_ERR_INVALID,
               "FAST thresholds must be in [0,255]");
  int ndev = 0;
  MORB_HIP_CHECK(hipGetDeviceCount(&ndev));
  MORB_REQUIRE(device >= 0 && device < ndev, MORB_ERR_INVALID, "no such HIP device");
  morb_extractor* e = new morb_extractor();
  e->nfeatures = nfeatures; e->scaleFactor = scaleFactor; e->nlevels = nlevels; e->iniTh = iniThFAST;
  e->minTh = minThFAST; e->device = device;
  // ORBextractor.cc:413-443
  e->scale.resize(nlevels); e->sigma2.resize(nlevels); e->invScale.resize(nlevels); e->invSigma2.resize(nlevels);
  e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) { e->scale[i] = e->scale[i - 1] * scaleFactor; e->sigma2[i] = e->scale[i] * e->scale[i]; }
  for (int i = 0; i < nlevels; i++) { e->invScale[i] = 1.0f / e->scale[i]; e->invSigma2[i] = 1.0f / e->sigma2[i]; }
  e->quota.resize(nlevels);
  float factor = 1.0f / scaleFactor;
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) { e->quota[l] = cvRoundF(nDesired); sum += e->quota[l]; nDesired *= factor; }
  e->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
  // umax (:451-463)
  {
    int v, v0, vmax = (int)std::floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(HALF_PATCH * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v <= vmax; ++v) e->umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
      while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
      e->umax[v] = v0;
      ++v0;
    }
  }
  {
    static const int kStd[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int i = 0; i < 16; ++i)
      if (e->umax[i] != kStd[i]) { set_error("umax table mismatch"); delete e; return MORB_ERR_UNSUPPORTED; }
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&e->sideStream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->evFork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->evPyr, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->evJoin, hipEventDisableTiming) != hipSuccess) {
    set_error("cannot create a stream on device %d", device);
    delete e;
    return MORB_ERR_HIP;
  }
  // status word the kernels can flag (pinned, device-mapped: the host reads it after a synchronisation without a copy)
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_status), sizeof(int), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_status), e->h_status, 0) != hipSuccess) {
    set_error("cannot allocate the status word");
    morb_extractor_destroy(e);
    return MORB_ERR_HIP;
  }
  *e->h_status = 0;
  *out = e;
  return MORB_OK;
}

int morb_extractor_status(morb_extractor* e, int* flags) {
  MORB_REQUIRE(e, MORB_ERR_INVALID, "extractor is NULL");
  const int f = __atomic_exchange_n(e->h_status, 0, __ATOMIC_ACQ_REL);
  if (flags) *flags = f;
  // (bit 0 was "a level held more than 65535 FAST candidates" until round 5; the quadtree no longer has that limit and no kernel raises a flag today)
  if (f) { set_error("an extraction was flagged on the device (flags 0x%x): the keypoints of that call are not valid", f); return MORB_ERR_UNSUPPORTED; }
  return MORB_OK;
}

void morb_extractor_destroy(morb_extractor* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  free_buffers(e);
  free_staging(e);
  for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  e->ev.clear();
  if (e->evFork) (void)hipEventDestroy(e->evFork);
  if (e->evPyr) (void)hipEventDestroy(e->evPyr);
  if (e->evJoin) (void)hipEventDestroy(e->evJoin);
  if (e->sideStream) (void)hipStreamDestroy(e->sideStream);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->h_status) (void)hipHostFree(e->h_status);
  delete e;
}

int morb_extractor_levels(const morb_extractor* e) { return e ? e->nlevels : MORB_ERR_INVALID; }
float morb_extractor_scale_factor(const morb_extractor* e) { return e ? e->scaleFactor : 0.f; }
int morb_extractor_tables(const morb_extractor* e, float* sc, float* isc, float* s2, float* is2, int* fpl) {
  MORB_REQUIRE(e, MORB_ERR_INVALID, "extractor is NULL");
  for (int i = 0; i < e->nlevels; ++i) {
    if (sc) sc[i] = e->scale[i];
    if (isc) isc[i] = e->invScale[i];
    if (s2) s2[i] = e->sigma2[i];
    if (is2) is2[i] = e->invSigma2[i];
    if (fpl) fpl[i] = e->quota[i];
  }
  return MORB_OK;
}
int morb_extractor_max_keypoints(const morb_extractor* e) {
  if (!e) return MORB_ERR_INVALID;
  int n = 0;
  for (int l = 0; l < e->nlevels; ++l) n += std::max(e->quota[l] + 3, 16) + 1;
  return n;
}
int morb_extractor_set_profiling(morb_extractor* e, int enable) {
  MORB_REQUIRE(e, MORB_ERR_INVALID, "extractor is NULL");
  MORB_HIP_CHECK(hipSetDevice(e->device));
  e->profiling = enable != 0;
  e->profCalls = 0;
  if (e->profiling && e->ev.empty()) {
    e->ev.resize((size_t)morb_extractor::kProfRing * 8, nullptr);
    for (auto& ev : e->ev) MORB_HIP_CHECK(hipEventCreate(&ev));
  }
  return MORB_OK;
}
int morb_extractor_event_after_fast(morb_extractor* e, void** event) {
  MORB_REQUIRE(e && event, MORB_ERR_INVALID, "NULL argument");
  *event = (void*)e->evFork;
  return MORB_OK;
}
int morb_extractor_event_after_pyramid(morb_extractor* e, void** event) {
  MORB_REQUIRE(e && event, MORB_ERR_INVALID, "NULL argument");
  e->wantPyrEvent = true;   // recorded by the extractions queued from now on
  *event = (void*)e->evPyr;
  return MORB_OK;
}
int morb_stream_wait_event(void* stream, void* event) {
  MORB_REQUIRE(event, MORB_ERR_INVALID, "event is NULL");
  MORB_HIP_CHECK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0));
  return MORB_OK;
}
int morb_extractor_stage_ms(morb_extractor* e, float* ms7) {
  MORB_REQUIRE(e && ms7, MORB_ERR_INVALID, "NULL argument");
  const int n = e->profCalls < morb_extractor::kProfRing ? e->profCalls : morb_extractor::kProfRing;
  for (int i = 0; i < 7; ++i) e->stageMs[i] = 0.f;
  for (int c = 0; c < n; ++c) {
    // events: 0 start, 1 pyramid done, 2 FAST done, 3 quadtree done, 4 layout done, 5 describe done (launch stream);
    // 6 / 7 around the blur on the side stream.  Stages: pyramid, blur, fast, distribute, layout, describe, total.
    hipEvent_t* ev = &e->ev[(size_t)c * 8];
    MORB_HIP_CHECK(hipEventSynchronize(ev[5]));
    MORB_HIP_CHECK(hipEventSynchronize(ev[7]));
    const int from[7] = {0, 6, 1, 2, 3, 4, 0}, to[7] = {1, 7, 2, 3, 4, 5, 5};
    for (int i = 0; i < 7; ++i) {
      float ms = 0.f;
      MORB_HIP_CHECK(hipEventElapsedTime(&ms, ev[from[i]], ev[to[i]]));
      e->stageMs[i] += ms / n;
    }
  }
  for (int i = 0; i < 7; ++i) ms7[i] = e->stageMs[i];
  e->profCalls = 0;
  return n;
}

int morb_extract_batch(morb_extractor* e, const uint8_t* d_images, int nimg, int width, int height, int stride,
                       size_t image_pitch, const int* lap, morb_keypoint* d_kps, uint8_t* d_desc, int cap,
                       int* d_count, int* d_mono, void* stream_) {
  MORB_REQUIRE(e && d_images && d_kps && d_desc && d_count && d_mono, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nimg > 0, MORB_ERR_INVALID, "nimg must be positive");
  if (width <= 0 || height <= 0) { set_error("empty image"); return MORB_ERR_EMPTY; }
  MORB_REQUIRE(stride >= width && image_pitch >= (size_t)stride * height, MORB_ERR_INVALID, "bad stride/pitch");
  // k_layout places the lapping-area keypoints from the back of [0, count): a smaller cap would cut the wrong end
  MORB_REQUIRE(cap >= morb_extractor_max_keypoints(e), MORB_ERR_CAPACITY, "cap must be at least morb_extractor_max_keypoints()");
  MORB_HIP_CHECK(hipSetDevice(e->device));
  int rc = configure(e, width, height, nimg);
  if (rc != MORB_OK) return rc;
  hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
  const int L = e->nlevels;
  e->nimgLast = nimg;

  {
    std::vector<int> lapv((size_t)nimg * 2, 0);
    if (lap) memcpy(lapv.data(), lap, sizeof(int) * nimg * 2);
    if (lapv != e->lapLast) {  // rare: the lapping areas are per-camera constants
      MORB_HIP_CHECK(hipStreamSynchronize(st));
      MORB_HIP_CHECK(hipMemcpy(e->d_lap, lapv.data(), sizeof(int) * nimg * 2, hipMemcpyHostToDevice));
      e->lapLast = lapv;
    }
  }

  hipEvent_t* evs = e->profiling ? &e->ev[(size_t)(e->profCalls % morb_extractor::kProfRing) * 8] : nullptr;
  auto mark = [&](int i) { if (evs) (void)hipEventRecord(evs[i], st); };
  mark(0);
  {
    const LevelGeom& g0 = e->geom[0];
    const PyrLevel0& p0 = e->pyrL0;
    Level0Args a0;
    a0.edge = e->d_pedge; a0.nInt = p0.nInt; a0.j0 = p0.j0; a0.nEdge = p0.nEdge; a0.nIntItems = p0.nIntItems; a0.nItems = p0.nItems;
    a0.H = g0.h + 2 * EDGE; a0.h = g0.h; a0.dpstride = g0.pstride; a0.magicInt = p0.magicInt; a0.magicEdge = p0.magicEdge; a0.dOff = g0.pyrOff; a0.dImg = g0.pyrImg;
    auto resize_args = [&](int l) {
      const LevelGeom& g = e->geom[l];
      const PyrLevel& pl = e->pyrLv[l];
      ResizeArgs a;
      a.col = e->d_pcol + pl.colOff; a.row = e->d_prow + pl.rowOff; a.nC = pl.nC; a.nItems = pl.nItems; a.H = g.h + 2 * EDGE; a.dpstride = g.pstride;
      a.magicC = pl.magicC; a.dOff = g.pyrOff; a.dImg = g.pyrImg;
      return a;
    };
    // Round 6: the eight dependent launches run over CHUNKS of images.  Level l reads level l - 1 back: with all 1024 images of a bench step in one
    // launch the producer's 400 MB have long left the L2 / Infinity Cache (256 MB) when the consumer starts and every level is read from HBM again;
    // chunk by chunk the previous level of the chunk is still on chip (profiles/r06/pyramid_chunk_sweep.txt).
    static const int pyrChunkEnv = [] { const char* v = getenv("MORB_PYR_CHUNK"); return v ? atoi(v) : 0; }();
    const int chunk = pyrChunkEnv > 0 ? pyrChunkEnv : kPyrChunkImages;
    for (int i0 = 0; i0 < nimg; i0 += chunk) {
      const int ni = nimg - i0 < chunk ? nimg - i0 : chunk;
      Level0Args a0c = a0;
      a0c.dOff += (unsigned long long)i0 * a0.dImg;
      const uint8_t* src = d_images + (size_t)i0 * image_pitch;
      if (e->pyrPacked) {
        // level 0 first, in its own launch: level 1 reads it back from the pyramid (dword-aligned rows with slack behind them — the wide
        // aligned window loads of k_resize cannot be pointed at a caller-owned buffer; the fused level-0 + level-1 kernel of round 2 did
        // that with byte-granular 8-byte loads at twice the memory-pipe cost)
        hipLaunchKernelGGL(k_level0, dim3(div_up(p0.nItems, 256), 1, ni), dim3(256), 0, st, src, stride, image_pitch, e->d_pyr, a0c);
        for (int l = 1; l < L; ++l) {
          const LevelGeom& gs = e->geom[l - 1];
          const PyrLevel& pl = e->pyrLv[l];
          ResizeArgs ra = resize_args(l);
          ra.dOff += (unsigned long long)i0 * ra.dImg;
          hipLaunchKernelGGL(k_resize, dim3(div_up(pl.nItems, 256), 1, ni), dim3(256), sizeof(PyrRow) * pl.ldsRows, st, e->d_pyr,
                             gs.pyrOff + (unsigned long long)i0 * gs.pyrImg + (unsigned long long)EDGE * gs.pstride, gs.pyrImg, gs.pstride, ra);
        }
      } else {
        hipLaunchKernelGGL(k_level0, dim3(div_up(p0.nItems, 256), 1, ni), dim3(256), 0, st, src, stride, image_pitch, e->d_pyr, a0c);
        for (int l = 1; l < L; ++l) {
          const LevelGeom& g = e->geom[l];
          dim3 gr(div_up(g.pstride / 4, 64), div_up(g.h + 2 * EDGE, 4 * PY_ROWS), ni);
          hipLaunchKernelGGL(k_resize_gather, gr, dim3(256), 0, st, e->d_pyr, e->geom[l - 1], g, e->d_tabs + g.xtabOff, e->d_tabs + g.ytabOff, i0);
        }
      }
    }
  }
  mark(1);
  if (e->wantPyrEvent) MORB_HIP_CHECK(hipEventRecord(e->evPyr, st));   // (only for callers that asked for it: an event between two launches costs a one-frame call ~10 us)
  // FAST: the two launch groups (cells of the big levels; the taller cells of the small top levels) follow each other in the launch
  // stream.  Grid x = image: hardware deals consecutive workgroups round-robin over the 8 XCDs, so with a multiple of 8 images all
  // cells of an image meet in one XCD's L2.
  for (int k = 0, s0 = 0; k < 2; s0 += e->fastSegs[k], ++k)
    if (e->fastSegs[k]) {
      const dim3 gr(nimg, div_up(div_up(e->fastSegs[k], FW_WAVES), FW_CPW)), bl(64 * FW_WAVES);
#define MORB_FW_LAUNCH(PP) hipLaunchKernelGGL(k_fastw<PP>, gr, bl, e->fastSmem[k], st, e->fastGeom, e->d_segTab + s0, e->fastSegs[k], e->d_pyr, e->d_cand, \
                                              e->d_candCnt, e->totalCells, e->cellCap, e->fastRows[k], e->iniTh, e->minTh)
      switch (e->fastP) {
        case 48: MORB_FW_LAUNCH(48); break;
        case 64: MORB_FW_LAUNCH(64); break;
        default: MORB_FW_LAUNCH(80); break;
      }
#undef MORB_FW_LAUNCH
    }
  mark(2);
  // The blur only feeds the descriptors and is VALU-bound; the quadtree is one latency-bound wave per (level, image)
  // that leaves the vector ALUs ~90 % idle.  Fork: the blur runs on the handle's side stream underneath the quadtree
  // and the layout, and the launch stream joins it again before k_describe.
  MORB_HIP_CHECK(hipEventRecord(e->evFork, st));
  // the quadtree is enqueued first so that its long-running waves get their slots before the blur fills the chip;
  // the workgroups with level 0 (the longest wave) first: grid x = image, y = group of levels
  // (which of the two is enqueued first makes no difference: measured both ways)
  // (developer switch MORB_TEAM_MID=n: calls of up to n images take the team packing with FOUR waves per team; measured, off — see profiles/r06/README.md)
  static const int teamMid = [] { const char* v = getenv("MORB_TEAM_MID"); return v ? atoi(v) : 0; }();
  if (nimg <= kTeamMaxImages && e->distGroupsTeam > 0)   // few images: latency matters, the big levels are worked by teams of waves
    hipLaunchKernelGGL(k_distribute, dim3(nimg, e->distGroupsTeam), dim3(64 * QT_TEAM_WAVES), e->distSmemTeam, st, e->d_geomTeam, e->d_cand, e->d_candCnt,
                       e->totalCells, e->cellCap, e->d_qt, e->d_sel, e->d_selCnt, e->selPerImg, L, 0, e->d_status);
  else if (nimg <= teamMid && e->distGroupsTeam > 0)
    hipLaunchKernelGGL(k_distribute, dim3(nimg, e->distGroupsTeam), dim3(64 * QT_MAX_WAVES), e->distSmemTeam, st, e->d_geomTeam, e->d_cand, e->d_candCnt,
                       e->totalCells, e->cellCap, e->d_qt, e->d_sel, e->d_selCnt, e->selPerImg, L, 0, e->d_status);
  else
    // (one launch per bin of levels, each with its own LDS size — all bins of one launch get the largest bin's — measured: the launches
    // follow each other on the stream, 128 -> 204 us per 128 images, 420 -> 435 per 512; round 4: a launch per LEVEL, single-wave workgroups with
    // exactly the level's LDS, the eight launches side by side on streams of their own: 399 -> 513 us per 512 images alone, bench 124.5 -> 112.8 k frames/s)
    hipLaunchKernelGGL(k_distribute, dim3(nimg, e->distGroups), dim3(64 * e->distWaves), e->distSmem, st, e->d_geom, e->d_cand, e->d_candCnt,
                       e->totalCells, e->cellCap, e->d_qt, e->d_sel, e->d_selCnt, e->selPerImg, L, 0, e->d_status);
  hipStream_t sideStream = e->sideStream;
  MORB_HIP_CHECK(hipStreamWaitEvent(sideStream, e->evFork, 0));
  if (evs) (void)hipEventRecord(evs[6], sideStream);
  hipLaunchKernelGGL(k_blur, dim3(e->blurTiles, nimg), dim3(256), 0, sideStream, e->d_geom, L, e->d_pyr, e->d_blur);
  if (evs) (void)hipEventRecord(evs[7], sideStream);
  MORB_HIP_CHECK(hipEventRecord(e->evJoin, sideStream));
  mark(3);
  if (nimg <= kTeamMaxImages) hipLaunchKernelGGL(k_layout<16>, dim3(nimg), dim3(64 * 16), 0, st, e->d_geom, L, e->d_sel, e->d_selCnt, e->selPerImg,
                                                 e->d_lap, e->d_kref, d_count, d_mono, cap);
  else hipLaunchKernelGGL(k_layout<4>, dim3(nimg), dim3(64 * 4), 0, st, e->d_geom, L, e->d_sel, e->d_selCnt, e->selPerImg,
                          e->d_lap, e->d_kref, d_count, d_mono, cap);
  mark(4);
  MORB_HIP_CHECK(hipStreamWaitEvent(st, e->evJoin, 0));
  constexpr int descRev = 1;   // images in reverse order: the blur wrote the last ones most recently (581 -> 565 us at 512 images)
  hipLaunchKernelGGL(k_describe, dim3(div_up(e->selPerImg, DESC_WAVES * DESC_KPW), nimg), dim3(64 * DESC_WAVES), 0, st, e->descGeom, e->d_pyr,
                     e->d_blur, e->d_kref, e->selPerImg, d_kps, d_desc, cap, descRev);
  mark(5);
  MORB_HIP_CHECK(hipGetLastError());
  if (e->profiling) ++e->profCalls;
  return MORB_OK;
}

int morb_extract(morb_extractor* e, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                 morb_keypoint* kps, uint8_t* desc, int cap, int* n) {
  MORB_REQUIRE(e && n, MORB_ERR_INVALID, "NULL argument");
  *n = 0;
  if (!image || width <= 0 || height <= 0) { set_error("empty image"); return MORB_ERR_EMPTY; }
  MORB_REQUIRE(kps && desc, MORB_ERR_INVALID, "NULL output");
  MORB_REQUIRE(stride >= width, MORB_ERR_INVALID, "bad stride");
  MORB_HIP_CHECK(hipSetDevice(e->device));
  const size_t bytes = (size_t)stride * height;
  const int maxk = morb_extractor_max_keypoints(e);
  if (e->imgBytes < bytes || !e->d_kps1) {
    MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->d_img) (void)hipFree(e->d_img);
    e->d_img = nullptr;
    MORB_HIP_CHECK(hipMalloc(&e->d_img, bytes));
    e->imgBytes = bytes;
    if (!e->d_kps1) {
      // ONE device block laid out like the pinned buffer of the way back: count | monoIndex | pad to 16 | keypoints | descriptors — one copy brings a call's results
      // home (four copies before round 6: ~6 us of launch latency each on a 0.2 ms call)
      const size_t out1 = 16 + (sizeof(morb_keypoint) + 32) * (size_t)maxk;
      MORB_HIP_CHECK(hipMalloc(&e->d_out1, out1));
      uint8_t* b = static_cast<uint8_t*>(e->d_out1);
      e->d_cnt1 = reinterpret_cast<int*>(b); e->d_mono1 = reinterpret_cast<int*>(b) + 1;
      e->d_kps1 = reinterpret_cast<morb_keypoint*>(b + 16); e->d_desc1 = b + 16 + sizeof(morb_keypoint) * (size_t)maxk;
    }
  }
  int rc = configure(e, width, height, 1);
  if (rc != MORB_OK) return rc;
  // Host <-> device through ONE pinned buffer: the image is copied into it and uploaded asynchronously, the results (count, monoIndex,
  // all keypoint / descriptor slots) come back in one asynchronous copy and one synchronisation; pageable copies straight from / to the
  // caller's buffers are staged and synchronised by the runtime one by one.
  const size_t outBytes = 16 + (sizeof(morb_keypoint) + 32) * (size_t)maxk, need = std::max(bytes, outBytes);
  if (e->ioBytes1 < need) {
    MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->h_io1) (void)hipHostFree(e->h_io1);
    e->h_io1 = nullptr; e->ioBytes1 = 0;
    MORB_HIP_CHECK(hipHostMalloc(&e->h_io1, need));
    e->ioBytes1 = need;
  }
  const int stale = __atomic_load_n(e->h_status, __ATOMIC_ACQUIRE);   // flags of earlier, unqueried calls on this handle
  {
    // the image goes up in pieces: while the copy engine moves piece k the host copies piece k + 1 into the pinned buffer (a 1920 x 1080 image is 2 MB:
    // ~0.1 ms of memcpy in front of a 0.06 ms upload when done in one go)
    const size_t piece = bytes > ((size_t)512 << 10) ? (bytes / 4 + 4095) & ~(size_t)4095 : bytes;
    for (size_t o = 0; o < bytes; o += piece) {
      const size_t nb = std::min(piece, bytes - o);
      memcpy(e->h_io1 + o, image + o, nb);
      MORB_HIP_CHECK(hipMemcpyAsync(e->d_img + o, e->h_io1 + o, nb, hipMemcpyHostToDevice, e->stream));
    }
  }
  int lap[2] = {lap0, lap1};
  rc = morb_extract_batch(e, e->d_img, 1, width, height, stride, bytes, lap, e->d_kps1, e->d_desc1, maxk, e->d_cnt1,
                          e->d_mono1, e->stream);
  if (rc != MORB_OK) return rc;
  int* hcnt = reinterpret_cast<int*>(e->h_io1);
  morb_keypoint* hkps = reinterpret_cast<morb_keypoint*>(e->h_io1 + 16);
  uint8_t* hdesc = e->h_io1 + 16 + sizeof(morb_keypoint) * (size_t)maxk;
  // (the upload is ordered before these copies on the same stream, so the buffer can be reused for the way back)
  MORB_HIP_CHECK(hipMemcpyAsync(e->h_io1, e->d_out1, outBytes, hipMemcpyDeviceToHost, e->stream));   // count | monoIndex | keypoints | descriptors: one copy
  MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
  {   // only THIS call's flags decide its result: what an earlier batched call left unqueried stays for morb_extractor_status
    const int own = __atomic_exchange_n(e->h_status, 0, __ATOMIC_ACQ_REL) & ~stale;
    if (stale) __atomic_fetch_or(e->h_status, stale, __ATOMIC_ACQ_REL);
    if (own) { set_error("the extraction was flagged on the device (flags 0x%x)", own); *n = 0; return MORB_ERR_UNSUPPORTED; }
  }
  const int cnt = hcnt[0], mono = hcnt[1];
  *n = cnt;
  MORB_REQUIRE(cnt <= cap, MORB_ERR_CAPACITY, "keypoint buffer too small");
  if (cnt) { memcpy(kps, hkps, sizeof(morb_keypoint) * (size_t)cnt); memcpy(desc, hdesc, 32 * (size_t)cnt); }
  return mono;
}

int morb_extractor_pyramid_level(const morb_extractor* e, int img, int lvl, const uint8_t** d_ptr, int* width,
                                 int* height, int* stride) {
  MORB_REQUIRE(e && e->W > 0, MORB_ERR_INVALID, "no batch has been extracted yet");
  MORB_REQUIRE(lvl >= 0 && lvl < e->nlevels && img >= 0 && img < e->nimgLast, MORB_ERR_INVALID, "bad level/image");
  const LevelGeom& g = e->geom[lvl];
  if (d_ptr) *d_ptr = e->d_pyr + g.pyrOff + (size_t)img * g.pyrImg + (size_t)EDGE * g.pstride + EDGE;
  if (width) *width = g.w;
  if (height) *height = g.h;
  if (stride) *stride = g.pstride;
  return MORB_OK;
}

int morb_extractor_pyramid_level_host(const morb_extractor* e, int img, int lvl, uint8_t* out) {
  const uint8_t* p; int w, h, s;
  int rc = morb_extractor_pyramid_level(e, img, lvl, &p, &w, &h, &s);
  if (rc != MORB_OK) return rc;
  MORB_HIP_CHECK(hipSetDevice(e->device));
  MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
  MORB_HIP_CHECK(hipMemcpy2D(out, w + 2 * EDGE, p - (size_t)EDGE * s - EDGE, s, w + 2 * EDGE, h + 2 * EDGE, hipMemcpyDeviceToHost));
  return MORB_OK;
}

int morb_extractor_blurred_level_host(const morb_extractor* e, int img, int lvl, uint8_t* out) {
  MORB_REQUIRE(e && e->W > 0, MORB_ERR_INVALID, "no batch has been extracted yet");
  MORB_REQUIRE(lvl >= 0 && lvl < e->nlevels && img >= 0 && img < e->nimgLast, MORB_ERR_INVALID, "bad level/image");
  const LevelGeom& g = e->geom[lvl];
  MORB_HIP_CHECK(hipSetDevice(e->device));
  MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
  MORB_HIP_CHECK(hipMemcpy2D(out, g.w, e->d_blur + g.blurOff + (size_t)img * g.blurImg, g.bstride, g.w, g.h, hipMemcpyDeviceToHost));
  return MORB_OK;
}

int morb_extractor_level_candidates_host(const morb_extractor* e, int img, int lvl, morb_keypoint* out, int cap, int* n) {
  MORB_REQUIRE(e && e->W > 0 && n, MORB_ERR_INVALID, "no batch has been extracted yet");
  MORB_REQUIRE(lvl >= 0 && lvl < e->nlevels && img >= 0 && img < e->nimgLast, MORB_ERR_INVALID, "bad level/image");
  const LevelGeom& g = e->geom[lvl];
  const int ncell = g.nCols * g.nRows;
  MORB_HIP_CHECK(hipSetDevice(e->device));
  MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
  std::vector<int> cnt(ncell);
  std::vector<uint32_t> keys((size_t)ncell * e->cellCap);
  MORB_HIP_CHECK(hipMemcpy(cnt.data(), e->d_candCnt + (size_t)img * e->totalCells + g.cellBase, sizeof(int) * ncell, hipMemcpyDeviceToHost));
  MORB_HIP_CHECK(hipMemcpy(keys.data(), e->d_cand + ((size_t)img * e->totalCells + g.cellBase) * e->cellCap,
                           sizeof(uint32_t) * keys.size(), hipMemcpyDeviceToHost));
  int m = 0;
  for (int c = 0; c < ncell; ++c)
    for (int i = 0; i < cnt[c]; ++i, ++m) {
      if (out && m < cap) {
        const uint32_t k = keys[(size_t)c * e->cellCap + i];
        out[m] = morb_keypoint{(float)morbqt::key_x(k), (float)morbqt::key_y(k), 7.f, -1.f, (float)morbqt::key_r(k), 0, -1};
      }
    }
  *n = m;
  return MORB_OK;
}

int morb_extractor_level_keypoints_host(const morb_extractor* e, int img, int lvl, morb_keypoint* out, int cap, int* n) {
  MORB_REQUIRE(e && e->W > 0 && n, MORB_ERR_INVALID, "no batch has been extracted yet");
  MORB_REQUIRE(lvl >= 0 && lvl < e->nlevels && img >= 0 && img < e->nimgLast, MORB_ERR_INVALID, "bad level/image");
  const LevelGeom& g = e->geom[lvl];
  MORB_HIP_CHECK(hipSetDevice(e->device));
  MORB_HIP_CHECK(hipStreamSynchronize(e->stream));
  int cnt = 0;
  MORB_HIP_CHECK(hipMemcpy(&cnt, e->d_selCnt + img * e->nlevels + lvl, sizeof(int), hipMemcpyDeviceToHost));
  std::vector<uint32_t> keys(std::max(cnt, 1));
  MORB_HIP_CHECK(hipMemcpy(keys.data(), e->d_sel + (size_t)img * e->selPerImg + g.selBase, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt && i < cap && out; ++i) {
    const uint32_t k = keys[i];
    out[i] = morb_keypoint{(float)(morbqt::key_x(k) + MINB), (float)(morbqt::key_y(k) + MINB), g.kpSize, -1.f,
                           (float)morbqt::key_r(k), lvl, -1};
  }
  *n = cnt;
  return MORB_OK;
}

}  // extern "C"
