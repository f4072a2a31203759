// Order-exact restatement of ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:540-738,
// ExtractorNode::DivideNode :475-523, compareNodes :525-538) for one wave64 per (image, level).
//
// Design (MI355X): the reference keeps a std::list<ExtractorNode> whose nodes own std::vector<KeyPoint>;
// here a node is a rectangle + a contiguous segment [begin, begin+count) of one packed key array, the list
// is an array of node ids that grows towards index 0 (push_front = --head, erase = tombstone, compaction
// between sweeps), and DivideNode is a stable in-segment 4-way partition done by the whole wave with
// ballots + popcounts.  The control flow runs on wave-uniform values; only lane 0 stores.  The result is
// identical to the reference including list order, the "largest first" phase and the behaviour of
// libstdc++'s std::sort on equivalent elements (introsort emulated below).
//
// The same source compiles for the host (tests/ build it with g++ to check it against the oracle's
// std::list/std::sort restatement on random inputs); the product only uses the device instantiation.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define QT_DEVICE 1
#else
#define QT_DEVICE 0
#endif
#if defined(__HIPCC__)
#define QT_HD __host__ __device__ __forceinline__
#else
#define QT_HD inline
#endif

#if defined(MORB_FAST_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_fastPhase[32];   // phase clocks of tools/fast_phases.py (timing build only)
#endif
#if defined(MORB_FAST_TIMING) && QT_DEVICE
#define QT_T0() unsigned long long q0_ = wall_clock64()
#define QT_MARK(k) do { if (threadIdx.x == 0 && blockIdx.y == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_fastPhase[k], now_ - q0_); q0_ = now_; } } while (0)
#else
#define QT_T0()
#define QT_MARK(k)
#endif
namespace morbqt {

// key = x | y << 12 | response << 24, x/y relative to (minBorderX, minBorderY); position in the key array =
// rank in vToDistributeKeys.
QT_HD int key_x(uint32_t k) { return (int)(k & 0xFFFu); }
QT_HD int key_y(uint32_t k) { return (int)((k >> 12) & 0xFFFu); }
QT_HD int key_r(uint32_t k) { return (int)(k >> 24); }
QT_HD uint32_t make_key(int x, int y, int r) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)r << 24); }

struct Node {
  int16_t x0, y0, x1, y1;  // UL=(x0,y0) UR=(x1,y0) BL=(x0,y1) BR=(x1,y1)
  uint32_t begin, count;
  uint16_t lit;     // position in the list array
  uint16_t noMore;  // bNoMore
};

struct Work {
  uint32_t* keys;  // [kcap]
  uint32_t* tmp;   // [kcap]
  Node* nodes;     // [nodeCap]
  uint16_t* freeIds;  // [nodeCap] stack of free node ids
  uint16_t* list;     // [listCap] node id or 0xFFFF (erased)
  uint64_t* vA;       // [nodeCap] vSizeAndPointerToNode: count << 32 | x0 << 16 | id
  uint64_t* vB;       // [nodeCap] vPrevSizeAndPointerToNode
  // batch split (device): the nodes one sweep divides, in processing order, with their child key counts and ranks
  uint16_t* order;    // [nodeCap] node ids
  uint64_t* bcnt;     // [nodeCap] keys per child, 4 x 16 bit (n1 | n2 << 16 | n3 << 32 | n4 << 48), each SATURATED at 0xFFFF: exact for nodes of
                      //           <= 65535 keys; the few larger nodes of a noise-like level are counted again when they are partitioned (qt_pack4)
  uint32_t* brank;    // [nodeCap] children pushed before this node | children with > 1 key before it << 16
  int nodeCap, listCap;
  bool hostFastForward = false;   // (host build with QT_HOST_FAST_FORWARD only: qt_distribute starts from qt_fast_forward_host's state)
};

// Live nodes never exceed N + 3 (a full sweep only runs when it cannot overshoot N, the "largest first" phase stops at
// N); the device splits a whole sweep at once and takes the children's ids before the parents' ids are released, so
// it needs up to one id per parent on top of that.
QT_HD int qt_node_cap(int N, int nIni) { int a = N + 3, b = 4 * nIni; return 2 * (a > b ? a : b) + 16; }
// Between two compactions the list holds the live nodes at the last compaction (<= N + 3) plus every child pushed since
// (<= N + 3 live afterwards + one tombstone per parent <= N + 3): 3 (N + 3) < 2 nodeCap.
QT_HD int qt_list_cap(int nodeCap) { return 2 * nodeCap; }

#define QT_MAX_WAVES 4   // waves (= levels) of one k_distribute workgroup in the packed form; nothing below synchronises across waves
#ifndef QT_TEAM_WAVES
#define QT_TEAM_WAVES 16  // waves of a workgroup that works ONE level as a team (calls with few images: latency).  Round 6: 4 -> 16, a 1920 x 1080 level 0
#endif                    // spends its time in dependent LDS round trips per node, and the nodes of a sweep are independent (profiles/r06/k_distribute_phases.txt)
#define QT_TEAM_SH (16 + 4 * QT_TEAM_WAVES)   // ints of Team::sh
#if QT_DEVICE
#define QT_LANE ((int)(threadIdx.x & 63))
#define QT_LANE0 (QT_LANE == 0)
#define QT_SYNC()                                          \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)
#else
#define QT_LANE 0
#define QT_LANE0 true
#define QT_SYNC() \
  do {            \
  } while (0)
#endif

// A level may be worked by a TEAM of QT_TEAM_WAVES waves (a whole workgroup; used for the big levels of a few images: one wave per
// (image, level) leaves a 1920 x 1080 level-0 quadtree at 0.7 ms).  The splits of one sweep are independent of each other, so the
// team divides a sweep's nodes among its waves — every wave still partitions the nodes it owns with the wave-level code below —
// and only the order-dependent steps (list compaction, ranks of the children, the std::sort emulation) stay with wave 0.  Between the
// phases the waves meet at workgroup barriers and exchange the wave-uniform state through a few LDS words.
struct Team {
  int nw, tw;   // waves in the team (1: the level has a single wave, no workgroup barrier is ever executed), this wave's index
  int* sh;      // [QT_TEAM_SH] shared words (LDS): 0-11 state exchanged between phases, 16.. per-wave slice counts of the team partition (4 per wave)
};
#if QT_DEVICE
#define QT_TEAM_SYNC(t) do { if ((t).nw > 1) __syncthreads(); else QT_SYNC(); } while (0)
#else
#define QT_TEAM_SYNC(t) do { } while (0)
#endif

// ---- wave-cooperative primitives ---------------------------------------------------------------------

// Stable partition of keys[begin, begin+count) into G (<= 4) groups given by cls(key) in group order;
// returns the group sizes in cnt[].  All lanes call this convergently.
// (device) knownCnt: cnt[] already holds the group sizes (the batch split counted them in its first phase): the counting pass is skipped.
template <typename Cls>
QT_HD void qt_partition(uint32_t* keys, uint32_t* tmp, uint32_t begin, uint32_t count, Cls cls, uint32_t cnt[4], bool knownCnt = false) {
#if QT_DEVICE
  const int lane = QT_LANE;
  uint32_t c0 = cnt[0], c1 = cnt[1], c2 = cnt[2], c3 = cnt[3];
  if (!knownCnt) {
    c0 = c1 = c2 = c3 = 0;
    for (uint32_t i = 0; i < count; i += 64) {
      int g = -1;
      if (i + lane < count) g = cls(keys[begin + i + lane]);
      c0 += __popcll(__ballot(g == 0));
      c1 += __popcll(__ballot(g == 1));
      c2 += __popcll(__ballot(g == 2));
      c3 += __popcll(__ballot(g == 3));
    }
    cnt[0] = c0; cnt[1] = c1; cnt[2] = c2; cnt[3] = c3;
  }
  uint32_t b0 = 0, b1 = c0, b2 = c0 + c1, b3 = c0 + c1 + c2;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (uint32_t i = 0; i < count; i += 64) {
    int g = -1;
    uint32_t k = 0;
    if (i + lane < count) { k = keys[begin + i + lane]; g = cls(k); }
    uint64_t m0 = __ballot(g == 0), m1 = __ballot(g == 1), m2 = __ballot(g == 2), m3 = __ballot(g == 3);
    uint32_t dst = 0;
    if (g == 0) dst = b0 + __popcll(m0 & lt);
    if (g == 1) dst = b1 + __popcll(m1 & lt);
    if (g == 2) dst = b2 + __popcll(m2 & lt);
    if (g == 3) dst = b3 + __popcll(m3 & lt);
    if (g >= 0) tmp[begin + dst] = k;
    b0 += __popcll(m0); b1 += __popcll(m1); b2 += __popcll(m2); b3 += __popcll(m3);
  }
  QT_SYNC();
  for (uint32_t i = lane; i < count; i += 64) keys[begin + i] = tmp[begin + i];
  QT_SYNC();
#else
  uint32_t c[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < count; ++i) c[cls(keys[begin + i])]++;
  uint32_t b[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
  for (uint32_t i = 0; i < count; ++i) { uint32_t k = keys[begin + i]; tmp[begin + b[cls(k)]++] = k; }
  for (uint32_t i = 0; i < count; ++i) keys[begin + i] = tmp[begin + i];
  for (int g = 0; g < 4; ++g) cnt[g] = c[g];
#endif
}

#if QT_DEVICE
constexpr uint32_t QT_HUGE = 2048;   // nodes with more keys are partitioned by the whole team together (the first sweeps of a large level)
// Group sizes of keys[begin, begin + count) by the whole team (all waves call this convergently): wave tw counts the tw-th slice;
// tm.sh[16 + 4 w + g] = keys of group g in wave w's slice, cnt[] = the totals.  Ends with the slice counts still valid in tm.sh.
template <typename Cls>
__device__ inline void qt_count_team(const uint32_t* keys, uint32_t begin, uint32_t count, Cls cls, uint32_t cnt[4], const Team& tm, uint32_t* lo_, uint32_t* hi_) {
  const int lane = QT_LANE;
  const uint32_t per = ((count + tm.nw * 64 - 1) / (tm.nw * 64)) * 64;
  const uint32_t lo = per * tm.tw < count ? per * tm.tw : count, hi = lo + per < count ? lo + per : count;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (uint32_t i = lo; i < hi; i += 64) {
    int g = -1;
    if (i + lane < hi) g = cls(keys[begin + i + lane]);
    c0 += __popcll(__ballot(g == 0)); c1 += __popcll(__ballot(g == 1));
    c2 += __popcll(__ballot(g == 2)); c3 += __popcll(__ballot(g == 3));
  }
  if (lane == 0) { int* o = tm.sh + 16 + 4 * tm.tw; o[0] = (int)c0; o[1] = (int)c1; o[2] = (int)c2; o[3] = (int)c3; }
  __syncthreads();
  for (int g = 0; g < 4; ++g) { uint32_t t = 0; for (int wv = 0; wv < tm.nw; ++wv) t += (uint32_t)tm.sh[16 + 4 * wv + g]; cnt[g] = t; }
  *lo_ = lo; *hi_ = hi;
}
// Stable partition by the whole team; same result as qt_partition.
template <typename Cls>
__device__ inline void qt_partition_team(uint32_t* keys, uint32_t* tmp, uint32_t begin, uint32_t count, Cls cls, uint32_t cnt[4], const Team& tm) {
  const int lane = QT_LANE;
  uint32_t lo, hi;
  qt_count_team(keys, begin, count, cls, cnt, tm, &lo, &hi);
  // this wave's first slot in every group = the groups before it + the earlier slices' keys of the group
  uint32_t b[4];
  uint32_t gb = 0;
  for (int g = 0; g < 4; ++g) {
    uint32_t before = 0;
    for (int wv = 0; wv < tm.tw; ++wv) before += (uint32_t)tm.sh[16 + 4 * wv + g];
    b[g] = gb + before;
    gb += cnt[g];
  }
  uint32_t b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (uint32_t i = lo; i < hi; i += 64) {
    int g = -1;
    uint32_t k = 0;
    if (i + lane < hi) { k = keys[begin + i + lane]; g = cls(k); }
    const uint64_t m0 = __ballot(g == 0), m1 = __ballot(g == 1), m2 = __ballot(g == 2), m3 = __ballot(g == 3);
    uint32_t dst = 0;
    if (g == 0) dst = b0 + __popcll(m0 & lt);
    if (g == 1) dst = b1 + __popcll(m1 & lt);
    if (g == 2) dst = b2 + __popcll(m2 & lt);
    if (g == 3) dst = b3 + __popcll(m3 & lt);
    if (g >= 0) tmp[begin + dst] = k;
    b0 += __popcll(m0); b1 += __popcll(m1); b2 += __popcll(m2); b3 += __popcll(m3);
  }
  __syncthreads();
  for (uint32_t i = lo + lane; i < hi; i += 64) keys[begin + i] = tmp[begin + i];
  __syncthreads();
}
#endif

// First key with the maximum response in keys[begin, begin+count)  (ORBextractor.cc:719-735).
QT_HD uint32_t qt_best_key(const uint32_t* keys, uint32_t begin, uint32_t count) {
#if QT_DEVICE
  const int lane = QT_LANE;
  // maximise (response, -position): pack response << 32 | (0xFFFFFFFF - position)
  uint64_t best = 0;
  for (uint32_t i = lane; i < count; i += 64) {
    uint64_t v = ((uint64_t)key_r(keys[begin + i]) << 32) | (uint64_t)(0xFFFFFFFFu - i);
    best = v > best ? v : best;
  }
  best = ~morbwave::min_u64(~best);   // wave maximum on DPP (all lanes active)
  uint32_t pos = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu);
  return keys[begin + pos];
#else
  uint32_t bk = keys[begin];
  for (uint32_t i = 1; i < count; ++i)
    if (key_r(keys[begin + i]) > key_r(bk)) bk = keys[begin + i];
  return bk;
#endif
}

// ---- libstdc++ std::sort(first, last, compareNodes) emulation on packed entries -------------------------
// entry = count << 32 | x0 << 16 | id ; compareNodes(e1, e2) == ((e1 >> 16) < (e2 >> 16)).
QT_HD bool qt_less(uint64_t a, uint64_t b) { return (a >> 16) < (b >> 16); }
QT_HD void qt_swap(uint64_t* v, int a, int b) { uint64_t t = v[a]; v[a] = v[b]; v[b] = t; }

QT_HD void qt_adjust_heap(uint64_t* v, int first, int holeIndex, int len, uint64_t value) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (qt_less(v[first + secondChild], v[first + (secondChild - 1)])) secondChild--;
    v[first + holeIndex] = v[first + secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    v[first + holeIndex] = v[first + (secondChild - 1)];
    holeIndex = secondChild - 1;
  }
  int parent = (holeIndex - 1) / 2;  // __push_heap
  while (holeIndex > topIndex && qt_less(v[first + parent], value)) {
    v[first + holeIndex] = v[first + parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  v[first + holeIndex] = value;
}

QT_HD void qt_heapsort(uint64_t* v, int first, int last) {  // std::__partial_sort(first, last, last)
  const int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      uint64_t value = v[first + parent];
      qt_adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {  // __sort_heap
    --last;
    uint64_t value = v[last];
    v[last] = v[first];
    qt_adjust_heap(v, first, 0, last - first, value);
  }
}

QT_HD void qt_unguarded_linear_insert(uint64_t* v, int last) {
  uint64_t val = v[last];
  int next = last - 1;
  while (qt_less(val, v[next])) { v[last] = v[next]; last = next; --next; }
  v[last] = val;
}

QT_HD void qt_insertion_sort(uint64_t* v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (qt_less(v[i], v[first])) {
      uint64_t val = v[i];
      for (int k = i; k > first; --k) v[k] = v[k - 1];  // move_backward(first, i, i+1)
      v[first] = val;
    } else
      qt_unguarded_linear_insert(v, i);
  }
}

// std::__introsort_loop(first, last, 2 * lg(n)): leaves every element within a <= 16-element partition of its
// sorted position.  Serial; call from ONE lane (or the host).
// st: 3 * 64 ints of stack storage (LDS on the device: a local array would live in scratch memory, i.e. a
// global-memory round trip per push / pop).
QT_HD void qt_introsort_loop(uint64_t* v, int n, int* st) {
  if (n <= 0) return;
  int* stF = st; int* stL = st + 64; int* stD = st + 128;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;  // std::__lg(n)
  // __introsort_loop with an explicit stack of (first, last, depth) for the recursive right halves
  int sp = 0;
  stF[0] = 0; stL[0] = n; stD[0] = lg * 2; sp = 1;
  while (sp > 0) {
    --sp;
    int first = stF[sp], last = stL[sp], depth = stD[sp];
    while (last - first > 16) {
      if (depth == 0) { qt_heapsort(v, first, last); break; }
      --depth;
      // __unguarded_partition_pivot
      int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        int a = first + 1, b = mid, c = last - 1;
        if (qt_less(v[a], v[b])) {
          if (qt_less(v[b], v[c])) qt_swap(v, first, b);
          else if (qt_less(v[a], v[c])) qt_swap(v, first, c);
          else qt_swap(v, first, a);
        } else if (qt_less(v[a], v[c])) qt_swap(v, first, a);
        else if (qt_less(v[b], v[c])) qt_swap(v, first, c);
        else qt_swap(v, first, b);
      }
      int lo = first + 1, hi = last;
      while (true) {  // __unguarded_partition(first+1, last, pivot = first)
        while (qt_less(v[lo], v[first])) ++lo;
        --hi;
        while (qt_less(v[first], v[hi])) --hi;
        if (!(lo < hi)) break;
        qt_swap(v, lo, hi);
        ++lo;
      }
      const int cut = lo;
      // recurse on [cut, last) first (the reference recursion order), then loop on [first, cut): the two
      // ranges are disjoint so the order of processing does not change the result; push the right half.
      stF[sp] = cut; stL[sp] = last; stD[sp] = depth; ++sp;
      last = cut;
    }
  }
}

// Serial; call from ONE lane (or the host).
QT_HD void qt_std_sort(uint64_t* v, int n) {
  if (n <= 0) return;
  int st[3 * 64];
  qt_introsort_loop(v, n, st);
  // __final_insertion_sort
  if (n > 16) {
    qt_insertion_sort(v, 0, 16);
    for (int i = 16; i != n; ++i) qt_unguarded_linear_insert(v, i);
  } else
    qt_insertion_sort(v, 0, n);
}

#if QT_DEVICE
// Whole wave, convergent: v[0..n) in LDS ends up exactly as std::sort(v, v + n, compareNodes) leaves it.
// std::__final_insertion_sort is a STABLE insertion sort (both __insertion_sort and __unguarded_linear_insert stop
// at the first element that is not greater), so after the serial introsort loop the result is the stable sort of
// the array by key: every lane ranks its entries (keys smaller + equal keys to the left) instead of one lane
// walking LDS element by element (that walk was ~25 % of k_distribute).
// std::__unguarded_partition_pivot(first, last) by the whole wave; returns the cut.  The serial loop swaps the k-th
// entry from the left that is not less than the pivot (list L, ascending positions) with the k-th entry from the right
// that is not greater (list R, descending positions, closed by the pivot slot `first` itself) for as long as the left
// position is below the right one -- so the lanes build both lists with ballots, perform all those swaps at once, and the
// cut is the position where the serial scan stops: min(L[K], R[K-1]) with K the number of swaps.
// posL / posR: scratch for n and n + 1 positions.
__device__ inline int qt_partition_pivot_wave(uint64_t* v, int first, int last, uint16_t* posL, uint16_t* posR) {
  const int lane = QT_LANE;
  {  // __move_median_to_first(first, first + 1, mid, last - 1)
    const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
    const uint64_t va = v[a], vb = v[b], vc = v[c];
    int m;
    if (qt_less(va, vb)) m = qt_less(vb, vc) ? b : (qt_less(va, vc) ? c : a);
    else m = qt_less(va, vc) ? a : (qt_less(vb, vc) ? c : b);
    QT_SYNC();
    if (lane == 0) qt_swap(v, first, m);
    QT_SYNC();
  }
  const uint64_t pivot = v[first];
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int nL = 0, nR = 0;
  for (int p0 = first + 1; p0 < last; p0 += 64) {
    const int p = p0 + lane;
    const bool in = p < last;
    const uint64_t e = in ? v[p] : 0;
    nL += __popcll(__ballot(in && !qt_less(e, pivot)));
    nR += __popcll(__ballot(in && !qt_less(pivot, e)));
  }
  int bl = 0, br = 0;   // entries of L / R at lower positions
  for (int p0 = first + 1; p0 < last; p0 += 64) {
    const int p = p0 + lane;
    const bool in = p < last;
    const uint64_t e = in ? v[p] : 0;
    const bool ge = in && !qt_less(e, pivot), le = in && !qt_less(pivot, e);
    const uint64_t mg = __ballot(ge), ml = __ballot(le);
    if (ge) posL[bl + __popcll(mg & below)] = (uint16_t)p;
    if (le) posR[nR - 1 - (br + __popcll(ml & below))] = (uint16_t)p;
    bl += __popcll(mg); br += __popcll(ml);
  }
  if (lane == 0) posR[nR] = (uint16_t)first;
  QT_SYNC();
  const int nPair = nL < nR + 1 ? nL : nR + 1;
  int K = 0;
  for (int k0 = 0; k0 < nPair; k0 += 64) {
    const int k = k0 + lane;
    bool act = false;
    int pl = 0, pr = 0;
    if (k < nPair) { pl = posL[k]; pr = posR[k]; act = pl < pr; }
    if (act) qt_swap(v, pl, pr);
    const uint64_t ma = __ballot(act);
    K += __popcll(ma);
    if (ma != ~0ull) break;   // the swapping pairs are a prefix of the pairing
  }
  int cut = last;
  if (K < nL) cut = posL[K];
  if (K > 0) { const int r = posR[K - 1]; cut = r < cut ? r : cut; }
  QT_SYNC();
  return cut;
}

// Stable rank of v[i] among the 33 entries around it (everything further left precedes it, everything further right follows: see below).  A fixed
// trip count, so the 33 LDS reads are in flight together instead of one round trip after the other.
__device__ __forceinline__ int qt_window_rank(const uint64_t* v, int n, int i) {
  const uint64_t k = v[i] >> 16;
  int rank = i > 16 ? i - 16 : 0;
#pragma unroll
  for (int o = -16; o <= 16; ++o) {
    const int j = i + o;
    const bool in = j >= 0 && j < n;
    const uint64_t kj = v[in ? j : i] >> 16;
    rank += (in && (kj < k || (kj == k && o < 0))) ? 1 : 0;
  }
  return rank;
}
// st: 3 * 64 ints of LDS, this wave's explicit stack
__device__ inline void qt_std_sort_wave(uint64_t* v, uint64_t* tmp, int n, uint16_t* posL, uint16_t* posR, int* st) {
  if (n > 16) {  // std::__introsort_loop with the recursion on an explicit (wave-uniform) stack, see qt_introsort_loop
    int* stF = st; int* stL = st + 64; int* stD = st + 128;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    int sp = 1;
    if (QT_LANE0) { stF[0] = 0; stL[0] = n; stD[0] = lg * 2; }
    QT_SYNC();
    while (sp > 0) {
      --sp;
      int first = stF[sp], last = stL[sp], depth = stD[sp];
      while (last - first > 16) {
        if (depth == 0) {
          QT_SYNC();
          if (QT_LANE0) qt_heapsort(v, first, last);
          QT_SYNC();
          break;
        }
        --depth;
        const int cut = qt_partition_pivot_wave(v, first, last, posL, posR);
        if (QT_LANE0) { stF[sp] = cut; stL[sp] = last; stD[sp] = depth; }
        ++sp;
        last = cut;
      }
      QT_SYNC();
    }
  }
  QT_SYNC();
  // std::__final_insertion_sort = a stable sort of what the introsort loop left: runs of <= 16 elements, every element of an earlier
  // run <= every element of a later one.  An element therefore ends up inside its own run, i.e. fewer than 16 places from where it
  // is: everything more than 16 places to its left precedes it, everything more than 16 to its right follows it, and its rank
  // only needs the 33-element window around it (the full n x n count cost 180 us of the 307 us a level-0 sort takes at 4000 features).
  for (int i = QT_LANE; i < n; i += 64) tmp[qt_window_rank(v, n, i)] = v[i];
  QT_SYNC();
  for (int i = QT_LANE; i < n; i += 64) v[i] = tmp[i];
  QT_SYNC();
}
#endif

// ---- the distribution -----------------------------------------------------------------------------

struct State {
  int head;      // list occupies [head, listCap)
  int size;      // live nodes
  int nFree;     // free-id stack height
  int nA;        // entries in vA
};

QT_HD int qt_alloc(Work& w, State& s) { return w.freeIds[--s.nFree]; }

// push_front of a child; returns nothing.  Wave-uniform arguments; lane 0 stores.
QT_HD void qt_push_child(Work& w, State& s, int x0, int y0, int x1, int y1, uint32_t begin, uint32_t count,
                         int* nToExpand) {
  if (count == 0) return;
  const int id = qt_alloc(w, s);
  --s.head;
  ++s.size;
  if (QT_LANE0) {
    Node nd;
    nd.x0 = (int16_t)x0; nd.y0 = (int16_t)y0; nd.x1 = (int16_t)x1; nd.y1 = (int16_t)y1;
    nd.begin = begin; nd.count = count; nd.lit = (uint16_t)s.head; nd.noMore = (count == 1) ? 1 : 0;
    w.nodes[id] = nd;
    w.list[s.head] = (uint16_t)id;
    if (count > 1) w.vA[s.nA] = ((uint64_t)count << 32) | ((uint64_t)(uint16_t)x0 << 16) | (uint64_t)id;
  }
  if (count > 1) { ++s.nA; if (nToExpand) ++*nToExpand; }
}

// DivideNode + the four push_front blocks + erase of the parent (ORBextractor.cc:609-650 / :671-708).
#if QT_DEVICE
// Device form: few dependent LDS round trips per split.  The four free ids are fetched while the keys are being
// partitioned, nodes of <= 64 keys are partitioned in registers in one pass (one load, four ballots, one store),
// and lanes 0..3 write the four children (node record, list slot, vA entry) in one step.
QT_HD void qt_split(Work& w, State& s, int id, int* nToExpand) {
  QT_SYNC();
  const int lane = QT_LANE;
  const Node nd = w.nodes[id];
  int freeReg = 0;
  if (lane < 4 && s.nFree - 1 - lane >= 0) freeReg = w.freeIds[s.nFree - 1 - lane];
  const int halfX = (nd.x1 - nd.x0 + 1) >> 1;  // ceil((UR.x-UL.x)/2.f) for non-negative ints
  const int halfY = (nd.y1 - nd.y0 + 1) >> 1;
  const int mx = nd.x0 + halfX, my = nd.y0 + halfY;
  uint32_t cnt[4];
  if (nd.count <= 64) {
    int g = -1;
    uint32_t k = 0;
    if ((uint32_t)lane < nd.count) { k = w.keys[nd.begin + lane]; g = (key_x(k) < mx ? 0 : 1) + (key_y(k) < my ? 0 : 2); }
    const uint64_t m0 = __ballot(g == 0), m1 = __ballot(g == 1), m2 = __ballot(g == 2), m3 = __ballot(g == 3);
    cnt[0] = __popcll(m0); cnt[1] = __popcll(m1); cnt[2] = __popcll(m2); cnt[3] = __popcll(m3);
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t dst = 0;
    if (g == 0) dst = __popcll(m0 & lt);
    if (g == 1) dst = cnt[0] + __popcll(m1 & lt);
    if (g == 2) dst = cnt[0] + cnt[1] + __popcll(m2 & lt);
    if (g == 3) dst = cnt[0] + cnt[1] + cnt[2] + __popcll(m3 & lt);
    if (g >= 0) w.keys[nd.begin + dst] = k;   // every lane's key is already in a register
  } else {
    qt_partition(w.keys, w.tmp, nd.begin, nd.count,
                 [mx, my](uint32_t k) -> int { return (key_x(k) < mx ? 0 : 1) + (key_y(k) < my ? 0 : 2); }, cnt);
  }
  // group order 0:n1 (x<mx,y<my) 1:n2 (x>=mx,y<my) 2:n3 (x<mx,y>=my) 3:n4; children with keys are pushed to the front
  // in that order; those with more than one key also enter vSizeAndPointerToNode
  const int has0 = cnt[0] > 0, has1 = cnt[1] > 0, has2 = cnt[2] > 0, has3 = cnt[3] > 0;
  const int ex0 = cnt[0] > 1, ex1 = cnt[1] > 1, ex2 = cnt[2] > 1, ex3 = cnt[3] > 1;
  const int nCh = has0 + has1 + has2 + has3, nEx = ex0 + ex1 + ex2 + ex3;
  if (lane < 4) {
    const int c = lane;
    const uint32_t myCnt = c == 0 ? cnt[0] : c == 1 ? cnt[1] : c == 2 ? cnt[2] : cnt[3];
    const int r = c == 0 ? 0 : c == 1 ? has0 : c == 2 ? has0 + has1 : has0 + has1 + has2;          // push rank
    const int re = c == 0 ? 0 : c == 1 ? ex0 : c == 2 ? ex0 + ex1 : ex0 + ex1 + ex2;               // rank in vA
    const uint32_t b = nd.begin + (c == 0 ? 0u : c == 1 ? cnt[0] : c == 2 ? cnt[0] + cnt[1] : cnt[0] + cnt[1] + cnt[2]);
    const int cid = __shfl(freeReg, r, 64);
    if (myCnt > 0) {
      Node ch;
      ch.x0 = (int16_t)((c & 1) ? mx : nd.x0); ch.x1 = (int16_t)((c & 1) ? nd.x1 : mx);
      ch.y0 = (int16_t)((c & 2) ? my : nd.y0); ch.y1 = (int16_t)((c & 2) ? nd.y1 : my);
      ch.begin = b; ch.count = myCnt; ch.lit = (uint16_t)(s.head - 1 - r); ch.noMore = (myCnt == 1) ? 1 : 0;
      w.nodes[cid] = ch;
      w.list[s.head - 1 - r] = (uint16_t)cid;
      if (myCnt > 1) w.vA[s.nA + re] = ((uint64_t)myCnt << 32) | ((uint64_t)(uint16_t)ch.x0 << 16) | (uint64_t)cid;
    }
  } else {
    (void)__shfl(freeReg, 0, 64);   // keep the shuffle convergent
  }
  s.head -= nCh;
  s.size += nCh - 1;
  s.nA += nEx;
  if (nToExpand) *nToExpand += nEx;
  s.nFree -= nCh;
  if (QT_LANE0) {
    w.list[nd.lit] = 0xFFFF;
    w.freeIds[s.nFree] = (uint16_t)id;   // the parent's id is released last
  }
  ++s.nFree;
  QT_SYNC();
}
#else
QT_HD void qt_split(Work& w, State& s, int id, int* nToExpand) {
  QT_SYNC();
  const Node nd = w.nodes[id];
  const int halfX = (nd.x1 - nd.x0 + 1) >> 1;  // ceil((UR.x-UL.x)/2.f) for non-negative ints
  const int halfY = (nd.y1 - nd.y0 + 1) >> 1;
  const int mx = nd.x0 + halfX, my = nd.y0 + halfY;
  uint32_t cnt[4];
  qt_partition(w.keys, w.tmp, nd.begin, nd.count,
               [mx, my](uint32_t k) -> int { return (key_x(k) < mx ? 0 : 1) + (key_y(k) < my ? 0 : 2); }, cnt);
  // group order 0:n1 (x<mx,y<my) 1:n2 (x>=mx,y<my) 2:n3 (x<mx,y>=my) 3:n4
  uint32_t b = nd.begin;
  // erase(parent) happens after the pushes in the reference; ids are not reused within a split because the
  // parent's id is released last.
  qt_push_child(w, s, nd.x0, nd.y0, mx, my, b, cnt[0], nToExpand); b += cnt[0];
  qt_push_child(w, s, mx, nd.y0, nd.x1, my, b, cnt[1], nToExpand); b += cnt[1];
  qt_push_child(w, s, nd.x0, my, mx, nd.y1, b, cnt[2], nToExpand); b += cnt[2];
  qt_push_child(w, s, mx, my, nd.x1, nd.y1, b, cnt[3], nToExpand);
  if (QT_LANE0) {
    w.list[nd.lit] = 0xFFFF;
    w.freeIds[s.nFree] = (uint16_t)id;
  }
  ++s.nFree;
  --s.size;
  QT_SYNC();
}
#endif

#if QT_DEVICE
// ---- one sweep at a time --------------------------------------------------------------------------------------
// The splits of one sweep are independent: a node's children only depend on its own keys, and where they go in the
// list / in vSizeAndPointerToNode only depends on how many children the nodes BEFORE it (in processing order)
// produce.  So: (1) count the keys per child for every node of the sweep, one node per lane (nodes with more than
// QT_SMALL keys by the whole wave), (2) prefix-sum children and expandable children in processing order — and, for the
// "largest first" phase, find the node whose split makes size >= N (the reference breaks right after it) —
// (3) partition the keys and write the children.  v1 did ~190 dependent splits per level-0 image (94 us).
constexpr uint32_t QT_SMALL = 64;
__device__ __forceinline__ int qt_class(uint32_t k, int mx, int my) { return (key_x(k) < mx ? 0 : 1) + (key_y(k) < my ? 0 : 2); }
__device__ __forceinline__ int qt_scan_incl(int v) {   // inclusive prefix sum over the wave (all lanes active): DPP, no LDS-crossbar round trips
  MORB_DPP_SCAN(v, 0, morbwave::op_add);
  return v;
}
// four child counts in 16 bits each, saturated: what the rank pass needs (is a child empty, does it hold more than one key) survives, and a
// node of <= 65535 keys cannot saturate.  Until round 5 the fields simply overflowed and a level of > 65535 candidates was refused.
__device__ __forceinline__ uint64_t qt_pack4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  auto sat = [](uint32_t c) -> uint64_t { return c < 0xFFFFu ? c : 0xFFFFu; };
  return sat(c0) | (sat(c1) << 16) | (sat(c2) << 32) | (sat(c3) << 48);
}
__device__ __forceinline__ void qt_unpack4(uint64_t c, uint32_t cnt[4]) {
  cnt[0] = (uint32_t)c & 0xFFFFu; cnt[1] = (uint32_t)(c >> 16) & 0xFFFFu; cnt[2] = (uint32_t)(c >> 32) & 0xFFFFu; cnt[3] = (uint32_t)(c >> 48) & 0xFFFFu;
}
__device__ __forceinline__ void qt_write_children(Work& w, const Node& nd, int mx, int my, const uint32_t (&cnt)[4], uint32_t rk, int oldHead,
                                                  int nA0, int nFree0, int c) {
  // child c (0..3) of nd: rank among the node's non-empty children = number of non-empty children before it (cnt: EXACT key counts)
  int r = 0, re = 0;
  uint32_t b = nd.begin;
#pragma unroll
  for (int q = 0; q < 4; ++q) if (q < c) { const uint32_t cq = cnt[q]; r += cq > 0; re += cq > 1; b += cq; }
  const uint32_t myCnt = c == 0 ? cnt[0] : c == 1 ? cnt[1] : c == 2 ? cnt[2] : cnt[3];
  if (myCnt == 0) return;
  const int gr = (int)(rk & 0xFFFFu) + r, ge = (int)(rk >> 16) + re;
  const int cid = w.freeIds[nFree0 - 1 - gr];
  Node ch;
  ch.x0 = (int16_t)((c & 1) ? mx : nd.x0); ch.x1 = (int16_t)((c & 1) ? nd.x1 : mx);
  ch.y0 = (int16_t)((c & 2) ? my : nd.y0); ch.y1 = (int16_t)((c & 2) ? nd.y1 : my);
  ch.begin = b; ch.count = myCnt; ch.lit = (uint16_t)(oldHead - 1 - gr); ch.noMore = (myCnt == 1) ? 1 : 0;
  w.nodes[cid] = ch;
  w.list[oldHead - 1 - gr] = (uint16_t)cid;
  if (myCnt > 1) w.vA[nA0 + ge] = ((uint64_t)myCnt << 32) | ((uint64_t)(uint16_t)ch.x0 << 16) | (uint64_t)cid;
}

// Splits w.order[0..m) in that order; cutoffN >= 0: stop after the split that makes s.size >= cutoffN.
__device__ inline void qt_split_batch(Work& w, State& s, int mAll, int cutoffN, int* nToExpand, const Team& tm) {
  const int lane = QT_LANE;
  // "largest first" (cutoffN >= 0) stops after the split that makes size >= N, and a split adds at most three nodes: the first
  // ceil((N - size) / 3) nodes are needed at least, and — the largest nodes rarely leave a child empty — almost always at most a few more.  Only
  // those are counted (phase 1 was 7 of the phase's 27 us at 1920 x 1080 with all ~480 candidates counted); should the cut lie beyond, the
  // batch is counted again in full.
  int m = mAll;
  if (cutoffN >= 0) { const int need = (cutoffN - s.size + 2) / 3 + 8; m = need < mAll ? need : mAll; }
  const int estart = tm.tw * 64, estep = tm.nw * 64;
  int runCh = 0, runEx = 0, runSize = s.size, mProc = m;
  const bool teamHuge = tm.nw > 1;
  // a lane walks a node's keys one LDS round trip after the other: fewest instructions (the packed form: many waves per SIMD), longest chain.  A team
  // (latency) gives every node of more than a few keys to a whole wave: with only the needed nodes counted (below) 4.9 + 10.4 us against 6.4 + 16.5
  const uint32_t smallMax = tm.nw > 1 ? 6u : QT_SMALL;
  QT_T0();
  for (;;) {
  // the team's waves take the sweep's nodes interleaved (node e -> wave e % nw): the first sweeps have a handful of huge nodes, one wave each
  QT_TEAM_SYNC(tm);
  // (1) keys per child.  A team first counts the huge nodes together, slice by slice (every wave walks the sweep's nodes: the choice is uniform)
  if (teamHuge) {
    for (int e0 = 0; e0 < m; e0 += 64) {
      uint32_t cntN = 0;
      if (e0 + lane < m) cntN = w.nodes[w.order[e0 + lane]].count;
      uint64_t hm = __ballot(cntN > QT_HUGE);
      while (hm) {
        const int bl = __ffsll((unsigned long long)hm) - 1;
        hm &= hm - 1;
        const Node nb = w.nodes[w.order[e0 + bl]];
        const int mx = nb.x0 + ((nb.x1 - nb.x0 + 1) >> 1), my = nb.y0 + ((nb.y1 - nb.y0 + 1) >> 1);
        uint32_t cnt[4], lo, hi;
        qt_count_team(w.keys, nb.begin, nb.count, [mx, my](uint32_t k) -> int { return qt_class(k, mx, my); }, cnt, tm, &lo, &hi);
        if (tm.tw == 0 && lane == 0) w.bcnt[e0 + bl] = qt_pack4(cnt[0], cnt[1], cnt[2], cnt[3]);
        __syncthreads();   // (the slice counts in tm.sh are rewritten by the next huge node)
      }
    }
  }
  for (int q0 = 0; q0 * tm.nw < m; q0 += 64) {
    const int e = (q0 + lane) * tm.nw + tm.tw;
    Node nd = {};
    if (e < m) nd = w.nodes[w.order[e]];
    const bool valid = e < m && !(teamHuge && nd.count > QT_HUGE);
    const bool small = valid && nd.count <= smallMax;
    if (small) {
      const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
      uint64_t c = 0;
      for (uint32_t k = 0; k < nd.count; ++k) c += 1ull << (16 * qt_class(w.keys[nd.begin + k], mx, my));
      w.bcnt[e] = c;
    }
    uint64_t big = __ballot(valid && !small);
    while (big) {
      const int bl = __ffsll((unsigned long long)big) - 1;
      big &= big - 1;
      const int eb = (q0 + bl) * tm.nw + tm.tw;
      const Node nb = w.nodes[w.order[eb]];
      const int mx = nb.x0 + ((nb.x1 - nb.x0 + 1) >> 1), my = nb.y0 + ((nb.y1 - nb.y0 + 1) >> 1);
      uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      for (uint32_t i = 0; i < nb.count; i += 64) {
        int g = -1;
        if (i + lane < nb.count) g = qt_class(w.keys[nb.begin + i + lane], mx, my);
        c0 += __popcll(__ballot(g == 0)); c1 += __popcll(__ballot(g == 1));
        c2 += __popcll(__ballot(g == 2)); c3 += __popcll(__ballot(g == 3));
      }
      if (lane == 0) w.bcnt[eb] = qt_pack4(c0, c1, c2, c3);
    }
  }
  QT_TEAM_SYNC(tm);
  QT_MARK(16);
  // (2) ranks in processing order, and where to stop (wave 0 of a team; the totals reach the others through tm.sh)
  runCh = 0; runEx = 0; runSize = s.size; mProc = m;
  if (tm.tw == 0) {
  for (int e0 = 0; e0 < m; e0 += 64) {
    const int e = e0 + lane;
    const uint64_t c = e < m ? w.bcnt[e] : 0ull;
    int nCh = 0, nEx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const uint32_t cq = (uint32_t)(c >> (16 * q)) & 0xFFFFu; nCh += cq > 0; nEx += cq > 1; }
    const int incCh = qt_scan_incl(nCh), incEx = qt_scan_incl(nEx), incSz = qt_scan_incl(e < m ? nCh - 1 : 0);
    int last = (m - e0 < 64 ? m - e0 : 64) - 1;
    bool cut = false;
    if (cutoffN >= 0) {
      const uint64_t reach = __ballot(e < m && runSize + incSz >= cutoffN);
      if (reach) { last = __ffsll((unsigned long long)reach) - 1; mProc = e0 + last + 1; cut = true; }
    }
    if (e < m) w.brank[e] = (uint32_t)(runCh + incCh - nCh) | ((uint32_t)(runEx + incEx - nEx) << 16);
    runCh += __builtin_amdgcn_readlane(incCh, last); runEx += __builtin_amdgcn_readlane(incEx, last); runSize += __builtin_amdgcn_readlane(incSz, last);   // (last is wave-uniform)
    if (cut) break;
  }
  if (tm.nw > 1 && lane == 0) { tm.sh[0] = runCh; tm.sh[1] = runEx; tm.sh[2] = mProc; }
  }
  QT_TEAM_SYNC(tm);
  if (tm.nw > 1) { runCh = tm.sh[0]; runEx = tm.sh[1]; mProc = tm.sh[2]; }
  QT_MARK(17);
  if (m < mAll && s.size + runCh - mProc < cutoffN) { m = mAll; continue; }   // the cut lies beyond the nodes counted (the same decision in every wave of a team)
  break;
  }
  // (3) partition the keys, write the children, erase the parents
  const int oldHead = s.head, nA0 = s.nA, nFree0 = s.nFree;
  if (teamHuge) {   // the huge nodes: partitioned by the whole team
    for (int e0 = 0; e0 < mProc; e0 += 64) {
      uint32_t cntN = 0;
      if (e0 + lane < mProc) cntN = w.nodes[w.order[e0 + lane]].count;
      uint64_t hm = __ballot(cntN > QT_HUGE);
      while (hm) {
        const int bl = __ffsll((unsigned long long)hm) - 1;
        hm &= hm - 1;
        const Node nb = w.nodes[w.order[e0 + bl]];
        const int bmx = nb.x0 + ((nb.x1 - nb.x0 + 1) >> 1), bmy = nb.y0 + ((nb.y1 - nb.y0 + 1) >> 1);
        uint32_t cnt[4];
        qt_partition_team(w.keys, w.tmp, nb.begin, nb.count, [bmx, bmy](uint32_t k) -> int { return qt_class(k, bmx, bmy); }, cnt, tm);
        if (tm.tw == 0) {
          const uint32_t brk = w.brank[e0 + bl];
          if (lane < 4) qt_write_children(w, nb, bmx, bmy, cnt, brk, oldHead, nA0, nFree0, lane);   // (cnt: the team partition's own exact counts)
          if (lane == 0) w.list[nb.lit] = 0xFFFF;
        }
      }
    }
  }
  for (int q0 = 0; q0 * tm.nw < mProc; q0 += 64) {
    const int e = (q0 + lane) * tm.nw + tm.tw;
    Node nd = {};
    uint64_t c = 0;
    uint32_t rk = 0;
    if (e < mProc) nd = w.nodes[w.order[e]];
    const bool valid = e < mProc && !(teamHuge && nd.count > QT_HUGE);
    if (valid) { c = w.bcnt[e]; rk = w.brank[e]; }
    const bool small = valid && nd.count <= smallMax;
    const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
    if (small) {
      // stable 4-way partition through tmp: group bases packed like the counts
      const uint32_t c0 = (uint32_t)c & 0xFFFFu, c1 = (uint32_t)(c >> 16) & 0xFFFFu, c2 = (uint32_t)(c >> 32) & 0xFFFFu;
      uint64_t bs = ((uint64_t)c0 << 16) | ((uint64_t)(c0 + c1) << 32) | ((uint64_t)(c0 + c1 + c2) << 48);
      for (uint32_t k = 0; k < nd.count; ++k) {
        const uint32_t key = w.keys[nd.begin + k];
        const int g = qt_class(key, mx, my);
        w.tmp[nd.begin + ((uint32_t)(bs >> (16 * g)) & 0xFFFFu)] = key;
        bs += 1ull << (16 * g);
      }
      for (uint32_t k = 0; k < nd.count; ++k) w.keys[nd.begin + k] = w.tmp[nd.begin + k];
      uint32_t c4[4];
      qt_unpack4(c, c4);
#pragma unroll
      for (int q = 0; q < 4; ++q) qt_write_children(w, nd, mx, my, c4, rk, oldHead, nA0, nFree0, q);
      w.list[nd.lit] = 0xFFFF;
    }
    uint64_t big = __ballot(valid && !small);
    while (big) {
      const int bl = __ffsll((unsigned long long)big) - 1;
      big &= big - 1;
      const int eb = (q0 + bl) * tm.nw + tm.tw;
      const Node nb = w.nodes[w.order[eb]];
      const int bmx = nb.x0 + ((nb.x1 - nb.x0 + 1) >> 1), bmy = nb.y0 + ((nb.y1 - nb.y0 + 1) >> 1);
      const uint64_t bc = w.bcnt[eb];
      const uint32_t brk = w.brank[eb];
      uint32_t cnt[4];
      qt_unpack4(bc, cnt);
      // (a node of more than 65535 keys — noise-like levels only — may have saturated fields: the partition counts again)
      qt_partition(w.keys, w.tmp, nb.begin, nb.count, [bmx, bmy](uint32_t k) -> int { return qt_class(k, bmx, bmy); }, cnt, nb.count <= 0xFFFFu);
      if (lane < 4) qt_write_children(w, nb, bmx, bmy, cnt, brk, oldHead, nA0, nFree0, lane);
      if (lane == 0) w.list[nb.lit] = 0xFFFF;
    }
  }
  QT_TEAM_SYNC(tm);
  QT_MARK(18);
  // (4) bookkeeping (every wave of a team computes the same state); the parents' ids go back on the free stack after every child id has been taken
  s.head = oldHead - runCh;
  s.size += runCh - mProc;
  s.nA = nA0 + runEx;
  if (nToExpand) *nToExpand += runEx;
  s.nFree = nFree0 - runCh;
  for (int e = estart + lane; e < mProc; e += estep) w.freeIds[s.nFree + e] = w.order[e];
  s.nFree += mProc;
  QT_TEAM_SYNC(tm);
  QT_MARK(19);
}
#endif

// Move the live entries to the top of the list array (order preserved) and refresh Node::lit.
QT_HD void qt_compact(Work& w, State& s) {
  QT_SYNC();
#if QT_DEVICE
  // 64 list slots per step, from the top down: a live entry moves up by the number of tombstones above it
  const int lane = QT_LANE;
  int wp = w.listCap;
  for (int top = w.listCap; top > s.head; top -= 64) {
    const int idx = top - 64 + lane;                    // lane 63 = highest slot of the chunk
    uint16_t id = 0xFFFF;
    if (idx >= s.head) id = w.list[idx];
    const uint64_t m = __ballot(id != 0xFFFF);
    const int above = __popcll(lane == 63 ? 0ull : (m >> (lane + 1)));
    QT_SYNC();                                           // every lane has read before any lane writes
    if (id != 0xFFFF) { const int np = wp - 1 - above; w.list[np] = id; w.nodes[id].lit = (uint16_t)np; }
    wp -= __popcll(m);
    QT_SYNC();
  }
  s.head = wp;
#else
  int wp = w.listCap;
  for (int rp = w.listCap - 1; rp >= s.head; --rp) {
    const uint16_t id = w.list[rp];
    if (id != 0xFFFF) {
      --wp;
      w.list[wp] = id;
      w.nodes[id].lit = (uint16_t)wp;
    }
  }
  s.head = wp;
#endif
  QT_SYNC();
}

enum { QT_FF_CONTINUE = 0, QT_FF_PHASE = 1, QT_FF_FINISH = 2 };
constexpr int QT_FF_MAXD = 5;

struct FfGeom { float hX; int last, width, height, nIni; };

// cell code of key k after d halvings: root * 4^d + sum of (bx + 2 by) digits, most significant first (DivideNode's arithmetic, :475-523)
QT_HD uint32_t qt_ff_code(uint32_t k, const FfGeom& fg, int d) {
  const int x = key_x(k), y = key_y(k);
  int g = (int)((float)x / fg.hX);
  g = g > fg.last ? fg.last : g;
  int x0 = (int)(fg.hX * (float)g), x1 = (int)(fg.hX * (float)(g + 1)), y0 = 0, y1 = fg.height;
  uint32_t c = (uint32_t)g;
  for (int j = 0; j < d; ++j) {
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
    const int bx = x >= mx ? 1 : 0, by = y >= my ? 1 : 0;
    c = c * 4 + (uint32_t)(bx + 2 * by);
    x0 = bx ? mx : x0; x1 = bx ? x1 : mx; y0 = by ? my : y0; y1 = by ? y1 : my;
  }
  return c;
}
// rectangle of the depth-d cell `code`
QT_HD void qt_ff_rect(uint32_t code, int d, const FfGeom& fg, int& x0, int& y0, int& x1, int& y1) {
  const int g = (int)(code >> (2 * d));
  x0 = (int)(fg.hX * (float)g); x1 = (int)(fg.hX * (float)(g + 1)); y0 = 0; y1 = fg.height;
  for (int j = 1; j <= d; ++j) {
    const int dg = (int)(code >> (2 * (d - j))) & 3;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
    if (dg & 1) x0 = mx; else x1 = mx;
    if (dg & 2) y0 = my; else y1 = my;
  }
}
// generation-order index t -> cell code at depth d (d >= 1)
QT_HD uint32_t qt_ff_untransform(uint32_t t, int d, int nIni) {
  const uint32_t pm = (1u << (2 * d)) - 1u;
  const uint32_t tr = t >> (2 * d);
  const uint32_t r = ((d - 1) & 1) ? (uint32_t)(nIni - 1) - tr : tr;
  return (r << (2 * d)) | ((t & pm) ^ (0xCCCCCCCCu & pm));
}


#if defined(QT_HOST_FAST_FORWARD) && !QT_DEVICE
// The fast forward restated serially for the host (tests/native/qt_host.cc defines QT_HOST_FAST_FORWARD): the SAME construction as the device's
// qt_fast_forward below — cell codes, per-depth counts, the reference's three conditions, one stable sort by the depth-d0 code, the list as
// (children of sweep d0 in reverse generation order) ++ (single-key nodes of the shallower sweeps) ++ (single-key roots) — in plain loops, so that
// the CPU suite can check the construction itself (generation order, what follows the sweeps, node rectangles) against the oracle's std::list
// restatement on thousands of random inputs.  Returns QT_FF_*.
inline int qt_fast_forward_host(Work& w, State& s, uint32_t nkeys, const FfGeom& fg, int N) {
  const int nIni = fg.nIni;
  int D = 1;
  while (D < QT_FF_MAXD && (nIni << (2 * D)) <= 2 * N) ++D;
  std::vector<std::vector<int>> cnt(D + 1);
  for (int d = 0; d <= D; ++d) cnt[d].assign((size_t)nIni << (2 * d), 0);
  std::vector<uint32_t> code(nkeys);
  for (uint32_t i = 0; i < nkeys; ++i) { code[i] = qt_ff_code(w.keys[i], fg, D); ++cnt[D][code[i]]; }
  for (int d = D - 1; d >= 0; --d)
    for (size_t c = 0; c < cnt[d].size(); ++c) cnt[d][c] = cnt[d + 1][4 * c] + cnt[d + 1][4 * c + 1] + cnt[d + 1][4 * c + 2] + cnt[d + 1][4 * c + 3];
  auto nonEmpty = [&](int d) { int n = 0; for (int c : cnt[d]) n += c > 0; return n; };
  auto multi = [&](int d) { int n = 0; for (int c : cnt[d]) n += c > 1; return n; };
  int d0 = D, outcome = QT_FF_CONTINUE, prev = nonEmpty(0);
  for (int d = 1; d <= D; ++d) {
    const int sz = nonEmpty(d), nEx = multi(d);
    if (sz >= N || sz == prev) { d0 = d; outcome = QT_FF_FINISH; break; }
    if (sz + 3 * nEx > N) { d0 = d; outcome = QT_FF_PHASE; break; }
    prev = sz;
  }
  // one stable sort of the keys by their depth-d0 cell
  {
    const int down = 2 * (D - d0);
    std::vector<uint32_t> idx(nkeys);
    for (uint32_t i = 0; i < nkeys; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (code[a] >> down) < (code[b] >> down); });
    for (uint32_t i = 0; i < nkeys; ++i) w.tmp[i] = w.keys[idx[i]];
    for (uint32_t i = 0; i < nkeys; ++i) w.keys[i] = w.tmp[i];
  }
  std::vector<int> P(cnt[d0].size() + 1, 0);
  for (size_t c = 0; c < cnt[d0].size(); ++c) P[c + 1] = P[c] + cnt[d0][c];
  // members of the list, segment by segment from the head
  struct Member { int d; uint32_t c; };
  std::vector<Member> order;
  for (int d = d0; d >= 0; --d) {
    std::vector<Member> seg;
    const uint32_t cells = (uint32_t)nIni << (2 * d);
    for (uint32_t t = 0; t < cells; ++t) {   // ascending generation order
      const uint32_t c = d == 0 ? t : qt_ff_untransform(t, d, nIni);
      const int k = cnt[d][c];
      if (d == d0 ? k < 1 : k != 1) continue;
      if (d > 0 && cnt[d - 1][c >> 2] <= 1) continue;
      seg.push_back(Member{d, c});
    }
    if (d > 0) std::reverse(seg.begin(), seg.end());   // push_front: the list holds a sweep's children in reverse generation order (roots: push_back)
    order.insert(order.end(), seg.begin(), seg.end());
  }
  const int size = (int)order.size();
  s.head = w.listCap - size;
  s.size = size;
  s.nFree = w.nodeCap - size;
  for (int i = 0; i < size; ++i) {
    const Member& m = order[i];
    int x0, y0, x1, y1;
    qt_ff_rect(m.c, m.d, fg, x0, y0, x1, y1);
    Node nd;
    nd.x0 = (int16_t)x0; nd.y0 = (int16_t)y0; nd.x1 = (int16_t)x1; nd.y1 = (int16_t)y1;
    nd.begin = (uint32_t)P[(size_t)m.c << (2 * (d0 - m.d))];
    nd.count = (uint32_t)cnt[m.d][m.c]; nd.lit = (uint16_t)(s.head + i); nd.noMore = nd.count == 1 ? 1 : 0;
    w.nodes[i] = nd;
    w.list[s.head + i] = (uint16_t)i;
  }
  // vSizeAndPointerToNode: the last sweep's children with more than one key, in generation order = the head segment backwards
  s.nA = 0;
  for (int i = size - 1; i >= 0; --i)
    if (order[i].d == d0 && w.nodes[i].count > 1)
      w.vA[s.nA++] = ((uint64_t)w.nodes[i].count << 32) | ((uint64_t)(uint16_t)w.nodes[i].x0 << 16) | (uint64_t)i;
  return outcome;
}
#endif

#if QT_DEVICE
// ---- the full sweeps at once ("fast forward", round 6) -----------------------------------------------------------
// While the reference's loop (:589-655) runs FULL sweeps — every node with more than one key is divided, no early exit — the state it
// reaches is a pure function of the keys' positions: after d sweeps a key lies in the cell named by its root and d (x, y) halving
// decisions, the d stable 4-way partitions of its ancestors amount to ONE stable sort of the key array by that cell code, the list
// is (children of sweep d in reverse generation order) ++ (single-key nodes of sweep d - 1, reverse generation order) ++ ... ++
// (single-key roots), and whether sweep d is followed by another full sweep, by the "largest first" phase or by the end only depends
// on how many cells of each depth hold one / several keys.  So instead of one partition pass over all keys per sweep (each with its
// count / rank / scatter phases and barriers: 4 sweeps = 113 of level 0's 207 us at 1920 x 1080 / 4000 features with 16 waves), the
// keys are histogrammed once by their depth-Dcap cell, the per-depth cell counts decide how many full sweeps d0 the reference runs,
// the keys are radix-sorted (stable, LSD, usually ONE pass with the whole cell code as the digit) by their depth-d0 cell, and the nodes, the list and
// vSizeAndPointerToNode are written directly in the order the sweeps would have left them.
//
// Generation order.  Sweep d visits the list from its head: the children of sweep d - 1 in REVERSE generation order (push_front),
// each parent emitting its non-empty children in the order n1..n4.  Two depth-d nodes whose paths first differ at digit j (root = 0)
// therefore compare by that digit ascending when d - j is even and descending when it is odd (the root digit: by d - 1): generation
// order = ascending order of the path code with those digits complemented (an XOR with 0b11 per child digit).
// lanes (among `valid`) that hold the same `bits`-bit digit as this lane: one ballot per digit bit, every lane keeps the lanes that agree with it
// on that bit.  (Round 6 also tried one step per DISTINCT value present — readfirstlane, compare, ballot — which is fewer steps on these spatially
// coherent keys but every step is a VALU -> SALU -> VALU round trip: 1.4 x slower per chunk.)
__device__ __forceinline__ uint64_t qt_peers(int dg, bool valid, int bits) {
  uint64_t peers = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool one = (dg >> b) & 1;
    const uint64_t m = __ballot(valid && one);
    peers &= m ^ (one ? 0ull : ~0ull);
  }
  return peers;
}
// number of set bits of m in the lanes below this one
__device__ __forceinline__ int qt_below(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// One stable LSD radix pass of src[0..n) -> dst by digit (code >> shift) & (2^bits - 1) by the whole team (contiguous slice per wave, so the order of
// equal digits is the input order).  Round 6, second form: the digit is as wide as the scratch allows — the whole cell code when it fits, i.e. ONE
// pass instead of two 6-bit ones (22 -> 20 us of level 0's quadtree at 1920 x 1080 / 4000 with 16 waves — nine ballots per chunk instead of six eat most of the saved pass —, 36 -> 22 us per wave in the one-wave-per-level packing).  cnt16: [2^bits][nw] 16-bit counters (digit-major), tot:
// [2^bits] ints; both in the node array, which nothing reads before the nodes are written.  n < 65536 (the caller checks).
template <typename Code>
__device__ inline void qt_radix_pass(const uint32_t* src, uint32_t* dst, uint32_t n, int shift, int bits, Code code, const Team& tm, uint16_t* cnt16,
                                     int* tot) {
  const int lane = QT_LANE, tid = tm.tw * 64 + lane, nth = tm.nw * 64;
  const int nd = 1 << bits, mask = nd - 1;
  uint32_t* cnt32 = reinterpret_cast<uint32_t*>(cnt16);
  for (int i = tid; i < (nd * tm.nw + 1) / 2; i += nth) cnt32[i] = 0u;
  const uint32_t per = ((n + tm.nw * 64 - 1) / (tm.nw * 64)) * 64;
  const uint32_t lo = per * tm.tw < n ? per * tm.tw : n, hi = lo + per < n ? lo + per : n;
  QT_TEAM_SYNC(tm);
  QT_T0();
#pragma unroll 4
  for (uint32_t i = lo + lane; i < hi; i += 64) {   // (LDS atomics without return value: fire and forget; two 16-bit counters per word, neither can carry: n < 65536)
    const int e = ((int)(code(src[i]) >> shift) & mask) * tm.nw + tm.tw;
    atomicAdd(&cnt32[e >> 1], 1u << (16 * (e & 1)));
  }
  QT_TEAM_SYNC(tm);
  QT_MARK(26);
  // first slot of (digit, wave) = keys of smaller digits + keys of this digit in earlier waves: per digit the waves' counts become an exclusive
  // prefix (in place) and the digit's total goes to tot[]; then wave 0 turns tot[] into its exclusive prefix
  for (int d = tid; d < nd; d += nth) {
    uint16_t* row = cnt16 + d * tm.nw;
    int run = 0;
    for (int wv = 0; wv < tm.nw; ++wv) { const int c = row[wv]; row[wv] = (uint16_t)run; run += c; }
    tot[d] = run;
  }
  QT_TEAM_SYNC(tm);
  if (tm.tw == 0) {
    int run = 0;
    for (int d0 = 0; d0 < nd; d0 += 64) {
      const int d = d0 + lane;
      const int v = d < nd ? tot[d] : 0;
      const int inc = qt_scan_incl(v);
      if (d < nd) tot[d] = run + inc - v;
      run += __builtin_amdgcn_readlane(inc, 63);
    }
  }
  QT_TEAM_SYNC(tm);
  QT_MARK(27);
  // scatter: the slot of a key = first slot of (digit, this wave) + keys of that digit in the wave's earlier chunks + such keys in lower lanes.  The
  // running slots make the chunks dependent; their keys and digits do not: loaded four chunks ahead.
  for (uint32_t i0 = lo; i0 < hi; i0 += 256) {
    uint32_t k[4];
    int dg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + 64 * u + lane; k[u] = i < hi ? src[i] : 0u; }
#pragma unroll
    for (int u = 0; u < 4; ++u) dg[u] = (int)(code(k[u]) >> shift) & mask;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + 64 * u >= hi) break;   // (wave-uniform)
      const bool valid = i0 + 64 * u + lane < hi;
      const uint64_t peers = qt_peers(dg[u], valid, bits);
      const int below = qt_below(peers);
      int off = 0;
      uint16_t* slot = cnt16 + dg[u] * tm.nw + tm.tw;
      if (valid) { off = *slot; dst[tot[dg[u]] + off + below] = k[u]; }
      QT_SYNC();
      if (valid && below == 0) *slot = (uint16_t)(off + __popcll(peers));
      QT_SYNC();
    }
  }
  QT_TEAM_SYNC(tm);
  QT_MARK(28);
}

// Replaces the root set-up and every full sweep the reference would run (up to QT_FF_MAXD, and as far as the cell counts fit `hist`): on
// return the keys are partitioned, nodes / list / vA / State are what the sweeps leave, and the result says how the loop goes on.
// hist: histCap ints of scratch (the vB region), aux: 16 + 12 * nw ints (the brank region); the code tables and the radix counters live in the node array.
__device__ inline int qt_fast_forward(Work& w, State& s, uint32_t nkeys, const FfGeom& fg, int N, const Team& tm, int* hist, int histCap, int* aux,
                                      int auxCap) {
  const int lane = QT_LANE, tid = tm.tw * 64 + lane, nth = tm.nw * 64;
  const int nIni = fg.nIni;
  // depth of the histogram: no deeper than the sweeps can go (a full sweep needs size <= N afterwards, and cells >= nodes), and what fits
  // level d's counts start at off(d) = sum over j < d of (nIni 4^j + 1): one slot more than cells (the prefix total of step 5)
  auto offOf = [nIni](int d) -> int { return nIni * (((1 << (2 * d)) - 1) / 3) + d; };
  int D = 1;
  while (D < QT_FF_MAXD && (nIni << (2 * D)) <= 2 * N && offOf(D + 2) <= histCap) ++D;
  const int total = offOf(D + 1);
  if (total > histCap || 16 + 12 * tm.nw > auxCap || nkeys >= 65536u || (size_t)w.nodeCap * sizeof(Node) < (size_t)4 * (2 * tm.nw + 4) + 16) return -1;   // (tiny quotas, or a level of > 65535 candidates: the caller runs the sweeps)
  for (int i = tid; i < total; i += nth) hist[i] = 0;
  for (int i = tid; i < 16 + 12 * tm.nw; i += nth) aux[i] = 0;
  QT_TEAM_SYNC(tm);
  QT_T0();
  // The depth-D code of a key is (x half) | (y half): the halving decisions of the two axes do not depend on each other.  Both halves are
  // tabulated once per level (x: root digit and the bx bits, y: the by bits, each already at its digit position) in the node array, which is
  // not written before step 6 — a code is then two LDS reads and an OR instead of a float division and D dependent select chains.
  const int tabW = fg.width + 1, tabH = fg.height + 1;
  uint16_t* xs = reinterpret_cast<uint16_t*>(w.nodes);
  uint16_t* ys = xs + tabW;
  // (beside the tables the radix pass wants room for at least 6-bit digits: 64 * (2 nw + 4) bytes)
  const bool useTab = (((size_t)(tabW + tabH) * 2 + 15) & ~(size_t)15) + (size_t)64 * (2 * tm.nw + 4) <= (size_t)w.nodeCap * sizeof(Node) && 2 * D + 2 <= 16;
  if (useTab) {
    for (int i = tid; i < tabW + tabH; i += nth) {
      if (i < tabW) xs[i] = (uint16_t)(qt_ff_code(make_key(i, 0, 0), fg, D) & (0x5555u | (~0u << (2 * D))));
      else ys[i - tabW] = (uint16_t)(qt_ff_code(make_key(0, i - tabW, 0), fg, D) & (0xAAAAu & ((1u << (2 * D)) - 1u)));
    }
    QT_TEAM_SYNC(tm);
  }
  const int Dc = D;
  auto codeD = [xs, ys, useTab, fg, Dc](uint32_t k) -> uint32_t {
    return useTab ? (uint32_t)xs[key_x(k)] | (uint32_t)ys[key_y(k)] : qt_ff_code(k, fg, Dc);
  };
  {  // (1) keys per depth-D cell (LDS atomics without return: nothing waits for them)
    int* hD = hist + offOf(D);
#pragma unroll 4
    for (uint32_t i = (uint32_t)tid; i < nkeys; i += (uint32_t)nth) atomicAdd(&hD[codeD(w.keys[i])], 1);
  }
  QT_TEAM_SYNC(tm);
  QT_MARK(20);
  // (2) the coarser depths, and per depth the number of non-empty cells (= list size after that many sweeps) and of cells with more than one key
  for (int d = D - 1; d >= 0; --d) {
    const int cells = nIni << (2 * d);
    const int* hc = hist + offOf(d + 1);
    int* hp = hist + offOf(d);
    int n1 = 0, n2 = 0;
    for (int c = tid; c < cells; c += nth) {
      const int a0 = hc[4 * c], a1 = hc[4 * c + 1], a2 = hc[4 * c + 2], a3 = hc[4 * c + 3];
      hp[c] = a0 + a1 + a2 + a3;
      n1 += (a0 > 0) + (a1 > 0) + (a2 > 0) + (a3 > 0);
      n2 += (a0 > 1) + (a1 > 1) + (a2 > 1) + (a3 > 1);
    }
    n1 = morbwave::sum_i32(n1); n2 = morbwave::sum_i32(n2);
    if (lane == 0 && (n1 | n2)) { atomicAdd(&aux[2 * (d + 1)], n1); atomicAdd(&aux[2 * (d + 1) + 1], n2); }
    QT_TEAM_SYNC(tm);
  }
  int size0 = 0;
  for (int r = 0; r < nIni; ++r) size0 += hist[offOf(0) + r] > 0;
  // (3) how many full sweeps the reference runs, and what follows (:652-667)
  int d0 = D, outcome = QT_FF_CONTINUE;
  {
    int prev = size0;
    for (int d = 1; d <= D; ++d) {
      const int sz = aux[2 * d], nEx = aux[2 * d + 1];
      if (sz >= N || sz == prev) { d0 = d; outcome = QT_FF_FINISH; break; }
      if (sz + 3 * nEx > N) { d0 = d; outcome = QT_FF_PHASE; break; }
      prev = sz;
    }
  }
  const int size = aux[2 * d0], nMulti = aux[2 * d0 + 1];
  QT_TEAM_SYNC(tm);   // (aux is rewritten below)
  QT_MARK(21);
  // (4) stable sort of the keys by their depth-d0 cell
  {
    int rootBits = 0;
    while ((1 << rootBits) < nIni) ++rootBits;
    const int bits = 2 * d0 + rootBits;
    const int down = 2 * (D - d0);
    auto code = [codeD, down](uint32_t k) -> uint32_t { return codeD(k) >> down; };
    // scratch of a pass with b-bit digits: 2^b * (2 nw + 4) bytes behind the code tables
    const size_t tabBytes = useTab ? (((size_t)(tabW + tabH) * 2 + 15) & ~(size_t)15) : 0;
    const size_t room = (size_t)w.nodeCap * sizeof(Node) - tabBytes;
    int maxDb = 0;
    while (maxDb < 12 && ((size_t)2 << maxDb) * (size_t)(2 * tm.nw + 4) <= room) ++maxDb;
    const int passes = (bits + maxDb - 1) / maxDb, db = (bits + passes - 1) / passes;   // (maxDb >= 1: checked before anything was written)
    uint16_t* cnt16 = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(w.nodes) + tabBytes);
    int* tot = reinterpret_cast<int*>(cnt16 + ((size_t)(1 << db) * tm.nw + 1) / 2 * 2);
    uint32_t* a = w.keys;
    uint32_t* b = w.tmp;
    for (int sh = 0; sh < bits; sh += db) {
      qt_radix_pass(a, b, nkeys, sh, bits - sh < db ? bits - sh : db, code, tm, cnt16, tot);
      uint32_t* t = a; a = b; b = t;
    }
    if (a != w.keys) {
      for (uint32_t i = (uint32_t)tid; i < nkeys; i += (uint32_t)nth) w.keys[i] = a[i];
      QT_TEAM_SYNC(tm);
    }
  }
  QT_MARK(22);
  // (5) first key of every depth-d0 cell: exclusive prefix of its counts, in place (slot `cells` = total)
  int* P = hist + offOf(d0);
  const int cells0 = nIni << (2 * d0);
  {
    const int perW = ((cells0 + 1 + tm.nw * 64 - 1) / (tm.nw * 64)) * 64;
    const int lo = perW * tm.tw < cells0 + 1 ? perW * tm.tw : cells0 + 1, hi = lo + perW < cells0 + 1 ? lo + perW : cells0 + 1;
    int run = 0;
    for (int c0 = lo; c0 < hi; c0 += 64) {
      const int c = c0 + lane;
      const int v = c < hi && c < cells0 ? P[c] : 0;
      const int inc = qt_scan_incl(v);
      if (c < hi) P[c] = run + inc - v;
      run += __builtin_amdgcn_readlane(inc, 63);
    }
    if (lane == 0) aux[tm.tw] = run;
    QT_TEAM_SYNC(tm);
    int before = 0;
    for (int wv = 0; wv < tm.tw; ++wv) before += aux[wv];
    if (before) for (int c = lo + lane; c < hi; c += 64) P[c] += before;
    QT_TEAM_SYNC(tm);
  }
  QT_MARK(23);
  // (6) the nodes: per depth the members of the list (depth d0: every non-empty cell under a divided parent; shallower: the single-key cells
  // under a divided parent), counted per wave slice of the generation order, then written
  int* segCnt = aux + 16;   // [(d * nw + wave) * 2 + {members, members with > 1 key}]
  auto member = [&](int d, uint32_t c, int& cntOut) -> bool {
    const int* h = hist + offOf(d);
    const int cnt = d == d0 ? h[c + 1] - h[c] : h[c];
    cntOut = cnt;
    if (d == d0 ? cnt < 1 : cnt != 1) return false;
    return d == 0 || hist[offOf(d - 1) + (c >> 2)] > 1;
  };
  // (depth d0 - 1's counts are intact: only level d0 was turned into a prefix)
  for (int pass = 0; pass < 2; ++pass) {
    int segBase = s.head;   // set below in pass 1
    for (int d = d0; d >= 0; --d) {
      const int cells = nIni << (2 * d);
      const int perW = ((cells + tm.nw * 64 - 1) / (tm.nw * 64)) * 64;
      const int lo = perW * tm.tw < cells ? perW * tm.tw : cells, hi = lo + perW < cells ? lo + perW : cells;
      int segTotal = 0, before = 0, beforeM = 0;
      if (pass == 1) {   // lane = wave of the team (nw <= 64)
        const int c = lane < tm.nw ? segCnt[(d * tm.nw + lane) * 2] : 0, cm = lane < tm.nw ? segCnt[(d * tm.nw + lane) * 2 + 1] : 0;
        segTotal = morbwave::sum_i32(c);
        before = morbwave::sum_i32(lane < tm.tw ? c : 0);
        beforeM = morbwave::sum_i32(lane < tm.tw ? cm : 0);
      }
      int run = 0, runM = 0;
      for (int t0 = lo; t0 < hi; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < hi;
        const uint32_t c = !in ? 0u : (d == 0 ? (uint32_t)t : qt_ff_untransform((uint32_t)t, d, nIni));
        int cnt = 0;
        const bool mem = in && member(d, c, cnt);
        const bool multi = mem && cnt > 1;
        const uint64_t mm = __ballot(mem), mx = __ballot(multi);
        if (pass == 1 && mem) {
          const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
          const int rankAsc = before + run + __popcll(mm & lt);
          const int pos = d == 0 ? segBase + rankAsc : segBase + (segTotal - 1 - rankAsc);
          const int id = pos - s.head;
          int x0, y0, x1, y1;
          qt_ff_rect(c, d, fg, x0, y0, x1, y1);
          Node nd;
          nd.x0 = (int16_t)x0; nd.y0 = (int16_t)y0; nd.x1 = (int16_t)x1; nd.y1 = (int16_t)y1;
          nd.begin = (uint32_t)(d == d0 ? P[c] : P[c << (2 * (d0 - d))]);
          nd.count = (uint32_t)cnt; nd.lit = (uint16_t)pos; nd.noMore = cnt == 1 ? 1 : 0;
          w.nodes[id] = nd;
          w.list[pos] = (uint16_t)id;
          if (multi) w.vA[beforeM + runM + __popcll(mx & lt)] = ((uint64_t)(uint32_t)cnt << 32) | ((uint64_t)(uint16_t)x0 << 16) | (uint64_t)id;
        }
        run += __popcll(mm); runM += __popcll(mx);
      }
      if (pass == 0) { if (lane == 0) { segCnt[(d * tm.nw + tm.tw) * 2] = run; segCnt[(d * tm.nw + tm.tw) * 2 + 1] = runM; } }
      else segBase += segTotal;
    }
    if (pass == 0) {
      s.head = w.listCap - size;
      QT_TEAM_SYNC(tm);
    }
  }
  s.size = size;
  s.nA = nMulti;
  s.nFree = w.nodeCap - size;   // ids 0 .. size - 1 are taken: exactly the top `size` entries of the initial free stack
  QT_TEAM_SYNC(tm);
  QT_MARK(24);
  return outcome;
}

// std::sort emulation by the whole team: the partitions of disjoint ranges do not depend on each other, so after the first cut the ranges of
// a recursion level are dealt to the waves (breadth first; two range lists in LDS, this level's and the next one's).  Same result as
// qt_std_sort_wave.  q: 1024 ints of scratch.  false: n too large for the lists (nothing touched).
__device__ inline bool qt_std_sort_team(uint64_t* v, uint64_t* tmp, int n, uint16_t* posL, uint16_t* posR, const Team& tm, int* q) {
  constexpr int QCAP = 168;
  static_assert(6 * QCAP + 2 <= 1024, "the range lists must fit qt_distribute's static scratch");   // ranges of one recursion level: disjoint, each > 16 entries -> at most n / 17
  if (n / 17 + 1 > QCAP) return false;
  const int lane = QT_LANE;
  int* cnt = q + 6 * QCAP;    // [0], [1]: ranges in list 0 / 1
  if (n > 16) {
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    if (tm.tw == 0 && lane == 0) { q[0] = 0; q[1] = n; q[2] = lg * 2; cnt[0] = 1; cnt[1] = 0; }
    __syncthreads();
    int cur = 0;
    while (true) {
      const int avail = cnt[cur];
      if (avail == 0) break;
      const int* qc = q + 3 * QCAP * cur;
      int* qn = q + 3 * QCAP * (cur ^ 1);
      for (int r = tm.tw; r < avail; r += tm.nw) {
        const int first = qc[3 * r], last = qc[3 * r + 1];
        int depth = qc[3 * r + 2];
        if (depth == 0) {
          QT_SYNC();
          if (lane == 0) qt_heapsort(v, first, last);
          QT_SYNC();
          continue;
        }
        --depth;
        const int cut = qt_partition_pivot_wave(v, first, last, posL + first, posR + 2 * first);
        if (lane == 0) {
          if (last - cut > 16) { const int k = atomicAdd(&cnt[cur ^ 1], 1); qn[3 * k] = cut; qn[3 * k + 1] = last; qn[3 * k + 2] = depth; }
          if (cut - first > 16) { const int k = atomicAdd(&cnt[cur ^ 1], 1); qn[3 * k] = first; qn[3 * k + 1] = cut; qn[3 * k + 2] = depth; }
        }
      }
      __syncthreads();
      if (tm.tw == 0 && lane == 0) cnt[cur] = 0;
      cur ^= 1;
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = tm.tw * 64 + lane; i < n; i += tm.nw * 64) tmp[qt_window_rank(v, n, i)] = v[i];   // the final insertion sort = windowed stable ranks
  __syncthreads();
  for (int i = tm.tw * 64 + lane; i < n; i += tm.nw * 64) v[i] = tmp[i];
  __syncthreads();
  return true;
}
#endif

// keys[0..nkeys) hold vToDistributeKeys in order.  Writes the selected keys (reference output order) to
// out[] and returns their number.  width = maxX-minX, height = maxY-minY.  tm: the team of waves working this level (device; see Team).
QT_HD int qt_distribute(Work& w, uint32_t nkeys, int width, int height, int N, uint32_t* out, int outCap, const Team& tm = Team{1, 0, nullptr}) {
  // nIni = round((float)width / height); hX = (float)width / nIni   (:545-547)
  const float ratio = (float)width / (float)height;
  const int nIni = (int)roundf(ratio);
  if (nIni <= 0 || nIni > 4 || nkeys == 0) return 0;
  const float hX = (float)width / (float)nIni;
  const bool w0 = tm.tw == 0;   // the wave that performs the order-dependent steps and every single-lane store

  State s;
  s.head = w.listCap;
  s.size = 0;
  s.nA = 0;
  s.nFree = w.nodeCap;
  for (int i = (QT_DEVICE ? tm.tw * 64 + QT_LANE : 0); i < w.nodeCap; i += (QT_DEVICE ? 64 * tm.nw : 1)) w.freeIds[i] = (uint16_t)(w.nodeCap - 1 - i);
  QT_TEAM_SYNC(tm);

  bool bFinish = false, enterPhase = false;
#if QT_DEVICE
  // scratch shared by the team sort's range lists and qt_std_sort_wave's stack (192 ints): a team uses all
  // of it, the unsynchronised waves of a packed workgroup a quarter each
  __shared__ int qtShared[1024];   // (a fixed size: the team sort's lists take 1010 ints and a packed workgroup's four waves 256 each, whatever QT_TEAM_WAVES an A/B build sets —
                                   // with 64 * QT_TEAM_WAVES a -DQT_TEAM_WAVES=4 build wrote past the array and, once in two runs, never came back)
  int* const shWave = qtShared + (tm.nw > 1 ? 0 : 256 * (int)(threadIdx.x >> 6));
  int ff = -1;
#ifndef QT_FAST_FORWARD
#define QT_FAST_FORWARD 1
#endif
  if (QT_FAST_FORWARD) {
    FfGeom fg;
    fg.hX = hX; fg.last = nIni - 1; fg.width = width; fg.height = height; fg.nIni = nIni;
    ff = qt_fast_forward(w, s, nkeys, fg, N, tm, reinterpret_cast<int*>(w.vB), 2 * w.nodeCap, reinterpret_cast<int*>(w.brank), w.nodeCap);
    bFinish = ff == QT_FF_FINISH;
    enterPhase = ff == QT_FF_PHASE;
  }
  if (ff < 0)
#elif defined(QT_HOST_FAST_FORWARD)
  int ffHost = -1;
  if (w.hostFastForward) {
    FfGeom fg;
    fg.hX = hX; fg.last = nIni - 1; fg.width = width; fg.height = height; fg.nIni = nIni;
    ffHost = qt_fast_forward_host(w, s, nkeys, fg, N);
    bFinish = ffHost == QT_FF_FINISH;
    enterPhase = ffHost == QT_FF_PHASE;
  }
  if (ffHost < 0)
#endif
  {
    // initial nodes, pushed BACK in order i = 0..nIni-1 (:555-567); keys go to node (int)(x / hX) (:570-573);
    // empty initial nodes are erased (:577-585).  nIni <= 4 (aspect ratio < 4.5:1) is enforced by the caller.
    {
      uint32_t cnt[4] = {0, 0, 0, 0};
      const int last = nIni - 1;
      auto rootOf = [hX, last](uint32_t k) -> int { int g = (int)((float)key_x(k) / hX); return g > last ? last : g; };
#if QT_DEVICE
      if (tm.nw > 1) qt_partition_team(w.keys, w.tmp, 0, nkeys, rootOf, cnt, tm);
      else qt_partition(w.keys, w.tmp, 0, nkeys, rootOf, cnt);
#else
      qt_partition(w.keys, w.tmp, 0, nkeys, rootOf, cnt);
#endif
      int live = 0;
      for (int i = 0; i < nIni; ++i) live += cnt[i] > 0 ? 1 : 0;
      s.head = w.listCap - live;
      s.size = live;
      int p = s.head;
      uint32_t begin = 0;
      for (int i = 0; i < nIni; ++i) {
        if (cnt[i] > 0) {
          const int id = qt_alloc(w, s);
          if (QT_LANE0 && w0) {
            Node nd;
            nd.x0 = (int16_t)(int)(hX * (float)i);
            nd.x1 = (int16_t)(int)(hX * (float)(i + 1));
            nd.y0 = 0;
            nd.y1 = (int16_t)height;
            nd.begin = begin; nd.count = cnt[i]; nd.lit = (uint16_t)p; nd.noMore = (cnt[i] == 1) ? 1 : 0;
            w.nodes[id] = nd;
            w.list[p] = (uint16_t)id;
          }
          ++p;
        }
        begin += cnt[i];
      }
      QT_TEAM_SYNC(tm);
    }
  }

  QT_T0();
  while (!bFinish) {
    if (!enterPhase) {
#if QT_DEVICE
    // the sweep visits the list from the (compacted) head on — children are pushed in front of it: not visited — and divides every
    // node that is not bNoMore, with no early exit (:589-655): collect them in list order (wave 0), then split them at once (the team)
    int m = 0;
    if (w0) {
      qt_compact(w, s);
      QT_MARK(13);
      const uint64_t lt = QT_LANE == 0 ? 0ull : (~0ull >> (64 - QT_LANE));
      for (int pos0 = s.head; pos0 < w.listCap; pos0 += 64) {
        const int pos = pos0 + QT_LANE;
        int id = 0xFFFF;
        bool todo = false;
        if (pos < w.listCap) {
          id = w.list[pos];
          if (id != 0xFFFF) todo = !w.nodes[id].noMore;
        }
        const uint64_t mk = __ballot(todo);
        if (todo) w.order[m + __popcll(mk & lt)] = (uint16_t)id;
        m += __popcll(mk);
      }
      if (tm.nw > 1 && QT_LANE0) { tm.sh[8] = s.head; tm.sh[9] = m; }
    }
    if (tm.nw > 1) { __syncthreads(); s.head = tm.sh[8]; m = tm.sh[9]; }
    const int prevSize = s.size;
    int nToExpand = 0;
    s.nA = 0;
    qt_split_batch(w, s, m, -1, &nToExpand, tm);
#else
    qt_compact(w, s);
    const int prevSize = s.size;
    int nToExpand = 0;
    s.nA = 0;
    const int oldHead = s.head;
    for (int pos = oldHead; pos < w.listCap; ++pos) {  // children are pushed in front of oldHead: not visited
      const uint16_t id = w.list[pos];
      if (id == 0xFFFF) continue;
      if (w.nodes[id].noMore) continue;
      qt_split(w, s, id, &nToExpand);
    }
#endif
    QT_MARK(14);
    if (s.size >= N || s.size == prevSize) bFinish = true;
    else if (s.size + nToExpand * 3 > N) enterPhase = true;
    }
    if (enterPhase && !bFinish) {
      enterPhase = false;
      while (!bFinish) {
        const int prevSize2 = s.size;
        // vPrev = vSize; vSize.clear(); sort(vPrev)
        const int nPrev = s.nA;
        s.nA = 0;
#if QT_DEVICE
        QT_TEAM_SYNC(tm);
        for (int i = tm.tw * 64 + QT_LANE; i < nPrev; i += tm.nw * 64) w.vB[i] = w.vA[i];
        QT_TEAM_SYNC(tm);
        QT_MARK(14);
        // vA was just cleared: free as scratch until the splits below refill it
        bool sorted = false;
        if (tm.nw > 1) sorted = qt_std_sort_team(w.vB, w.vA, nPrev, w.order, (uint16_t*)w.brank, tm, qtShared);
        if (!sorted && w0) qt_std_sort_wave(w.vB, w.vA, nPrev, w.order, (uint16_t*)w.brank, shWave);
        QT_MARK(15);
        if (w0) {
          // compaction keeps ids stable (vB holds ids), only Node::lit moves
          qt_compact(w, s);
          for (int j = QT_LANE; j < nPrev; j += 64) w.order[j] = (uint16_t)(w.vB[nPrev - 1 - j] & 0xFFFF);   // largest first
          if (tm.nw > 1 && QT_LANE0) tm.sh[8] = s.head;
        }
        if (tm.nw > 1) { __syncthreads(); s.head = tm.sh[8]; }
        qt_split_batch(w, s, nPrev, N, nullptr, tm);
#else
        for (int i = 0; i < nPrev; ++i) w.vB[i] = w.vA[i];
        qt_std_sort(w.vB, nPrev);
        qt_compact(w, s);
        for (int j = nPrev - 1; j >= 0; --j) {
          const int id = (int)(w.vB[j] & 0xFFFF);
          qt_split(w, s, id, nullptr);
          if (s.size >= N) break;
        }
#endif
        if (s.size >= N || s.size == prevSize2) bFinish = true;
      }
    }
  }

  // retain the best point in each node, list order (:716-737)
  QT_TEAM_SYNC(tm);
  QT_MARK(14);
  int nOut = 0;
#if QT_DEVICE
  // wave 0 lists the live nodes in list order; then one node per lane over the whole team (most nodes hold a handful of keys),
  // nodes with many keys are finished by the whole wave that owns them
  if (w0) {
    const uint64_t lt = QT_LANE == 0 ? 0ull : (~0ull >> (64 - QT_LANE));
    for (int pos0 = s.head; pos0 < w.listCap; pos0 += 64) {
      const int pos = pos0 + QT_LANE;
      int id = 0xFFFF;
      if (pos < w.listCap) id = w.list[pos];
      const uint64_t mlive = __ballot(id != 0xFFFF);
      if (id != 0xFFFF) w.order[nOut + __popcll(mlive & lt)] = (uint16_t)id;
      nOut += __popcll(mlive);
    }
    if (tm.nw > 1 && QT_LANE0) tm.sh[10] = nOut;
  }
  QT_TEAM_SYNC(tm);
  if (tm.nw > 1) nOut = tm.sh[10];
  for (int q0 = 0; q0 * tm.nw < nOut; q0 += 64) {
    const int rank = (q0 + QT_LANE) * tm.nw + tm.tw;
    const bool live = rank < nOut;
    uint32_t begin = 0, count = 0;
    if (live) { const int id = w.order[rank]; begin = w.nodes[id].begin; count = w.nodes[id].count; }
    const bool big = live && count > 32;
    if (live && !big) {
      uint32_t bk = w.keys[begin];
      for (uint32_t k = 1; k < count; ++k) { const uint32_t kk = w.keys[begin + k]; if (key_r(kk) > key_r(bk)) bk = kk; }
      if (rank < outCap) out[rank] = bk;
    }
    uint64_t mb = __ballot(big);
    while (mb) {
      const int b = __ffsll((unsigned long long)mb) - 1;
      mb &= mb - 1;
      const uint32_t bk = qt_best_key(w.keys, (uint32_t)__builtin_amdgcn_readlane((int)begin, b), (uint32_t)__builtin_amdgcn_readlane((int)count, b));
      const int r = (q0 + b) * tm.nw + tm.tw;
      if (QT_LANE0 && r < outCap) out[r] = bk;
    }
  }
#else
  for (int pos = s.head; pos < w.listCap; ++pos) {
    const uint16_t id = w.list[pos];
    if (id == 0xFFFF) continue;
    const Node nd = w.nodes[id];
    const uint32_t bk = qt_best_key(w.keys, nd.begin, nd.count);
    if (nOut < outCap) out[nOut] = bk;
    ++nOut;
  }
#endif
  QT_TEAM_SYNC(tm);
  QT_MARK(25);
  return nOut;
}

}  // namespace morbqt
