// Frames sharded over several GPUs from C++, without torch: the loop of INTEGRATION.md section 5 compiled and run.
// A stream of stereo frames is dealt round-robin over N "GPUs" (global frame g -> GPU g % N, slot g / N); every GPU extracts its
// frames, converts the left images' descriptors (ComputeBoW), packs the left images' features into ONE slab (morb_feature_slab_pack),
// the slab travels one GPU up the ring with hipMemcpyPeerAsync, and every frame is matched against its predecessor with
// SearchByBoW on the [own; received] pool.  The box this runs on has one GPU, so the N virtual GPUs are N sets of handles, streams
// and buffers on device 0 and the peer copy is device 0 -> device 0: the code path is the multi-GPU one, the xGMI link is not there.
// The match table of every global frame must be the same for N = 1, 2 and 3.
//   usage: shard_ring_check <dir>   (dir holds dims.bin = {W, H, nfeat, frames, k, L, levelsup} int32, imgs.bin [frames][2][H][W] u8,
//                                    voc_desc.bin [nodes][32] u8, voc_first.bin [nodes] int32; writes out_match.bin / out_nmatch.bin)
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/morb_hip.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
#define MCK(x) do { int r_ = (x); if (r_ < 0) { std::fprintf(stderr, "%s:%d %s -> %d\n", __FILE__, __LINE__, #x, r_); std::exit(3); } } while (0)

template <class T>
static std::vector<T> slurp(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(4); }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<T> v((size_t)n / sizeof(T));
  if (std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) std::exit(4);
  std::fclose(f);
  return v;
}
template <class T>
static T* dalloc(size_t n) { void* p = nullptr; HIPCK(hipMalloc(&p, n * sizeof(T) + 16)); return (T*)p; }
template <class T>
static T* dupload(const std::vector<T>& h) { T* d = dalloc<T>(h.size()); HIPCK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

struct Gpu {   // everything one GPU of the node owns
  int dev = 0;
  hipStream_t st = nullptr;
  hipEvent_t packed = nullptr;
  morb_extractor* ext = nullptr;
  morb_matcher* m = nullptr;
  uint8_t *images = nullptr, *desc = nullptr, *poolDesc = nullptr, *hasMP = nullptr, *vocDesc = nullptr;
  morb_keypoint *kps = nullptr, *poolKps = nullptr;
  int *cnt = nullptr, *cntLeft = nullptr, *mono = nullptr, *word = nullptr, *node = nullptr, *poolNode = nullptr, *poolCnt = nullptr, *vocFirst = nullptr;
  int *leftRows = nullptr, *rows0 = nullptr, *rowsS = nullptr, *kfRow = nullptr, *fRow = nullptr, *match = nullptr, *nmatch = nullptr;
  void *send = nullptr, *recv = nullptr;
};

static bool has_mp(int g, int i) { return ((unsigned)(g * 131 + i * 7) % 5u) != 0u; }   // 80 % of a frame's features hold a MapPoint

int main(int argc, char** argv) {
  if (argc < 2) return 64;
  const std::string dir = argv[1];
  const std::vector<int> dims = slurp<int>(dir + "/dims.bin");
  const int W = dims[0], H = dims[1], nfeat = dims[2], F = dims[3], VK = dims[4], VL = dims[5], VUP = dims[6];
  const std::vector<uint8_t> imgs = slurp<uint8_t>(dir + "/imgs.bin");
  const std::vector<uint8_t> vocDesc = slurp<uint8_t>(dir + "/voc_desc.bin");
  const std::vector<int> vocFirst = slurp<int>(dir + "/voc_first.bin");
  if (imgs.size() != (size_t)F * 2 * W * H) { std::fprintf(stderr, "imgs.bin: size\n"); return 4; }
  int ndev = 0;
  HIPCK(hipGetDeviceCount(&ndev));
  std::vector<std::vector<int>> tables;   // per N: [F][cap] match rows
  std::vector<std::vector<int>> counts;   // per N: [F]
  int cap = 0;
  const int worlds[3] = {1, 2, 3};
  for (int wi = 0; wi < 3; ++wi) {
    const int N = worlds[wi];
    if (F % N) { std::fprintf(stderr, "frames must divide by %d\n", N); return 5; }
    const int S = F / N;   // frames per GPU (one step)
    std::vector<Gpu> g(N);
    for (int d = 0; d < N; ++d) {
      Gpu& G = g[d];
      G.dev = d % ndev;   // (a node with >= N GPUs gives every rank its own)
      HIPCK(hipSetDevice(G.dev));
      HIPCK(hipStreamCreateWithFlags(&G.st, hipStreamNonBlocking));
      HIPCK(hipEventCreateWithFlags(&G.packed, hipEventDisableTiming));
      MCK(morb_extractor_create(&G.ext, nfeat, 1.2f, 8, 20, 7, G.dev));
      MCK(morb_matcher_create(&G.m, G.dev));
      cap = morb_extractor_max_keypoints(G.ext);
      // this GPU's frames: slot s holds global frame s * N + d; image 2 s = left, 2 s + 1 = right
      std::vector<uint8_t> mine((size_t)S * 2 * W * H);
      for (int s = 0; s < S; ++s)
        std::copy(imgs.begin() + (size_t)(s * N + d) * 2 * W * H, imgs.begin() + (size_t)(s * N + d + 1) * 2 * W * H, mine.begin() + (size_t)s * 2 * W * H);
      G.images = dupload(mine);
      G.kps = dalloc<morb_keypoint>((size_t)2 * S * cap); G.desc = dalloc<uint8_t>((size_t)2 * S * cap * 32);
      G.cnt = dalloc<int>(2 * S); G.cntLeft = dalloc<int>(2 * S); G.mono = dalloc<int>(2 * S);
      G.word = dalloc<int>((size_t)2 * S * cap); G.node = dalloc<int>((size_t)2 * S * cap);
      G.poolKps = dalloc<morb_keypoint>((size_t)2 * S * cap); G.poolDesc = dalloc<uint8_t>((size_t)2 * S * cap * 32);
      G.poolNode = dalloc<int>((size_t)2 * S * cap); G.poolCnt = dalloc<int>(2 * S);
      G.vocDesc = dupload(vocDesc); G.vocFirst = dupload(vocFirst);
      HIPCK(hipMalloc(&G.send, morb_feature_slab_bytes(S, cap))); HIPCK(hipMalloc(&G.recv, morb_feature_slab_bytes(S, cap)));
      std::vector<int> leftRows(S), rows0(S), rowsS(S), kf(S), fr(S);
      const int prev = (d + N - 1) % N;
      std::vector<uint8_t> hm((size_t)2 * S * cap);
      for (int s = 0; s < S; ++s) {
        leftRows[s] = 2 * s; rows0[s] = s; rowsS[s] = S + s;
        const int gl = s * N + d;
        // pool rows: [0, S) own frames, [S, 2 S) the previous GPU's frames of the same slot (parallel.neighbour_pairs)
        kf[s] = gl == 0 ? s : N == 1 ? s - 1 : d > 0 ? S + s : S + s - 1;
        fr[s] = s;
        for (int i = 0; i < cap; ++i) { hm[(size_t)s * cap + i] = has_mp(gl, i); hm[(size_t)(S + s) * cap + i] = has_mp(s * N + prev, i); }
      }
      G.leftRows = dupload(leftRows); G.rows0 = dupload(rows0); G.rowsS = dupload(rowsS); G.kfRow = dupload(kf); G.fRow = dupload(fr);
      G.hasMP = dupload(hm);
      G.match = dalloc<int>((size_t)S * cap); G.nmatch = dalloc<int>(S);
    }
    // ---- the step: extraction, ComputeBoW and the pack on every GPU ...
    for (int d = 0; d < N; ++d) {
      Gpu& G = g[d];
      HIPCK(hipSetDevice(G.dev));
      MCK(morb_extract_batch(G.ext, G.images, 2 * S, W, H, W, (size_t)W * H, nullptr, G.kps, G.desc, cap, G.cnt, G.mono, G.st));
      // Frame::ComputeBoW converts the LEFT image's descriptors only: the right images' counts are zeroed for the BoW kernels
      std::vector<int> c(2 * S);
      HIPCK(hipMemcpyAsync(c.data(), G.cnt, sizeof(int) * 2 * S, hipMemcpyDeviceToHost, G.st));
      HIPCK(hipStreamSynchronize(G.st));
      MCK(morb_extractor_status(G.ext, nullptr));
      for (int s = 0; s < S; ++s) c[2 * s + 1] = 0;
      HIPCK(hipMemcpyAsync(G.cntLeft, c.data(), sizeof(int) * 2 * S, hipMemcpyHostToDevice, G.st));
      HIPCK(hipStreamSynchronize(G.st));   // (c leaves scope)
      MCK(morb_bow_transform_batch(G.m, 2 * S, G.desc, G.cntLeft, cap, G.vocDesc, G.vocFirst, VK, VL, VUP, G.word, G.node, G.st));
      MCK(morb_feature_slab_pack(G.m, S, cap, G.leftRows, G.kps, G.desc, G.node, G.cntLeft, G.send, G.st));
      HIPCK(hipEventRecord(G.packed, G.st));
    }
    // ---- ... the ring step: GPU d's slab goes to GPU (d + 1) % N, point to point ...
    for (int d = 0; d < N; ++d) {
      const int nx = (d + 1) % N;
      HIPCK(hipSetDevice(g[nx].dev));
      HIPCK(hipStreamWaitEvent(g[nx].st, g[d].packed, 0));
      HIPCK(hipMemcpyPeerAsync(g[nx].recv, g[nx].dev, g[d].send, g[d].dev, morb_feature_slab_bytes(S, cap), g[nx].st));
    }
    // ---- ... and the cross-frame matcher on the [own; received] pool
    tables.emplace_back((size_t)F * cap, -2);
    counts.emplace_back(F, -1);
    for (int d = 0; d < N; ++d) {
      Gpu& G = g[d];
      HIPCK(hipSetDevice(G.dev));
      MCK(morb_feature_slab_unpack(G.m, S, cap, G.send, G.rows0, G.poolKps, G.poolDesc, G.poolNode, G.poolCnt, G.st));
      MCK(morb_feature_slab_unpack(G.m, S, cap, G.recv, G.rowsS, G.poolKps, G.poolDesc, G.poolNode, G.poolCnt, G.st));
      MCK(morb_search_by_bow_batch(G.m, S, G.kfRow, G.fRow, 2 * S, G.poolKps, G.poolDesc, G.poolNode, G.poolCnt, G.hasMP, cap, 0.7f, 1, G.match, G.nmatch, G.st));
      std::vector<int> mt((size_t)S * cap), nm(S);
      HIPCK(hipMemcpyAsync(mt.data(), G.match, sizeof(int) * mt.size(), hipMemcpyDeviceToHost, G.st));
      HIPCK(hipMemcpyAsync(nm.data(), G.nmatch, sizeof(int) * S, hipMemcpyDeviceToHost, G.st));
      HIPCK(hipStreamSynchronize(G.st));
      for (int s = 0; s < S; ++s) {
        const int gl = s * N + d;
        std::copy(mt.begin() + (size_t)s * cap, mt.begin() + (size_t)(s + 1) * cap, tables.back().begin() + (size_t)gl * cap);
        counts.back()[gl] = nm[s];
      }
    }
    for (int d = 0; d < N; ++d) {
      Gpu& G = g[d];
      HIPCK(hipSetDevice(G.dev));
      morb_extractor_destroy(G.ext); morb_matcher_destroy(G.m);
      void* bufs[] = {G.images, G.desc, G.poolDesc, G.hasMP, G.vocDesc, G.kps, G.poolKps, G.cnt, G.cntLeft, G.mono, G.word, G.node, G.poolNode, G.poolCnt, G.vocFirst,
                      G.leftRows, G.rows0, G.rowsS, G.kfRow, G.fRow, G.match, G.nmatch, G.send, G.recv};
      for (void* b : bufs) HIPCK(hipFree(b));
      HIPCK(hipEventDestroy(G.packed)); HIPCK(hipStreamDestroy(G.st));
    }
  }
  long total = 0;
  for (int gl = 0; gl < F; ++gl) total += counts[0][gl];
  for (int wi = 1; wi < 3; ++wi) {
    for (int gl = 0; gl < F; ++gl) {
      if (counts[wi][gl] != counts[0][gl]) { std::fprintf(stderr, "N = %d: frame %d has %d matches, one GPU gave %d\n", worlds[wi], gl, counts[wi][gl], counts[0][gl]); return 1; }
      for (int i = 0; i < cap; ++i)
        if (tables[wi][(size_t)gl * cap + i] != tables[0][(size_t)gl * cap + i]) { std::fprintf(stderr, "N = %d: frame %d feature %d differs\n", worlds[wi], gl, i); return 1; }
    }
  }
  FILE* f = std::fopen((dir + "/out_match.bin").c_str(), "wb");
  std::fwrite(tables[0].data(), sizeof(int), tables[0].size(), f); std::fclose(f);
  f = std::fopen((dir + "/out_nmatch.bin").c_str(), "wb");
  std::fwrite(counts[0].data(), sizeof(int), counts[0].size(), f); std::fclose(f);
  std::printf("shard ring ok: %d frames, cap %d, %ld matches, identical for 1 / 2 / 3 GPUs (%d device(s) present)\n", F, cap, total, ndev);
  return 0;
}
