// Can the FP64 matrix core carry PoseOptimization's edge-order sums?  g2o adds the edges' contributions to H, b and chi2 one edge after the
// other in double precision; k_pose_opt2 repeats that order with one dependent v_add_f64 per term (28 chains side by side in 28 lanes,
// ~13 - 17 cycles per term: tools/micro/f64_chain.hip).  v_mfma_f64_4x4x4_4b_f64 computes D[b][i][j] = C[b][i][j] + sum_k A[b][i][k] B[b][k][j]:
// with B = 1 it adds FOUR terms to 16 independent accumulators (4 blocks x 4 rows) in one 4-pass instruction — IF the hardware adds the four
// products one after the other, each rounded to double (c + a0, + a1, + a2, + a3), which no document states.  This probe
//   1. finds the operand layout with one-hot operands,
//   2. checks the result against every order of a rounded sequential sum on random terms of mixed sign and magnitude (cancellation, ties),
//   3. times dependent MFMAs (one chain, two interleaved chains).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/micro/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_probe(const double* A, const double* B, const double* C, double* D) {
  const size_t o = (size_t)blockIdx.x * 64 + threadIdx.x;
  D[o] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[o], B[o], C[o], 0, 0, 0);
}

__global__ __launch_bounds__(64) void k_time(int n, int mode, double* out, long long* cyc) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0;
  double d0 = 0.0, d1 = 0.5;
  const long long t0 = clock64();
  if (mode == 0) {
    for (int i = 0; i < n; i += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < n; i += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = d0 + d1;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

static double rnd_term() {
  // mixed magnitudes (2^-30 .. 2^30), both signs, full mantissas; now and then an exact negative of nothing in particular
  const int e = rand() % 61 - 30;
  const double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 0x1p-31;
  return std::ldexp((rand() & 1) ? m : -m, e);
}

int main() {
  double *dA, *dB, *dC, *dD;
  const int NP = 64 * 64;
  std::vector<double> hA((size_t)NP * 64, 0.0), hB((size_t)NP * 64, 0.0), hC((size_t)NP * 64, 0.0), hD((size_t)NP * 64);
  CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dB, hA.size() * 8)); CK(hipMalloc(&dC, hA.size() * 8)); CK(hipMalloc(&dD, hA.size() * 8));
  // 1. layout: A one-hot in lane la, B one-hot in lane lb
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) { const size_t o = (size_t)(la * 64 + lb) * 64; hA[o + la] = 1.0; hB[o + lb] = 1.0; }
  CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hA.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, hC.data(), hA.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe, dim3(NP), dim3(64), 0, 0, dA, dB, dC, dD);
  CK(hipMemcpy(hD.data(), dD, hA.size() * 8, hipMemcpyDeviceToHost));
  int outLane[64][64];
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
    outLane[la][lb] = -1;
    for (int l = 0; l < 64; ++l) if (hD[(size_t)(la * 64 + lb) * 64 + l] != 0.0) outLane[la][lb] = l;
  }
  printf("layout (A lane -> the B lanes it meets : the D lanes they land in)\n");
  for (int la = 0; la < 64; la += 1) {
    if (la >= 8 && la < 56 && (la % 16) > 1) continue;
    printf("  A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb) if (outLane[la][lb] >= 0) printf("  B %2d -> D %2d", lb, outLane[la][lb]);
    printf("\n");
  }
  // contributors of every D lane when B is all ones: the A lanes whose one-hot reaches it
  std::vector<int> contrib[64];
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (outLane[la][lb] >= 0) {
    auto& v = contrib[outLane[la][lb]];
    if (std::find(v.begin(), v.end(), la) == v.end()) v.push_back(la);
  }
  for (int l = 0; l < 64; ++l) if (contrib[l].size() != 4) { printf("D lane %d has %zu contributors: not the layout this probe expects\n", l, contrib[l].size()); return 1; }
  // 2. order of the additions: B = 1 everywhere, random A and C
  const int NT = 4096;
  srand(12345);
  for (size_t i = 0; i < (size_t)NT * 64; ++i) { hA[i] = rnd_term(); hB[i] = 1.0; hC[i] = (rand() % 8 == 0) ? 0.0 : rnd_term(); }
  // a quarter of the items: terms that nearly cancel the running sum (the rounding of every step matters)
  for (int it = 0; it < NT; it += 4) for (int l = 0; l < 64; ++l) {
    const size_t o = (size_t)it * 64;
    const auto& c = contrib[l];
    hA[o + c[1]] = -(hC[o + l] + hA[o + c[0]]) * (1.0 + 0x1p-40 * (rand() % 1024));
  }
  CK(hipMemcpy(dA, hA.data(), (size_t)NT * 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), (size_t)NT * 64 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, hC.data(), (size_t)NT * 64 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe, dim3(NT), dim3(64), 0, 0, dA, dB, dC, dD);
  CK(hipMemcpy(hD.data(), dD, (size_t)NT * 64 * 8, hipMemcpyDeviceToHost));
  int perm[4] = {0, 1, 2, 3};
  long best = -1; int bestPerm[4] = {0, 0, 0, 0}, bestForm = 0;
  do {
    for (int form = 0; form < 3; ++form) {   // 0: ((((c + a) + a) + a) + a); 1: c + (((a + a) + a) + a); 2: fused exact sum rounded once (long double stand-in)
      long ok = 0;
      for (int it = 0; it < NT; ++it) for (int l = 0; l < 64; ++l) {
        const size_t o = (size_t)it * 64;
        const auto& c = contrib[l];
        volatile double s;
        if (form == 0) { s = hC[o + l]; for (int k = 0; k < 4; ++k) s = s + hA[o + c[perm[k]]]; }
        else if (form == 1) { s = hA[o + c[perm[0]]]; for (int k = 1; k < 4; ++k) s = s + hA[o + c[perm[k]]]; s = hC[o + l] + s; }
        else { long double t = hC[o + l]; for (int k = 0; k < 4; ++k) t += (long double)hA[o + c[perm[k]]]; s = (double)t; }
        ok += std::memcmp((const void*)&s, &hD[o + l], 8) == 0;
      }
      if (ok > best) { best = ok; std::memcpy(bestPerm, perm, sizeof perm); bestForm = form; }
      if (ok == (long)NT * 64) printf("  EXACT: form %d, contributor order %d %d %d %d (A lanes of D lane 0: %d %d %d %d)\n", form, perm[0], perm[1], perm[2], perm[3],
                                       contrib[0][perm[0]], contrib[0][perm[1]], contrib[0][perm[2]], contrib[0][perm[3]]);
    }
  } while (std::next_permutation(perm, perm + 4));
  printf("best hypothesis: form %d order %d %d %d %d: %ld of %ld results bit-equal\n", bestForm, bestPerm[0], bestPerm[1], bestPerm[2], bestPerm[3], best, (long)NT * 64);
  // 3. timing
  double* dOut; long long* dCyc; long long hc;
  CK(hipMalloc(&dOut, 64 * 8)); CK(hipMalloc(&dCyc, 8));
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
    const int n = 4096;
    hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, n, mode, dOut, dCyc);
    CK(hipMemcpy(&hc, dCyc, 8, hipMemcpyDeviceToHost));
    if (rep) printf("%s: %.1f cycles per MFMA (clock64, the unit of f64_chain.hip), four terms of 16 chains each\n", mode ? "two interleaved chains" : "one dependent chain",
                    (double)hc / (n * (mode ? 2 : 1)));
  }
  return 0;
}
