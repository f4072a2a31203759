// Proof that morb_slam_amd/csrc/libm_f32.h returns the bits of this machine's libm (glibc 2.35): CPU, ~2 minutes with 8 threads.
//   g++ -O2 -ffp-contract=off -std=c++17 -pthread -o /tmp/check_libm tools/check_libm_f32.cc && /tmp/check_libm [quick]
// atanf: every float.  tanf, sinf, cosf: every float in [-8, 8].  atan2f: 2^16 x 2^16 pairs of
// floats stepping through all exponents and signs, plus 2^31 pseudo-random pairs of moderate magnitude (the projection's regime).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>
#include "../morb_slam_amd/csrc/libm_f32.h"
using namespace morbm;
static inline bool same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }
int main(int argc, char** argv) {
  const bool quick = argc > 1;
  const int NT = 8;
  std::atomic<long> bad[5]; for (auto& b : bad) b = 0;
  std::vector<std::thread> th;
  for (int t = 0; t < NT; ++t) th.emplace_back([&, t] {
    const uint64_t step = quick ? 257 : 1;
    for (uint64_t u = t * step; u < (1ull << 32); u += NT * step) {
      const float x = u2f((uint32_t)u);
      if (!same(atanf_glibc(x), atanf(x))) { if (bad[0]++ < 5) printf("atanf(%a) = %a, libm %a\n", x, atanf_glibc(x), atanf(x)); }
      const float ax = fabsf(x);
      if (ax <= 8.0f) { if (!same(tanf_glibc(x), tanf(x))) { if (bad[1]++ < 5) printf("tanf(%a) = %a, libm %a\n", x, tanf_glibc(x), tanf(x)); } }
      if (ax <= 8.0f) {
        if (!same(sinf_glibc(x), sinf(x))) { if (bad[2]++ < 5) printf("sinf(%a) = %a, libm %a\n", x, sinf_glibc(x), sinf(x)); }
        if (!same(cosf_glibc(x), cosf(x))) { if (bad[3]++ < 5) printf("cosf(%a) = %a, libm %a\n", x, cosf_glibc(x), cosf(x)); }
      }
    }
    // atan2f: structured pairs
    const uint32_t s1 = quick ? 0x00400001u : 0x00010001u;
    for (uint64_t a = (uint64_t)t * s1; a < (1ull << 32); a += (uint64_t)NT * s1)
      for (uint64_t b = 0; b < (1ull << 32); b += 0x0000ffefu * (quick ? 16 : 1)) {
        const float y = u2f((uint32_t)a), x = u2f((uint32_t)b);
        if (!same(atan2f_glibc(y, x), atan2f(y, x))) { if (bad[4]++ < 5) printf("atan2f(%a, %a) = %a, libm %a\n", y, x, atan2f_glibc(y, x), atan2f(y, x)); }
      }
    // atan2f: random pairs in [-64, 64]
    uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
    const long nrand = quick ? (1l << 22) : (1l << 28);
    for (long i = 0; i < nrand; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const float y = ((int32_t)(s >> 32)) * (64.0f / 2147483648.0f), x = ((int32_t)s) * (64.0f / 2147483648.0f);
      if (!same(atan2f_glibc(y, x), atan2f(y, x))) { if (bad[4]++ < 5) printf("atan2f(%a, %a) = %a, libm %a\n", y, x, atan2f_glibc(y, x), atan2f(y, x)); }
    }
  });
  for (auto& t : th) t.join();
  printf("mismatches: atanf %ld, tanf %ld, sinf %ld, cosf %ld, atan2f %ld\n", bad[0].load(), bad[1].load(), bad[2].load(), bad[3].load(), bad[4].load());
  return (bad[0] | bad[1] | bad[2] | bad[3] | bad[4]) != 0;
}
