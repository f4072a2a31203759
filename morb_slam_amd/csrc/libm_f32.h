// Single-precision libm functions of glibc 2.35 (the reference's CPU libm: KannalaBrandt8.cpp calls atan2f / tanf and the float
// overloads of cos / sin) restated operation for operation, so that the KB8 projection / unprojection on the device returns the
// same bits as the reference's — and with them the same match indices, frustum flags and outlier flags:
//   atanf   sysdeps/ieee754/flt-32/s_atanf.c    (fdlibm: argument reduction to [0, 7/16], odd / even polynomial)
//   atan2f  sysdeps/ieee754/flt-32/e_atan2f.c   (fdlibm: quadrant logic around atanf(|y / x|))
//   tanf    sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c  (fdlibm kernel behind the double-precision reduce_fast of sincosf.h; |x| < 120)
//   sinf / cosf  sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h (double-precision polynomials; restated for |x| < 120)
// tools/check_libm_f32.cc compares every one of them against this container's libm: atanf over ALL floats, tanf, sinf and cosf over
// all floats in [-8, 8], atan2f over 2^32 structured + random pairs — bit-equal.
// Everything is plain IEEE arithmetic; the library is built with -ffp-contract=off, which these sequences rely on.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MORB_LIBM_FN __host__ __device__ __forceinline__
#else
#define MORB_LIBM_FN inline
#endif

namespace morbm {

MORB_LIBM_FN uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MORB_LIBM_FN float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MORB_LIBM_FN float fabsf_(float x) { return u2f(f2u(x) & 0x7fffffffu); }

MORB_LIBM_FN float atanf_glibc(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                        6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {   // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;   // NaN
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {   // |x| < 0.4375
    if (ix < 0x31000000) return x;   // |x| < 2^-29
    id = -1;
  } else {
    x = fabsf_(x);
    if (ix < 0x3f980000) {   // |x| < 1.1875
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }   // 7/16 <= |x| < 11/16
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                          // 11/16 <= |x| < 19/16
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }   // |x| < 2.4375
      else { id = 3; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -r : r;
}

MORB_LIBM_FN float atan2f_glibc(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   // NaN
  if (hx == 0x3f800000) return atanf_glibc(y);            // x = 1
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);      // 2 sign(x) + sign(y)
  if (iy == 0) {
    switch (m) {
      case 0: case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = atanf_glibc(fabsf_(y / x));
  switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

// __kernel_tanf(x, y, iy): tan(x + y) for iy = 1, -1 / tan(x + y) for iy = -1, |x| <= pi/4
MORB_LIBM_FN float kernel_tanf_glibc(float x, float y, int iy) {
  const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
  const float T[13] = {3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f, 3.5920790397e-03f,
                       1.4562094584e-03f, 5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f, 7.1407252108e-05f, -1.8558637748e-05f,
                       2.5907305826e-05f};
  const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff;
  if (ix < 0x39000000) {   // |x| < 2^-13
    if ((int)x == 0) {
      if ((ix | (iy + 1)) == 0) return 1.0f / fabsf_(x);
      else if (iy == 1) return x;
      else return -1.0f / (x + y);
    }
  }
  if (ix >= 0x3f2ca140) {   // |x| >= 0.6744
    if (hx < 0) { x = -x; y = -y; }
    const float z = pio4 - x, w = pio4lo - y;
    x = z + w; y = 0.0f;
    if (fabsf_(x) < 0x1p-13f) return (float)(1 - ((hx >> 30) & 2)) * (float)iy * (1.0f - 2.0f * (float)iy * x);
  }
  float z = x * x, w = z * z;
  float r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
  float v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
  float s = z * x;
  r = y + z * (s * (r + v) + y);
  r += T[0] * s;
  w = x + r;
  if (ix >= 0x3f2ca140) {
    v = (float)iy;
    return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
  }
  if (iy == 1) return w;
  // -1 / (x + r) accurately
  z = u2f(f2u(w) & 0xfffff000u);
  v = r - (z - x);
  const float a = -1.0f / w;
  const float t = u2f(f2u(a) & 0xfffff000u);
  s = 1.0f + t * z;
  return t + a * (s + t * v);
}
// tanf (s_tanf.c of glibc 2.35, read off this libm's disassembly): |x| <= pi/4 goes to the kernel directly; otherwise the
// argument is reduced in DOUBLE with sincosf.h's reduce_fast (n = round(x 2/pi), x - n pi/2) and the remainder is handed to the
// float kernel as a head / tail pair.  Restated for |x| < 120; NaN beyond (KannalaBrandt8::unproject passes [0, pi/2]).
MORB_LIBM_FN float tanf_glibc(float x) {
  const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff;
  if (ix <= 0x3f490fda) return kernel_tanf_glibc(x, 0.0f, 1);
  if (((f2u(x) >> 20) & 0x7ff) > 0x42e) return u2f(0x7fc00000u);
  double xd = (double)x;
  const double r = xd * 0x1.45F306DC9C883p+23;
  const int n = ((int32_t)r + 0x800000) >> 24;
  xd = xd - n * 0x1.921FB54442D18p0;
  const float y0 = (float)xd, y1 = (float)(xd - (double)y0);
  return kernel_tanf_glibc(y0, y1, 1 - ((n & 1) << 1));
}

// sinf / cosf (sincosf.h: __sincosf_table, sinf_poly, reduce_fast); valid for |y| < 120
// glibc keeps two tables (__sincosf_table[0 / 1]) that differ only in the sign of the cosine coefficients, and a sign[4] array indexed by
// the quadrant.  Both are restated as arithmetic: a per-thread array with a dynamic index ends up in LDS or scratch on the GPU (k_describe
// carried 8 KB of LDS and 23 LDS instructions per wave for it).  Negating every coefficient of the cosine polynomial negates its value
// exactly (IEEE rounding is symmetric), so "table 1" is "table 0, cosine negated"; sign[q] = {1, -1, -1, 1}[q] = -1 iff (q + 1) & 2.
struct SinCosTab { double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
MORB_LIBM_FN double sc_poly(double x, double x2, const SinCosTab& p, int n) {
  if ((n & 1) == 0) {
    const double x3 = x * x2, s1 = p.s2 + x2 * p.s3, x7 = x3 * x2, s = x + x3 * p.s1;
    return s + x7 * s1;
  }
  const double x4 = x2 * x2, c2 = p.c3 + x2 * p.c4, c1 = p.c0 + x2 * p.c1, x6 = x4 * x2, c = c1 + x4 * p.c2;
  const double v = c + x6 * c2;
  return (n & 2) ? -v : v;   // (__sincosf_table[1]: the same polynomial with every coefficient negated)
}
MORB_LIBM_FN void sincosf_glibc(float y, float* sn, float* cs) {
  const SinCosTab t0 = {0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0,
                        -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10,
                        0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7,
                        -0x1.994eb3774cf24p-13};
  const uint32_t top = (f2u(y) >> 20) & 0x7ff;
  const uint32_t topPio4 = (f2u(0x1.921FB6p-1f) >> 20) & 0x7ff;
  const uint32_t topTiny = (f2u(0x1p-12f) >> 20) & 0x7ff;
  double x = (double)y;
  if (top < topPio4) {
    const double x2 = x * x;
    if (top < topTiny) { *cs = 1.0f; *sn = y; return; }
    *cs = (float)sc_poly(x, x2, t0, 1);
    *sn = (float)sc_poly(x, x2, t0, 0);
    return;
  }
  const double r = x * t0.hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  x = x - n * t0.hpi;
  const double s = ((n + 1) & 2) ? -1.0 : 1.0;
  // glibc: p = &__sincosf_table[(n >> 1) & 1]; cos = poly(x * s, x2, p, n ^ 1), sin = poly(x * s, x2, p, n) — the table index is bit 1 of n,
  // which sc_poly applies to the cosine-type branch
  const int tb = n & 2;
  *cs = (float)sc_poly(x * s, x * x, t0, ((n ^ 1) & 1) | tb);
  *sn = (float)sc_poly(x * s, x * x, t0, (n & 1) | tb);
}
MORB_LIBM_FN float sinf_glibc(float y) { float s, c; sincosf_glibc(y, &s, &c); return s; }
MORB_LIBM_FN float cosf_glibc(float y) { float s, c; sincosf_glibc(y, &s, &c); return c; }

}  // namespace morbm
