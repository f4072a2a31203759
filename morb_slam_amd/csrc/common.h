// Shared host-side helpers for libmorb_hip.so (error reporting, HIP status checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "morb_hip.h"

namespace morb {

std::string& last_error();
void set_error(const char* fmt, ...);

#define MORB_HIP_CHECK(expr)                                                                        \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      ::morb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MORB_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

#define MORB_REQUIRE(cond, code, msg)                                   \
  do {                                                                  \
    if (!(cond)) {                                                      \
      ::morb::set_error("%s (%s:%d)", msg, __FILE__, __LINE__);         \
      return code;                                                      \
    }                                                                   \
  } while (0)

static inline int div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace morb
