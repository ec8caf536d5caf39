// Shared helpers for libslic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/slic_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

void slic_set_error(const char* fmt, ...);

#define SLIC_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      slic_set_error(__VA_ARGS__);              \
      return SLIC_EINVAL;                       \
    }                                           \
  } while (0)

#define SLIC_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      slic_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return SLIC_EHIP;                                                             \
    }                                                                               \
  } while (0)

#define SLIC_LAUNCH_CHECK() SLIC_HIP_CHECK(hipGetLastError())

static inline size_t slic_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t slic_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// carve consecutive 256-byte aligned regions out of a caller workspace
struct SlicCarver {
  char* p;
  size_t off;
  explicit SlicCarver(void* base) : p((char*)base), off(0) {}
  template <typename T>
  T* take(size_t n) {
    T* r = (T*)(p + off);
    off += slic_align_up(n * sizeof(T), 256);
    return r;
  }
};
