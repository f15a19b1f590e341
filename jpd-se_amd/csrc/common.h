// Shared host/device helpers for libjpdse_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/jpdse.h"

namespace jpdse {

// ---- error plumbing ---------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);

#define JPDSE_REQUIRE(cond, ...)                              \
  do {                                                        \
    if (!(cond)) return ::jpdse::set_error(JPDSE_EINVAL, __VA_ARGS__); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// opt-in timer of the HBM-bound calls (api.cpp; jpdse_prof_hbm_select / _collect): begin returns a slot or -1 (off / full)
int hbm_prof_begin(hipStream_t s);
void hbm_prof_end(int slot, int cls, double algorithmic_bytes, hipStream_t s);   // cls: JPDSE_HBM_INORM_FWD / _BWD / _ADAM
static inline int cpad(int c) { return (c + 7) & ~7; }
static inline size_t esize(int dtype) { return dtype == JPDSE_BF16 ? 2 : 4; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Kernel-selection switches: compile-time constants in the shipped library, run-time variables in the developer build
// (libjpdse_hip_dev.so, -DJPDSE_DEV; set through include/jpdse_dev.h).
#ifdef JPDSE_DEV
#define JPDSE_SWITCH(type, name, value) static type name = value
extern int g_norm_fused;     // norm.hip: InstanceNorm form, 1 = two register-held kernels (default), 0 = three kernels (mode 27), 2 = one kernel (mode 28)
#else
#define JPDSE_SWITCH(type, name, value) static constexpr type name = value
#endif

// ---- device scalar types ------------------------------------------------------------------
typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserving
  return __builtin_bit_cast(bf16_t, h);
}

template <typename T> struct ElemOps;
template <> struct ElemOps<float> {
  static constexpr int kDtype = JPDSE_F32;
  static constexpr int kVec = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemOps<bf16_t> {
  static constexpr int kDtype = JPDSE_BF16;
  static constexpr int kVec = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector of T unpacked to floats and back
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void unpack(u32x4 t, float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(t[i]);
  }
  __device__ static __forceinline__ u32x4 pack(const float (&v)[4]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = __float_as_uint(v[i]);
    return t;
  }
  __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
  // streaming forms (nontemporal hint): the last use of a tensor far larger than the caches
  __device__ static __forceinline__ void load_nt(const float* p, float (&v)[4]) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static __forceinline__ void store_nt(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void unpack(u32x4 t, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(t[i] << 16);
      v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ u32x4 pack(const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      t[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    return t;
  }
  __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(t[i] << 16);
      v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      t[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = t;
  }
  __device__ static __forceinline__ void load_nt(const bf16_t* p, float (&v)[8]) {
    unpack(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)), v);
  }
  __device__ static __forceinline__ void store_nt(bf16_t* p, const float (&v)[8]) {
    __builtin_nontemporal_store(pack(v), reinterpret_cast<u32x4*>(p));
  }
};

// last read of a large tensor in a streaming kernel (nontemporal hint; -DJPDSE_NO_NT builds the A/B variant without it)
#ifndef JPDSE_NO_NT
#define JPDSE_LOAD_LAST(T, ptr, v) Vec16<T>::load_nt(ptr, v)
#else
#define JPDSE_LOAD_LAST(T, ptr, v) Vec16<T>::load(ptr, v)
#endif

// ---- InstanceNorm moments written by conv epilogues (conv_rows.h, dgrad2_rows.h, thin_fwd.h -> norm.hip) ----------------
// Every block writes, per (image, channel), ONE slot = (mean, M2) of the values it produced -- of the bf16-ROUNDED values,
// the ones that are stored and that the norm's backward re-reads -- with M2 = sum (v - mean)^2.  Each lane sums about its own
// pilot (its first value; no cross-lane operation inside the hand-scheduled loops), lanes that share a channel are merged
// pairwise with Chan's formula after the loop, so neither the sums nor the merge of the slots
// (finalize_slots_kernel: Chan's parallel-variance formula, all slots cover the same number of pixels) ever forms
// E[v^2] - mean^2 of un-shifted values: a channel with |mean| >> std keeps its variance.
__device__ __forceinline__ float bf16_round(float v) { return bf2f(f2bf(v)); }
// Chan's merge of two (mean, M2) pairs that cover `n_each` values each: (mean, m2) <- merged over 2 n_each values
__device__ __forceinline__ void chan_merge_equal(float& mean, float& m2, float mean_o, float m2_o, float n_each) {
  const float d = mean_o - mean;
  m2 = m2 + m2_o + d * d * (0.5f * n_each);
  mean = mean + 0.5f * d;
}
// (sum, sum of squares) about `pilot` over `count` values -> (mean, M2)
__device__ __forceinline__ void shifted_to_mean_m2(float s1, float s2, float pilot, float count, float& mean, float& m2) {
  const float d = s1 / count;
  mean = pilot + d;
  m2 = fmaxf(s2 - s1 * d, 0.f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// grid-stride launch geometry for HBM-bound kernels: <= 256 CUs x 8 blocks
static inline int ew_blocks(int64_t work_items, int threads = 256) {
  int64_t b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

}  // namespace jpdse
