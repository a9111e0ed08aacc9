// Device helpers shared by the gfx950 kernels: bf16 conversion, the two MFMA
// precisions behind one 16-byte-fragment interface, Philox4x32-10.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iqlhip {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int WAVE = 64;

// ---------------------------------------------------------------- bf16 ----
// Round-to-nearest-even on the f32 bits (torch's c10::BFloat16 rounding); NaN
// stays NaN.  Returns the 16-bit pattern.
__device__ __forceinline__ uint16_t f2bf(float x) {
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// float -> bf16 -> float
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// ------------------------------------------------------------ precision ----
// One "fragment" is 16 bytes per lane for both precisions.
//   BF16: 8 bf16 = k-slice [8*(lane>>4), +8) of a 32-deep MFMA step
//         (v_mfma_f32_16x16x32_bf16: A[row=lane&15][k], B[k][col=lane&15]).
//   FP32: 4 f32  = k-slice [4*(lane>>4), +4) of a 16-deep macro step executed as
//         four v_mfma_f32_16x16x4_f32; component c of every lane feeds MFMA c,
//         i.e. the K index is permuted identically for A and B (exact f32).
// C/D layout (both): col = lane&15, row = 4*(lane>>4) + reg.
template <bool BF16>
struct Prec;

template <>
struct Prec<true> {
  typedef uint16_t T;
  static constexpr int KM = 32;  // K covered by one fragment pair
  static constexpr int EPV = 8;  // elements per 16-byte fragment
  static __device__ __forceinline__ T from_f32(float x) { return f2bf(x); }
  static __device__ __forceinline__ float to_f32(T x) { return bf2f(x); }
  static __device__ __forceinline__ float round(float x) { return rbf(x); }
  static __device__ __forceinline__ void mma(const uint4 &a, const uint4 &b, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};

template <>
struct Prec<false> {
  typedef float T;
  static constexpr int KM = 16;
  static constexpr int EPV = 4;
  static __device__ __forceinline__ T from_f32(float x) { return x; }
  static __device__ __forceinline__ float to_f32(T x) { return x; }
  static __device__ __forceinline__ float round(float x) { return x; }
  static __device__ __forceinline__ void mma(const uint4 &a, const uint4 &b, f32x4 &acc) {
    const float4 af = __builtin_bit_cast(float4, a), bf = __builtin_bit_cast(float4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc, 0, 0, 0);
  }
};

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// --------------------------------------------------------------- Philox ----
// Philox4x32-10; stream definition in oracle/philox.py.
struct Philox4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
constexpr uint32_t STREAM_INDEX = 0, STREAM_DROPOUT1 = 1, STREAM_DROPOUT2 = 2;

__device__ __forceinline__ int64_t philox_index(uint64_t seed, uint64_t step, uint32_t row,
                                                uint64_t n_rows) {
  const Philox4 r = philox4x32_10(row, (uint32_t)step, (uint32_t)(step >> 32), STREAM_INDEX,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
  return (int64_t)((uint64_t)r.x % n_rows);
}

}  // namespace iqlhip
