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
// gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, RNE, NaN preserving).
__device__ __forceinline__ uint16_t f2bf(float x) {
  return __builtin_bit_cast(uint16_t, (__bf16)x);
}
__device__ __forceinline__ float bf2f(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// float -> bf16 -> float
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// ------------------------------------------------------- global accesses ----
// Pointers that the kernels read out of descriptor structs in memory are
// generic ("flat") to the compiler; flat accesses tie LDS and vector-memory
// counters together.  Every access to device memory goes through these helpers,
// which cast to the global address space (global_load / global_store).
#define IQL_AS1 __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T ldg(const T *p) {
  return *(const T IQL_AS1 *)p;
}
template <class T>
__device__ __forceinline__ void stg(T *p, T v) {
  *(T IQL_AS1 *)p = v;
}
// (HIP's uint4/float4 are classes whose copy constructors take generic references,
// which would turn the access back into a flat one: load builtin vectors instead)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ldg16(const void *p) {
  return __builtin_bit_cast(uint4, *(const u32x4_t IQL_AS1 *)p);
}
__device__ __forceinline__ uint2 ldg8(const void *p) {
  return __builtin_bit_cast(uint2, *(const u32x2_t IQL_AS1 *)p);
}
__device__ __forceinline__ void stg16(void *p, float4 v) {
  *(u32x4_t IQL_AS1 *)p = __builtin_bit_cast(u32x4_t, v);
}
// Write-through store (sc0 sc1): the bytes leave for memory as the store executes and the line is
// NOT kept in this XCD's L2 -- for data whose next reader is another kernel on another XCD (or the
// same kernel one step later), so that nothing of it is dirty when the kernel ends
// (MI355X_MICROARCH.md: "stores of each flavour"; nt is not write-through).  Issued as a raw buffer
// store with the cache-policy bits set (aux: bit 0 = sc0, bit 4 = sc1) -- a uniform base in a
// buffer descriptor + a 32-bit per-lane byte offset -- and NOT as inline assembly: the compiler
// must see the instruction to keep the wait state a 16-byte store needs before its data registers
// are overwritten (an asm store without it corrupted the group-launch variant of k_update).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void stg16_wt(const void *base, uint32_t byte_off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), buffer_of(base), byte_off, 0, 0x11);
}
__device__ __forceinline__ void stg16(const void *base, uint32_t byte_off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), buffer_of(base), byte_off, 0, 0);
}
__device__ __forceinline__ void stg8_wt(const void *base, uint32_t byte_off, uint2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), buffer_of(base), byte_off, 0, 0x11);
}
__device__ __forceinline__ void stg4_wt(const void *base, uint32_t byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), buffer_of(base), byte_off, 0, 0x11);
}
__device__ __forceinline__ void stg8(void *p, uint2 v) {
  *(u32x2_t IQL_AS1 *)p = __builtin_bit_cast(u32x2_t, v);
}

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

// ---- cross-lane sums without the LDS crossbar ----
// __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip, >100 cycles, and a butterfly
// is a chain of them).  Within a row of 16 lanes the DPP permutes are plain VALU operands; across
// rows gfx950 has v_permlane16_swap / v_permlane32_swap.  (The clang of ROCm 7.2 returns the
// wrong register for the second result of __builtin_amdgcn_permlane*_swap: inline asm instead.)
template <int CTRL, class T>
__device__ __forceinline__ T dpp_mov(T v) {
  static_assert(sizeof(T) == 4, "32-bit lanes");
  return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <class T>
__device__ __forceinline__ T xor16_sum(T v) {  // v[lane] + v[lane ^ 16]
  T a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
template <class T>
__device__ __forceinline__ T xor32_sum(T v) {  // v[lane] + v[lane ^ 32]
  T a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// v_permlane16_swap_b32 a, b: rows 1 and 3 of a (a row = 16 lanes) trade places with rows 0 and 2 of b.
// With a, b = the packed C-layout registers of two 16-row MFMA tiles that are neighbours along the
// row dimension, a lane of an even row then holds [its own 4 rows of tile a | the next 4 rows of
// tile a] and a lane of an odd row the same of tile b: 16 contiguous bytes per lane of a feature-major
// image, ONE 16-byte store where the plain epilogue issues two 8-byte ones -- these epilogues are
// bound by the number of store instructions, not by their bytes (MI355X_MICROARCH "store tail").
__device__ __forceinline__ void permlane16_swap(uint32_t &a, uint32_t &b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// sum over aligned groups of W consecutive lanes (W a power of two <= 64), result in every lane
template <int W, class T>
__device__ __forceinline__ T lane_sum(T v) {
  if constexpr (W >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (W >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (W >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror (quads already uniform)
  if constexpr (W >= 16) v += dpp_mov<0x140>(v);  // row_mirror (halves already uniform)
  if constexpr (W >= 32) v = xor16_sum(v);
  if constexpr (W >= 64) v = xor32_sum(v);
  return v;
}
// the same for a wave-uniform run-time width
template <class T>
__device__ __forceinline__ T lane_sum_rt(T v, int w) {
  if (w >= 2) v += dpp_mov<0xB1>(v);
  if (w >= 4) v += dpp_mov<0x4E>(v);
  if (w >= 8) v += dpp_mov<0x141>(v);
  if (w >= 16) v += dpp_mov<0x140>(v);
  if (w >= 32) v = xor16_sum(v);
  if (w >= 64) v = xor32_sum(v);
  return v;
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Fragment-major layout of an [F][K] operand matrix (F padded to 16, K to KM).
// The 16-byte fragment lane l needs for (16-row tile ft, k-step ks) sits at
// ((ft*nk + ks)*64 + l)*EPV, so one wave-wide fragment load is ONE contiguous
// 1 KiB read (8 full 128-B lines) instead of 16 half-used lines of a row-major
// image.  Element (f, k): lane = ((k % KM) / EPV) * 16 + f % 16, slot = k % EPV.
// Runs of 4 consecutive k (4-aligned) of one row stay contiguous.
template <class P>
__host__ __device__ inline size_t fidx(int f, int k, int nk) {
  return ((((size_t)(f >> 4) * nk + k / P::KM) * 64) + ((k % P::KM) / P::EPV) * 16 + (f & 15)) * P::EPV +
         (k % P::EPV);
}
template <class P>
__host__ __device__ inline size_t frag_off(int ft, int ks, int nk, int lane) {
  return (((size_t)ft * nk + ks) * 64 + lane) * P::EPV;
}

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
constexpr uint32_t STREAM_INDEX = 0, STREAM_DROPOUT1 = 1, STREAM_DROPOUT2 = 2, STREAM_MLP_DROPOUT = 3;

__device__ __forceinline__ int64_t philox_index(uint64_t seed, uint64_t step, uint32_t row,
                                                uint64_t n_rows) {
  const Philox4 r = philox4x32_10(row, (uint32_t)step, (uint32_t)(step >> 32), STREAM_INDEX,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
  return (int64_t)((uint64_t)r.x % n_rows);
}

}  // namespace iqlhip
