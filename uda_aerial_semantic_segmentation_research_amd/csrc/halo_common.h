// Pieces shared by the halo-resident convolution kernels (conv_halo_bf16.hip, conv_halo_f32x3.hip).
#pragma once
#include <stdlib.h>

#include "common.h"

namespace udaseg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HALO_STATS_REPLICAS = 16;   // == udaseg_bn_replicas()
constexpr int HALO_SCR_REPLICAS = 256;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
__device__ __forceinline__ float bf_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float bf_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }

// Sum over the 32 lanes of a half-wave (lane bits 0..4), in the vector ALU: four rotations inside each 16-lane row, then lane 15
// of rows 0 / 2 is broadcast into rows 1 / 3 -- the totals of the low / high half-wave end up in lanes 16..31 / 48..63.
// (The first version used five __shfl_xor steps = ds_bpermute: 320 LDS crossbar operations per wave per tile; the in-kernel
// timeline showed the epilogue at 6.5 us of a 15 us block, profiles/r03_halo_timeline.txt.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float halfwave_sum(float x) {
  x += dpp_mov<0x128, 0xf>(0.f, x);   // row_ror:8
  x += dpp_mov<0x124, 0xf>(0.f, x);   // row_ror:4
  x += dpp_mov<0x122, 0xf>(0.f, x);   // row_ror:2
  x += dpp_mov<0x121, 0xf>(0.f, x);   // row_ror:1   -> every lane holds its row's total
  x += dpp_mov<0x142, 0xa>(0.f, x);   // row_bcast:15 into rows 1 and 3 (rows 0 and 2 add 0)
  return x;
}
// The same reduction over N values at once with the add and the lane movement in ONE instruction (v_add_f32_dpp; hipcc emits
// v_mov_b32_dpp + v_add_f32 for the form above).  Step-major order: two dependent DPP operations on a register are N
// instructions apart, which covers the two wait states a DPP read needs behind a VALU write of its source (the compiler
// does not see into the asm).
template <int N>
__device__ __forceinline__ void halfwave_sum_n(float (&x)[N]) {
  static_assert(N >= 4, "spacing of dependent DPP operations");
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x[v]));
#pragma unroll
  for (int v = 0; v < N; ++v) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x[v]));
}

// x = p0 + p1 + p2 exactly, eight values at a time (two 16-byte fp32 pieces -> three bf16x8 fragment pieces).
// Eleven vector instructions per pair of values: v_cvt_pk_bf16_f32, a shift and a mask to widen the two halves again, two subtractions,
// twice, and the last conversion.  The subtractions are inline asm: left to itself hipcc packs the two of a pair into v_pk_add_f32,
// which wants its operands in adjacent registers -- the ISA of round 3's kernels shows ~65 instructions per eight values (v_mov
// copies, conversions paired with zero, v_and_or / sdwa / alignbit repacking) against the 44 of this form, and packed fp32 adds are
// slow beside MFMAs (MI355X_MICROARCH.md, per-instruction constants).  Bit-identical: tests/test_gpu_f32x3.py compares the planes
// with oracle/f32x3_ref.py bit for bit.
typedef float f32x2h __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2h v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2h));
}
__device__ __forceinline__ float sub_f32(float a, unsigned bits) {      // a - as_float(bits), kept out of the SLP vectoriser's reach
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bits));
  return r;
}
__device__ __forceinline__ void split3(const u32x4 lo, const u32x4 hi, u32x4& p0, u32x4& p1, u32x4& p2) {
  const f32x4 lf = __builtin_bit_cast(f32x4, lo), hf = __builtin_bit_cast(f32x4, hi);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = e < 2 ? lf[2 * e] : hf[2 * e - 4];
    float b = e < 2 ? lf[2 * e + 1] : hf[2 * e - 3];
    const unsigned h = cvt_pk_bf16(a, b);
    a = sub_f32(a, h << 16);
    b = sub_f32(b, h & 0xffff0000u);
    const unsigned m = cvt_pk_bf16(a, b);
    a = sub_f32(a, m << 16);
    b = sub_f32(b, m & 0xffff0000u);
    p0[e] = h;
    p1[e] = m;
    p2[e] = cvt_pk_bf16(a, b);
  }
}

// four values (one 16-byte fp32 piece -> three 8-byte bf16x4 pieces): the shared implicit-GEMM kernels stage 16 bytes per thread and row
__device__ __forceinline__ void split3_4(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float a = v[2 * e], b = v[2 * e + 1];
    const unsigned h = cvt_pk_bf16(a, b);
    a = sub_f32(a, h << 16);
    b = sub_f32(b, h & 0xffff0000u);
    const unsigned m = cvt_pk_bf16(a, b);
    a = sub_f32(a, m << 16);
    b = sub_f32(b, m & 0xffff0000u);
    p0[e] = h;
    p1[e] = m;
    p2[e] = cvt_pk_bf16(a, b);
  }
}

// Sign pattern of the halo-resident three-term kernels (conv_halo_f32x3.hip).  v_mfma_f32_32x32x16_bf16 TRUNCATES its sum toward minus
// infinity where the fp32 MFMA rounds to nearest (measured, tools/x3_bias_check.py: mean signed error -3.8e-8 of mean |y| on a 64-channel
// 3x3 layer, 1e-10 on the fp32 pipe).  A sum accumulated as -y is biased the other way, so the K loop of G = 3 x (16-channel chunks)
// (chunk, dx) groups runs + - - +: the groups [q1, q3) are multiplied with NEGATED weights (the packer flips their sign bits, nothing
// at run time) on an accumulator negated at q1 and again at q3 -- two 16-register sign flips per wave and kernel.  The partial sum grows
// like sqrt(k), the truncation step with it: quarter points leave ~5 % of the bias of a long loop (30 % at 32 channels).
__host__ __device__ inline void f3_negated_groups(int nk16, int& q1, int& q3) {
  const int ng = 3 * nk16;
  q1 = (ng + 2) / 4;
  q3 = ng - q1;
}
inline bool f3_signs_on() {       // UDASEG_OPT_F3_SIGNS = 0 (A/B, bias measurements): every group positive -- read by the packer AND the kernels
  return opt_get(UDASEG_OPT_F3_SIGNS) != 0;
}

// The environment's UDASEG_F32_SPLIT=0 is the default of TWO keys: UDASEG_OPT_F32_SPLIT (the shared-source kernels' three-term
// mode: conv_igemm X3, conv_wgrad_x3) and UDASEG_OPT_F32_HALO (whether the halo-resident three-term kernels report themselves
// applicable).  Tests switch the first off alone to grade the second against the fp32 pipe on the same launch.
inline bool f32_split_enabled() { return opt_get(UDASEG_OPT_F32_SPLIT) != 0; }
inline bool f32_halo_enabled() { return opt_get(UDASEG_OPT_F32_HALO) != 0; }

// f64 partial-sum scratch of launches with more than 1024 blocks (udaseg_set_stats_scratch): the current device's, when it holds
// HALO_SCR_REPLICAS x 2 x co doubles, else nullptr; and the launch that folds it into the [R][2][co] accumulators
// One scratch per device, used by ONE stream: the first stream that asks owns it until the scratch is re-bound
// (udaseg_set_stats_scratch); a launch on any other stream gets nullptr and adds into the 16 replicas directly (correct, slower)
// instead of mixing its partial sums with the owner's in-flight launch (the contract was only written down before round 5).
double* halo_stats_scratch(int co, hipStream_t s);
void launch_halo_stats_fold(double* sscr, int co, double* stats, hipStream_t s);

}  // namespace udaseg
