// Device-side helpers shared by the kernel translation units: LDS-DMA issue (hand-counted vmcnt) and the
// L1-bypassing (`sc1`) loads of mutable data used inside the persistent launch.
#pragma once
#include <hip/hip_runtime.h>

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform `lds_dst` + lane*16 (M0 holds the
// base), global source per lane.  Issued through asm so that hipcc neither counts it nor drains
// it with vmcnt(0) at the next LDS read: the waits below are counted by hand.
// (lds_dst goes through readfirstlane: a wave-uniform value that hipcc happens to hold in a vector register - e.g. the
// result of an integer division - is otherwise handed to the "s" operand as a VGPR: "invalid operand for instruction";
// for a value already in a scalar register the builtin folds away)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// ... with the source as wave-uniform base (scalar register pair) + per-lane 32-bit byte offset: no 64-bit vector address
// arithmetic per instruction (the weight DMAs of the k-loop)
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  {   // (a wave-uniform value that hipcc happens to hold in vector registers does not satisfy the "s" constraint)
    const unsigned long long b = reinterpret_cast<unsigned long long>(sbase);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    sbase = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
  }
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_sc1(const void* gsrc, unsigned lds_dst) {   // L1-bypassing variant
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Hand-over of data between workgroups of ONE launch (split-K finish): the producer writes through (`sc0 sc1`: the
// line does not stay dirty in its XCD's L2), the consumer - after it has seen the producer's agent-scope atomic -
// loads past its L1 at system scope.  (tools/micro/xcd_barrier.hip: plain loads return stale L1 lines.)
// Written as relaxed system-scope atomics on 64-bit halves (the compiler emits global_store/load_dwordx2 sc0 sc1 and
// tracks the loads' completion itself; an asm load whose result is consumed after a separate asm wait is not safe -
// the compiler may copy the destination registers before the wait).
// (the STORE is one 16-byte instruction in assembly - a store has no result for the compiler to mishandle, the caller's
// wait_vmcnt<0> covers it; as two 8-byte atomic stores a lane-linear dump was 8-byte pieces at a 16-byte stride: partial 32-byte
// sectors, the slow ~240-cycles-per-instruction form of tools/micro/store_pattern.hip)
__device__ __forceinline__ void st_handover16(float4* p, float4 v) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");   // (s_nop: store-data hazard, see dv_st16)
}
__device__ __forceinline__ float4 ld_handover16(const float4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
// plain-value loads of mutable data: default policy, or `sc1` (L2-served) inside a persistent launch
template <bool SC1>
__device__ __forceinline__ float4 ld_mut4(const float* p) {
  if (!SC1) return *reinterpret_cast<const float4*>(p);
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return v;
}
template <bool SC1>
__device__ __forceinline__ float2 ld_mut2(const float2* p) {
  if (!SC1) return *p;
  float2 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return v;
}
template <bool SC1>
__device__ __forceinline__ float ld_mut1(const float* p) {
  if (!SC1) return *p;
  float v;
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return v;
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {   // 4 bytes per lane
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_sc1(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#ifndef DV_ATTN_PLO
// 1: the probabilities P of an attention tile enter P V as split bf16 (hi + lo: three products); 0: as one bf16 (two products:
// P_hi V_hi + P_hi V_lo) - experiment knob, see docs/HISTORY.md
#define DV_ATTN_PLO 1
#endif
// Bulk result stores (activations, planes, fragments) of the epilogues: one place for their cache policy.  DV_WT_STORES=1
// (experiment, round 4): written THROUGH the XCD's L2 (`sc0 sc1`), so that nothing of a tensor stays dirty for the
// end-of-kernel write-back.  tools/micro/overlap.hip: a chain of kernels that only WRITE 8 MiB each takes 2.35 us per kernel
// that way against 2.95 us - but in the forward it LOSES 3.4 % (profiles/r04_ab_write_through.txt): the default stays plain.
#ifndef DV_WT_STORES
#define DV_WT_STORES 0
#endif
typedef unsigned dv_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dv_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dv_st16(void* p, uint4 v) {
#if DV_WT_STORES
  const dv_u32x4 r = {v.x, v.y, v.z, v.w};
  // (s_nop: a VMEM store of more than 64 bits reads its data registers late - a VALU write to them needs wait states the
  // compiler's hazard recogniser inserts for its own stores but cannot see for one inside an asm block)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}
__device__ __forceinline__ void dv_st16(void* p, float4 v) {
  dv_st16(p, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}
__device__ __forceinline__ void dv_st8(void* p, uint2 v) {
#if DV_WT_STORES
  const dv_u32x2 r = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
#else
  *reinterpret_cast<uint2*>(p) = v;
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Two floats -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32.  Written as a vector conversion, NOT
// inline assembly: the compiler's hazard recogniser does not look inside asm blocks, and a conversion scheduled right
// behind the transcendental-unit instruction (v_exp_f32) that produces its operand read a stale register.
typedef float dv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dv_cvt_pk_bf16(float lo, float hi) {
  const dv_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dv_bf16x2));
}

// P V of the attention kernels (parity mode).  DV_ATTN_PF16 = 1 (default since round 4): the probabilities enter as ONE fp16
// plane (11 significant bits, round to nearest even: v_cvt_pk_f16_f32) and V as split fp16 (hi + lo): two products
// P Vhi + P Vlo on v_mfma_f32_32x32x16_f16.  [Rounds 1-3: P and V both split bf16, three products; the softmax loop is bound
// by vector-ALU issue and a third of its work was splitting P.  P as ONE bf16 plane (8 bits) was measured in round 3: +1.9 %
// end to end but 1e-4 of error per forward - rejected; fp16 keeps 8x the precision for the same instruction count.]
// 0: the three-product split-bf16 form.  Writers of V^T fragments (kernels_chain.hip) and the kernels that multiply them agree
// through this one macro.
// Row-block chains: with NS >= 2 output fragments per wave the two k-groups of a stage each OWN half of the fragments (hand the
// others over, run the epilogue on their own) instead of k-group 1 handing everything to k-group 0 (kernels_chain.hip).  0: off.
#ifndef DV_CHAIN_OWN
#define DV_CHAIN_OWN 1
#endif
#ifndef DV_ATTN_PF16
#define DV_ATTN_PF16 1
#endif
typedef _Float16 dv_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dv_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned dv_cvt_pk_f16(float lo, float hi) {
  const dv_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dv_f16x2));
}
// two floats -> packed fp16 pair `h` (round to nearest even) and the packed fp16 pair `l` of what the rounding left
__device__ __forceinline__ void dv_split_pk_f16(float x0, float x1, unsigned& h, unsigned& l) {
  // (clamped to the fp16 range: a |V| beyond 65504 - no checkpoint seen has one - saturates instead of becoming inf / NaN in the
  // P V product; v_med3_f32, two instructions per pair in the kernels that WRITE V fragments, none in the attention loops)
  x0 = __builtin_amdgcn_fmed3f(x0, -65504.0f, 65504.0f); x1 = __builtin_amdgcn_fmed3f(x1, -65504.0f, 65504.0f);
  const dv_f32x2 v = {x0, x1};
  const dv_f16x2 hh = __builtin_convertvector(v, dv_f16x2);
  const dv_f32x2 back = __builtin_convertvector(hh, dv_f32x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(v - back, dv_f16x2));
}

// ---- cross-lane helpers ----
// Sum over the 64 lanes of a wave, the same value in every lane, fixed order (deterministic): four DPP adds inside each row of 16
// lanes, then the four row sums through scalar registers - instead of six dependent ds_bpermute round trips through the LDS
// crossbar (~100 cycles each on a wave's critical path in the epilogues).
template <int CTRL>
__device__ __forceinline__ float dv_dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float dv_readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum64(float v) {
  v += dv_dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dv_dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dv_dpp_f32<0x141>(v);    // row_half_mirror
  v += dv_dpp_f32<0x140>(v);    // row_mirror
  return (dv_readlane_f32(v, 0) + dv_readlane_f32(v, 16)) + (dv_readlane_f32(v, 32) + dv_readlane_f32(v, 48));
}
template <int CTRL>
__device__ __forceinline__ double dv_dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double dv_readlane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum64(double v) {
  v += dv_dpp_f64<0xB1>(v);
  v += dv_dpp_f64<0x4E>(v);
  v += dv_dpp_f64<0x141>(v);
  v += dv_dpp_f64<0x140>(v);
  return (dv_readlane_f64(v, 0) + dv_readlane_f64(v, 16)) + (dv_readlane_f64(v, 32) + dv_readlane_f64(v, 48));
}
// Lane pair (l, l + 32) of a wave, each holding dword `a` (its part of column group g) and dword `b` (... of group g + 1):
// afterwards the lower lane holds (its own a, the upper lane's a) and the upper lane (the lower lane's b, its own b) - one
// v_permlane32_swap (gfx950).  The transposed-accumulator epilogues use it so that a lane owns 8 consecutive columns of a bf16
// plane instead of 4: 16-byte stores, 32 contiguous bytes per row and instruction.  [8-byte stores that leave 16 B per row are
// ~240 cycles of issue EACH (partial 32-byte sectors); 16-byte ones ~100: tools/micro/store_pattern.hip.]
__device__ __forceinline__ void pair_swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
// value op value-of-the-lane-32-away (the two half-waves of a 32x32 MFMA accumulator hold the two halves of a row): one
// v_permlane32_swap on two copies of the value instead of a ds_bpermute round trip (~100+ cycles of dependent latency - it sits
// inside the key loop of every attention kernel).  After the swap the lower lanes hold (own, partner), the upper (partner, own):
// a commutative op on the pair needs no select.
__device__ __forceinline__ float pair_sum32(float v) {
  unsigned a = __float_as_uint(v), b = a;
  pair_swap32(a, b);
  return __uint_as_float(a) + __uint_as_float(b);
}
__device__ __forceinline__ float pair_max32(float v) {
  unsigned a = __float_as_uint(v), b = a;
  pair_swap32(a, b);
  return fmaxf(__uint_as_float(a), __uint_as_float(b));
}
// Split bf16 planes of 16 values of one row in the transposed-accumulator layout (v[4g + e] = column 8g + 4lh + e of a 32-column
// fragment, lh = lane >> 5): `o` = element offset of the fragment's first column in this lane's row.  Both lanes of a pair must be active.
// `pairs` (wave-uniform): bit 0 / bit 1 = the fragment's columns 0-15 / 16-31 exist (attention heads of 16 / 48 channels).
__device__ __forceinline__ void store_planes16(unsigned short* hi, unsigned short* lo, size_t o, int lh, const float* v, int pairs = 3) {
  unsigned h[8], l[8];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    h[2 * g] = dv_cvt_pk_bf16(v[4 * g], v[4 * g + 1]);
    h[2 * g + 1] = dv_cvt_pk_bf16(v[4 * g + 2], v[4 * g + 3]);
    if (lo) {
      l[2 * g] = dv_cvt_pk_bf16(v[4 * g] - __uint_as_float(h[2 * g] << 16), v[4 * g + 1] - __uint_as_float(h[2 * g] & 0xffff0000u));
      l[2 * g + 1] = dv_cvt_pk_bf16(v[4 * g + 2] - __uint_as_float(h[2 * g + 1] << 16), v[4 * g + 3] - __uint_as_float(h[2 * g + 1] & 0xffff0000u));
    }
  }
#pragma unroll
  for (int G = 0; G < 4; G += 2) {     // groups (G, G + 1): the lower lane takes columns 8G .. 8G + 7, the upper lane 8G + 8 .. 8G + 15
    if (!((pairs >> (G >> 1)) & 1)) continue;
    pair_swap32(h[2 * G], h[2 * G + 2]);
    pair_swap32(h[2 * G + 1], h[2 * G + 3]);
    dv_st16(hi + o + 8 * (G + lh), make_uint4(h[2 * G], h[2 * G + 1], h[2 * G + 2], h[2 * G + 3]));
    if (lo) {
      pair_swap32(l[2 * G], l[2 * G + 2]);
      pair_swap32(l[2 * G + 1], l[2 * G + 3]);
      dv_st16(lo + o + 8 * (G + lh), make_uint4(l[2 * G], l[2 * G + 1], l[2 * G + 2], l[2 * G + 3]));
    }
  }
}

// ... of 8 values: v[4g + e] = column 8g + 4lh + e of a 16-column half fragment (g < 2); `o` = element offset of the half's first column
__device__ __forceinline__ void store_planes8(unsigned short* hi, unsigned short* lo, size_t o, int lh, const float* v) {
  unsigned h[4], l[4];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    h[2 * g] = dv_cvt_pk_bf16(v[4 * g], v[4 * g + 1]);
    h[2 * g + 1] = dv_cvt_pk_bf16(v[4 * g + 2], v[4 * g + 3]);
    if (lo) {
      l[2 * g] = dv_cvt_pk_bf16(v[4 * g] - __uint_as_float(h[2 * g] << 16), v[4 * g + 1] - __uint_as_float(h[2 * g] & 0xffff0000u));
      l[2 * g + 1] = dv_cvt_pk_bf16(v[4 * g + 2] - __uint_as_float(h[2 * g + 1] << 16), v[4 * g + 3] - __uint_as_float(h[2 * g + 1] & 0xffff0000u));
    }
  }
  pair_swap32(h[0], h[2]);
  pair_swap32(h[1], h[3]);
  dv_st16(hi + o + 8 * lh, make_uint4(h[0], h[1], h[2], h[3]));
  if (lo) {
    pair_swap32(l[0], l[2]);
    pair_swap32(l[1], l[3]);
    dv_st16(lo + o + 8 * lh, make_uint4(l[0], l[1], l[2], l[3]));
  }
}

// GELU with the exact-erf definition (reference unet1d/activations / F.gelu default), erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7): branch-free, one v_exp_f32 and one v_rcp_f32, ~20 issue slots instead of erff()'s two divergent
// polynomial branches.  1 + erf is formed without cancellation on the negative side: measured max |error| of the GELU
// 4.2e-7 over [-12, 12] (docs/HISTORY.md), two orders below the split-bf16 product error.
__device__ __forceinline__ float gelu_erf(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = p * t * __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);   // = 1 - erf(z)
  return 0.5f * v * (v < 0.f ? y : 2.0f - y);
}
