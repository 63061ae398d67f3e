// Device-side helpers shared by the kernel translation units: LDS-DMA issue (hand-counted vmcnt) and the
// L1-bypassing (`sc1`) loads of mutable data used inside the persistent launch.
#pragma once
#include <hip/hip_runtime.h>

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform `lds_dst` + lane*16 (M0 holds the
// base), global source per lane.  Issued through asm so that hipcc neither counts it nor drains
// it with vmcnt(0) at the next LDS read: the waits below are counted by hand.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_sc1(const void* gsrc, unsigned lds_dst) {   // L1-bypassing variant
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Hand-over of data between workgroups of ONE launch (split-K finish): the producer writes through (`sc0 sc1`: the
// line does not stay dirty in its XCD's L2), the consumer - after it has seen the producer's agent-scope atomic -
// loads past its L1 at system scope.  (tools/micro/xcd_barrier.hip: plain loads return stale L1 lines.)
typedef float dv_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_handover16(float4* p, float4 v) {
  const dv_f32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ float4 ld_handover16(const float4* p) {   // caller waits (vmcnt) before using the value
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
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
