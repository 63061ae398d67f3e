// Micro-benchmark (GPU box): per-CU fill rate of LDS from L2-resident data, LDS-DMA vs register staging.
//   hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate && ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA.  MODE 1: global_load_dwordx4 -> registers only.  MODE 2: global_load -> ds_write_b128.
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_fill(const char* src, size_t span, int iters, unsigned long long* ticks, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (unsigned)(size_t)smem;
  // each iteration every wave moves 4 KiB (4 instructions of 1 KiB); the workgroup moves NW * 4 KiB
  size_t off = ((size_t)blockIdx.x * 65536 + wave * 4096 + lane * 16) % span;
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned dst = base + (unsigned)(((it & 3) * NW + wave) * 4096);
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(src + off + q * 1024, dst + q * 1024);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      u32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4*>(src + off + q * 1024);
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= v[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(smem + (dst - base) + q * 1024 + lane * 16) = v[q];
      }
    }
    off += (size_t)NW * 4096;
    if (off >= span) off -= span;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (MODE != 0) { if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = reinterpret_cast<unsigned*>(smem)[threadIdx.x]; }
  else if (iters < 0) sink[0] = reinterpret_cast<unsigned*>(smem)[threadIdx.x];
}

template <int MODE, int NW>
int run(const char* name, const char* src, size_t span, int wgs) {
  unsigned long long* ticks; unsigned* sink;
  CK(hipMalloc(&ticks, wgs * 8)); CK(hipMalloc(&sink, 64));
  const int iters = 256, smem = 4 * NW * 4096;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_fill<MODE, NW>), dim3(wgs), dim3(64 * NW), smem, 0, src, span, iters, ticks, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  unsigned long long h[1024]; CK(hipMemcpy(h, ticks, wgs * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < wgs; ++i) avg += (double)h[i]; avg /= wgs;
  const double bytes_wg = (double)iters * NW * 4096;
  printf("%-38s waves/WG %d  WGs %4d  span %5.1f MB: %6.1f B/tick/WG   %7.1f GB/s/CU(by event)  chip %.2f TB/s\n", name, NW, wgs,
         span / 1048576.0, bytes_wg / avg, bytes_wg * wgs / (ms * 1e-3) / 1e9 / (wgs < 256 ? wgs : 256), bytes_wg * wgs / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  char* src; const size_t cap = 256u << 20;
  CK(hipMalloc(&src, cap)); CK(hipMemset(src, 1, cap));
  for (size_t span : {(size_t)2 << 20, (size_t)24 << 20, (size_t)200 << 20}) {
    if (run<0, 8>("LDS-DMA global_load_lds_dwordx4", src, span, 256)) return 1;
    if (run<0, 4>("LDS-DMA global_load_lds_dwordx4", src, span, 256)) return 1;
    if (run<0, 4>("LDS-DMA global_load_lds_dwordx4", src, span, 512)) return 1;
    if (run<1, 8>("global_load_dwordx4 -> registers", src, span, 256)) return 1;
    if (run<1, 4>("global_load_dwordx4 -> registers", src, span, 256)) return 1;
    if (run<2, 8>("global_load_dwordx4 -> ds_write_b128", src, span, 256)) return 1;
    if (run<2, 4>("global_load_dwordx4 -> ds_write_b128", src, span, 512)) return 1;
  }
  return 0;
}
