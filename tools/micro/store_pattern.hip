// Micro-benchmark (GPU box): what a GEMM epilogue's store pattern costs.  Each workgroup (4 waves, 2 x 2) owns a 64 x 64 tile of
// out[M][N] (fp32); a wave holds a 32 x 32 fragment in the transposed-accumulator layout of gemm_tile.h (lane = row l31, half lh;
// register 4g + e = column 8g + 4lh + e).
//   MODE 0: four 16-byte stores per lane straight from that layout: one instruction writes 32 rows x 32 B (partial lines)
//   MODE 1: through LDS, then four 16-byte stores where 8 consecutive lanes write one row's 128 B (8 full lines per instruction)
//   MODE 2: + split bf16 planes (hi, lo) from the register layout (8-byte stores: 16 B per row per instruction)
//   MODE 3: + split planes through LDS (16-byte stores, 4 lanes per 64-byte row of a plane)
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_store(float* out, unsigned short* hi, unsigned short* lo, int M, int N, unsigned long long* ticks) {
  __shared__ __attribute__((aligned(16))) float stage[4][32 * 36];     // per wave: 32 rows x (32 + 4 pad) floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int tn = N / 64, m0 = (blockIdx.x / tn) * 64, n0 = (blockIdx.x % tn) * 64;
  const int wm = wave >> 1, wn = wave & 1;
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = (float)(m0 + l31) * 0.001f + (float)(r + lh);
  const int m = m0 + wm * 32 + l31, nf = n0 + wn * 32 + 4 * lh;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0 || MODE == 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const size_t o = (size_t)m * N + nf + 8 * g;
      *reinterpret_cast<float4*>(out + o) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
      if (MODE == 2) {
        const unsigned h01 = pk_bf16(v[4 * g], v[4 * g + 1]), h23 = pk_bf16(v[4 * g + 2], v[4 * g + 3]);
        *reinterpret_cast<uint2*>(hi + o) = make_uint2(h01, h23);
        const unsigned l01 = pk_bf16(v[4 * g] - __uint_as_float(h01 << 16), v[4 * g + 1] - __uint_as_float(h01 & 0xffff0000u));
        const unsigned l23 = pk_bf16(v[4 * g + 2] - __uint_as_float(h23 << 16), v[4 * g + 3] - __uint_as_float(h23 & 0xffff0000u));
        *reinterpret_cast<uint2*>(lo + o) = make_uint2(l01, l23);
      }
    }
  } else {
    float* st = stage[wave];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(st + l31 * 36 + 8 * g + 4 * lh) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    // (same wave writes and reads its own stage: no barrier, only the LDS counter)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int rr = lane >> 3, cc = (lane & 7) * 4;
    float4 w[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) w[it] = *reinterpret_cast<const float4*>(st + (it * 8 + rr) * 36 + cc);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const size_t o = (size_t)(m0 + wm * 32 + it * 8 + rr) * N + n0 + wn * 32 + cc;
      *reinterpret_cast<float4*>(out + o) = w[it];
    }
    if (MODE == 3) {
      // planes: a lane takes 8 consecutive columns of a row (two float4) -> 16 bytes of hi, 16 bytes of lo; 4 lanes per row, 16 rows per instruction
      const int r2 = lane >> 2, c2 = (lane & 3) * 8;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const float4 a = *reinterpret_cast<const float4*>(st + (it * 16 + r2) * 36 + c2);
        const float4 b = *reinterpret_cast<const float4*>(st + (it * 16 + r2) * 36 + c2 + 4);
        const unsigned h0 = pk_bf16(a.x, a.y), h1 = pk_bf16(a.z, a.w), h2 = pk_bf16(b.x, b.y), h3 = pk_bf16(b.z, b.w);
        const size_t o = (size_t)(m0 + wm * 32 + it * 16 + r2) * N + n0 + wn * 32 + c2;
        *reinterpret_cast<uint4*>(hi + o) = make_uint4(h0, h1, h2, h3);
        const unsigned l0 = pk_bf16(a.x - __uint_as_float(h0 << 16), a.y - __uint_as_float(h0 & 0xffff0000u));
        const unsigned l1 = pk_bf16(a.z - __uint_as_float(h1 << 16), a.w - __uint_as_float(h1 & 0xffff0000u));
        const unsigned l2 = pk_bf16(b.x - __uint_as_float(h2 << 16), b.y - __uint_as_float(h2 & 0xffff0000u));
        const unsigned l3 = pk_bf16(b.z - __uint_as_float(h3 << 16), b.w - __uint_as_float(h3 & 0xffff0000u));
        *reinterpret_cast<uint4*>(lo + o) = make_uint4(l0, l1, l2, l3);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { ticks[(blockIdx.x * 4 + wave) * 2] = t1 - t0; ticks[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

__global__ void k_flush(float* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f; }

template <int MODE>
int run(const char* name, int M, int N) {
  float* out; unsigned short *hi, *lo; unsigned long long* ticks; float* fl;
  const int wgs = (M / 64) * (N / 64);
  CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&hi, (size_t)M * N * 2)); CK(hipMalloc(&lo, (size_t)M * N * 2));
  CK(hipMalloc(&ticks, wgs * 4 * 2 * 8)); CK(hipMalloc(&fl, 64u << 20));
  CK(hipMemset(fl, 0, 64u << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  std::vector<unsigned long long> h(wgs * 8);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_flush, dim3(1024), dim3(256), 0, 0, fl, (size_t)(64u << 20) / 4);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_store<MODE>, dim3(wgs), dim3(256), 0, 0, out, hi, lo, M, N, ticks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  CK(hipMemcpy(h.data(), ticks, wgs * 8 * 8, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> a, b;
  for (int i = 0; i < wgs * 4; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("%-44s M=%5d N=%4d wgs=%4d  issue median %6llu p90 %6llu | complete median %6llu p90 %6llu cyc | kernel %.2f us\n", name, M, N, wgs,
         a[a.size() / 2], a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() * 9 / 10], best * 1e3);
  hipFree(out); hipFree(hi); hipFree(lo); hipFree(ticks); hipFree(fl);
  return 0;
}

int main() {
  const int shapes[][2] = {{2048, 384}, {8192, 128}, {4096, 256}, {1024, 512}};
  for (auto& s : shapes) {
    if (run<0>("fp32, register layout (32 B per row)", s[0], s[1])) return 1;
    if (run<1>("fp32, through LDS (full 128 B lines)", s[0], s[1])) return 1;
    if (run<2>("fp32 + planes, register layout", s[0], s[1])) return 1;
    if (run<3>("fp32 + planes, through LDS", s[0], s[1])) return 1;
  }
  return 0;
}
