// Micro-benchmark (GPU box): can a dependent kernel's predecessor-independent prologue overlap its predecessor's tail?
//   hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap && ./overlap
// A chain of N "work" kernels (256 workgroups x 256 threads, 64 KiB of LDS so that two workgroups share a CU at most):
//   prologue  - ~P us of work that does not depend on the predecessor (stands for argument fetch, row geometry and the
//               first weight k-tiles of a GEMM),
//   wait      - (flag variants) poll the predecessor's completion counter (agent scope),
//   body      - read 16 KiB of the predecessor's output written by ANOTHER workgroup, add 1, write 16 KiB (write-through
//               in the flag variants), stands for the k-loop + epilogue,
//   signal    - (flag variants) release + one atomic add on this launch's completion counter.
// Variants: (a) plain stream launches (barrier bit set: the baseline of every launch in the engine), eager and in a graph;
// (b) hipExtLaunchKernel(..., hipExtAnyOrderLaunch) + the flags, eager and captured; (c) the plain launch with the flags
// (what the flag protocol itself costs).  Every variant checks the final values (a stale read or a lost dependency shows
// as a wrong count) and reports whether launch i+1's first workgroup started before launch i's last one ended
// (s_memrealtime stamps, 100 MHz).
// Second part: what the end-of-kernel L2 write-back costs - a chain of kernels that write 8 MiB each with plain,
// write-through (sc0 sc1) and non-temporal stores.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct WorkArgs {
  const float4* in; float4* out;
  unsigned* done_prev; unsigned* done_self; int n_prev;
  const float4* weights; int prologue_iters;
  unsigned long long* stamps;   // [2] per launch: min start, max end (atomicMin / atomicMax)
  unsigned* timeout;
  int flags;                    // 1: use the completion flags
};

__global__ __launch_bounds__(256) void k_work(WorkArgs a) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  if (tid == 0) atomicMin(a.stamps, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  // prologue: predecessor-independent loads into LDS (16 B per thread per iteration from a 4 MiB span)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < a.prologue_iters; ++i) {
    const float4 w = a.weights[((size_t)(wg * 131 + i * 7) % 1024) * 256 + tid];
    acc.x += w.x; acc.y += w.y;
  }
  lds[tid] = acc;
  __syncthreads();
  if (a.flags && a.done_prev) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(a.done_prev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.n_prev) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { atomicExch(a.timeout, 1u); break; }
      }
    }
    __syncthreads();
  }
  // body: 16 KiB written by workgroup (wg + 97) % nwg of the predecessor
  const int src = (wg + 97) % nwg;
  for (int i = 0; i < 4; ++i) {
    const size_t o_in = (size_t)src * 1024 + i * 256 + tid, o_out = (size_t)wg * 1024 + i * 256 + tid;
    float4 v;
    if (a.flags) {
      const unsigned long long* q = reinterpret_cast<const unsigned long long*>(a.in + o_in);
      const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
    } else v = a.in[o_in];
    v.x += 1.f; v.y += 1.f; v.z += lds[(tid + i) & 255].x * 0.f; v.w += 1.f;
    if (a.flags) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t r = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a.out + o_out), "v"(r) : "memory");
    } else a.out[o_out] = v;
  }
  if (a.flags) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.done_self, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) atomicMax(a.stamps + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

template <int MODE>   // 0 plain, 1 write-through, 2 non-temporal
__global__ __launch_bounds__(256) void k_write(float4* out, int n4_per_wg, float val) {
  float4* q = out + (size_t)blockIdx.x * n4_per_wg;
  const float4 v = make_float4(val, val, val, val);
  for (int i = threadIdx.x; i < n4_per_wg; i += 256) {
    if (MODE == 0) q[i] = v;
    else if (MODE == 1) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t r = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q + i), "v"(r) : "memory");
    } else {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t r = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(r, reinterpret_cast<f32x4_t*>(q + i));
    }
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NWG = 256, N = 100;
  float4 *bufA, *bufB, *weights; unsigned* done; unsigned long long* stamps; unsigned* timeout;
  CK(hipMalloc(&bufA, (size_t)NWG * 1024 * 16)); CK(hipMalloc(&bufB, (size_t)NWG * 1024 * 16));
  CK(hipMalloc(&weights, (size_t)1024 * 256 * 16)); CK(hipMemset(weights, 0, (size_t)1024 * 256 * 16));
  CK(hipMalloc(&done, (N + 1) * sizeof(unsigned))); CK(hipMalloc(&stamps, (size_t)N * 2 * 8)); CK(hipMalloc(&timeout, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_work), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> init(N * 2);
  for (int i = 0; i < N; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }

  auto reset = [&]() -> int {
    CK(hipMemsetAsync(bufA, 0, (size_t)NWG * 1024 * 16, st)); CK(hipMemsetAsync(bufB, 0, (size_t)NWG * 1024 * 16, st));
    CK(hipMemsetAsync(done, 0, (N + 1) * sizeof(unsigned), st)); CK(hipMemsetAsync(timeout, 0, 4, st));
    CK(hipMemcpyAsync(stamps, init.data(), (size_t)N * 16, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    return 0;
  };
  auto check = [&](const char* name, float us) -> int {
    std::vector<float> h((size_t)NWG * 1024 * 4);
    CK(hipMemcpy(h.data(), ((N - 1) % 2) ? bufB : bufA, h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); i += 4) if (h[i] != (float)N || h[i + 1] != (float)N || h[i + 3] != (float)N) ++bad;
    std::vector<unsigned long long> s(N * 2);
    CK(hipMemcpy(s.data(), stamps, (size_t)N * 16, hipMemcpyDeviceToHost));
    unsigned to; CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
    int overlapped = 0; double ov_ticks = 0, life = 0;
    for (int i = 0; i + 1 < N; ++i) {
      if (s[2 * (i + 1)] < s[2 * i + 1]) { ++overlapped; ov_ticks += (double)(s[2 * i + 1] - s[2 * (i + 1)]); }
      life += (double)(s[2 * i + 1] - s[2 * i]);
    }
    printf("%-58s %7.2f us/kernel  wrong=%zu timeout=%u  launches starting before the predecessor ended: %d/%d (avg %.2f us)  kernel life %.2f us\n",
           name, us, bad, to, overlapped, N - 1, overlapped ? ov_ticks / overlapped / 100.0 : 0.0, life / (N - 1) / 100.0);
    return 0;
  };
  auto args_of = [&](int i, int flags, int piters) {
    WorkArgs a{};
    a.in = (i % 2) ? bufA : bufB; a.out = (i % 2) ? bufB : bufA;
    a.done_prev = i ? done + i - 1 : nullptr; a.done_self = done + i; a.n_prev = NWG;
    a.weights = weights; a.prologue_iters = piters; a.stamps = stamps + 2 * i; a.timeout = timeout; a.flags = flags;
    return a;
  };
  for (int piters : {0, 64, 256}) {
    printf("---- prologue: %d dependent-free 4 KiB loads per workgroup\n", piters);
    for (int variant = 0; variant < 3; ++variant) {
      const int flags = variant != 0, anyorder = variant == 1;
      const char* vn = variant == 0 ? "plain launches" : (variant == 1 ? "any-order launches + completion flags" : "plain launches + completion flags");
      auto launch_all = [&]() -> int {
        for (int i = 0; i < N; ++i) {
          WorkArgs a = args_of(i, flags, piters);
          void* params[] = {&a};
          if (anyorder) CK(hipExtLaunchKernel(reinterpret_cast<const void*>(k_work), dim3(NWG), dim3(256), params, 64 << 10, st, nullptr, nullptr, hipExtAnyOrderLaunch));
          else CK(hipLaunchKernel(reinterpret_cast<const void*>(k_work), dim3(NWG), dim3(256), params, 64 << 10, st));
        }
        return 0;
      };
      char name[128];
      float ms;
      // eager
      if (reset()) return 1;
      if (launch_all()) return 1;                   // warm
      CK(hipStreamSynchronize(st));
      if (reset()) return 1;
      CK(hipEventRecord(e0, st));
      if (launch_all()) return 1;
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      snprintf(name, sizeof(name), "%s, eager", vn);
      if (check(name, ms * 1e3f / N)) return 1;
      // graph
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      if (launch_all()) { printf("  (capture of this variant failed)\n"); (void)hipStreamEndCapture(st, &g); continue; }
      hipError_t ce = hipStreamEndCapture(st, &g);
      if (ce != hipSuccess) { printf("  %s: capture failed: %s\n", vn, hipGetErrorString(ce)); (void)hipGetLastError(); continue; }
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      if (reset()) return 1;
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      if (reset()) return 1;
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      snprintf(name, sizeof(name), "%s, hipGraph", vn);
      if (check(name, ms * 1e3f / N)) return 1;
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }

  // ---- end-of-kernel write-back: 8 MiB written per kernel, chain of 100 in a graph
  printf("---- a chain of kernels that each write 8 MiB (256 workgroups x 32 KiB), hipGraph\n");
  float4* big; CK(hipMalloc(&big, 8 << 20));
  for (int mode = 0; mode < 3; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      if (mode == 0) hipLaunchKernelGGL(k_write<0>, dim3(256), dim3(256), 0, st, big, 2048, (float)i);
      else if (mode == 1) hipLaunchKernelGGL(k_write<1>, dim3(256), dim3(256), 0, st, big, 2048, (float)i);
      else hipLaunchKernelGGL(k_write<2>, dim3(256), dim3(256), 0, st, big, 2048, (float)i);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float ms;
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); }
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-30s %6.2f us/kernel\n", mode == 0 ? "plain stores" : (mode == 1 ? "write-through (sc0 sc1)" : "non-temporal"), ms * 1e3f / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
