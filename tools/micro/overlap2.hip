// Micro-benchmark (GPU box), second form of tools/micro/overlap.hip: the CHEAP dependency protocol.
//   hipcc --offload-arch=gfx950 -O3 overlap2.hip -o overlap2 && ./overlap2
// overlap.hip paid for its hand-over with write-through stores and L2-bypassing loads on every byte.  Here a producer
// workgroup stores normally and releases ONCE at its end (`buffer_wbl2 sc1` + vmcnt(0) + one agent-scope atomic add on the
// launch's completion counter); a consumer workgroup polls the counter, acquires ONCE (`buffer_inv sc1`) and then loads
// normally - what the command processor does at a kernel boundary, moved into the kernels, so that a launch WITHOUT the
// barrier bit (hipExtAnyOrderLaunch) can run its predecessor-independent prologue (arguments, geometry, weight tiles) under
// its predecessor's tail.  Workgroups shaped like the engine's GEMMs: 128 KiB of LDS (one per CU), 512 threads, 256 of
// them; prologue = 96 KiB of weight loads per workgroup, body = 16 KiB of the predecessor's output (written by another
// workgroup) + a timed spin (the k-loop) + 24 KiB of stores.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Args {
  const float4* in; float4* out; const float4* weights;
  unsigned* done_prev; unsigned* done_self; int n_prev;
  unsigned long long* stamps; unsigned* timeout;
  int proto;          // 0: none (plain launches), 1: completion counter + wbl2 / inv
  int body_spin;      // iterations of the body's spin (s_sleep 8 each)
};

__global__ __launch_bounds__(512) void k_work(Args a) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  if (tid == 0) atomicMin(a.stamps, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  // prologue: 96 KiB of weights (12 x 16 B per thread), independent of the predecessor
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float4 w = a.weights[((size_t)((wg * 7 + i) & 255) * 12 + i) * 512 + tid];
    acc.x += w.x; acc.y += w.y;
  }
  lds[tid] = acc;
  if (a.proto && a.done_prev) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(a.done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.n_prev) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { atomicExch(a.timeout, 1u); break; }
      }
    }
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");                 // acquire, agent scope: drop what this CU / XCD may hold of the predecessor's output
  }
  __syncthreads();
  // body: 16 KiB of the predecessor's output written by workgroup (wg + 97) % nwg, spin, 24 KiB of output
  const int src = (wg + 97) % nwg;
  float4 v0 = a.in[(size_t)src * 1536 + tid], v1 = a.in[(size_t)src * 1536 + 512 + tid];
  for (int i = 0; i < a.body_spin; ++i) __builtin_amdgcn_s_sleep(8);
  v0.x += 1.f; v0.y += lds[(tid + 1) & 511].x * 0.f; v1.x += 1.f;
  a.out[(size_t)wg * 1536 + tid] = v0;
  a.out[(size_t)wg * 1536 + 512 + tid] = v1;
  a.out[(size_t)wg * 1536 + 1024 + tid] = v0;
  if (a.proto) {
    asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");   // release, agent scope
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.done_self, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) atomicMax(a.stamps + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NWG = 256, N = 100;
  float4 *bufA, *bufB, *weights; unsigned* done; unsigned long long* stamps; unsigned* timeout;
  const size_t act = (size_t)NWG * 1536 * 16;
  CK(hipMalloc(&bufA, act)); CK(hipMalloc(&bufB, act));
  CK(hipMalloc(&weights, (size_t)256 * 12 * 512 * 16)); CK(hipMemset(weights, 0, (size_t)256 * 12 * 512 * 16));
  CK(hipMalloc(&done, (N + 1) * 4)); CK(hipMalloc(&stamps, (size_t)N * 16)); CK(hipMalloc(&timeout, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_work), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> init(N * 2);
  for (int i = 0; i < N; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
  auto reset = [&]() -> int {
    CK(hipMemsetAsync(bufA, 0, act, st)); CK(hipMemsetAsync(bufB, 0, act, st));
    CK(hipMemsetAsync(done, 0, (N + 1) * 4, st)); CK(hipMemsetAsync(timeout, 0, 4, st));
    CK(hipMemcpyAsync(stamps, init.data(), (size_t)N * 16, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    return 0;
  };
  auto check = [&](const char* name, float us) -> int {
    std::vector<float> h(act / 4);
    CK(hipMemcpy(h.data(), ((N - 1) % 2) ? bufB : bufA, act, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t w = 0; w < (size_t)NWG; ++w)
      for (size_t i = 0; i < 1024; ++i) if (h[(w * 1536 + i) * 4] != (float)N) ++bad;
    std::vector<unsigned long long> s(N * 2);
    CK(hipMemcpy(s.data(), stamps, (size_t)N * 16, hipMemcpyDeviceToHost));
    unsigned to; CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
    int ov = 0; double life = 0;
    for (int i = 0; i + 1 < N; ++i) { if (s[2 * (i + 1)] < s[2 * i + 1]) ++ov; life += (double)(s[2 * i + 1] - s[2 * i]); }
    printf("%-52s %7.2f us/kernel  wrong=%zu timeout=%u  early starts %d/%d  kernel life %.2f us\n", name, us, bad, to, ov, N - 1, life / (N - 1) / 100.0);
    return 0;
  };
  for (int spin : {0, 40, 160}) {
    printf("---- body spin %d x s_sleep(8)\n", spin);
    for (int variant = 0; variant < 3; ++variant) {
      const int proto = variant != 0, anyorder = variant == 1;
      const char* vn = variant == 0 ? "plain launches" : (variant == 1 ? "any-order launches + counter, wbl2 / inv" : "plain launches + counter, wbl2 / inv");
      auto launch_all = [&]() -> int {
        for (int i = 0; i < N; ++i) {
          Args a{};
          a.in = (i % 2) ? bufA : bufB; a.out = (i % 2) ? bufB : bufA; a.weights = weights;
          a.done_prev = i ? done + i - 1 : nullptr; a.done_self = done + i; a.n_prev = NWG;
          a.stamps = stamps + 2 * i; a.timeout = timeout; a.proto = proto; a.body_spin = spin;
          void* params[] = {&a};
          if (anyorder) CK(hipExtLaunchKernel(reinterpret_cast<const void*>(k_work), dim3(NWG), dim3(512), params, 128 << 10, st, nullptr, nullptr, hipExtAnyOrderLaunch));
          else CK(hipLaunchKernel(reinterpret_cast<const void*>(k_work), dim3(NWG), dim3(512), params, 128 << 10, st));
        }
        return 0;
      };
      float ms;
      if (reset()) return 1;
      if (launch_all()) return 1;
      CK(hipStreamSynchronize(st));
      if (reset()) return 1;
      CK(hipEventRecord(e0, st));
      if (launch_all()) return 1;
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      char name[128];
      snprintf(name, sizeof(name), "%s, eager", vn);
      if (check(name, ms * 1e3f / N)) return 1;
      if (variant == 1) continue;                      // (capture drops the flag: overlap.hip)
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      if (launch_all()) return 1;
      CK(hipStreamEndCapture(st, &g));
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
  return 0;
}
