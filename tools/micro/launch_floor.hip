// Micro-benchmark (GPU box): cost of a dependent chain of small kernels, eager vs hipGraph.
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor && ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f; }
// touches `bytes_per_wg` per workgroup: read + write
__global__ void k_touch(float4* p, int n4_per_wg) {
  float4* q = p + (size_t)blockIdx.x * n4_per_wg;
  for (int i = threadIdx.x; i < n4_per_wg; i += blockDim.x) { float4 v = q[i]; v.x += 1.f; q[i] = v; }
}
struct Big { char pad[320]; float* p; };
__global__ void k_bigarg(Big b) { if (b.p == nullptr && threadIdx.x == 9999) b.p[0] = 1.f; }

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float4* buf; CK(hipMalloc(&buf, 64 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 200;
  auto time_chain = [&](const char* name, auto launch) -> int {
    for (int rep = 0; rep < 2; ++rep) {           // eager
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) launch();
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const float eager = ms * 1e3f / N;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s eager %6.2f us/kernel   graph %6.2f us/kernel\n", name, eager, ms * 1e3f / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
  };
  if (time_chain("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (float*)buf); })) return 1;
  if (time_chain("empty <<<256,256>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, (float*)buf); })) return 1;
  if (time_chain("empty <<<256,512>>> 128 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 128 << 10, st, (float*)buf); })) return 1;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
  if (time_chain("empty <<<2048,256>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st, (float*)buf); })) return 1;
  Big b; b.p = (float*)buf;
  if (time_chain("empty, 328-byte argument <<<256,256>>>", [&] { hipLaunchKernelGGL(k_bigarg, dim3(256), dim3(256), 0, st, b); })) return 1;
  if (time_chain("touch 16 KB / workgroup <<<256,256>>> (4 MB)", [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, buf, 1024); })) return 1;
  if (time_chain("touch 16 KB / workgroup <<<1024,256>>> (16 MB)", [&] { hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, st, buf, 1024); })) return 1;
  if (time_chain("touch 4 KB / workgroup <<<4096,256>>> (16 MB)", [&] { hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st, buf, 256); })) return 1;
  return 0;
}
