// Persistent per-XCD schedule for gfx950 (MI355X): the body of one denoiser forward (everything between the input
// packing and conv_out) as ONE launch of 256 workgroups, one per CU.
//
// Why: with ~280 dependent launches per forward, each kernel boundary costs ~4 us of launch / drain / L2 write-back and
// leaves the next kernel a cold L2 (docs/HISTORY.md §4, profiles/r01_gemm_phase_trace.txt) - more than most of the
// contractions themselves.  The batch maps one-to-one onto the chip: utterance b is processed by XCD b % 8 (32 CUs,
// its own 4 MB L2); utterances never exchange data (no BatchNorm, attention / norms are per utterance), so the only
// synchronisation between consecutive operations is an XCD-LOCAL barrier (agent-scope atomic counter, ~1.2 us:
// tools/micro/xcd_barrier.hip) and the activations of an utterance stay in that XCD's L2 from producer to consumer.
//
// Rules that make this correct (verified by tools/micro/xcd_barrier.hip):
//  * every load of data written earlier in the launch uses `sc1` (bypasses the CU's L1, served by the XCD's L2):
//    the tile routines are instantiated with SC1 = true;
//  * a workgroup's stores are complete (s_waitcnt vmcnt(0)) before it arrives at the barrier;
//  * utterances own disjoint rows (>= 128-byte aligned) of every buffer, so XCDs never share a cache line;
//  * all 256 workgroups must be co-resident (1 per CU: 512 threads, > 80 KB LDS); the poll loop is bounded and
//    raises an error flag instead of hanging.
#include "dv_common.h"
#include "gemm_tile.h"
#include "attn_tile.h"
#include "misc_body.h"

__device__ __forceinline__ void xcd_barrier(PersistSync* s, unsigned xcc, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's stores have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&s->arrive[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_fetch_add(&s->arrive[xcc][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 17) || __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(&s->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // never hang the device
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

template <int BM, int BN, int BK, int WM, int WN>
__device__ __forceinline__ void run_gemm(const GemmParams& g, int b, int rank, int nwg, char* smem) {
  const int mt = g.T_out / BM, nt = (g.N + BN - 1) / BN;
  for (int t = rank; t < mt * nt; t += nwg) {
    gemm_tile<BM, BN, BK, WM, WN, 3, 2, true>(g, b * g.T_out + (t / nt) * BM, (t % nt) * BN, smem);
    __syncthreads();          // the LDS ring and the staged bias are reused by the next tile
  }
}

template <int DP>
__device__ __forceinline__ void run_attn(const AttnParams& a, int b, int rank, int nwg, char* smem) {
  const int qb = (a.Tq + 255) / 256;
  for (int t = rank; t < qb * a.H; t += nwg) {
    attn_tile<DP, 8, 3, true>(a, t % qb, t / qb, b, smem);
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void k_persist(const PersistOp* __restrict__ ops, int n_ops, PersistSync* sync, int B) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  __shared__ unsigned s_xcc, s_rank;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    s_xcc = xcc;
    s_rank = atomicAdd(&sync->rank[xcc][0], 1u);
  }
  __syncthreads();
  const unsigned xcc = __builtin_amdgcn_readfirstlane(s_xcc);
  const int rank = (int)__builtin_amdgcn_readfirstlane(s_rank);
  const int nwg = (int)(gridDim.x >> 3);                  // workgroups per XCD (round-robin dispatch: 32)
  if (rank >= nwg) {                                      // dispatch was not balanced over the XCDs: refuse to run
    if (threadIdx.x == 0) __hip_atomic_store(&sync->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  unsigned phase = 0;
  const bool stamp = xcc == 0 && rank == 0 && threadIdx.x == 0;
  if (stamp) sync->ticks[0] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n_ops; ++i) {
    const PersistOp& o = ops[i];
    for (int b = (int)xcc; b < B; b += 8) {
      switch (o.type) {
        case POP_GEMM:
          switch (o.cfg) {
            case 0: run_gemm<128, 128, 32, 2, 2>(o.g, b, rank, nwg, smem); break;    // any channel count (BK = 32)
            case 1: run_gemm<128, 64, 64, 4, 1>(o.g, b, rank, nwg, smem); break;     // GEGLU (FN = 2), channels % 64 == 0
            default: run_gemm<64, 64, 64, 2, 2>(o.g, b, rank, nwg, smem); break;     // channels % 64 == 0
          }
          break;
        case POP_ATTN:
          switch (o.cfg) {
            case 16: run_attn<16>(o.a, b, rank, nwg, smem); break;
            case 32: run_attn<32>(o.a, b, rank, nwg, smem); break;
            case 48: run_attn<48>(o.a, b, rank, nwg, smem); break;
            default: run_attn<64>(o.a, b, rank, nwg, smem); break;
          }
          break;
        case POP_GN: {
          const int tasks = o.gn_chunks * o.gn.groups;
          for (int t = rank; t < tasks; t += nwg) {
            gn_apply_body<512, true>(o.gn, o.gn_rpb, t % o.gn_chunks, t / o.gn_chunks, b);
            __syncthreads();
          }
          break;
        }
        default: {   // POP_SPLIT
          const int64_t i0 = (int64_t)b * o.n4_per_item;
          split_body<true>(o.sp_in, o.sp_hi, o.sp_lo, i0, i0 + o.n4_per_item, (int64_t)rank * 512 + threadIdx.x, (int64_t)nwg * 512);
          break;
        }
      }
    }
    ++phase;
    if (i + 1 < n_ops) xcd_barrier(sync, xcc, phase * (unsigned)nwg);
    if (stamp && i + 1 < 1024) sync->ticks[i + 1] = __builtin_amdgcn_s_memtime();
  }
}

static size_t persist_smem() {
  // GEMM rings (split planes): 128x128x32 and 64x64x64 = 4 stages x 32 KB; 128x64x64 = 3 stages x 48 KB
  size_t need = 3 * (128 + 64) * 64 * 2 * 2;
  const size_t att[] = {(size_t)AttnGeom<16, 8>::smem_bytes(2), (size_t)AttnGeom<32, 8>::smem_bytes(2),
                        (size_t)AttnGeom<48, 8>::smem_bytes(2), (size_t)AttnGeom<64, 8>::smem_bytes(2)};
  for (size_t a : att) need = a > need ? a : need;
  return need;
}

hipError_t persist_init() {
  static bool done = false;
  if (done) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)persist_smem());
  if (e != hipSuccess) return e;
  int dev = 0, cus = 0, occ = 0;
  if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
  if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
  if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist, 512, persist_smem())) != hipSuccess) return e;
  if (cus != 256 || occ < 1) return hipErrorNotSupported;                   // needs one resident workgroup on each of 256 CUs
  done = true;
  return hipSuccess;
}

hipError_t launch_persist(const PersistOp* ops_dev, int n_ops, PersistSync* sync, int B, hipStream_t st) {
  hipError_t e = hipMemsetAsync(sync, 0, sizeof(PersistSync), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_persist, dim3(256), dim3(512), persist_smem(), st, ops_dev, n_ops, sync, B);
  return hipGetLastError();
}
