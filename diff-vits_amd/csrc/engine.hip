// Plan-compile-run engine of the denoiser behind the C ABI of include/dvits_hip.h.
//
// dv_unet_prepare() turns (constructor config, state dict, B, T, L) into a static schedule
// of fused HIP kernel launches over a channels-last fp32 activation arena (SURVEY.md
// Appendix A gives the block order; reference unet1d/unet_1d_condition.py:743-1037).
// Step-invariant work (pooled-text embedding, the cross-attention K/V projections, the
// mask bias) is a second schedule run by dv_unet_set_cond().
#include "../../include/dvits_hip.h"
#include "dv_common.h"

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

// ----------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
int dv_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* dv_last_error(void) { return g_err; }
#ifndef DV_SRC_HASH
#define DV_SRC_HASH "unknown"
#endif
// the build identity: profiles/*_pmc_roofline.json is stamped with it and bench.py reports those figures only while it
// still is the loaded library's (csrc/Makefile hashes every kernel source into DV_SRC_HASH)
extern "C" const char* dv_version(void) { return "dvits_hip 0.3 gfx950 src=" DV_SRC_HASH; }
// Schedule generations are PROCESS-global and monotonic: a captured sampler graph is keyed on (handle address,
// generation), and a destroyed handle's address can be handed out again by `new` - with a per-handle counter the new
// handle's first prepare would reproduce the old key and replay a graph that points into freed memory.
static std::atomic<int64_t> g_generation{0};

#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return dv_fail(DV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ----------------------------------------------------------------------------- arena
// Plan-time allocator over one device slab: the schedule is static and stream-ordered, so a
// buffer released at plan time may be handed to any later op.
struct Arena {
  struct Blk { size_t off, size; };
  std::vector<Blk> free_;
  std::map<size_t, size_t> live_;   // off -> size
  size_t top = 0, high = 0;
  bool reuse = true;
  // exact: a released block is only handed to a later request of exactly the same byte count, and blocks are never
  // split or merged.  Every tensor is utterance-major ([B][...] contiguous), so two tensors of equal size give each
  // utterance the same byte range: an address then belongs to ONE utterance for the whole schedule, which is what lets
  // the persistent per-XCD launch keep utterance b in XCD b's (non-coherent) L2 without ever sharing a line.
  bool exact = false;
  std::map<size_t, std::vector<size_t>> classes_;   // exact mode: true byte count -> free offsets
  std::map<size_t, size_t> true_size_;              // exact mode: off -> requested byte count
  size_t alloc(size_t bytes) {
    const size_t want = bytes;
    bytes = (bytes + 255) / 256 * 256;
    if (reuse && exact) {
      auto it = classes_.find(want);
      if (it != classes_.end() && !it->second.empty()) {
        size_t off = it->second.back();
        it->second.pop_back();
        live_[off] = bytes; true_size_[off] = want;
        return off;
      }
      size_t off = top;
      top += bytes;
      if (top > high) high = top;
      live_[off] = bytes; true_size_[off] = want;
      return off;
    }
    if (reuse) {
      int best = -1;
      for (int i = 0; i < (int)free_.size(); ++i)
        if (free_[i].size >= bytes && (best < 0 || free_[i].size < free_[best].size)) best = i;
      if (best >= 0) {
        size_t off = free_[best].off;
        if (free_[best].size == bytes) free_.erase(free_.begin() + best);
        else { free_[best].off += bytes; free_[best].size -= bytes; }
        live_[off] = bytes;
        return off;
      }
    }
    size_t off = top;
    top += bytes;
    if (top > high) high = top;
    live_[off] = bytes;
    return off;
  }
  void release(size_t off) {
    auto it = live_.find(off);
    if (it == live_.end()) return;
    size_t size = it->second;
    live_.erase(it);
    if (!reuse) return;
    if (exact) { classes_[true_size_[off]].push_back(off); return; }
    // insert sorted by offset and coalesce
    size_t i = 0;
    while (i < free_.size() && free_[i].off < off) ++i;
    free_.insert(free_.begin() + i, Blk{off, size});
    if (i + 1 < free_.size() && free_[i].off + free_[i].size == free_[i + 1].off) {
      free_[i].size += free_[i + 1].size;
      free_.erase(free_.begin() + i + 1);
    }
    if (i > 0 && free_[i - 1].off + free_[i - 1].size == free_[i].off) {
      free_[i - 1].size += free_[i].size;
      free_.erase(free_.begin() + i);
    }
    if (!free_.empty() && free_.back().off + free_.back().size == top) {
      top = free_.back().off;
      free_.pop_back();
    }
  }
};

// ----------------------------------------------------------------------------- data
struct RawW { float* p = nullptr; std::vector<int64_t> shape; size_t numel = 0; };

struct PackedW {
  bf16_t* hi = nullptr; bf16_t* lo = nullptr; float* bias = nullptr;
  bf16_t* fhi = nullptr; bf16_t* flo = nullptr;   // fragment-major copies for the row-block chains (made on demand)
  float* u = nullptr;           // sum_k gamma[k]*W[n,k] (fused-LayerNorm consumers), packed row order
  int Kp = 0, N = 0, N_pad = 0;
};

struct Act {   // channels-last fp32 [B*Tp, C] (+ GN statistics) (+ split planes when the consumer is a resampling conv)
  float* p = nullptr; int C = 0, T = 0;   // T: frames that exist per utterance
  int Tp = 0;                 // row pitch per utterance (Builder::pitch: T, or T rounded up to whole 32-frame blocks - GemmParams Tv_out)
  float* stat = nullptr;      // per-column slab [B*T/32, C, 2] (consumer: k_gn_apply), or
  float* stat16 = nullptr;    // per 32x16-block statistics [B*T/32, C/16, 2] (consumer: a conv that normalises its own operand)
  bf16_t* pl_hi = nullptr; bf16_t* pl_lo = nullptr;
  // GroupNorm + SiLU of this tensor for ONE consumer (`n_pre` = its norm's weight prefix), written by the producer GEMM's
  // own epilogue (GnxParams); the consumer releases them
  bf16_t* n_hi = nullptr; bf16_t* n_lo = nullptr; std::string n_pre;
  // ... of [this tensor | a skip tensor] when that consumer is an up-path resnet block (GnxParams sk_*): the skip's share of
  // the normalised planes and of the raw planes (the folded 1x1 shortcut's operand), [B*Tp, skip channels]
  bf16_t* sn_hi = nullptr; bf16_t* sn_lo = nullptr; bf16_t* sr_hi = nullptr; bf16_t* sr_lo = nullptr;
};

typedef std::function<hipError_t(hipStream_t)> OpFn;

struct Probe { std::string name; float* p; int T, C, Tp; };

struct dv_unet {
  dv_unet_cfg cfg{};
  dv_penc_cfg pcfg{};                        // prompt-encoder handles (dv_penc wraps a dv_unet core)
  std::map<std::string, RawW> w;
  bool weights_dirty = true;
  // prepared state
  bool prepared = false, cond_set = false;
  int B = 0, T = 0, L = 0, precision = 0, force_up = 0;
  std::vector<void*> owned;                  // hipMalloc'ed per prepared shape (tables, tickets, scratch)
  std::vector<void*> owned_w;                // hipMalloc'ed packed weights: kept across prepares while the weights and the
                                             // precision stay the same (a new (B, T, L) only re-plans the schedule)
  int packed_prec = -1;
  std::map<std::string, PackedW> packed;
  char* slab = nullptr; size_t slab_bytes = 0;
  bf16_t* zero_page = nullptr;               // DV_ZERO_PAGE_BYTES zero bytes: source of padded rows for the LDS-DMA
  unsigned* sk_tickets = nullptr;            // per-tile arrival counters of the fused split-K pairs (zero between launches)
  // GroupNorm finished in the producer GEMM's epilogue (GnxParams): the ops' exchange words (one pool, reset to EMPTY by
  // the first kernel of every forward) and the time-out flag, in host memory the device writes through (read without a
  // synchronisation)
  static constexpr size_t GNX_POOL = (size_t)1 << 19;   // 8-byte words (4 MiB)
  unsigned long long* gnx_pool = nullptr;
  size_t gnx_words = 0;                      // words in use by the prepared schedule
  unsigned* gnx_status = nullptr;
  int gnx_ops = 0;
  bool exclusive = true;                     // false: other streams may run kernels beside this handle's (no in-launch waits)
  std::vector<OpFn> step_ops, cond_ops;
  // GEMM launch parameters live here (stable addresses): the prepare-time tuner rewrites their tile choice in place
  std::vector<std::unique_ptr<GemmParams>> gemm_store;
  int tuned_gemms = 0, tuned_changed = 0;
  struct OpMeta {
  const char* kind; double flops; std::string desc;
  const GemmParams* gp;         // the operation's GEMM parameters (tile / split-K chosen at prepare time), or null
  int launches() const { return gp && gp->sk_buf && gp->sk_split >= 2 && !(gp->sk_ticket && gp->sk_split == 2) ? 2 : 1; }
};
  std::vector<OpMeta> step_meta;          // parallel to step_ops (profiling / roofline report)
  std::vector<Probe> probes;
  // persistent per-XCD schedule (persist.hip): descriptors of the step ops that can run inside it
  std::vector<PersistOp> pops;
  std::vector<int> step_pop;              // parallel to step_ops: index into pops, or -1
  PersistOp* pops_dev = nullptr; PersistSync* psync = nullptr;
  int p_begin = 0, p_end = 0;             // step_ops[p_begin, p_end) run as one persistent launch when persist_on
  bool persist_on = false;
  double flops = 0;
  bool keep_intermediates = false;
  // per-call I/O (read by the ops when they are enqueued)
  struct { const float* x = nullptr; int cx = 0; const float* cond = nullptr; const float* t = nullptr; float* y = nullptr;
           const float* enc = nullptr; const float* mask = nullptr;
           const float* tp_base = nullptr; } io;   // tp_base: this evaluation's rows of the batched time_emb_proj table, or null
  // Time-embedding chain of ALL evaluations of a sampler run in one pass (dv_unet_temb_all): the timesteps of a compiled
  // loop are known up front, so sincos -> linear_1 -> linear_2 (+ pooled text) -> the 22 time_emb_proj GEMVs leave the
  // per-step schedule (4 launches per step) for 5 launches per run.  Row r of every buffer = (evaluation r / B, item r % B).
  struct TembCtx {
    bool ok = false;
    int begin = 0, end = 0;                  // step_ops [begin, end): the per-step chain, skipped when tp_base is set
    int C0 = 0, E = 0, tt = 0;
    const float* w1T = nullptr; const float* b1 = nullptr; const float* w2T = nullptr; const float* b2 = nullptr;
    const float* WtT = nullptr; const float* bt = nullptr; const float* aug_emb = nullptr;
    const float* tproj_arena = nullptr;      // the per-step table the schedule's operations point into
    float* tsin = nullptr; float* h1 = nullptr; float* emb = nullptr; float* tproj_all = nullptr; int rows_cap = 0;
  } temb;
  int64_t generation = 0;
};

static void unet_release_packed(dv_unet* u) {
  for (void* p : u->owned_w) (void)hipFree(p);
  u->owned_w.clear();
  u->packed.clear();
  u->packed_prec = -1;
}
static void unet_release_prepared(dv_unet* u, bool keep_packed = false) {
  for (void* p : u->owned) (void)hipFree(p);
  u->owned.clear();
  u->zero_page = nullptr;
  u->sk_tickets = nullptr;
  u->gnx_pool = nullptr; u->gnx_words = 0; u->gnx_ops = 0;
  for (float* b : {u->temb.tsin, u->temb.h1, u->temb.emb, u->temb.tproj_all}) if (b) (void)hipFree(b);
  u->temb = dv_unet::TembCtx{};
  u->io.tp_base = nullptr;
  if (!keep_packed) unet_release_packed(u);
  if (u->slab) (void)hipFree(u->slab);
  u->slab = nullptr; u->slab_bytes = 0;
  u->step_ops.clear(); u->cond_ops.clear(); u->probes.clear(); u->step_meta.clear(); u->gemm_store.clear();
  u->pops.clear(); u->step_pop.clear(); u->pops_dev = nullptr; u->psync = nullptr; u->persist_on = false; u->p_begin = u->p_end = 0;
  u->prepared = false; u->cond_set = false; u->flops = 0;
}

// ----------------------------------------------------------------------------- C ABI: lifetime
extern "C" int dv_unet_create(const dv_unet_cfg* cfg, dv_unet** out) {
  if (!cfg || !out) return dv_fail(DV_ERR_INVALID, "dv_unet_create: null argument");
  if (cfg->n_levels < 2 || cfg->n_levels > 6) return dv_fail(DV_ERR_INVALID, "n_levels must be 2..6");
  for (int i = 0; i < cfg->n_levels; ++i)
    if (cfg->block_out_channels[i] % 32 != 0 || cfg->block_out_channels[i] <= 0)
      return dv_fail(DV_ERR_INVALID, "block_out_channels[%d]=%d must be a positive multiple of 32", i,
                     cfg->block_out_channels[i]);
  if (cfg->cross_attention_dim % 32 != 0) return dv_fail(DV_ERR_INVALID, "cross_attention_dim must be a multiple of 32");
  if (cfg->num_heads <= 0 || cfg->norm_num_groups <= 0 || cfg->norm_num_groups > 64)
    return dv_fail(DV_ERR_INVALID, "bad num_heads / norm_num_groups");
  for (int i = 0; i < cfg->n_levels; ++i) {
    int c = cfg->block_out_channels[i];
    if (c % cfg->num_heads != 0 || (c / cfg->num_heads) % 4 != 0 || c / cfg->num_heads > 64)
      return dv_fail(DV_ERR_INVALID, "head dim %d/%d must be a multiple of 4 and <= 64", c, cfg->num_heads);
    if (c % cfg->norm_num_groups != 0 || (c / cfg->norm_num_groups) % 4 != 0)
      return dv_fail(DV_ERR_INVALID, "channels per GroupNorm group must be a multiple of 4");
  }
  if (cfg->cross_attention_dim % cfg->add_embed_heads != 0 || cfg->cross_attention_dim / cfg->add_embed_heads > 8)
    return dv_fail(DV_ERR_INVALID, "add_embed_heads must divide cross_attention_dim with <= 8 dims per head");
  dv_unet* u = new dv_unet();
  u->cfg = *cfg;
  *out = u;
  return DV_OK;
}

extern "C" void dv_unet_destroy(dv_unet* u) {
  if (!u) return;
  (void)hipDeviceSynchronize();
  unet_release_prepared(u);
  if (u->gnx_status) (void)hipHostFree(u->gnx_status);
  for (auto& kv : u->w)
    if (kv.second.p) (void)hipFree(kv.second.p);
  delete u;
}

extern "C" int dv_unet_set_weight(dv_unet* u, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim) {
  if (!u || !name || !dev_ptr || !shape || ndim < 1 || ndim > 4) return dv_fail(DV_ERR_INVALID, "dv_unet_set_weight: bad argument");
  size_t n = 1;
  std::vector<int64_t> sh(shape, shape + ndim);
  for (auto s : sh) n *= (size_t)s;
  RawW& r = u->w[name];
  if (r.numel != n) {
    if (r.p) (void)hipFree(r.p);
    r.p = nullptr;
    HIPCHK(hipMalloc((void**)&r.p, n * sizeof(float)));
  }
  r.shape = sh;
  r.numel = n;
  HIPCHK(hipMemcpy(r.p, dev_ptr, n * sizeof(float), hipMemcpyDeviceToDevice));
  u->weights_dirty = true;
  return DV_OK;
}

// ----------------------------------------------------------------------------- plan builder
static bool autotune_on() {
  const char* e = getenv("DVITS_GEMM_AUTOTUNE");       // opt-in: "1", or "2" to also report how many GEMMs moved
  return e && (e[0] == '1' || e[0] == '2');
}

// Prepare-time tile tuner.  The denoiser's GEMMs sit where per-workgroup fixed cost, rounds of workgroups and k-loop
// efficiency trade against each other shape by shape (docs/HISTORY.md §4), so every GEMM of the schedule is timed on its real
// operands with each tile of the menu that can run it (and, where scratch was offered, with and without the split-K
// pair), behind an L2 flush as in the real sequence (its inputs were written by the previous kernel and the L2 does
// not survive a kernel boundary), and keeps the fastest.  ~0.1 s per prepare.  Off by default (DVITS_GEMM_AUTOTUNE=1):
// measured at the bench shape it moves 17-18 of 180 GEMMs off the heuristic's tile for +0.5 % end to end, and it makes
// the schedule (hence the float32 rounding of the result) depend on timing noise.
static int autotune_gemms(dv_unet* u, int precision) {
  u->tuned_gemms = u->tuned_changed = 0;
  if (!autotune_on() || u->gemm_store.empty()) return DV_OK;
  HIPCHK(hipDeviceSynchronize());                      // weight packing (pack stream) has finished
  hipStream_t st = nullptr;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  const size_t flush_bytes = 48u << 20;
  void* flush = nullptr;
  HIPCHK(hipMalloc(&flush, flush_bytes));
  auto time_one = [&](const GemmParams& t, float& best) -> bool {
    if (launch_gemm(t, precision, st) != hipSuccess) { (void)hipGetLastError(); return false; }   // also warms the code
    best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      if (hipMemsetAsync(flush, r, flush_bytes, st) != hipSuccess) return false;
      if (hipEventRecord(e0, st) != hipSuccess) return false;
      if (launch_gemm(t, precision, st) != hipSuccess) return false;
      if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return false;
      float ms = 0;
      if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
      best = ms < best ? ms : best;
    }
    return true;
  };
  for (auto& up : u->gemm_store) {
    GemmParams* gp = up.get();
    if (!gp->out && !gp->out_hi) continue;             // output pointer patched per call (conv_out): heuristic
    int cands[12];
    const int n = gemm_candidates(*gp, cands, 12);
    float best = 1e30f;
    int best_tile = gp->force_tile, best_split = gp->sk_split;
    const int heur_split = gp->sk_split;
    {   // the heuristic's own choice is the incumbent: a candidate must beat it by 3 % to replace it
      float ms;
      if (time_one(*gp, ms)) best = ms * 0.97f;
    }
    for (int c = 0; c < n; ++c)
      for (int sp = 0; sp < (gp->sk_buf ? 2 : 1); ++sp) {
        GemmParams t = *gp;
        t.force_tile = cands[c];
        t.sk_split = sp ? (heur_split > 2 ? heur_split : 2) : 0;
        float ms;
        if (time_one(t, ms) && ms < best) { best = ms; best_tile = cands[c]; best_split = t.sk_split; }
      }
    if (best_tile != gp->force_tile || best_split != heur_split) u->tuned_changed++;
    gp->force_tile = best_tile;
    gp->sk_split = best_split;
    u->tuned_gemms++;
  }
  if (const char* e = getenv("DVITS_GEMM_AUTOTUNE"))
    if (e[0] == '2') fprintf(stderr, "[dvits] GEMM tuner: %d of %d GEMMs moved off the heuristic tile\n", u->tuned_changed, u->tuned_gemms);
  (void)hipFree(flush);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  HIPCHK(hipDeviceSynchronize());
  return DV_OK;
}

struct Builder {
  dv_unet* u;
  Arena arena;
  bool dry;                       // first pass: measure the arena only
  int B, T, L, prec;
  std::string err;
  hipStream_t pack_stream = nullptr;
  bool fuse_ln = [] { const char* e = getenv("DVITS_FUSE_LN"); return !(e && e[0] == '0'); }();   // default on
  bool merge_ff = [] { const char* e = getenv("DVITS_MERGE_FF"); return !(e && e[0] == '0'); }();  // default on

  // ---- weights
  const RawW* raw(const std::string& name) {
    auto it = u->w.find(name);
    if (it == u->w.end()) { if (err.empty()) err = "missing weight: " + name; return nullptr; }
    return &it->second;
  }
  const float* W(const std::string& name) { const RawW* r = raw(name); return r ? r->p : nullptr; }
  bool has(const std::string& name) { return u->w.count(name) != 0; }

  // ---- arena
  float* alloc(size_t floats) {
    size_t off = arena.alloc(floats * sizeof(float));
    return dry ? reinterpret_cast<float*>(0x1000 + off) : reinterpret_cast<float*>(u->slab + off);
  }
  void release(const void* p) {
    if (u->keep_intermediates) return;
    size_t off = dry ? (reinterpret_cast<size_t>(p) - 0x1000) : (size_t)(reinterpret_cast<const char*>(p) - u->slab);
    arena.release(off);
  }
  const char* cur_kind = "misc";
  const GemmParams* cur_gp = nullptr;     // GEMM parameters of the operation being emitted
  double cur_flops = 0;
  std::string cur_desc;
  void emit(std::vector<OpFn>& ops, OpFn f, const PersistOp* pop = nullptr) {
    if (dry) return;
    ops.push_back(std::move(f));
    if (&ops == &u->step_ops) {
      u->step_meta.push_back({cur_kind, cur_flops, cur_desc, cur_gp});
      if (pop) { u->step_pop.push_back((int)u->pops.size()); u->pops.push_back(*pop); }
      else u->step_pop.push_back(-1);
    }
    cur_kind = "misc"; cur_flops = 0; cur_desc.clear(); cur_gp = nullptr;
  }
  // descriptor of a GEMM for the persistent schedule, or false if its shape is outside what persist.hip instantiates
  bool persist_gemm(const GemmParams& gin, PersistOp& po) {
    if (prec != DV_PREC_BF16X3 || gin.epi == EPI_STORE_NCT || !gin.w_lo) return false;
    bool k64 = true;
    for (int s2 = 0; s2 < gin.nseg; ++s2) k64 = k64 && gin.seg[s2].c0 % 64 == 0 && gin.seg[s2].c1 % 64 == 0;
    int cfg, bm, bk;
    if (gin.epi == EPI_GEGLU) { if (!k64) return false; cfg = 1; bm = 128; bk = 64; }
    else {
      const bool big = gin.T_out % 128 == 0 && (gin.T_out / 128) * ((gin.N + 127) / 128) >= 24;
      if (big || !k64) { cfg = 0; bm = 128; bk = 32; } else { cfg = 2; bm = 64; bk = 64; }
    }
    if (gin.T_out % bm != 0 || gin.M != B * gin.T_out) return false;
    if (gin.T_out < 1 || (unsigned long long)(gin.M > 0 ? gin.M : 1) * (unsigned long long)gin.T_out >= (1ull << 32)) return false;   // (tout_magic: launch_gemm's guard)
    po = PersistOp{};
    po.type = POP_GEMM; po.cfg = cfg; po.g = gin;
    po.g.tout_magic = gemm_tout_magic(gin.T_out);      // (launch_gemm's normalisations: the persistent launch runs the tile routine directly)
    if (po.g.Tv_out <= 0) po.g.Tv_out = po.g.T_out;
    if (po.g.Tv_in <= 0) po.g.Tv_in = po.g.T_in;
    for (int s2 = 0; s2 < gin.nseg; ++s2) po.g.seg[s2].nkt = gin.seg[s2].taps * (gin.seg[s2].c0 + gin.seg[s2].c1) / bk;
    return true;
  }
  void probe(const std::string& name, const float* p, int T_, int C_) {
    if (!dry && u->keep_intermediates) u->probes.push_back(Probe{name, const_cast<float*>(p), T_, C_, T_ > 1 ? pitch(T_) : T_});
  }

  // ---- weight packing (only in the real pass; device work on pack_stream)
  struct Piece { std::string w; int kind, C, taps, c_pad, k_off, n_off; std::string kscale; int geglu; };
  struct BiasPiece { std::string bias, bias2, foldW, foldBeta; int N, C, n_off, geglu; };

  const PackedW* pack(const std::string& key, int N, int Kp, const std::vector<Piece>& pieces,
                      const std::vector<BiasPiece>& biases) {
    if (dry) { static PackedW dummy; return &dummy; }
    auto it = u->packed.find(key);
    if (it != u->packed.end()) return &it->second;
    PackedW pw;
    pw.N = N; pw.Kp = Kp; pw.N_pad = rup(N, 128);
    const size_t elems = (size_t)pw.N_pad * Kp;
    if (hipMalloc((void**)&pw.hi, elems * 2) != hipSuccess) { err = "hipMalloc(packed weights) failed"; return nullptr; }
    u->owned_w.push_back(pw.hi);
    (void)hipMemsetAsync(pw.hi, 0, elems * 2, pack_stream);
    if (prec == DV_PREC_BF16X3) {
      if (hipMalloc((void**)&pw.lo, elems * 2) != hipSuccess) { err = "hipMalloc(packed weights) failed"; return nullptr; }
      u->owned_w.push_back(pw.lo);
      (void)hipMemsetAsync(pw.lo, 0, elems * 2, pack_stream);
    }
    for (const Piece& pc : pieces) {
      const RawW* r = raw(pc.w);
      if (!r) return nullptr;
      PackSpec s{};
      s.src = r->p; s.N = (int)r->shape[0]; s.kind = pc.kind; s.C = pc.C; s.taps = pc.taps; s.c_pad = pc.c_pad;
      s.k_off = pc.k_off; s.n_off = pc.n_off; s.geglu = pc.geglu;
      s.kscale = pc.kscale.empty() ? nullptr : W(pc.kscale);
      if ((size_t)s.N * s.C * s.taps != r->numel) { err = "weight shape mismatch: " + pc.w; return nullptr; }
      if (launch_pack_weight(s, pw.hi, pw.lo, Kp, pack_stream) != hipSuccess) { err = "pack_weight launch failed"; return nullptr; }
      if (s.kscale) {   // LayerNorm-folded piece: u[n] = sum_c gamma[c] * W[n, c]
        if (!pw.u) {
          if (hipMalloc((void**)&pw.u, (size_t)pw.N_pad * 4) != hipSuccess) { err = "hipMalloc(u) failed"; return nullptr; }
          u->owned_w.push_back(pw.u);
          (void)hipMemsetAsync(pw.u, 0, (size_t)pw.N_pad * 4, pack_stream);
        }
        if (launch_fold_bias(r->p, nullptr, s.kscale, pw.u, s.N, s.C, pc.n_off, pc.geglu, pack_stream) != hipSuccess) {
          err = "fold(u) launch failed"; return nullptr;
        }
      }
    }
    if (!biases.empty()) {
      if (hipMalloc((void**)&pw.bias, (size_t)pw.N_pad * 4) != hipSuccess) { err = "hipMalloc(bias) failed"; return nullptr; }
      u->owned_w.push_back(pw.bias);
      (void)hipMemsetAsync(pw.bias, 0, (size_t)pw.N_pad * 4, pack_stream);
      for (const BiasPiece& bp : biases) {
        const float* b1 = bp.bias.empty() ? nullptr : W(bp.bias);
        const float* fw = bp.foldW.empty() ? nullptr : W(bp.foldW);
        const float* fb = bp.foldBeta.empty() ? nullptr : W(bp.foldBeta);
        if (launch_fold_bias(fw, b1, fb, pw.bias, bp.N, bp.C, bp.n_off, bp.geglu, pack_stream) != hipSuccess) {
          err = "fold_bias launch failed"; return nullptr;
        }
        if (!bp.bias2.empty()) {   // second additive bias (conv2 + shortcut): accumulate on the host side is
                                   // avoided by a tiny axpy through lincomb: bias += bias2
          static const float one_one[8] = {1.f, 1.f, 0, 0, 0, 0, 0, 0};
          float* dcoef = nullptr;
          if (hipMalloc((void**)&dcoef, sizeof(one_one)) != hipSuccess) { err = "hipMalloc failed"; return nullptr; }
          u->owned_w.push_back(dcoef);
          (void)hipMemcpyAsync(dcoef, one_one, sizeof(one_one), hipMemcpyHostToDevice, pack_stream);
          (void)launch_lincomb(pw.bias + bp.n_off, pw.bias + bp.n_off, W(bp.bias2), nullptr, nullptr, nullptr, dcoef, bp.N,
                               pack_stream);
        }
      }
    }
    u->packed[key] = pw;
    return &u->packed[key];
  }

  // ---- op emitters
  struct Planes { bf16_t* hi = nullptr; bf16_t* lo = nullptr; };
  Planes alloc_planes(size_t elems) {
    Planes pl;
    pl.hi = reinterpret_cast<bf16_t*>(alloc((elems + 1) / 2));
    if (prec == DV_PREC_BF16X3) pl.lo = reinterpret_cast<bf16_t*>(alloc((elems + 1) / 2));
    return pl;
  }
  void release(const Planes& pl) { if (pl.hi) release((const void*)pl.hi); if (pl.lo) release((const void*)pl.lo); }
  // GroupNorm statistics of a GEMM output for its consumer are written by the GEMM's epilogue (k_gn_apply / the chain kernels /
  // the in-epilogue GroupNorm reduce them).  [Round 2 also had the CONSUMER conv normalise its own operand ("AF" tiles,
  // DVITS_FUSE_GN=1): parity-green but slower - an N-tiled conv repeats the elementwise GroupNorm + SiLU + hi/lo split of its
  // rows in every 64-column tile - and removed in round 3; docs/HISTORY.md section 4 keeps the measurements.]
  void alloc_stat(Act& a, bool want16 = false) {
    // 32-row blocks of the flat [B*Tp] row space must not span utterances: Tp % 32 == 0 (the padded row space of pitch()
    // makes that so for any T), or ONE utterance whose last block is partial (rows beyond M contribute zeros: the per-column
    // slab of configurations without whole 16-channel blocks per group)
    if (a.Tp % 32 != 0) {
      if (B == 1) a.stat = alloc((size_t)((a.Tp + 31) / 32) * a.C * 2);
      return;
    }
    if (want16 || stat16_everywhere()) a.stat16 = alloc((size_t)(B * a.Tp / 32) * (a.C / 16) * 2);
    else a.stat = alloc((size_t)(B * a.Tp / 32) * a.C * 2);
  }
  // Row pitch per utterance of a level with Tl frames.  Real utterances have any length (reference tts_infer.py:46-74,
  // resnet.py:157-160: T is arbitrary): wherever the fused schedule can run at all (32x16 block statistics at every level,
  // no exact-size arena) every level is padded to whole 32-frame blocks INSIDE the engine - the row-block chains, the block
  // statistics and the in-epilogue GroupNorm then run for any T and any B; the padding rows are kept out of every frame
  // count (GemmParams Tv_out).  DVITS_PAD_T=0 restores the unpadded row space (general-shape GroupNorm kernels, no chains).
  bool pad_on = [] { const char* e = getenv("DVITS_PAD_T"); return !(e && e[0] == '0'); }();
  int pitch(int Tl) const { return (pad_on && Tl % 32 != 0 && stat16_everywhere()) ? rup(Tl, 32) : Tl; }
  // every level has whole 16-channel blocks per GroupNorm group: all tensors carry block statistics (16x fewer entries
  // for k_gn_apply to reduce, cheaper producer epilogue); otherwise (tiny / duration-predictor configurations) per-column slabs
  bool stat16_everywhere() const {
    const char* e16 = getenv("DVITS_STAT16");      // (read per prepare: tests flip it inside one process)
    if ((e16 && e16[0] == '0') || arena.exact) return false;
    const int G = u->cfg.norm_num_groups;
    for (int i = 0; i < u->cfg.n_levels; ++i)
      if (u->cfg.block_out_channels[i] % G != 0 || (u->cfg.block_out_channels[i] / G) % 16 != 0) return false;
    return true;
  }
  void stat_out(GemmParams& g, const Act& a) { g.stats = a.stat; g.stats16 = a.stat16; }
  // Row-block chains of a transformer block (kernels_chain.hip; DVITS_CHAIN=0 restores one launch per GEMM):
  // norm -> proj_in -> LN -> to_q/k/v, and attention -> to_out + residual -> LN -> to_q of the next attention
  bool chain_on = [] { const char* e = getenv("DVITS_CHAIN"); return !(e && e[0] == '0'); }();
  // Levels with fewer than four 32-row blocks in the whole batch run one launch per GEMM instead: a chain kernel there is three
  // workgroups (x the column split) streaming whole weight matrices through one CU each, while 32x32 GEMM tiles spread the
  // same weights over 36-108 CUs.  One utterance of 300 frames, level 2 (96 rows, C = 384): the two chains take 35.6 + 24.7 us,
  // the five launches they replace ~45 us but the 30-step run drops 58.8 -> 55.2 ms; at 160 rows (level 1) the chains still
  // win (56.4 ms with both levels unchained).  DVITS_CHAIN_MIN_ROWS=<rows> moves the limit (0: always chain).
  int chain_min_rows = [] { const char* e = getenv("DVITS_CHAIN_MIN_ROWS"); return e ? atoi(e) : 128; }();
  bool chain_ok(int Tn, int C) const {
    const int G = u->cfg.norm_num_groups;
    return chain_on && fuse_ln && !arena.exact && prec == DV_PREC_BF16X3 && (C == 128 || C == 256 || C == 384) && Tn % 32 == 0 &&
           B * Tn >= chain_min_rows &&
           G > 0 && G <= 64 && (G & (G - 1)) == 0 && C % G == 0 && (C / G) % 16 == 0 && (Tn / 32) * (C / 16) <= (C / 128) * 2048;
  }
  // fragment-major copies of a packed weight (kernels_chain.hip k_relayout_frag), made once per prepare
  bool frag(const PackedW* cw) {
    if (dry) return true;
    PackedW* w = const_cast<PackedW*>(cw);
    if (w->fhi) return true;
    const size_t elems = (size_t)w->N_pad * w->Kp;
    if (!w->lo || hipMalloc((void**)&w->fhi, elems * 2) != hipSuccess || hipMalloc((void**)&w->flo, elems * 2) != hipSuccess) {
      err = "hipMalloc(fragment-major weights) failed"; return false;
    }
    u->owned_w.push_back(w->fhi); u->owned_w.push_back(w->flo);
    if (launch_relayout_frag(w->hi, w->fhi, w->N_pad, w->Kp, pack_stream) != hipSuccess ||
        launch_relayout_frag(w->lo, w->flo, w->N_pad, w->Kp, pack_stream) != hipSuccess) { err = "relayout launch failed"; return false; }
    return true;
  }
  // `extra_flops`: work of the launch beyond its two K = C contractions (the in-chain cross attention and its to_out GEMM) - part
  // of the operation's own count, so that the per-operation table and the family sums agree with dv_unet_stats
  void chain(std::vector<OpFn>& ops, const ChainParams& cp, const char* what, double extra_flops = 0.0) {
    cur_kind = "chain";
    cur_flops = 2.0 * (double)cp.M * cp.C * cp.C * (1 + cp.passes) + extra_flops;
    char buf[96];
    snprintf(buf, sizeof(buf), "%s M=%d C=%d N2=%d", what, cp.M, cp.C, cp.passes * cp.C);
    cur_desc = buf;
    if (!dry) u->flops += cur_flops;
    const int pr = prec;
    emit(ops, [cp, pr](hipStream_t st) { return launch_chain2(cp, pr, st); });
  }
  // GroupNorm (+ temb scale/shift) (+ SiLU) of a GEMM's own output finished in ITS epilogue (gemm_tile.h GNX; DVITS_GNX=0
  // restores the k_gn_apply launch): fills g.gnx and allocates the normalised planes; false if launch_gemm would refuse.
  // Call after g's segments, epilogue and statistics slab are set.
  // (a CU mask - HSA_CU_MASK / ROC_GLOBAL_CU_MASK - takes CUs away without hipDeviceAttributeMultiprocessorCount knowing:
  // the residency bound of the in-launch hand-over would be wrong, so it is not planned at all then)
  bool sk_fused = [] { const char* e = getenv("DVITS_SPLITK_FUSED"); return !(e && e[0] == '0'); }();   // 0: split-K as two launches
  bool gnx_on = [] { const char* e = getenv("DVITS_GNX"); return !(e && e[0] == '0') && !getenv("HSA_CU_MASK") && !getenv("ROC_GLOBAL_CU_MASK"); }();
  size_t gnx_used = 0;
  // polls before an in-launch wait gives up; DVITS_GNX_SPIN=<n> is a test hook (1: every wait that is not satisfied at once
  // times out - exercises the fallback path of engine.py deterministically)
  int gnx_spin = [] { const char* e = getenv("DVITS_GNX_SPIN"); const int v = e ? atoi(e) : 0; return v != 0 ? v : (1 << 18); }();   // (-1: give up at the first unsatisfied poll)
  // (DVITS_CU_BUDGET=<n>: plan as if the device had n CUs - for engines that are driven side by side on one device and still keep
  // their in-launch hand-overs: k engines with a budget of CUs / k each are co-resident launch by launch; bench.py --streams)
  int n_cu = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 0;
                  const char* e = getenv("DVITS_CU_BUDGET"); const int b = e ? atoi(e) : 0; return b > 0 && b < n ? b : n; }();
  // `skip`: the consumer normalises [g's output | skip] (an up-path resnet block's norm1); the skip's share of the planes goes
  // to *sk_y (and *sk_raw, if the consumer's shortcut wants the raw tensor as planes).  DVITS_GNX_CONCAT=0: not planned.
  bool gnx_cat_on = [] { const char* e = getenv("DVITS_GNX_CONCAT"); return !(e && e[0] == '0'); }();
  bool gnx_setup(GemmParams& g, const std::string& pre, float eps, const float* tscale, const float* tshift, int ld_t, bool silu,
                 Planes* y, const Act* skip = nullptr, Planes* sk_y = nullptr, Planes* sk_raw = nullptr) {
    if (!gnx_on || !u->exclusive || arena.exact || autotune_on() || !g.stats16 || n_cu <= 0) return false;
    if (skip && (!gnx_cat_on || !skip->p || !skip->stat16 || skip->Tp != g.T_out || tscale)) return false;
    GemmParams t = g;
    t.B = B;
    t.gnx = GnxParams{};
    t.gnx.groups = u->cfg.norm_num_groups;
    t.gnx.sk_c = skip ? skip->C : 0;
    int k_pad = 0;
    for (int s2 = 0; s2 < t.nseg; ++s2) k_pad += t.seg[s2].taps * (t.seg[s2].c0 + t.seg[s2].c1);
    t.c3_route = conv3_route(t);
    t.sk_split = t.c3_route ? gemm_conv3_split(t, n_cu) : gemm_splitk_plan(t.M, t.N, k_pad, t.epi);
    if (t.sk_split >= 2) {
      if (!sk_fused || ((t.M + 31) / 32) * ((t.N + 31) / 32) > 4096) return false;   // (two launches: see gemm())
      t.sk_buf = reinterpret_cast<float*>(0x1000); t.sk_ticket = reinterpret_cast<unsigned*>(0x1000);
    }
    const int nw = gemm_gnx_plan(t, n_cu);
    if (nw <= 0 || gnx_used + (size_t)nw > dv_unet::GNX_POOL) return false;
    g.gnx = t.gnx;
    gnx_fill(g.gnx, nw, g.M, g.N, pre, eps, tscale, tshift, ld_t, silu, y, skip, sk_y, sk_raw);
    return true;
  }
  // the part every producer kernel shares: `nw` exchange words of the pool, the norm's affine parameters, the planes it writes
  void gnx_fill(GnxParams& gx, int nw, int M, int N, const std::string& pre, float eps, const float* tscale, const float* tshift, int ld_t,
                bool silu, Planes* y, const Act* skip, Planes* sk_y, Planes* sk_raw) {
    gx.gamma = W(pre + ".weight"); gx.beta = W(pre + ".bias"); gx.eps = eps;
    gx.tscale = tscale; gx.tshift = tshift; gx.ld_t = ld_t; gx.silu = silu ? 1 : 0;
    gx.spin_max = gnx_spin;
    gx.xchg = dry ? reinterpret_cast<unsigned long long*>(0x1000) : u->gnx_pool + gnx_used;
    gx.status = dry ? reinterpret_cast<unsigned*>(0x1000) : u->gnx_status;
    gnx_used += ((size_t)nw + 1) & ~(size_t)1;
    if (!dry) u->gnx_words = gnx_used;
    *y = alloc_planes((size_t)M * N);
    gx.y_hi = y->hi; gx.y_lo = y->lo;
    if (skip) {
      gx.sk_x = skip->p; gx.sk_stat16 = skip->stat16;
      *sk_y = alloc_planes((size_t)M * skip->C);
      gx.sk_y_hi = sk_y->hi; gx.sk_y_lo = sk_y->lo;
      if (sk_raw) { *sk_raw = alloc_planes((size_t)M * skip->C); gx.sk_raw_hi = sk_raw->hi; gx.sk_raw_lo = sk_raw->lo; }
    }
    if (!dry) u->gnx_ops++;
  }

  // The build loop announces the single GroupNorm consumer of the next producer's output (a resnet block's norm1, or
  // conv_norm_out); the producer's last GEMM takes the offer if it can finish that GroupNorm itself.
  // (`skip`: that consumer normalises [the output | skip] - the up path's concatenation)
  struct NextNorm { std::string pre; bool raw = false; bool set = false; Act skip; float eps = 0.f; bool silu = true; } next_norm;
  void announce_norm(const std::string& pre, bool raw_planes, const Act& skip = Act{}) {
    next_norm.pre = pre; next_norm.raw = raw_planes; next_norm.skip = skip; next_norm.set = true;
    next_norm.eps = u->cfg.norm_eps; next_norm.silu = true;
  }
  // ... a Transformer2DModel's own GroupNorm (eps 1e-6, no activation; reference transformer_1d.py:262) when its first GEMM runs
  // as a launch of its own (channel counts the row-block chain does not take)
  void announce_transformer_norm(const std::string& tp, int Tp, int C) {
    if (chain_ok(Tp, C)) return;
    announce_norm(tp + "norm", false);
    next_norm.eps = 1e-6f; next_norm.silu = false;
  }
  void offer_next(GemmParams& g, Act& out) {
    if (!next_norm.set) return;
    const NextNorm nn = next_norm;
    next_norm.set = false;
    Planes y, sy, sr;
    const bool cat = nn.skip.C > 0;
    if (!gnx_setup(g, nn.pre, nn.eps, nullptr, nullptr, 0, nn.silu, &y, cat ? &nn.skip : nullptr, &sy, nn.raw ? &sr : nullptr)) return;
    out.n_hi = y.hi; out.n_lo = y.lo; out.n_pre = nn.pre;
    if (cat) { out.sn_hi = sy.hi; out.sn_lo = sy.lo; out.sr_hi = sr.hi; out.sr_lo = sr.lo; }
    // (an up-path tensor has this one consumer, which reads it through the planes alone when its shortcut is a convolution)
    const bool fp32_unread = cat && nn.raw && !u->keep_intermediates;
    if (nn.raw && !g.out_hi) {   // the consumer's folded 1x1 shortcut reads the raw tensor as split planes
      Planes pl = alloc_planes((size_t)g.M * g.N);
      out.pl_hi = pl.hi; out.pl_lo = pl.lo; g.out_hi = pl.hi; g.out_lo = pl.lo;
    }
    if (fp32_unread && cat_drop_fp32) g.out = nullptr;
  }
  // ... when the producer is the C = 128 feed-forward chain (k_chain_ff: one workgroup per row block)
  void offer_next(ChainFFParams& fp, Act& out) {
    if (!next_norm.set) return;
    const NextNorm nn = next_norm;
    next_norm.set = false;
    const bool cat = nn.skip.C > 0;
    if (!gnx_on || !gnx_ff_on || !u->exclusive || arena.exact || !fp.stats16 || n_cu <= 0) return;
    if (cat && (!gnx_cat_on || !nn.skip.p || !nn.skip.stat16 || nn.skip.Tp != fp.T)) return;
    fp.gnx = GnxParams{};
    fp.gnx.groups = u->cfg.norm_num_groups;
    fp.gnx.sk_c = cat ? nn.skip.C : 0;
    const int nw = chain_ff_gnx_plan(fp, n_cu);
    if (nw <= 0 || gnx_used + (size_t)nw > dv_unet::GNX_POOL) { fp.gnx = GnxParams{}; return; }
    Planes y, sy, sr;
    gnx_fill(fp.gnx, nw, fp.M, fp.C, nn.pre, nn.eps, nullptr, nullptr, 0, nn.silu, &y, cat ? &nn.skip : nullptr, &sy, nn.raw ? &sr : nullptr);
    out.n_hi = y.hi; out.n_lo = y.lo; out.n_pre = nn.pre;
    if (cat) { out.sn_hi = sy.hi; out.sn_lo = sy.lo; out.sr_hi = sr.hi; out.sr_lo = sr.lo; }
    if (nn.raw && !fp.out_hi) {
      Planes pl = alloc_planes((size_t)fp.M * fp.C);
      out.pl_hi = pl.hi; out.pl_lo = pl.lo; fp.out_hi = pl.hi; fp.out_lo = pl.lo;
    }
    if (cat && nn.raw && !u->keep_intermediates && cat_drop_fp32) fp.out = nullptr;
  }
  // ... when the producer is the split feed-forward launch of the C = 256 / 384 blocks (k_ff_split: its finishing tiles are
  // [64 rows x C / nspl columns], all resident)
  void offer_next(FFSplitParams& fp, Act& out) {
    if (!next_norm.set) return;
    const NextNorm nn = next_norm;
    next_norm.set = false;
    const bool cat = nn.skip.C > 0;
    if (!gnx_on || !gnx_ff_on || !u->exclusive || arena.exact || !fp.stats16 || n_cu <= 0) return;
    if (cat && (!gnx_cat_on || !nn.skip.p || !nn.skip.stat16 || nn.skip.Tp != fp.T)) return;
    fp.gnx = GnxParams{};
    fp.gnx.groups = u->cfg.norm_num_groups;
    fp.gnx.sk_c = cat ? nn.skip.C : 0;
    const int nw = ff_split_gnx_plan(fp, n_cu);
    if (nw <= 0 || gnx_used + (size_t)nw > dv_unet::GNX_POOL) { fp.gnx = GnxParams{}; return; }
    Planes y, sy, sr;
    gnx_fill(fp.gnx, nw, fp.M, fp.C, nn.pre, nn.eps, nullptr, nullptr, 0, nn.silu, &y, cat ? &nn.skip : nullptr, &sy, nn.raw ? &sr : nullptr);
    out.n_hi = y.hi; out.n_lo = y.lo; out.n_pre = nn.pre;
    if (cat) { out.sn_hi = sy.hi; out.sn_lo = sy.lo; out.sr_hi = sr.hi; out.sr_lo = sr.lo; }
    if (nn.raw && !fp.out_hi) {
      Planes pl = alloc_planes((size_t)fp.M * fp.C);
      out.pl_hi = pl.hi; out.pl_lo = pl.lo; fp.out_hi = pl.hi; fp.out_lo = pl.lo;
    }
    if (cat && nn.raw && !u->keep_intermediates && cat_drop_fp32) fp.out = nullptr;
  }
  bool ff_split_on = [] { const char* e = getenv("DVITS_FF_SPLIT"); return !(e && e[0] == '0'); }();
  bool qkv_split_on = [] { const char* e = getenv("DVITS_QKV_SPLIT"); return !(e && e[0] == '0'); }();
  int qkv_split_min_wg = [] { const char* e = getenv("DVITS_QKV_SPLIT_MIN_WG"); return e ? atoi(e) : 96; }();
  // (at C = 128 a workgroup of the 32-row chain streams 0.26 MB of weights: nothing to save - 18.1 us against 19.3 us on the split launch)
  int qkv_split_min_c = [] { const char* e = getenv("DVITS_QKV_SPLIT_MIN_C"); return e ? atoi(e) : 256; }();
  bool qkv_xa_on = [] { const char* e = getenv("DVITS_QKV_XA"); return !(e && e[0] == '0'); }();   // the cross-attention chains on k_qkv_split (MODE 2)
  // (C = 128: eight (head, row fragment) jobs walk all eight key tiles one after the other - 20 k cycles of key loop, 27.7 us against 24.2 us on the chain)
  int qkv_xa_min_c = [] { const char* e = getenv("DVITS_QKV_XA_MIN_C"); return e ? atoi(e) : 256; }();
  // ... from this many workgroups: one utterance (B = 1, T = 300 or 1024: 16-64 workgroups per launch) is bound by the chain of
  // dependent launches, and a launch with two in-launch hand-overs costs there what the two GEMM launches did (54.4 vs 53.6 ms per
  // 30-step run, 58.5 vs 57.6 at T = 1024: measured, round 5); from 128 workgroups (B = 4) the launch wins (+4.9 %)
  int ff_split_min_wg = [] { const char* e = getenv("DVITS_FF_SPLIT_MIN_WG"); return e ? atoi(e) : 96; }();
  bool gnx_ff_on = [] { const char* e = getenv("DVITS_GNX_FF"); return !(e && e[0] == '0'); }();
  bool cat_drop_fp32 = [] { const char* e = getenv("DVITS_GNX_CONCAT_KEEP_FP32"); return !(e && e[0] == '1'); }();

  // Stride-1 three-tap convolutions over 128-512 input channels run on the resident-operand kernel (k_conv3, kernels_conv.hip;
  // DVITS_CONV3=0: k_gemm): launch_gemm dispatches there when the fragment-major weights are given.  Such a launch is never split
  // over K (64 rows x all input channels are resident: the k-loop runs at the MFMA rate on the tiles there are).
  int conv3_min_tiles = [] { const char* e = getenv("DVITS_CONV3_MIN_TILES"); return e ? atoi(e) : 64; }();   // (below: k_gemm's 64x32 / 32x32 tiles spread the launch over more CUs; B = 4 / 2: 2.34 / 2.09 -> 2.24 / 2.06 ms per forward with 64 instead of 128, 32: the same)
  // 0: k_gemm; 1: k_conv3 / k_conv3s (64 x 64 tiles); 2: k_conv3u (128 x 64 tiles) - GemmParams c3_route.  Any row pitch of whole
  // 32-frame blocks (the padded row space of real utterance lengths included) and any grid size: the tiles are laid out per
  // utterance, and whether the launch may also finish its consumer's GroupNorm is gemm_gnx_plan's business (round 6; rounds 5's
  // window was 64-row multiples, 64-256 tiles).
  int conv3_route(const GemmParams& g) const {
    if (arena.exact || autotune_on() || prec != DV_PREC_BF16X3) return 0;
    GemmParams t = g;
    t.B = B;
    if (gemm_conv3_up_ok(t)) {   // the upsampling form: 128 x 64 tiles
      const int tu = gemm_conv3_row_tiles(t, 128) * (t.N / 64);
      return tu >= (conv3_min_tiles < 64 ? conv3_min_tiles : 96) ? 2 : 0;   // (a lowered DVITS_CONV3_MIN_TILES - tests, one utterance - lowers this bound too)
    }
    if (!gemm_conv3_shape_ok(t)) return 0;
    return gemm_conv3_row_tiles(t, 64) * (t.N / 64) >= conv3_min_tiles ? 1 : 0;
  }
  bool conv3_takes(const GemmParams& g) const { return conv3_route(g) != 0; }
  void gemm(std::vector<OpFn>& ops, GemmParams g, const PackedW* pw, int k_real) {
    g.w_hi = pw->hi; g.w_lo = pw->lo; g.Kp = pw->Kp; g.N_pad = pw->N_pad;
    if (!g.bias) g.bias = pw->bias;
    g.B = B;
    g.zero_page = u->zero_page;
    const int p = prec;
    const int route = conv3_route(g);
    // (the planning pass packs nothing - its PackedW is a dummy - and must size the split-K scratch exactly as the real pass will)
    const bool c3 = route != 0 && (dry || g.Kp == gemm_conv3_k(g)) && frag(pw);   // (the upsampling form has one segment too: the same K)
    g.c3_route = c3 ? route : 0;
    if (c3) { g.wf_hi = dry ? reinterpret_cast<const bf16_t*>(0x1000) : pw->fhi; g.wf_lo = dry ? reinterpret_cast<const bf16_t*>(0x1000) : pw->flo; }
    cur_kind = "gemm"; cur_flops = 2.0 * (double)g.M * (double)g.N * (double)k_real;
    {
      char buf[128];
      snprintf(buf, sizeof(buf), "M=%d N=%d K=%d taps=%d nseg=%d epi=%d stride=%d up=%d%s%s%s", g.M, g.N, k_real, g.seg[0].taps,
               g.nseg, g.epi, g.stride, g.up_mode, (g.stats || g.stats16) ? " +stats" : "", c3 ? " resident" : "",
               g.gnx.xchg ? " +gnx" : "");
      cur_desc = buf;
    }
    u->flops += dry ? 0.0 : cur_flops;
    // long K on few tiles: offer scratch for a two-launch split-K (kernels_gemm.hip decides with the same predicate)
    int k_pad = 0;
    for (int s2 = 0; s2 < g.nseg; ++s2) k_pad += g.seg[s2].taps * (g.seg[s2].c0 + g.seg[s2].c1);
    g.sk_split = arena.exact ? 0 : (c3 ? gemm_conv3_split(g, n_cu) : gemm_splitk_plan(g.M, g.N, k_pad, g.epi));
    // with the tuner on, scratch is also offered to GEMMs the heuristic would not split (the tuner times both ways)
    const bool offer = !arena.exact && autotune_on() && gemm_splitk_plan(64, 64, 1 << 20, EPI_STORE) != 0 && k_pad >= 768 && (g.epi == EPI_STORE || g.epi == EPI_RESIDUAL) &&
                       ((g.M + 63) / 64) * ((g.N + 63) / 64) <= 256;
    if (g.sk_split >= 2 || offer) {
      g.sk_buf = alloc((c3 && g.sk_split == 2 ? gemm_conv3_split_bytes(g) : gemm_splitk_bytes(g.M, g.N, g.sk_split > 2 ? g.sk_split : 2)) / sizeof(float));
      // null: two launches (DVITS_SPLITK_FUSED=0); the counters are indexed by tile id of any tile shape >= 32x32
      g.sk_ticket = (sk_fused && ((g.M + 31) / 32) * ((g.N + 31) / 32) <= 4096) ? u->sk_tickets : nullptr;
    }
    PersistOp po;
    const bool pok = !dry && persist_gemm(g, po);
    if (dry) emit(ops, OpFn{}, nullptr);
    else {
      u->gemm_store.emplace_back(new GemmParams(g));
      const GemmParams* gp = u->gemm_store.back().get();
      cur_gp = gp;
      dv_unet* uu = u;
      emit(ops, [gp, p, uu](hipStream_t st) {
        if (!uu->io.tp_base || !gp->gnx.xchg || !gp->gnx.tscale) return launch_gemm(*gp, p, st);
        GemmParams g2 = *gp;                 // this evaluation's rows of the batched time_emb_proj table
        g2.gnx.tscale = uu->io.tp_base + (gp->gnx.tscale - uu->temb.tproj_arena);
        g2.gnx.tshift = uu->io.tp_base + (gp->gnx.tshift - uu->temb.tproj_arena);
        return launch_gemm(g2, p, st);
      }, pok ? &po : nullptr);
    }
    if (g.sk_buf) release((const void*)g.sk_buf);
  }

  static GemmSeg seg(Planes a0, int c0, Planes a1, int c1, int taps, int pad) {
    GemmSeg s{};
    s.a0_hi = a0.hi; s.a0_lo = a0.lo; s.a1_hi = a1.hi; s.a1_lo = a1.lo;
    s.c0 = c0; s.c1 = c1; s.taps = taps; s.pad = pad;
    return s;
  }

  // GroupNorm (+ temb scale/shift) (+ SiLU) of [a0 | a1] -> split planes [B*T, C] for the consumer GEMM.
  // Fast path: statistics from the producers' epilogue slabs (a0.stat / a1.stat), one launch.
  // General path (T % 32 != 0): k_gn_partial + k_gn_finalize build the per-(b,c) affine first.
  Planes norm_apply(std::vector<OpFn>& ops, Act a0, Act a1, const std::string& pre, float eps, const float* tscale,
                    const float* tshift, int ld_t, bool silu, Planes* raw_out) {
    const int G = u->cfg.norm_num_groups, C = a0.C + a1.C, Tn = a0.T, Tp = a0.Tp;
    GnApplyParams gp{};
    gp.a0 = a0.p; gp.a1 = a1.p; gp.c0 = a0.C; gp.c1 = a1.C;
    gp.gamma = W(pre + ".weight"); gp.beta = W(pre + ".bias"); gp.eps = eps; gp.groups = G;
    gp.tscale = tscale; gp.tshift = tshift; gp.ld_t = ld_t; gp.silu = silu ? 1 : 0;
    gp.B = B; gp.T = Tp; gp.Tv = Tn;
    float* sc = nullptr; float* sh = nullptr;
    const bool fast16 = a0.stat16 && (a1.C == 0 || a1.stat16) && ((a0.C + a1.C) / G) % 16 == 0 && a0.C % 16 == 0;
    const bool fast = fast16 || (a0.stat && (a1.C == 0 || a1.stat));
    if (fast16) { gp.st16_0 = a0.stat16; gp.st16_1 = a1.stat16; }
    else if (fast) { gp.slab0 = a0.stat; gp.slab1 = a1.stat; }
    else {
      const int nchunk = std::max(1, std::min(64, (Tn + 63) / 64));
      double* part = reinterpret_cast<double*>(alloc((size_t)B * nchunk * G * 2 * 2));
      sc = alloc((size_t)B * C); sh = alloc((size_t)B * C);
      const int Bn = B;
      const float* gamma = gp.gamma; const float* beta = gp.beta;
      cur_kind = "gn_partial";
      emit(ops, [=](hipStream_t st) { return launch_gn_partial(a0.p, a0.C, a1.p, a1.C, part, Bn, Tn, G, nchunk, st); });
      cur_kind = "gn_finalize";
      dv_unet* uu = u;
      emit(ops, [=](hipStream_t st) {
        const float* ts = tscale; const float* tb = tshift;
        if (uu->io.tp_base && ts) {          // this evaluation's rows of the batched time_emb_proj table
          ts = uu->io.tp_base + (tscale - uu->temb.tproj_arena);
          tb = uu->io.tp_base + (tshift - uu->temb.tproj_arena);
        }
        return launch_gn_finalize(part, nchunk, gamma, beta, ts, tb, ld_t, sc, sh, nullptr, nullptr, Bn, Tn, C, G, eps, st);
      });
      release(part);
      gp.scale_in = sc; gp.shift_in = sh;
    }
    Planes out = alloc_planes((size_t)B * Tp * C);
    gp.out_hi = out.hi; gp.out_lo = out.lo;
    if (raw_out) { *raw_out = alloc_planes((size_t)B * Tp * C); gp.raw_hi = raw_out->hi; gp.raw_lo = raw_out->lo; }
    cur_kind = "gn_apply";
    {
      char buf[96];
      snprintf(buf, sizeof(buf), "T=%d C=%d%s%s", Tn, C, fast16 ? " blockstats" : (fast ? " slab" : " table"), raw_out ? " +raw" : "");
      cur_desc = buf;
    }
    PersistOp po{};
    if (fast) {
      po.type = POP_GN; po.gn = gp;
      po.gn_chunks = std::max(1, std::min(Tp / 4, 64 / G));
      po.gn_rpb = (Tp + po.gn_chunks - 1) / po.gn_chunks;
      po.gn_chunks = (Tp + po.gn_rpb - 1) / po.gn_rpb;
    }
    {
      dv_unet* uu = u;
      emit(ops, [gp, uu](hipStream_t st) {
        if (!uu->io.tp_base || !gp.tscale) return launch_gn_apply(gp, st);
        GnApplyParams g2 = gp;               // this evaluation's rows of the batched time_emb_proj table
        g2.tscale = uu->io.tp_base + (gp.tscale - uu->temb.tproj_arena);
        g2.tshift = uu->io.tp_base + (gp.tshift - uu->temb.tproj_arena);
        return launch_gn_apply(g2, st);
      }, (fast && !fast16) ? &po : nullptr);
    }
    if (sc) { release(sc); release(sh); }
    return out;
  }

  Planes ln_apply(std::vector<OpFn>& ops, const float* x, int M, int C) {
    Planes out = alloc_planes((size_t)M * C);
    cur_kind = "ln_apply";
    emit(ops, [=](hipStream_t st) { return launch_ln_apply(x, out.hi, out.lo, M, C, 1e-5f, st); });
    return out;
  }

  Planes split(std::vector<OpFn>& ops, const float* x, size_t n) {
    Planes out = alloc_planes(n);
    cur_kind = "split";
    PersistOp po{};
    po.type = POP_SPLIT; po.sp_in = x; po.sp_hi = out.hi; po.sp_lo = out.lo; po.n4_per_item = (int64_t)(n / 4 / (size_t)B);
    emit(ops, [=](hipStream_t st) { return launch_split(x, out.hi, out.lo, (int64_t)n, st); }, (n % (4 * (size_t)B) == 0) ? &po : nullptr);
    return out;
  }

  // GEMM over a level of Tn frames per utterance: row pitch pitch(Tn) (M = B * pitch), Tn of them exist
  GemmParams gp_base(int Tn, int M, int N) {
    GemmParams g{};
    g.nseg = 1; g.T_out = g.T_in = pitch(Tn); g.Tv_out = g.Tv_in = g.T_virt = Tn; g.stride = 1; g.up_mode = UP_NONE;
    g.M = M; g.N = N; g.epi = EPI_STORE; g.ldo = N; g.ldres = N;
    return g;
  }
  // ... over a plain row space of `rows` rows per item (conditioning schedule, prompt encoder: nothing is padded there)
  GemmParams gp_rows(int rows, int M, int N) {
    GemmParams g = gp_base(rows, M, N);
    g.T_out = g.T_in = rows;
    return g;
  }

  // temb projection table offsets
  std::map<std::string, int> tproj_off;
  int tproj_total = 0;
  float* tproj = nullptr;    // [B, tproj_total]

  // ResnetBlock2D (reference resnet.py:591-641): apply(norm1) -> conv1 -> apply(norm2, temb) -> conv2 (+1x1
  // shortcut as a second K-segment | + identity residual)
  Act resnet(std::vector<OpFn>& ops, const std::string& p, Act x0, Act x1, int cout, bool want_planes = false,
             bool stat16_out = false) {
    const int cin = x0.C + x1.C, Tn = x0.T, Tp = x0.Tp, M = B * Tp;
    const float eps = u->cfg.norm_eps;
    const bool shortcut = has(p + "conv_shortcut.weight");
    const PackedW* w1 = pack(p + "conv1", cout, 3 * cin, {{p + "conv1.weight", 1, cin, 3, cin, 0, 0, "", 0}},
                             {{p + "conv1.bias", "", "", "", cout, 0, 0, 0}});
    if (!w1) return Act{};
    Act h{};
    h.p = alloc((size_t)M * cout); h.C = cout; h.T = Tn; h.Tp = Tp; alloc_stat(h);
    Planes raw, raw1;             // the raw operand of the folded 1x1 shortcut: one tensor, or [x0 | x1] as two
    bool raw_made = false;
    const int toff = tproj_off[p];
    Planes n2x;                   // norm2(h) planes written by conv1's own epilogue (gnx_setup), if it can
    bool gnx1 = false;
    {
      GemmParams g = gp_base(Tn, M, cout);
      g.out = h.p; stat_out(g, h);
      {
        Planes n1, n1b;
        if (x0.n_hi && x0.n_pre == p + "norm1" && (!shortcut || x0.pl_hi) && (x1.C == 0 || (x0.sn_hi && (!shortcut || x0.sr_hi)))) {
          // normalised by its producer: x0 alone, or [x0 | x1] as two tensors of planes
          n1.hi = x0.n_hi; n1.lo = x0.n_lo;
          if (shortcut) { raw.hi = x0.pl_hi; raw.lo = x0.pl_lo; }
          if (x1.C) {
            n1b.hi = x0.sn_hi; n1b.lo = x0.sn_lo;
            if (shortcut) { raw1.hi = x0.sr_hi; raw1.lo = x0.sr_lo; }
          }
          g.seg[0] = x1.C ? seg(n1, x0.C, n1b, x1.C, 3, 1) : seg(n1, cin, Planes{}, 0, 3, 1);
        } else {
          if (x0.n_hi) { Planes stale; stale.hi = x0.n_hi; stale.lo = x0.n_lo; release(stale); }
          if (x0.sn_hi) { Planes stale; stale.hi = x0.sn_hi; stale.lo = x0.sn_lo; release(stale); }
          if (x0.sr_hi) { Planes stale; stale.hi = x0.sr_hi; stale.lo = x0.sr_lo; release(stale); }
          n1 = norm_apply(ops, x0, x1, p + "norm1", eps, nullptr, nullptr, 0, true, shortcut ? &raw : nullptr);
          g.seg[0] = seg(n1, cin, Planes{}, 0, 3, 1);
        }
        raw_made = shortcut;
        gnx1 = gnx_setup(g, p + "norm2", eps, tproj + toff, tproj + toff + cout, tproj_total, true, &n2x);
        if (gnx1 && !u->keep_intermediates) g.out = nullptr;   // h is only ever read through norm2
        gemm(ops, g, w1, 3 * cin);
        release(n1);
        if (n1b.hi) release(n1b);
      }
    }
    probe(p + "conv1", h.p, Tn, cout);

    const int K2 = 3 * cout + (shortcut ? cin : 0);
    std::vector<Piece> pcs = {{p + "conv2.weight", 1, cout, 3, cout, 0, 0, "", 0}};
    std::vector<BiasPiece> bps = {{p + "conv2.bias", shortcut ? p + "conv_shortcut.bias" : "", "", "", cout, 0, 0, 0}};
    if (shortcut) pcs.push_back({p + "conv_shortcut.weight", 1, cin, 1, cin, 3 * cout, 0, "", 0});
    const PackedW* w2 = pack(p + "conv2", cout, K2, pcs, bps);
    if (!w2) return Act{};
    Act out{};
    out.p = alloc((size_t)M * cout); out.C = cout; out.T = Tn; out.Tp = Tp; alloc_stat(out, stat16_out);   // (its consumer is a chain)
    {
      GemmParams g = gp_base(Tn, M, cout);
      if (!shortcut) { g.epi = EPI_RESIDUAL; g.res = x0.p; g.ldres = cout; }
      g.out = out.p; stat_out(g, out);
      if (want_planes) {   // the next op is a resampling conv: hand it split planes instead of a k_split launch
        Planes pl = alloc_planes((size_t)M * cout);
        out.pl_hi = pl.hi; out.pl_lo = pl.lo; g.out_hi = pl.hi; g.out_lo = pl.lo;
      }
      const bool offered = next_norm.set;
      {
        Planes n2 = gnx1 ? n2x : norm_apply(ops, h, Act{}, p + "norm2", eps, tproj + toff, tproj + toff + cout, tproj_total, true, nullptr);
        g.seg[0] = seg(n2, cout, Planes{}, 0, 3, 1);
        g.nseg = 1;
        if (shortcut) { g.seg[1] = raw1.hi ? seg(raw, x0.C, raw1, x1.C, 1, 0) : seg(raw, cin, Planes{}, 0, 1, 0); g.nseg = 2; }
        if (offered) offer_next(g, out);
        gemm(ops, g, w2, K2);
        release(n2);
      }
    }
    release_act(h);
    if (raw_made) release(raw);
    if (raw1.hi) release(raw1);
    probe(p.substr(0, p.size() - 1), out.p, Tn, cout);
    return out;
  }

  // cross-attention K/V of every transformer block (filled by the cond schedule)
  std::map<std::string, float*> cross_kv;
  float* mask_bias = nullptr;    // [B, L]
  // the same as MFMA fragments for the blocks whose cross attention runs inside the chain kernel (kernels_chain.hip)
  struct XFrag { bf16_t* kf_hi = nullptr; bf16_t* kf_lo = nullptr; bf16_t* vf_hi = nullptr; bf16_t* vf_lo = nullptr; };
  std::map<std::string, XFrag> cross_frag;
  float* xbias = nullptr;        // [B, nT * 32] key bias in the log2 domain
  bool xa_on = [] { const char* e = getenv("DVITS_CHAIN_XATTN"); return !(e && e[0] == '0'); }();
  bool xa_ok(int Tn, int C) const {
    const int H = u->cfg.num_heads;
    return xa_on && chain_ok(Tn, C) && H == 8 && C % H == 0 && (C / H == 16 || C / H == 32);
  }

  Planes attention(std::vector<OpFn>& ops, const float* q, int ldq, const float* k, const float* v, int ldkv, const float* bias,
                   int Tq, int Tk, int C, int Tk_pitch = 0) {
    Planes o = alloc_planes((size_t)B * Tq * C);
    AttnParams a{};
    a.Tk_pitch = Tk_pitch;
    a.q = q; a.k = k; a.v = v; a.bias = bias; a.o = nullptr; a.o_hi = o.hi; a.o_lo = o.lo;
    a.ldq = ldq; a.ldk = ldkv; a.ldv = ldkv; a.ldo = C;
    a.B = B; a.H = u->cfg.num_heads; a.Tq = Tq; a.Tk = Tk; a.d = C / u->cfg.num_heads;
    a.scale = 1.0f / sqrtf((float)a.d);
    a.nsplit = prec == DV_PREC_BF16X3 ? 3 : 1;
    cur_kind = "attn"; cur_flops = 4.0 * B * a.H * (double)Tq * Tk * a.d;
    {
      char buf[96];
      snprintf(buf, sizeof(buf), "Tq=%d Tk=%d d=%d H=%d", Tq, Tk, a.d, a.H);
      cur_desc = buf;
    }
    if (!dry) u->flops += cur_flops;
    PersistOp po{};
    po.type = POP_ATTN; po.cfg = (a.d + 15) / 16 * 16; po.a = a;
    emit(ops, [a](hipStream_t st) { return launch_attention(a, st); }, (prec == DV_PREC_BF16X3 && a.d % 4 == 0 && a.d <= 64) ? &po : nullptr);
    return o;
  }

  // Attention over K / V fragments (k_attention_frag): self attention (fragments written by the q|k|v chain) or cross
  // attention (the prompt's hoisted fragments)
  Planes attention_frag(std::vector<OpFn>& ops, AttnFragParams a, int Tq, int Tk, int C) {
    Planes o = alloc_planes((size_t)B * Tq * C);
    a.o = nullptr; a.o_hi = o.hi; a.o_lo = o.lo; a.ldo = C;
    a.B = B; a.H = u->cfg.num_heads; a.Tq = Tq; a.Tk = Tk; a.d = C / u->cfg.num_heads;
    a.scale = 1.0f / sqrtf((float)a.d);
    a.nsplit = prec == DV_PREC_BF16X3 ? 3 : 1;
    cur_kind = "attn"; cur_flops = 4.0 * B * a.H * (double)Tq * Tk * a.d;
    {
      char buf[96];
      snprintf(buf, sizeof(buf), "Tq=%d Tk=%d d=%d H=%d frag", Tq, Tk, a.d, a.H);
      cur_desc = buf;
    }
    if (!dry) u->flops += cur_flops;
    emit(ops, [a](hipStream_t st) { return launch_attention_frag(a, st); });
    return o;
  }
  bool attn_frag_on = [] { const char* e = getenv("DVITS_ATTN_FRAG"); return !(e && e[0] == '0'); }();
  // cross attention of block `p` over the prompt's K / V as hoisted MFMA fragments (cross_frag, key bias xbias)
  Planes cross_attention_frag(std::vector<OpFn>& ops, const std::string& p, const float* q2, int Tn, int C) {
    const XFrag& xf = cross_frag[p];
    const int H = u->cfg.num_heads, dh = C / H, nT = (L + 31) / 32, KSq = dh / 16, NBv = (dh + 31) / 32;
    AttnFragParams a{};
    a.q = q2; a.ldq = C;
    a.kf_hi = xf.kf_hi; a.kf_lo = xf.kf_lo; a.vf_hi = xf.vf_hi; a.vf_lo = xf.vf_lo;
    a.k_b = H * nT * KSq; a.k_h = nT * KSq; a.k_t = KSq;
    a.v_b = H * nT * 2 * NBv; a.v_h = nT * 2 * NBv; a.v_t = 2 * NBv; a.v_kb = NBv; a.v_nb = 1; a.self_layout = 0;
    a.bias = xbias; a.bias_ld = nT * 32;
    return attention_frag(ops, a, Tn, L, C);
  }

  // Transformer2DModel + BasicTransformerBlock (reference transformer_1d.py:191-326, attention.py:130-203)
  Act transformer(std::vector<OpFn>& ops, const std::string& p, Act x, bool want_planes = false) {
    const int C = x.C, Tn = x.T, Tp = x.Tp, M = B * Tp, D = u->cfg.cross_attention_dim;
    const std::string tb = p + "transformer_blocks.0.";
    const PackedW* w_in = pack(p + "proj_in", C, C, {{p + "proj_in.weight", 1, C, 1, C, 0, 0, "", 0}},
                               {{p + "proj_in.bias", "", "", "", C, 0, 0, 0}});
    const PackedW* w_qkv = pack(tb + "qkv1", 3 * C, C,
                                {{tb + "attn1.to_q.weight", 0, C, 1, C, 0, 0, tb + "norm1.weight", 0},
                                 {tb + "attn1.to_k.weight", 0, C, 1, C, 0, C, tb + "norm1.weight", 0},
                                 {tb + "attn1.to_v.weight", 0, C, 1, C, 0, 2 * C, tb + "norm1.weight", 0}},
                                {{"", "", tb + "attn1.to_q.weight", tb + "norm1.bias", C, C, 0, 0},
                                 {"", "", tb + "attn1.to_k.weight", tb + "norm1.bias", C, C, C, 0},
                                 {"", "", tb + "attn1.to_v.weight", tb + "norm1.bias", C, C, 2 * C, 0}});
    const PackedW* w_o1 = pack(tb + "out1", C, C, {{tb + "attn1.to_out.0.weight", 0, C, 1, C, 0, 0, "", 0}},
                               {{tb + "attn1.to_out.0.bias", "", "", "", C, 0, 0, 0}});
    const PackedW* w_q2 = pack(tb + "q2", C, C, {{tb + "attn2.to_q.weight", 0, C, 1, C, 0, 0, tb + "norm2.weight", 0}},
                               {{"", "", tb + "attn2.to_q.weight", tb + "norm2.bias", C, C, 0, 0}});
    const PackedW* w_o2 = pack(tb + "out2", C, C, {{tb + "attn2.to_out.0.weight", 0, C, 1, C, 0, 0, "", 0}},
                               {{tb + "attn2.to_out.0.bias", "", "", "", C, 0, 0, 0}});
    const PackedW* w_gg = pack(tb + "geglu", 8 * C, C, {{tb + "ff.net.0.proj.weight", 0, C, 1, C, 0, 0, tb + "norm3.weight", 1}},
                               {{tb + "ff.net.0.proj.bias", "", tb + "ff.net.0.proj.weight", tb + "norm3.bias", 8 * C, C, 0, 1}});
    const bool merged_ffproj = merge_ff && fuse_ln && !u->keep_intermediates;
    const PackedW* w_ff = merged_ffproj ? w_gg : pack(tb + "ffout", C, 4 * C, {{tb + "ff.net.2.weight", 0, 4 * C, 1, 4 * C, 0, 0, "", 0}},
                               {{tb + "ff.net.2.bias", "", "", "", C, 0, 0, 0}});
    const PackedW* w_out = merged_ffproj ? w_gg : pack(p + "proj_out", C, C, {{p + "proj_out.weight", 1, C, 1, C, 0, 0, "", 0}},
                                {{p + "proj_out.bias", "", "", "", C, 0, 0, 0}});
    if (!w_in || !w_qkv || !w_o1 || !w_q2 || !w_o2 || !w_gg || !w_ff || !w_out) return Act{};
    (void)D;

    // Two ways to feed the three LayerNorms: (default) the producer GEMM also emits raw planes + per-row partial
    // statistics and the consumer finishes the normalisation in its epilogue, 48 launches fewer per forward;
    // (DVITS_FUSE_LN=0) k_ln_apply writes normalised split planes for the consumer GEMM.  With the transposed
    // accumulator the row partials are in-lane sums and the fused schedule measures 4.27 vs 4.34 ms per forward.
    const int nblk = C / 32;
    struct LnIn { Planes pl; float* stat = nullptr; };
    auto ln_produce = [&](GemmParams& g) {   // producer side (fused mode only)
      LnIn in;
      if (!fuse_ln) return in;
      in.pl = alloc_planes((size_t)M * C);
      in.stat = alloc((size_t)M * nblk * 2);
      g.out_hi = in.pl.hi; g.out_lo = in.pl.lo; g.rowstat_out = in.stat;
      return in;
    };
    auto ln_consume = [&](GemmParams& g, LnIn& in, const float* x32, const PackedW* w) {
      if (!fuse_ln) in.pl = ln_apply(ops, x32, M, C);
      else { g.ln_stat = in.stat; g.ln_nblk = nblk; g.ln_u = w->u; g.ln_eps = 1e-5f; }
      g.seg[0] = seg(in.pl, C, Planes{}, 0, 1, 0);
    };
    auto ln_release = [&](LnIn& in) { release(in.pl); if (in.stat) release(in.stat); };

    const bool chained = chain_ok(Tp, C) && x.stat16;
    float* h3 = nullptr;
    LnIn l3;
    if (chained && !(frag(w_in) && frag(w_qkv) && frag(w_o1) && frag(w_q2))) return Act{};
    if (chained) {
      // chain 1: GN(eps 1e-6) -> proj_in -> LN1 -> to_q | to_k | to_v   (3 launches -> 1)
      float* h = alloc((size_t)M * C);
      // K and V leave the chain as MFMA fragments of 32-key tiles (k_attention_frag multiplies them as they are):
      // DVITS_ATTN_FRAG=0 restores fp32 q | k | v and the converting attention kernel
      const int dh = C / u->cfg.num_heads;
      const bool sa_frag = attn_frag_on && !arena.exact && C % u->cfg.num_heads == 0 && dh % 16 == 0 && dh <= 64;
      float* qkv = alloc((size_t)M * (sa_frag ? 1 : 3) * C);
      Planes kf, vf;
      if (sa_frag) { kf = alloc_planes((size_t)M * C); vf = alloc_planes((size_t)M * C); }
      {
        ChainParams cp{};
        cp.M = M; cp.C = C; cp.T = Tp; cp.Tv = Tn; cp.amode = 1;
        cp.x = x.p; cp.stat16 = x.stat16; cp.gamma = W(p + "norm.weight"); cp.beta = W(p + "norm.bias"); cp.gn_eps = 1e-6f;
        cp.groups = u->cfg.norm_num_groups;
        cp.w1_hi = w_in->fhi; cp.w1_lo = w_in->flo; cp.Kp1 = w_in->Kp; cp.b1 = w_in->bias; cp.res = nullptr; cp.out1 = h;
        cp.w2_hi = w_qkv->fhi; cp.w2_lo = w_qkv->flo; cp.Kp2 = w_qkv->Kp; cp.b2 = w_qkv->bias; cp.u2 = w_qkv->u;
        cp.passes = 3; cp.out2 = qkv; cp.ldo2 = sa_frag ? C : 3 * C; cp.ln_eps = 1e-5f;
        if (sa_frag) { cp.sa_kf_hi = kf.hi; cp.sa_kf_lo = kf.lo; cp.sa_vf_hi = vf.hi; cp.sa_vf_lo = vf.lo; }
        {   // few row blocks: share the three passes out (DVITS_CHAIN_SPLIT=0: one workgroup per row block)
          const char* es = getenv("DVITS_CHAIN_SPLIT");
          const bool off = es && es[0] == '0';
          const int rbs = M / 32, cus = n_cu > 0 ? n_cu : 256;
          cp.nsplit = off ? 1 : (rbs * 3 <= cus ? 3 : (rbs * 2 <= cus ? 2 : 1));
        }
        // 64-row blocks with the output columns of both contractions split over C / 64 workgroups per row block (k_qkv_split,
        // kernels_qkv.hip: an all-gather of h inside the launch - planned like k_ff_split's hand-over; DVITS_QKV_SPLIT=0: the
        // 32-row chain above) wherever that gives the launch at least qkv_split_min_wg workgroups
        const int n_qflags = qkv_split_flags(cp);
        if (qkv_split_on && C >= qkv_split_min_c && sa_frag && gnx_on && u->exclusive && !arena.exact && !autotune_on() && n_cu > 0 && qkv_split_supported(cp, prec) &&
            n_qflags <= n_cu && n_qflags >= qkv_split_min_wg && gnx_used + (size_t)n_qflags <= dv_unet::GNX_POOL) {   // (one round of workgroups: at B = 16 - 384 / 512 of them - the 32-row chain is the faster one, 4.09 vs 4.19 ms per forward)
          cp.nsplit = 1;
          cp.qs_flags = dry ? reinterpret_cast<unsigned long long*>(0x1000) : u->gnx_pool + gnx_used;
          cp.qs_status = dry ? reinterpret_cast<unsigned*>(0x1000) : u->gnx_status;
          cp.qs_spin = gnx_spin;
          gnx_used += ((size_t)n_qflags + 1) & ~(size_t)1;
          if (!dry) { u->gnx_words = gnx_used; u->gnx_ops++; }
          cur_kind = "chain";
          cur_flops = 2.0 * (double)cp.M * cp.C * cp.C * 4.0;
          char buf[96];
          snprintf(buf, sizeof(buf), "norm+proj_in+LN+q|Kfrag|Vfrag (%d wg / 64 rows) M=%d C=%d N2=%d", C / 64, cp.M, cp.C, 3 * cp.C);
          cur_desc = buf;
          if (!dry) u->flops += cur_flops;
          const int pr = prec;
          emit(ops, [cp, pr](hipStream_t st) { return launch_qkv_split(cp, pr, st); });
        } else
        chain(ops, cp, sa_frag ? "norm+proj_in+LN+q|Kfrag|Vfrag" : "norm+proj_in+LN+qkv");
      }
      probe(p + "proj_in", h, Tn, C);
      Planes ao;
      if (sa_frag) {
        AttnFragParams a{};
        a.q = qkv; a.ldq = C;
        a.kf_hi = kf.hi; a.kf_lo = kf.lo; a.vf_hi = vf.hi; a.vf_lo = vf.lo;
        const int nT = Tp / 32;
        a.k_b = nT * (C / 16); a.k_h = dh / 16; a.k_t = C / 16;
        a.v_b = nT * (C / 32) * 2; a.v_h = 0; a.v_t = (C / 32) * 2; a.v_kb = 1; a.v_nb = 2; a.self_layout = 1;
        a.bias = nullptr; a.bias_ld = 0;
        ao = attention_frag(ops, a, Tp, Tn, C);       // (queries: every row of the padded space; keys: the frames that exist)
        release(kf); release(vf);
      } else ao = attention(ops, qkv, 3 * C, qkv + C, qkv + 2 * C, 3 * C, nullptr, Tp, Tn, C, Tp);
      release(qkv);
      if (xa_ok(Tp, C) && cross_frag.count(p)) {
        // chain 2 with the cross attention inside: to_out + residual -> LN2 -> to_q -> cross attention (wave = head, K / V
        // of the prompt as hoisted MFMA fragments) -> to_out + residual -> LN3 partials   (4 launches -> 1)
        if (!frag(w_o2)) return Act{};
        float* h2 = alloc((size_t)M * C);
        h3 = alloc((size_t)M * C);
        l3 = LnIn{};
        l3.pl = alloc_planes((size_t)M * C);
        l3.stat = alloc((size_t)M * nblk * 2);
        {
          const XFrag& xf = cross_frag[p];
          ChainParams cp{};
          cp.M = M; cp.C = C; cp.T = Tp; cp.Tv = Tn; cp.amode = 0;
          cp.a_hi = ao.hi; cp.a_lo = ao.lo;
          cp.w1_hi = w_o1->fhi; cp.w1_lo = w_o1->flo; cp.Kp1 = w_o1->Kp; cp.b1 = w_o1->bias; cp.res = h; cp.out1 = h2;
          cp.w2_hi = w_q2->fhi; cp.w2_lo = w_q2->flo; cp.Kp2 = w_q2->Kp; cp.b2 = w_q2->bias; cp.u2 = w_q2->u;
          cp.passes = 1; cp.out2 = nullptr; cp.ldo2 = C; cp.ln_eps = 1e-5f;
          cp.xa_kf_hi = xf.kf_hi; cp.xa_kf_lo = xf.kf_lo; cp.xa_vf_hi = xf.vf_hi; cp.xa_vf_lo = xf.vf_lo; cp.xa_bias = xbias;
          cp.xa_nT = (L + 31) / 32; cp.xa_d = C / u->cfg.num_heads;
          cp.xa_qscale = (1.0f / sqrtf((float)cp.xa_d)) * 1.44269504088896340736f;
          cp.w3_hi = w_o2->fhi; cp.w3_lo = w_o2->flo; cp.b3 = w_o2->bias;
          cp.out3 = h3; cp.out3_hi = l3.pl.hi; cp.out3_lo = l3.pl.lo; cp.rowstat3 = l3.stat;
          {   // C = 256 with few row blocks (M = 4096 at the bench shape: 128 for 256 CUs): two workgroups per row block, heads
              // 0-3 / 4-7, two waves per head, stage 3 handed over like a fused split-K pair (DVITS_CHAIN_XSPLIT=0: one)
            const char* es = getenv("DVITS_CHAIN_XSPLIT");
            const int rbs = M / 32, cus = n_cu > 0 ? n_cu : 256;
            if (!(es && es[0] == '0') && C == 256 && rbs * 2 <= cus && rbs <= 4096 && (dry || u->sk_tickets)) {
              cp.nsplit = 2;
              cp.xs_buf = alloc((size_t)rbs * 2 * C * 32);
              cp.xs_ticket = dry ? reinterpret_cast<unsigned*>(0x1000) : u->sk_tickets;
            }
          }
          // The same block as ONE column-split launch of 64-row blocks (k_qkv_split MODE 2, kernels_qkv.hip: the slice's heads attend
          // inside, h2 and the attention output handed over through the XCD's L2): one round of workgroups, whole multiples of 8
          // row blocks (DVITS_QKV_XA=0: the row-block chain above)
          ChainParams cq = cp;
          cq.nsplit = 1; cq.xs_buf = nullptr; cq.xs_ticket = nullptr;
          Planes oxa;
          const int n_wg = (M / 64) * (C / 64);
          bool split_xa = qkv_split_on && qkv_xa_on && C >= qkv_xa_min_c && gnx_on && u->exclusive && !arena.exact && !autotune_on() && n_cu > 0 &&
                          Tp % 64 == 0 && (M / 64) % 8 == 0 && n_wg <= n_cu && n_wg >= qkv_split_min_wg && u->cfg.num_heads == 8 &&
                          gnx_used + (size_t)2 * n_wg <= dv_unet::GNX_POOL;
          if (split_xa) {
            oxa = alloc_planes((size_t)M * C);
            cq.qs_o_hi = oxa.hi; cq.qs_o_lo = oxa.lo;
            split_xa = qkv_split_supported(cq, prec);
            if (!split_xa) release(oxa);
          }
          if (split_xa) {
            const int n_qflags = qkv_split_flags(cq);
            cq.qs_flags = dry ? reinterpret_cast<unsigned long long*>(0x1000) : u->gnx_pool + gnx_used;
            cq.qs_status = dry ? reinterpret_cast<unsigned*>(0x1000) : u->gnx_status;
            cq.qs_spin = gnx_spin;
            gnx_used += ((size_t)n_qflags + 1) & ~(size_t)1;
            if (!dry) { u->gnx_words = gnx_used; u->gnx_ops++; }
            if (cp.xs_buf) { release(cp.xs_buf); cp.xs_buf = nullptr; }
            cur_kind = "chain";
            cur_flops = 2.0 * (double)M * C * C * 2.0 + 4.0 * B * u->cfg.num_heads * (double)Tp * L * cq.xa_d + 2.0 * (double)M * C * C;
            char buf[96];
            snprintf(buf, sizeof(buf), "to_out+res+LN+to_q+xattn+to_out+res (%d wg / 64 rows) M=%d C=%d", C / 64, M, C);
            cur_desc = buf;
            if (!dry) u->flops += cur_flops;
            const int pr = prec;
            emit(ops, [cq, pr](hipStream_t st) { return launch_qkv_split(cq, pr, st); });
            release(oxa);
          } else {
          chain(ops, cp, cp.nsplit == 2 ? "to_out+res+LN+to_q+xattn(2 wg)+to_out+res" : "to_out+res+LN+to_q+xattn+to_out+res",
                4.0 * B * u->cfg.num_heads * (double)Tp * L * cp.xa_d + 2.0 * (double)M * C * C);
          if (cp.xs_buf) release(cp.xs_buf);
          }
        }
        release(ao); release(h);
        probe(tb + "attn1", h2, Tn, C);
        release(h2);
        probe(tb + "attn2", h3, Tn, C);
      } else {
      // chain 2: to_out + residual -> LN2 -> to_q of the cross attention   (2 launches -> 1)
      float* h2 = alloc((size_t)M * C);
      float* q2 = alloc((size_t)M * C);
      {
        ChainParams cp{};
        cp.M = M; cp.C = C; cp.T = Tp; cp.Tv = Tn; cp.amode = 0;
        cp.a_hi = ao.hi; cp.a_lo = ao.lo;
        cp.w1_hi = w_o1->fhi; cp.w1_lo = w_o1->flo; cp.Kp1 = w_o1->Kp; cp.b1 = w_o1->bias; cp.res = h; cp.out1 = h2;
        cp.w2_hi = w_q2->fhi; cp.w2_lo = w_q2->flo; cp.Kp2 = w_q2->Kp; cp.b2 = w_q2->bias; cp.u2 = w_q2->u;
        cp.passes = 1; cp.out2 = q2; cp.ldo2 = C; cp.ln_eps = 1e-5f;
        {   // few row blocks: the 128-column groups of to_q are shared out over C / 128 workgroups per row block
          const char* es = getenv("DVITS_CHAIN_SPLIT");
          const bool off = es && es[0] == '0';
          const int rbs = M / 32, cus = n_cu > 0 ? n_cu : 256, nsg = C / 128;
          cp.nsplit = (!off && nsg >= 2 && rbs * nsg <= cus) ? nsg : 1;
        }
        // (the same 64-row column-split launch as the block head - k_qkv_split, MODE 1: one round of workgroups)
        const int n_qflags = qkv_split_flags(cp);
        if (qkv_split_on && C >= qkv_split_min_c && gnx_on && u->exclusive && !arena.exact && !autotune_on() && n_cu > 0 && qkv_split_supported(cp, prec) &&
            n_qflags <= n_cu && n_qflags >= qkv_split_min_wg && gnx_used + (size_t)n_qflags <= dv_unet::GNX_POOL) {
          cp.nsplit = 1;
          cp.qs_flags = dry ? reinterpret_cast<unsigned long long*>(0x1000) : u->gnx_pool + gnx_used;
          cp.qs_status = dry ? reinterpret_cast<unsigned*>(0x1000) : u->gnx_status;
          cp.qs_spin = gnx_spin;
          gnx_used += ((size_t)n_qflags + 1) & ~(size_t)1;
          if (!dry) { u->gnx_words = gnx_used; u->gnx_ops++; }
          cur_kind = "chain";
          cur_flops = 2.0 * (double)cp.M * cp.C * cp.C * 2.0;
          char buf[96];
          snprintf(buf, sizeof(buf), "to_out+res+LN+to_q (%d wg / 64 rows) M=%d C=%d N2=%d", C / 64, cp.M, cp.C, cp.C);
          cur_desc = buf;
          if (!dry) u->flops += cur_flops;
          const int pr = prec;
          emit(ops, [cp, pr](hipStream_t st) { return launch_qkv_split(cp, pr, st); });
        } else
        chain(ops, cp, "to_out+res+LN+to_q");
      }
      release(ao); release(h);
      probe(tb + "attn1", h2, Tn, C);
      float* kv = cross_kv[p];
      if (attn_frag_on && !arena.exact && cross_frag.count(p)) ao = cross_attention_frag(ops, p, q2, Tp, C);
      else ao = attention(ops, q2, C, kv, kv + C, 2 * C, mask_bias, Tp, L, C);
      release(q2);
      h3 = alloc((size_t)M * C);
      {
        GemmParams g = gp_base(Tn, M, C); g.seg[0] = seg(ao, C, Planes{}, 0, 1, 0);
        g.epi = EPI_RESIDUAL; g.res = h2; g.out = h3; l3 = ln_produce(g);
        gemm(ops, g, w_o2, C);
      }
      release(ao); release(h2);
      probe(tb + "attn2", h3, Tn, C);
      }
    } else {
    // GN(eps 1e-6) -> 1x1 proj_in
    float* h = alloc((size_t)M * C);
    LnIn l1;
    {
      GemmParams g = gp_base(Tn, M, C);
      g.out = h; l1 = ln_produce(g);
      {
        Planes gn;
        if (x.n_hi && x.n_pre == p + "norm") { gn.hi = x.n_hi; gn.lo = x.n_lo; }    // normalised by its producer's epilogue
        else gn = norm_apply(ops, x, Act{}, p + "norm", 1e-6f, nullptr, nullptr, 0, false, nullptr);
        g.seg[0] = seg(gn, C, Planes{}, 0, 1, 0);
        gemm(ops, g, w_in, C);
        release(gn);
      }
    }
    probe(p + "proj_in", h, Tn, C);

    // self-attention
    float* qkv = alloc((size_t)M * 3 * C);
    {
      GemmParams g = gp_base(Tn, M, 3 * C);
      g.out = qkv; ln_consume(g, l1, h, w_qkv); gemm(ops, g, w_qkv, C);
    }
    ln_release(l1);
    Planes ao = attention(ops, qkv, 3 * C, qkv + C, qkv + 2 * C, 3 * C, nullptr, Tp, Tn, C, Tp);   // (keys: the frames that exist)
    release(qkv);
    float* h2 = alloc((size_t)M * C);
    LnIn l2;
    {
      GemmParams g = gp_base(Tn, M, C); g.seg[0] = seg(ao, C, Planes{}, 0, 1, 0);
      g.epi = EPI_RESIDUAL; g.res = h; g.out = h2; l2 = ln_produce(g);
      gemm(ops, g, w_o1, C);
    }
    release(ao); release(h);
    probe(tb + "attn1", h2, Tn, C);

    // cross-attention (K/V hoisted: projected once per set_cond)
    float* q2 = alloc((size_t)M * C);
    {
      GemmParams g = gp_base(Tn, M, C);
      g.out = q2; ln_consume(g, l2, h2, w_q2); gemm(ops, g, w_q2, C);
    }
    ln_release(l2);
    float* kv = cross_kv[p];
    if (attn_frag_on && !arena.exact && cross_frag.count(p)) ao = cross_attention_frag(ops, p, q2, Tp, C);
    else ao = attention(ops, q2, C, kv, kv + C, 2 * C, mask_bias, Tp, L, C);
    release(q2);
    h3 = alloc((size_t)M * C);
    {
      GemmParams g = gp_base(Tn, M, C); g.seg[0] = seg(ao, C, Planes{}, 0, 1, 0);
      g.epi = EPI_RESIDUAL; g.res = h2; g.out = h3; l3 = ln_produce(g);
      gemm(ops, g, w_o2, C);
    }
    release(ao); release(h2);
    probe(tb + "attn2", h3, Tn, C);
    }

    // C = 128 blocks: LN3 -> GEGLU -> merged ff.net.2 + proj_out + residual as ONE row-block launch (k_chain_ff; the
    // 4C-wide product never leaves LDS).  DVITS_CHAIN_FF=0 restores the two GEMMs.
    {
      const char* eff = getenv("DVITS_CHAIN_FF");
      const bool ff_off = eff && eff[0] == '0';
      ChainFFParams fp{};
      fp.M = M; fp.C = C; fp.T = Tp; fp.Tv = Tn;
      if (!ff_off && merged_ffproj && chain_on && !arena.exact && l3.stat && x.stat16 && chain_ff_supported(fp, prec)) {
        const std::string mw = tb + "__ffproj.weight", mb = tb + "__ffproj.bias";
        if (!dry && !u->packed.count(p + "ffproj")) {
          const float* Wo = W(p + "proj_out.weight"); const float* W2 = W(tb + "ff.net.2.weight");
          const float* bo = W(p + "proj_out.bias"); const float* b2 = W(tb + "ff.net.2.bias");
          float* dw = derived(mw, {C, 4 * C});
          float* db = derived(mb, {C});
          if (!Wo || !W2 || !bo || !b2 || !dw || !db) return Act{};
          (void)launch_matmul_f32(Wo, W2, dw, C, C, 4 * C, pack_stream);
          (void)launch_fold_bias(Wo, bo, b2, db, C, C, 0, 0, pack_stream);
        }
        const PackedW* w_m = pack(p + "ffproj", C, 5 * C, {{p + "proj_out.weight", 1, C, 1, C, 0, 0, "", 0}, {mw, 0, 4 * C, 1, 4 * C, C, 0, "", 0}},
                                  {{mb, "", "", "", C, 0, 0, 0}});
        if (!w_m || !frag(w_gg) || !frag(w_m)) return Act{};
        Act out{};
        out.p = alloc((size_t)M * C); out.C = C; out.T = Tn; out.Tp = Tp; alloc_stat(out, true);
        fp.a_hi = l3.pl.hi; fp.a_lo = l3.pl.lo; fp.rowstat = l3.stat; fp.ln_eps = 1e-5f;
        fp.wg_hi = w_gg->fhi; fp.wg_lo = w_gg->flo; fp.bg = w_gg->bias; fp.ug = w_gg->u;
        fp.wm_hi = w_m->fhi; fp.wm_lo = w_m->flo; fp.bm = w_m->bias;
        fp.res = x.p; fp.out = out.p; fp.stats16 = out.stat16;
        if (want_planes) {
          Planes pl = alloc_planes((size_t)M * C);
          out.pl_hi = pl.hi; out.pl_lo = pl.lo; fp.out_hi = pl.hi; fp.out_lo = pl.lo;
        }
        offer_next(fp, out);
        cur_kind = "chain";
        cur_flops = 2.0 * (double)M * C * (8.0 * C + 5.0 * C);
        {
          char buf[96];
          snprintf(buf, sizeof(buf), "LN+GEGLU+ffproj+res%s M=%d C=%d", fp.gnx.xchg ? "+gnx" : "", M, C);
          cur_desc = buf;
        }
        if (!dry) u->flops += cur_flops;
        const int pr = prec;
        emit(ops, [fp, pr](hipStream_t st) { return launch_chain_ff(fp, pr, st); });
        ln_release(l3);
        release(h3);
        probe(p.substr(0, p.size() - 1), out.p, Tn, C);
        return out;
      }
    }
    // C = 256 / 384 blocks: the same three steps as ONE launch of 64-row blocks whose product columns are split over 4 / 8
    // workgroups (k_ff_split, kernels_ffsplit.hip: partial ffproj sums handed over inside the launch - every workgroup resident,
    // like the in-launch GroupNorm; DVITS_FF_SPLIT=0 restores the two GEMMs)
    {
      FFSplitParams fp{};
      fp.M = M; fp.C = C; fp.T = Tp; fp.Tv = Tn; fp.nspl = C == 256 ? 4 : 8;
      const int ff_rows = fp.rows = ff_split_rows(C, M, Tp, fp.nspl, n_cu);   // 64 rows per workgroup; 32 at C = 512, odd pitches, small inputs
      const size_t n_flags = (size_t)(M / ff_rows) * fp.nspl * 8;       // one word per wave (kernels_ffsplit.hip)
      if (ff_split_on && merged_ffproj && chain_on && gnx_on && u->exclusive && !arena.exact && !autotune_on() && l3.stat && x.stat16 &&
          n_cu > 0 && (C == 256 || C == 384 || C == 512) && ff_split_supported(fp, prec) && ((M / ff_rows) * fp.nspl <= n_cu || gemm_handover_rounds()) &&
          (M / ff_rows) * fp.nspl >= ff_split_min_wg &&
          gnx_used + n_flags <= dv_unet::GNX_POOL) {
        const std::string mw = tb + "__ffproj.weight", mb = tb + "__ffproj.bias";
        if (!dry && !u->packed.count(p + "ffproj")) {
          const float* Wo = W(p + "proj_out.weight"); const float* W2 = W(tb + "ff.net.2.weight");
          const float* bo = W(p + "proj_out.bias"); const float* b2 = W(tb + "ff.net.2.bias");
          float* dw = derived(mw, {C, 4 * C});
          float* db = derived(mb, {C});
          if (!Wo || !W2 || !bo || !b2 || !dw || !db) return Act{};
          (void)launch_matmul_f32(Wo, W2, dw, C, C, 4 * C, pack_stream);
          (void)launch_fold_bias(Wo, bo, b2, db, C, C, 0, 0, pack_stream);
        }
        const PackedW* w_m = pack(p + "ffproj", C, 5 * C, {{p + "proj_out.weight", 1, C, 1, C, 0, 0, "", 0}, {mw, 0, 4 * C, 1, 4 * C, C, 0, "", 0}},
                                  {{mb, "", "", "", C, 0, 0, 0}});
        if (!w_m || !frag(w_gg) || !frag(w_m)) return Act{};
        Act out{};
        out.p = alloc((size_t)M * C); out.C = C; out.T = Tn; out.Tp = Tp; alloc_stat(out, true);
        fp.a_hi = l3.pl.hi; fp.a_lo = l3.pl.lo; fp.rowstat = l3.stat; fp.ln_eps = 1e-5f;
        fp.wg_hi = w_gg->fhi; fp.wg_lo = w_gg->flo; fp.bg = w_gg->bias; fp.ug = w_gg->u;
        fp.wm_hi = w_m->fhi; fp.wm_lo = w_m->flo; fp.bm = w_m->bias;
        fp.res = x.p; fp.out = out.p; fp.stats16 = out.stat16;
        fp.flags = dry ? reinterpret_cast<unsigned long long*>(0x1000) : u->gnx_pool + gnx_used;
        fp.status = dry ? reinterpret_cast<unsigned*>(0x1000) : u->gnx_status;
        fp.spin_max = gnx_spin;
        gnx_used += n_flags;
        if (!dry) { u->gnx_words = gnx_used; u->gnx_ops++; }
        fp.xbuf = alloc(ff_split_xbuf_floats(M, C, fp.nspl, fp.rows));
        if (want_planes) {
          Planes pl = alloc_planes((size_t)M * C);
          out.pl_hi = pl.hi; out.pl_lo = pl.lo; fp.out_hi = pl.hi; fp.out_lo = pl.lo;
        }
        offer_next(fp, out);
        cur_kind = "chain";
        cur_flops = 2.0 * (double)M * C * (8.0 * C + 5.0 * C);
        {
          char buf[96];
          snprintf(buf, sizeof(buf), "LN+GEGLU+ffproj+res%s (%d wg / %d rows) M=%d C=%d", fp.gnx.xchg ? "+gnx" : "", fp.nspl, ff_rows, M, C);
          cur_desc = buf;
        }
        if (!dry) u->flops += cur_flops;
        const int pr = prec;
        emit(ops, [fp, pr](hipStream_t st) { return launch_ff_split(fp, pr, st); });
        release(fp.xbuf);
        ln_release(l3);
        release(h3);
        probe(p.substr(0, p.size() - 1), out.p, Tn, C);
        return out;
      }
    }
    // GEGLU feed-forward: the GEGLU product and the FF output only feed GEMMs -> split planes only
    Planes gg = alloc_planes((size_t)M * 4 * C);
    {
      GemmParams g = gp_base(Tn, M, 8 * C);
      g.epi = EPI_GEGLU; g.out_hi = gg.hi; g.out_lo = gg.lo; g.ldo = 4 * C; ln_consume(g, l3, h3, w_gg); gemm(ops, g, w_gg, C);
    }
    // ff.net.2 and proj_out are two k=1 contractions with only the residual add between them:
    //   proj_out(h3 + gg W2^T + b2) + x = h3 Wo^T + gg (Wo W2)^T + (Wo b2 + bo) + x
    // -> ONE contraction over two K-segments [raw planes of h3 | GEGLU product] (16 launches fewer per forward).
    // Needs the raw h3 planes of the fused-LayerNorm schedule; the per-layer probe build keeps the two-step form.
    if (merged_ffproj) {
      const std::string mw = tb + "__ffproj.weight", mb = tb + "__ffproj.bias";
      if (!dry && !u->packed.count(p + "ffproj")) {   // (kept across prepares with the packed weights)
        const float* Wo = W(p + "proj_out.weight"); const float* W2 = W(tb + "ff.net.2.weight");
        const float* bo = W(p + "proj_out.bias"); const float* b2 = W(tb + "ff.net.2.bias");
        float* dw = derived(mw, {C, 4 * C});
        float* db = derived(mb, {C});
        if (!Wo || !W2 || !bo || !b2 || !dw || !db) return Act{};
        (void)launch_matmul_f32(Wo, W2, dw, C, C, 4 * C, pack_stream);
        (void)launch_fold_bias(Wo, bo, b2, db, C, C, 0, 0, pack_stream);
      }
      const PackedW* w_m = pack(p + "ffproj", C, 5 * C, {{p + "proj_out.weight", 1, C, 1, C, 0, 0, "", 0}, {mw, 0, 4 * C, 1, 4 * C, C, 0, "", 0}},
                                {{mb, "", "", "", C, 0, 0, 0}});
      if (!w_m) return Act{};
      Act out{};
      out.p = alloc((size_t)M * C); out.C = C; out.T = Tn; out.Tp = Tp; alloc_stat(out);
      {
        GemmParams g = gp_base(Tn, M, C);
        g.seg[0] = seg(l3.pl, C, Planes{}, 0, 1, 0);
        g.seg[1] = seg(gg, 4 * C, Planes{}, 0, 1, 0); g.nseg = 2;
        g.epi = EPI_RESIDUAL; g.res = x.p; g.out = out.p; stat_out(g, out);
        if (want_planes) {
          Planes pl = alloc_planes((size_t)M * C);
          out.pl_hi = pl.hi; out.pl_lo = pl.lo; g.out_hi = pl.hi; g.out_lo = pl.lo;
        }
        offer_next(g, out);
        gemm(ops, g, w_m, 5 * C);
      }
      ln_release(l3);
      release(gg); release(h3);
      probe(p.substr(0, p.size() - 1), out.p, Tn, C);
      return out;
    }
    ln_release(l3);
    Planes h4 = alloc_planes((size_t)M * C);
    float* h4f = u->keep_intermediates ? alloc((size_t)M * C) : nullptr;
    {
      GemmParams g = gp_base(Tn, M, C); g.seg[0] = seg(gg, 4 * C, Planes{}, 0, 1, 0);
      g.epi = EPI_RESIDUAL; g.res = h3; g.out = h4f; g.out_hi = h4.hi; g.out_lo = h4.lo; gemm(ops, g, w_ff, 4 * C);
    }
    release(gg); release(h3);
    if (h4f) probe(tb + "ff", h4f, Tn, C);

    Act out{};
    out.p = alloc((size_t)M * C); out.C = C; out.T = Tn; out.Tp = Tp; alloc_stat(out);
    {
      GemmParams g = gp_base(Tn, M, C); g.seg[0] = seg(h4, C, Planes{}, 0, 1, 0);
      g.epi = EPI_RESIDUAL; g.res = x.p; g.out = out.p; stat_out(g, out);
      if (want_planes) {
        Planes pl = alloc_planes((size_t)M * C);
        out.pl_hi = pl.hi; out.pl_lo = pl.lo; g.out_hi = pl.hi; g.out_lo = pl.lo;
      }
      offer_next(g, out);
      gemm(ops, g, w_out, C);
    }
    release(h4);
    probe(p.substr(0, p.size() - 1), out.p, Tn, C);
    return out;
  }

  // Downsample2D / Upsample2D (reference resnet.py:138-223): stride-2 conv, or nearest upsample folded
  // into the conv's row gather
  Act resample(std::vector<OpFn>& ops, const std::string& p, Act x, bool down, int T_target) {
    const int C = x.C;
    const PackedW* w = pack(p + "conv", C, 3 * C, {{p + "conv.weight", 1, C, 3, C, 0, 0, "", 0}},
                            {{p + "conv.bias", "", "", "", C, 0, 0, 0}});
    if (!w) return Act{};
    Planes xs;
    if (x.pl_hi) { xs.hi = x.pl_hi; xs.lo = x.pl_lo; }
    else xs = split(ops, x.p, (size_t)B * x.Tp * C);
    GemmParams g = gp_base(x.T, 0, C);
    g.seg[0] = seg(xs, C, Planes{}, 0, 3, 1);
    g.T_in = x.Tp; g.Tv_in = x.T;
    int T_new;                                           // frames of the result that exist
    if (down) { g.T_virt = x.T; g.stride = 2; g.up_mode = UP_NONE; T_new = (x.T + 2 - 3) / 2 + 1; }
    else {
      g.stride = 1; T_new = g.T_virt = T_target;
      // (interpolate(size=...) to exactly twice the length IS the x2 form - floor(t * 0.5f) = t >> 1 - and the x2 form has the
      // resident-operand kernel: T = 300 upsamples 75 -> 150 -> 300 that way, only 38 -> 75 needs the general gather)
      if (u->force_up && T_target != 2 * x.T) { g.up_mode = UP_SIZE; g.up_scale = (float)x.T / (float)T_target; }
      else g.up_mode = UP_X2;
    }
    g.Tv_out = T_new; g.T_out = pitch(T_new);
    Act out{};
    out.p = alloc((size_t)B * g.T_out * C); out.C = C; out.T = T_new; out.Tp = g.T_out; alloc_stat(out);
    g.M = B * g.T_out; g.out = out.p; stat_out(g, out);
    offer_next(g, out);
    gemm(ops, g, w, 3 * C);
    release(xs);
    probe(p.substr(0, p.size() - 1), out.p, T_new, C);
    return out;
  }

  void release_act(const Act& a) { if (a.p) release(a.p); if (a.stat) release(a.stat); if (a.stat16) release(a.stat16); }

  // ---- whole network
  int build() {
    const dv_unet_cfg& c = u->cfg;
    const int n = c.n_levels, lpb = c.layers_per_block, E = c.block_out_channels[0] * 4, D = c.cross_attention_dim;
    const int C0 = c.block_out_channels[0];
    std::vector<OpFn>& S = u->step_ops;
    std::vector<OpFn>& K = u->cond_ops;
    dv_unet* uu = u;
    const int Bn = B, Ln = L, Tn = T;

    // ---------- enumerate resnets (for the batched time_emb_proj table) ----------
    std::vector<std::pair<std::string, int>> resnets;   // prefix, cout
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < lpb; ++j) resnets.push_back({"down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", c.block_out_channels[i]});
    resnets.push_back({"mid_block.resnets.0.", c.block_out_channels[n - 1]});
    resnets.push_back({"mid_block.resnets.1.", c.block_out_channels[n - 1]});
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < lpb + 1; ++j) resnets.push_back({"up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", c.block_out_channels[n - 1 - i]});
    tproj_total = 0;
    for (auto& r : resnets) { tproj_off[r.first] = tproj_total; tproj_total += 2 * r.second; }

    // concatenated time_emb_proj weights/biases (fp32, exact path)
    float* Wt = nullptr; float* bt = nullptr;
    if (!dry) {
      if (hipMalloc((void**)&Wt, (size_t)tproj_total * E * 4) != hipSuccess || hipMalloc((void**)&bt, (size_t)tproj_total * 4) != hipSuccess)
        return dv_fail(DV_ERR_HIP, "hipMalloc(time_emb_proj table) failed");
      u->owned.push_back(Wt); u->owned.push_back(bt);
      for (auto& r : resnets) {
        const RawW* w = raw(r.first + "time_emb_proj.weight");
        const RawW* b = raw(r.first + "time_emb_proj.bias");
        if (!w || !b) break;
        if (w->numel != (size_t)2 * r.second * E) { err = "shape mismatch: " + r.first + "time_emb_proj.weight"; break; }
        (void)hipMemcpyAsync(Wt + (size_t)tproj_off[r.first] * E, w->p, w->numel * 4, hipMemcpyDeviceToDevice, pack_stream);
        (void)hipMemcpyAsync(bt + tproj_off[r.first], b->p, b->numel * 4, hipMemcpyDeviceToDevice, pack_stream);
      }
      if (!u->zero_page) {
        void* z = nullptr;
        if (hipMalloc(&z, DV_ZERO_PAGE_BYTES) != hipSuccess) return dv_fail(DV_ERR_HIP, "hipMalloc(zero page) failed");
        (void)hipMemsetAsync(z, 0, DV_ZERO_PAGE_BYTES, pack_stream);
        u->owned.push_back(z);
        u->zero_page = reinterpret_cast<bf16_t*>(z);
      }
      if (!u->gnx_pool) {
        void* z = nullptr;
        if (hipMalloc(&z, dv_unet::GNX_POOL * 8) != hipSuccess) return dv_fail(DV_ERR_HIP, "hipMalloc(GroupNorm exchange words) failed");
        (void)hipMemsetAsync(z, 0xff, dv_unet::GNX_POOL * 8, pack_stream);
        u->owned.push_back(z);
        u->gnx_pool = reinterpret_cast<unsigned long long*>(z);
      }
      if (!u->gnx_status) {
        if (hipHostMalloc((void**)&u->gnx_status, 64, hipHostMallocMapped) != hipSuccess) return dv_fail(DV_ERR_HIP, "hipHostMalloc(status word) failed");
        *u->gnx_status = 0;
      }
      if (!u->sk_tickets) {      // (fused split-K pairs and the two-workgroup cross-attention chain share them: both leave zeros behind)
        void* z = nullptr;
        if (hipMalloc(&z, 4096 * sizeof(unsigned)) != hipSuccess) return dv_fail(DV_ERR_HIP, "hipMalloc(split-K tickets) failed");
        (void)hipMemsetAsync(z, 0, 4096 * sizeof(unsigned), pack_stream);
        u->owned.push_back(z);
        u->sk_tickets = reinterpret_cast<unsigned*>(z);
      }
    }

    // ---------- persistent buffers (live across calls: allocated first, never released) ----------
    float* aug_emb = alloc((size_t)B * E);
    mask_bias = alloc((size_t)B * L);
    std::vector<std::pair<std::string, int>> xformers;   // prefix, C
    for (int i = 0; i < n - 1; ++i)
      for (int j = 0; j < lpb; ++j) xformers.push_back({"down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".", c.block_out_channels[i]});
    xformers.push_back({"mid_block.attentions.0.", c.block_out_channels[n - 1]});
    for (int i = 1; i < n; ++i)
      for (int j = 0; j < lpb + 1; ++j) xformers.push_back({"up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j) + ".", c.block_out_channels[n - 1 - i]});
    for (auto& x : xformers) cross_kv[x.first] = alloc((size_t)B * L * 2 * x.second);
    {
      const int H = c.num_heads, nT = (L + 31) / 32;
      xbias = alloc((size_t)B * nT * 32);
      int lvl_T = T;
      std::map<int, int> T_of_C;   // frames at the level of each channel count (the transformer of C runs at T_l)
      for (int i = 0; i < n; ++i) { T_of_C[c.block_out_channels[i]] = pitch(lvl_T); lvl_T = (lvl_T + 2 - 3) / 2 + 1; }
      for (auto& x : xformers) {
        const int C = x.second;
        // fragments feed the in-chain cross attention (xa_ok) or k_attention_frag (any block with 16-channel head groups)
        const bool frag_attn = attn_frag_on && !arena.exact && prec == DV_PREC_BF16X3 && C % H == 0 && (C / H) % 16 == 0 && C / H <= 64;
        if (!xa_ok(T_of_C[C], C) && !frag_attn) continue;
        const int d = C / H, KSq = d / 16, NBv = (d + 31) / 32;
        const size_t kel = (size_t)B * H * nT * KSq * 64 * 8, vel = (size_t)B * H * nT * 2 * NBv * 64 * 8;
        XFrag xf;
        xf.kf_hi = reinterpret_cast<bf16_t*>(alloc((kel + 1) / 2)); xf.kf_lo = reinterpret_cast<bf16_t*>(alloc((kel + 1) / 2));
        xf.vf_hi = reinterpret_cast<bf16_t*>(alloc((vel + 1) / 2)); xf.vf_lo = reinterpret_cast<bf16_t*>(alloc((vel + 1) / 2));
        cross_frag[x.first] = xf;
      }
    }

    // ---------- cond schedule ----------
    {
      // mask bias (or zeros)
      float* mb = mask_bias;
      emit(K, [=](hipStream_t st) {
        return uu->io.mask ? launch_copy_f32(uu->io.mask, mb, (int64_t)Bn * Ln, st) : launch_fill_f32(mb, 0.f, (int64_t)Bn * Ln, st);
      });
      // pooled-text embedding: LN -> [mean+pos | x] -> k,v proj -> 1-query attention -> proj -> LN
      const std::string a = "add_embedding.";
      float* seq = alloc((size_t)B * (L + 1) * D);
      const float* n1w = W(a + "norm1.weight"); const float* n1b = W(a + "norm1.bias");
      emit(K, [=](hipStream_t st) { return launch_layernorm_rows_into(uu->io.enc, n1w, n1b, seq, Bn * Ln, D, 1e-5f, Ln, Ln + 1, 1, st); });
      const float* pos = W(a + "pool.positional_embedding");
      emit(K, [=](hipStream_t st) { return launch_mean_token(seq, pos, Bn, Ln, D, st); });
      const PackedW* wkv = pack(a + "pool.kv", 2 * D, D,
                                {{a + "pool.k_proj.weight", 0, D, 1, D, 0, 0, "", 0}, {a + "pool.v_proj.weight", 0, D, 1, D, 0, D, "", 0}},
                                {{a + "pool.k_proj.bias", "", "", "", D, 0, 0, 0}, {a + "pool.v_proj.bias", "", "", "", D, 0, D, 0}});
      if (!wkv) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
      Planes seqs = split(K, seq, (size_t)B * (L + 1) * D);
      float* kvp = alloc((size_t)B * (L + 1) * 2 * D);
      {
        GemmParams g = gp_rows(L + 1, B * (L + 1), 2 * D);
        g.seg[0] = seg(seqs, D, Planes{}, 0, 1, 0);
        g.out = kvp;
        gemm(K, g, wkv, D);
      }
      float* qp = alloc((size_t)B * D);
      const float* wq = W(a + "pool.q_proj.weight"); const float* bq = W(a + "pool.q_proj.bias");
      emit(K, [=](hipStream_t st) { return launch_small_linear(seq, (Ln + 1) * D, wq, bq, nullptr, qp, D, Bn, D, D, 0, 0, st); });
      float* pooled = alloc((size_t)B * D);
      const int heads = c.add_embed_heads;
      emit(K, [=](hipStream_t st) { return launch_pool_attn(qp, kvp, pooled, Bn, Ln + 1, D, heads, st); });
      float* pe = alloc((size_t)B * E);
      const float* wp = W(a + "proj.weight"); const float* bp = W(a + "proj.bias");
      emit(K, [=](hipStream_t st) { return launch_small_linear(pooled, D, wp, bp, nullptr, pe, E, Bn, D, E, 0, 0, st); });
      const float* n2w = W(a + "norm2.weight"); const float* n2b = W(a + "norm2.bias");
      emit(K, [=](hipStream_t st) { return launch_layernorm_rows(pe, n2w, n2b, aug_emb, Bn, E, 1e-5f, st); });
      release(seq); release(seqs); release(kvp); release(qp); release(pooled); release(pe);
      // cross-attention K/V projections of every transformer block: split the encoder states once
      Planes encs = alloc_planes((size_t)B * L * D);
      emit(K, [=](hipStream_t st) { return launch_split(uu->io.enc, encs.hi, encs.lo, (int64_t)Bn * Ln * D, st); });
      for (auto& x : xformers) {
        const std::string tb = x.first + "transformer_blocks.0.";
        const int C = x.second;
        const PackedW* w = pack(tb + "kv2", 2 * C, D,
                                {{tb + "attn2.to_k.weight", 0, D, 1, D, 0, 0, "", 0}, {tb + "attn2.to_v.weight", 0, D, 1, D, 0, C, "", 0}}, {});
        if (!w) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
        GemmParams g = gp_rows(L, B * L, 2 * C);
        g.seg[0] = seg(encs, D, Planes{}, 0, 1, 0);
        g.out = cross_kv[x.first];
        gemm(K, g, w, D);
        if (cross_frag.count(x.first)) {   // the same K / V as MFMA fragments for the in-chain cross attention
          const XFrag xf = cross_frag[x.first];
          const float* kvp2 = cross_kv[x.first];
          const int H = c.num_heads;
          emit(K, [=](hipStream_t st) { return launch_kv_frag(kvp2, xf.kf_hi, xf.kf_lo, xf.vf_hi, xf.vf_lo, Bn, Ln, C, H, st); });
        }
      }
      {
        float* xb = xbias; const float* mbp = mask_bias; const int nT = (L + 31) / 32;
        emit(K, [=](hipStream_t st) { return launch_xbias(mbp, xb, Bn, Ln, nT, st); });
      }
      release(encs);
    }

    // ---------- step schedule ----------
    u->flops = 0;   // count the per-step schedule only (the cond schedule is step-invariant)
    tproj = alloc((size_t)B * tproj_total);
    float* emb = alloc((size_t)B * E);
    {
      float* tsin = alloc((size_t)B * C0);
      float* h1 = alloc((size_t)B * E);
      const int temb_begin = (int)S.size();
      emit(S, [=](hipStream_t st) { return launch_timestep_sincos(uu->io.t, tsin, Bn, C0, st); });
      const float* w1 = W("time_embedding.linear_1.weight"); const float* b1 = W("time_embedding.linear_1.bias");
      const float* w2 = W("time_embedding.linear_2.weight"); const float* b2 = W("time_embedding.linear_2.bias");
      if (Bn <= 16) {   // lane-per-column kernel on transposed weights (built once): a quarter of the row-per-wave kernel's time
        float* w1T = nullptr; float* w2T = nullptr;
        if (!dry) {
          if (hipMalloc((void**)&w1T, (size_t)E * C0 * 4) != hipSuccess || hipMalloc((void**)&w2T, (size_t)E * E * 4) != hipSuccess)
            return dv_fail(DV_ERR_HIP, "hipMalloc(time_embedding^T) failed");
          u->owned.push_back(w1T); u->owned.push_back(w2T);
          (void)launch_transpose_f32(w1, w1T, E, C0, pack_stream);
          (void)launch_transpose_f32(w2, w2T, E, E, pack_stream);
        }
        emit(S, [=](hipStream_t st) { return launch_small_linear_t(tsin, C0, w1T, b1, nullptr, h1, E, Bn, C0, E, 0, 1, st); });
        emit(S, [=](hipStream_t st) { return launch_small_linear_t(h1, E, w2T, b2, aug_emb, emb, E, Bn, E, E, 0, 0, st); });
        if (!dry) { u->temb.w1T = w1T; u->temb.w2T = w2T; u->temb.b1 = b1; u->temb.b2 = b2; u->temb.aug_emb = aug_emb; }
      } else {
        emit(S, [=](hipStream_t st) { return launch_small_linear(tsin, C0, w1, b1, nullptr, h1, E, Bn, C0, E, 0, 1, st); });
        emit(S, [=](hipStream_t st) { return launch_small_linear(h1, E, w2, b2, aug_emb, emb, E, Bn, E, E, 0, 0, st); });
      }
      float* tp = tproj; const int tt = tproj_total;
      if (Bn <= 16) {   // lane-per-column kernel on the transposed table (built once, below the schedule's critical path)
        float* WtT = nullptr;
        if (!dry) {
          if (hipMalloc((void**)&WtT, (size_t)tt * E * 4) != hipSuccess) return dv_fail(DV_ERR_HIP, "hipMalloc(time_emb_proj^T) failed");
          u->owned.push_back(WtT);
          (void)launch_transpose_f32(Wt, WtT, tt, E, pack_stream);
        }
        emit(S, [=](hipStream_t st) { return launch_small_linear_t(emb, E, WtT, bt, nullptr, tp, tt, Bn, E, tt, 1, 0, st); });
        if (!dry && !arena.exact) {   // (B <= 16: the batched chain runs the same kernels on row chunks - identical bits per row)
          u->temb.WtT = WtT; u->temb.bt = bt; u->temb.C0 = C0; u->temb.E = E; u->temb.tt = tt; u->temb.tproj_arena = tp;
          u->temb.begin = temb_begin; u->temb.end = (int)S.size(); u->temb.ok = true;
        }
      } else {
        emit(S, [=](hipStream_t st) { return launch_small_linear(emb, E, Wt, bt, nullptr, tp, tt, Bn, E, tt, 1, 0, st); });
      }
      release(tsin); release(h1);
      probe("emb", emb, 1, E);
    }
    // (input channels padded to whole 64-deep k-tiles: 208 -> 256 instead of 224 - conv_in then runs the 64-deep tile menu,
    // 12 k-tiles instead of 21 half-depth ones: 20.7 k -> ~13 k cycles of k-loop at the bench shape, profiles/r05_gemm_per_launch_trace.txt)
    const int cin = c.in_channels, cpad = cin > 64 ? rup(cin, 64) : rup(cin, 32);
    const int Tp0 = pitch(T);                            // row pitch of the first level (padding rows: zeros)
    Planes xin = alloc_planes((size_t)B * Tp0 * cpad);
    emit(S, [=](hipStream_t st) {
      // (it precedes every GEMM of the forward: it also resets the exchange words of the in-epilogue GroupNorms, GnxParams)
      return launch_pack_input(uu->io.x, uu->io.cx, uu->io.cond, cin - uu->io.cx, xin.hi, xin.lo, cpad, Bn, Tn, st, uu->gnx_pool, uu->gnx_words, Tp0);
    });
    const PackedW* wci = pack("conv_in", C0, 3 * cpad, {{"conv_in.weight", 1, cin, 3, cpad, 0, 0, "", 0}},
                              {{"conv_in.bias", "", "", "", C0, 0, 0, 0}});
    if (!wci) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
    Act h{};
    h.p = alloc((size_t)B * Tp0 * C0); h.C = C0; h.T = T; h.Tp = Tp0; alloc_stat(h);
    {
      GemmParams g = gp_base(T, B * Tp0, C0);
      g.seg[0] = seg(xin, cpad, Planes{}, 0, 3, 1);
      g.out = h.p; stat_out(g, h);
      announce_norm("down_blocks.0.resnets.0.norm1", has("down_blocks.0.resnets.0.conv_shortcut.weight"));
      offer_next(g, h);
      gemm(S, g, wci, 3 * cin);
    }
    release(xin);
    probe("conv_in", h.p, T, C0);

    // a tensor kept for the up path's concat: without the planes that belong to its immediate consumer
    auto skip_of = [](Act a) { a.pl_hi = a.pl_lo = a.n_hi = a.n_lo = a.sn_hi = a.sn_lo = a.sr_hi = a.sr_lo = nullptr; a.n_pre.clear(); return a; };
    // the resnet block that consumes a down-path tensor alone: its norm1 can be finished by the tensor's producer
    auto announce_resnet = [&](const std::string& rp) { announce_norm(rp + "norm1", has(rp + "conv_shortcut.weight")); };
    std::vector<Act> skips{skip_of(h)};
    // the up-path resnet block (i, j) normalises [h | the skip on top of the stack]: announced to h's producer
    auto announce_up = [&](int i, int j) {
      const std::string rp = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".";
      if (!skips.empty()) announce_norm(rp + "norm1", has(rp + "conv_shortcut.weight"), skips.back());
    };
    for (int i = 0; i < n; ++i) {
      const std::string bp = "down_blocks." + std::to_string(i) + ".";
      const bool attn = i < n - 1;
      for (int j = 0; j < lpb; ++j) {
        const bool feeds_resampler = i < n - 1 && j == lpb - 1;     // its output is the downsampler's input
        // consumer of this (resnet [+ transformer]) pair's output: the block's next resnet, the downsampler, or the mid block
        const std::string next_rp = j + 1 < lpb ? bp + "resnets." + std::to_string(j + 1) + "." : (i == n - 1 ? std::string("mid_block.resnets.0.") : std::string());
        if (!attn && !next_rp.empty()) announce_resnet(next_rp);
        if (attn) announce_transformer_norm(bp + "attentions." + std::to_string(j) + ".", h.Tp, c.block_out_channels[i]);
        Act r = resnet(S, bp + "resnets." + std::to_string(j) + ".", h, Act{}, c.block_out_channels[i], feeds_resampler && !attn,
                       attn && chain_ok(h.Tp, c.block_out_channels[i]));
        next_norm.set = false;
        if (!r.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
        if (attn) {
          if (!next_rp.empty()) announce_resnet(next_rp);
          Act a = transformer(S, bp + "attentions." + std::to_string(j) + ".", r, feeds_resampler);
          next_norm.set = false;
          if (!a.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
          release_act(r);
          r = a;
        }
        h = r;
        skips.push_back(skip_of(h));
      }
      if (i < n - 1) {
        announce_resnet("down_blocks." + std::to_string(i + 1) + ".resnets.0.");
        h = resample(S, bp + "downsamplers.0.", h, true, 0);
        next_norm.set = false;
        if (!h.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
        skips.push_back(skip_of(h));
      }
    }
    {
      announce_transformer_norm("mid_block.attentions.0.", h.Tp, h.C);
      Act r0 = resnet(S, "mid_block.resnets.0.", h, Act{}, h.C, false, chain_ok(h.Tp, h.C));
      next_norm.set = false;
      if (!r0.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
      announce_resnet("mid_block.resnets.1.");
      Act a = transformer(S, "mid_block.attentions.0.", r0);
      next_norm.set = false;
      if (!a.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
      release_act(r0);
      announce_up(0, 0);
      h = resnet(S, "mid_block.resnets.1.", a, Act{}, a.C);
      next_norm.set = false;
      if (!h.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
      release_act(a);
    }
    for (int i = 0; i < n; ++i) {
      const std::string bp = "up_blocks." + std::to_string(i) + ".";
      const bool attn = i > 0, last = i == n - 1;
      const int cout = c.block_out_channels[n - 1 - i];
      for (int j = 0; j < lpb + 1; ++j) {
        Act sk = skips.back();
        skips.pop_back();
        const bool feeds_resampler = !last && j == lpb;              // its output is the upsampler's input
        const bool final_op = last && j == lpb;                      // its output goes to conv_norm_out alone
        if (final_op && !attn) announce_norm("conv_norm_out", false);
        if (j < lpb && !attn) announce_up(i, j + 1);
        if (attn) announce_transformer_norm(bp + "attentions." + std::to_string(j) + ".", h.Tp, cout);
        Act r = resnet(S, bp + "resnets." + std::to_string(j) + ".", h, sk, cout, feeds_resampler && !attn, attn && chain_ok(h.Tp, cout));
        next_norm.set = false;
        if (!r.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
        release_act(h);
        release_act(sk);
        if (attn) {
          if (final_op) announce_norm("conv_norm_out", false);
          if (j < lpb) announce_up(i, j + 1);
          Act a = transformer(S, bp + "attentions." + std::to_string(j) + ".", r, feeds_resampler);
          next_norm.set = false;
          if (!a.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
          release_act(r);
          r = a;
        }
        h = r;
      }
      if (!last) {
        announce_up(i + 1, 0);
        Act up = resample(S, bp + "upsamplers.0.", h, false, skips.back().T);
        next_norm.set = false;
        if (!up.p) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
        release_act(h);
        h = up;
      }
    }
    // conv_norm_out -> SiLU -> conv_out, written channels-first straight into y
    {
      const int co = c.out_channels;
      const PackedW* wo = pack("conv_out", co, 3 * C0, {{"conv_out.weight", 1, C0, 3, C0, 0, 0, "", 0}},
                               {{"conv_out.bias", "", "", "", co, 0, 0, 0}});
      if (!wo) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
      GemmParams g = gp_base(T, B * Tp0, co);
      Planes nf;
      if (h.n_hi && h.n_pre == "conv_norm_out") {   // normalised by its producer's epilogue
        nf.hi = h.n_hi; nf.lo = h.n_lo;
        g.seg[0] = seg(nf, C0, Planes{}, 0, 3, 1);
      } else {
        nf = norm_apply(S, h, Act{}, "conv_norm_out", c.norm_eps, nullptr, nullptr, 0, true, nullptr);
        g.seg[0] = seg(nf, C0, Planes{}, 0, 3, 1);
      }
      g.epi = EPI_STORE_NCT; g.ldo = co;
      g.w_hi = wo->hi; g.w_lo = wo->lo; g.Kp = wo->Kp; g.N_pad = wo->N_pad; g.bias = wo->bias; g.B = B;
      g.zero_page = u->zero_page;
      const int pr = prec;
      cur_kind = "gemm"; cur_flops = 2.0 * g.M * (double)co * 3 * C0;
      cur_desc = "conv_out (NCT store)";
      if (!dry) u->flops += cur_flops;
      emit(S, [g, pr, uu](hipStream_t st) { GemmParams gg = g; gg.out = uu->io.y; return launch_gemm(gg, pr, st); });
      if (nf.hi) release(nf);
      release_act(h);
    }
    if (!err.empty()) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
    return DV_OK;
  }

  // engine-made weight (a product of two state-dict tensors), registered under `name` like a state-dict entry and
  // recomputed at every prepare: returns the device buffer to fill, or null in the dry pass / on failure
  float* derived(const std::string& name, const std::vector<int64_t>& shape) {
    if (dry) return nullptr;
    size_t n = 1;
    for (auto v : shape) n *= (size_t)v;
    RawW& r = u->w[name];
    if (r.numel != n) {
      if (r.p) (void)hipFree(r.p);
      r.p = nullptr;
      if (hipMalloc((void**)&r.p, n * 4) != hipSuccess) { err = "hipMalloc(derived weight) failed"; return nullptr; }
    }
    r.shape = shape; r.numel = n;
    return r.p;
  }

  // engine-made constant vector registered like a weight (per-source-channel scale for pack())
  void const_vec(const std::string& name, int n, float value) {
    if (dry || u->w.count(name)) return;
    RawW r;
    if (hipMalloc((void**)&r.p, (size_t)n * 4) != hipSuccess) { err = "hipMalloc(const) failed"; return; }
    r.shape = {n}; r.numel = (size_t)n;
    (void)launch_fill_f32(r.p, value, n, pack_stream);
    u->w[name] = r;
  }

  // ---- PromptEncoder (reference model3.py:382-433): ConvLayer `pre` (k=1) -> n x EncSALayer
  // (operations.py:784-821) -> ConvLayer `out_proj` (k=1) -> LayerNorm, the keep mask re-applied after every
  // sub-layer.  Channels-last [B*L, C] throughout; the result [B, L, C_out] is exactly the encoder_hidden_states
  // layout the denoiser takes (model3.py:912 transposes the reference's [B, C, L] back).
  int build_penc() {
    const dv_penc_cfg& c = u->pcfg;
    const int H = c.hidden_channels, Cin = c.in_channels, Cout = c.out_channels, Ln = T, M = B * Ln, Bn = B;
    const int cpad = rup(Cin, 32), KS9 = c.ffn_kernel;
    std::vector<OpFn>& S = u->step_ops;
    dv_unet* uu = u;
    if (!dry && !u->zero_page) {
      void* z = nullptr;
      if (hipMalloc(&z, DV_ZERO_PAGE_BYTES) != hipSuccess) return dv_fail(DV_ERR_HIP, "hipMalloc(zero page) failed");
      (void)hipMemsetAsync(z, 0, DV_ZERO_PAGE_BYTES, pack_stream);
      u->owned.push_back(z);
      u->zero_page = reinterpret_cast<bf16_t*>(z);
    }
    const int nblk = H / 32;
    struct LnIn { Planes pl; float* stat = nullptr; };
    auto ln_produce = [&](GemmParams& g, float* x32) {   // producer of a k=1 LayerNorm consumer
      LnIn in;
      if (!fuse_ln) { (void)x32; return in; }
      in.pl = alloc_planes((size_t)M * H);
      in.stat = alloc((size_t)M * nblk * 2);
      g.out_hi = in.pl.hi; g.out_lo = in.pl.lo; g.rowstat_out = in.stat;
      return in;
    };
    auto ln_consume = [&](GemmParams& g, LnIn& in, const float* x32, const PackedW* w) {
      if (!fuse_ln) in.pl = ln_apply(S, x32, M, H);
      else { g.ln_stat = in.stat; g.ln_nblk = nblk; g.ln_u = w->u; g.ln_eps = 1e-5f; }
      g.seg[0] = seg(in.pl, H, Planes{}, 0, 1, 0);
    };
    auto ln_release = [&](LnIn& in) { release(in.pl); if (in.stat) release(in.stat); };

    // keep mask of the call -> arena (the GEMM epilogues read it), key bias, masked-LayerNorm'd input planes
    float* keep = alloc((size_t)M);
    float* kbias = alloc((size_t)M);
    emit(S, [=](hipStream_t st) { return launch_copy_f32(uu->io.mask, keep, (int64_t)Bn * Ln, st); });
    Planes p0 = alloc_planes((size_t)M * cpad);
    {
      const float* g0 = W("pre.layer_norm.weight"); const float* b0 = W("pre.layer_norm.bias");
      emit(S, [=](hipStream_t st) {
        return launch_prompt_pre(uu->io.x, keep, g0, b0, p0.hi, p0.lo, kbias, Bn, Cin, Ln, cpad, 1e-5f, st);
      });
    }
    const PackedW* w_pre = pack("pre", H, cpad, {{"pre.conv.weight", 0, Cin, 1, cpad, 0, 0, "", 0}},
                                {{"pre.conv.bias", "", "", "", H, 0, 0, 0}});
    if (!w_pre) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
    float* x = alloc((size_t)M * H);
    LnIn lin;
    {
      GemmParams g = gp_rows(Ln, M, H); g.seg[0] = seg(p0, cpad, Planes{}, 0, 1, 0);
      g.out = x; g.rowmask = keep; lin = ln_produce(g, x); gemm(S, g, w_pre, Cin);
    }
    release(p0);
    probe("pre", x, Ln, H);

    const_vec("__const.ffn_scale", 4 * H, 1.0f / sqrtf((float)KS9));
    for (int i = 0; i < c.n_layers; ++i) {
      const std::string p = "layers." + std::to_string(i) + ".op.";
      const PackedW* w_qkv = pack(p + "qkv", 3 * H, H, {{p + "self_attn.in_proj_weight", 0, H, 1, H, 0, 0, p + "layer_norm1.weight", 0}},
                                  {{"", "", p + "self_attn.in_proj_weight", p + "layer_norm1.bias", 3 * H, H, 0, 0}});
      const PackedW* w_o = pack(p + "out", H, H, {{p + "self_attn.out_proj.weight", 0, H, 1, H, 0, 0, "", 0}}, {});
      std::vector<Piece> f1;
      for (int j = 0; j < KS9; ++j) f1.push_back({p + "ffn.ffn_1." + std::to_string(j) + ".weight", 0, H, 1, H, H * j, 0, "", 0});
      const PackedW* w_f1 = pack(p + "ffn1", 4 * H, KS9 * H, f1, {{p + "ffn.ffn_1.0.bias", "", "", "", 4 * H, 0, 0, 0}});
      // relu(s * a) = s * relu(a), s = k^-1/2 > 0: the scale rides on ffn_2's input channels
      const PackedW* w_f2 = pack(p + "ffn2", H, 4 * H, {{p + "ffn.ffn_2.weight", 0, 4 * H, 1, 4 * H, 0, 0, "__const.ffn_scale", 0}},
                                 {{p + "ffn.ffn_2.bias", "", "", "", H, 0, 0, 0}});
      if (!w_qkv || !w_o || !w_f1 || !w_f2) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());

      // self-attention: LayerNorm1 finished in the qkv GEMM, padded keys masked by the key bias
      float* qkv = alloc((size_t)M * 3 * H);
      {
        GemmParams g = gp_rows(Ln, M, 3 * H);
        g.out = qkv; ln_consume(g, lin, x, w_qkv); gemm(S, g, w_qkv, H);
      }
      ln_release(lin);
      Planes ao = attention(S, qkv, 3 * H, qkv + H, qkv + 2 * H, 3 * H, kbias, Ln, Ln, H);
      release(qkv);
      float* x2 = alloc((size_t)M * H);
      {
        GemmParams g = gp_rows(Ln, M, H); g.seg[0] = seg(ao, H, Planes{}, 0, 1, 0);
        g.epi = EPI_RESIDUAL; g.res = x; g.out = x2; g.rowmask = keep; gemm(S, g, w_o, H);
      }
      release(ao); release(x);

      // feed-forward: LayerNorm2 WITH affine (the k=9 zero padding pads the normalised tensor), then the nine
      // shifted Linears as one contraction over two K-segments of the same planes: tap 0 at offset 0 (the
      // reference multiplies the unpadded x there, operations.py:678) and taps 1..8 at offsets -3..+4
      Planes n2 = alloc_planes((size_t)M * H);
      {
        const float* g2 = W(p + "layer_norm2.weight"); const float* b2 = W(p + "layer_norm2.bias");
        const float* xin = x2;
        cur_kind = "ln_apply";
        emit(S, [=](hipStream_t st) { return launch_ln_affine(xin, g2, b2, nullptr, nullptr, n2.hi, n2.lo, Bn * Ln, H, 1e-5f, st); });
      }
      Planes hh = alloc_planes((size_t)M * 4 * H);
      {
        GemmParams g = gp_rows(Ln, M, 4 * H);
        g.seg[0] = seg(n2, H, Planes{}, 0, 1, 0);
        if (KS9 > 1) { g.seg[1] = seg(n2, H, Planes{}, 0, KS9 - 1, (KS9 - 1) / 2 - 1); g.nseg = 2; }
        g.relu = 1; g.out_hi = hh.hi; g.out_lo = hh.lo;
        gemm(S, g, w_f1, KS9 * H);
      }
      release(n2);
      float* x3 = alloc((size_t)M * H);
      {
        GemmParams g = gp_rows(Ln, M, H); g.seg[0] = seg(hh, 4 * H, Planes{}, 0, 1, 0);
        g.epi = EPI_RESIDUAL; g.res = x2; g.out = x3; g.rowmask = keep; lin = ln_produce(g, x3); gemm(S, g, w_f2, 4 * H);
      }
      release(hh); release(x2);
      x = x3;
      probe("layer" + std::to_string(i), x, Ln, H);
    }

    // out_proj ConvLayer (no masked_fill: model3.py:427), then the last LayerNorm with affine, both re-masked
    const PackedW* w_op = pack("out_proj", Cout, H, {{"out_proj.conv.weight", 0, H, 1, H, 0, 0, "out_proj.layer_norm.weight", 0}},
                               {{"out_proj.conv.bias", "", "out_proj.conv.weight", "out_proj.layer_norm.bias", Cout, H, 0, 0}});
    if (!w_op) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
    float* z = alloc((size_t)M * Cout);
    {
      GemmParams g = gp_rows(Ln, M, Cout);
      g.out = z; g.rowmask = keep; ln_consume(g, lin, x, w_op); gemm(S, g, w_op, H);
    }
    ln_release(lin); release(x);
    if (has("layer_norm.weight")) {
      const float* gl = W("layer_norm.weight"); const float* bl = W("layer_norm.bias");
      cur_kind = "ln_apply";
      emit(S, [=](hipStream_t st) { return launch_ln_affine(z, gl, bl, keep, uu->io.y, nullptr, nullptr, Bn * Ln, Cout, 1e-5f, st); });
    } else {
      emit(S, [=](hipStream_t st) { return launch_copy_f32(z, uu->io.y, (int64_t)Bn * Ln * Cout, st); });
    }
    release(z);
    if (!err.empty()) return dv_fail(DV_ERR_MISSING_WEIGHT, "%s", err.c_str());
    return DV_OK;
  }
};

extern "C" int dv_unet_prepare(dv_unet* u, int32_t B, int32_t T, int32_t L, int32_t precision, int32_t force_upsample_size) {
  if (!u) return dv_fail(DV_ERR_INVALID, "dv_unet_prepare: null handle");
  if (B <= 0 || T <= 0 || L <= 0) return dv_fail(DV_ERR_INVALID, "dv_unet_prepare: B, T, L must be positive");
  if (precision != DV_PREC_BF16X3 && precision != DV_PREC_BF16) return dv_fail(DV_ERR_INVALID, "unknown precision %d", precision);
  {
    // the coarsest level must keep at least one frame
    int t = T;
    for (int i = 0; i < u->cfg.n_levels - 1; ++i) t = (t + 2 - 3) / 2 + 1;
    if (t < 1) return dv_fail(DV_ERR_INVALID, "T=%d too short for %d levels", T, u->cfg.n_levels);
  }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(gemm_init());
  gemm_env_refresh();
  HIPCHK(attn_init());
  HIPCHK(chain_init());
  HIPCHK(ff_split_init());
  HIPCHK(qkv_split_init());
  // a new shape re-plans the schedule; the packed weights survive unless the weights or the precision changed
  unet_release_prepared(u, !u->weights_dirty && u->packed_prec == precision);
  u->packed_prec = precision;
  u->B = B; u->T = T; u->L = L; u->precision = precision; u->force_up = force_upsample_size;
  const char* keep = getenv("DVITS_KEEP_INTERMEDIATES");
  u->keep_intermediates = keep && keep[0] == '1';

  const bool persist_env = [] { const char* e = getenv("DVITS_PERSIST"); return e && e[0] == '1'; }();
  // pass 1: measure the arena
  size_t need = 0;
  {
    Builder b{};
    b.u = u; b.dry = true; b.B = B; b.T = T; b.L = L; b.prec = precision;
    b.arena.reuse = !u->keep_intermediates;
    b.arena.exact = persist_env;
    int rc = b.build();
    if (rc != DV_OK) return rc;
    need = b.arena.high;
  }
  HIPCHK(hipMalloc((void**)&u->slab, need + 256));
  u->slab_bytes = need;
  // pass 2: emit the schedule and pack the weights
  {
    Builder b{};
    b.u = u; b.dry = false; b.B = B; b.T = T; b.L = L; b.prec = precision;
    b.arena.reuse = !u->keep_intermediates;
    b.arena.exact = persist_env;
    int rc = b.build();
    if (rc != DV_OK) { unet_release_prepared(u); return rc; }
  }
  if (!persist_env) {
    int rc = autotune_gemms(u, precision);
    if (rc != DV_OK) { unet_release_prepared(u); return rc; }
  }
  // Persistent per-XCD schedule (DVITS_PERSIST=1): the longest run of consecutive step operations that persist.hip
  // can execute (normally conv_in .. the final GroupNorm; input packing, the conditioning GEMVs and conv_out, whose
  // I/O pointers change per call, stay ordinary launches around it).
  {
    const char* pe = getenv("DVITS_PERSIST");
    const int n = (int)u->step_ops.size();
    int best0 = 0, best1 = 0;
    for (int i = 0; i < n;) {
      if (u->step_pop[i] < 0) { ++i; continue; }
      int j = i;
      while (j < n && u->step_pop[j] >= 0) ++j;
      if (j - i > best1 - best0) { best0 = i; best1 = j; }
      i = j;
    }
    if (pe && pe[0] == '1' && best1 - best0 >= 8 && persist_init() == hipSuccess) {
      const int np = best1 - best0;
      HIPCHK(hipMalloc((void**)&u->pops_dev, (size_t)np * sizeof(PersistOp)));
      u->owned.push_back(u->pops_dev);
      HIPCHK(hipMemcpy(u->pops_dev, u->pops.data() + u->step_pop[best0], (size_t)np * sizeof(PersistOp), hipMemcpyHostToDevice));
      HIPCHK(hipMalloc((void**)&u->psync, sizeof(PersistSync)));
      u->owned.push_back(u->psync);
      HIPCHK(hipMemset(u->psync, 0, sizeof(PersistSync)));
      u->p_begin = best0; u->p_end = best1; u->persist_on = true;
    }
  }
  HIPCHK(hipDeviceSynchronize());
  u->prepared = true;
  u->weights_dirty = false;
  u->generation = ++g_generation;
  return DV_OK;
}

// ----------------------------------------------------------------------------- C ABI: prompt encoder
struct dv_penc { dv_unet core; };

extern "C" int dv_penc_create(const dv_penc_cfg* cfg, dv_penc** out) {
  if (!cfg || !out) return dv_fail(DV_ERR_INVALID, "dv_penc_create: null argument");
  if (cfg->in_channels <= 0 || cfg->in_channels > 512) return dv_fail(DV_ERR_INVALID, "in_channels must be 1..512");
  if (cfg->hidden_channels <= 0 || cfg->hidden_channels % 32 != 0 || cfg->hidden_channels > 512)
    return dv_fail(DV_ERR_INVALID, "hidden_channels must be a positive multiple of 32, <= 512");
  if (cfg->out_channels <= 0 || cfg->out_channels % 4 != 0 || cfg->out_channels > 2048)
    return dv_fail(DV_ERR_INVALID, "out_channels must be a positive multiple of 4");
  if (cfg->n_layers < 0 || cfg->n_layers > 64) return dv_fail(DV_ERR_INVALID, "n_layers out of range");
  if (cfg->num_heads <= 0 || cfg->hidden_channels % cfg->num_heads != 0 || (cfg->hidden_channels / cfg->num_heads) % 4 != 0 ||
      cfg->hidden_channels / cfg->num_heads > 64)
    return dv_fail(DV_ERR_INVALID, "head dim must be a multiple of 4 and <= 64");
  if (cfg->ffn_kernel < 1 || cfg->ffn_kernel % 2 != 1) return dv_fail(DV_ERR_INVALID, "ffn_kernel must be odd ('SAME' padding)");
  dv_penc* p = new dv_penc();
  p->core.pcfg = *cfg;
  p->core.cfg.num_heads = cfg->num_heads;
  *out = p;
  return DV_OK;
}

extern "C" void dv_penc_destroy(dv_penc* p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  unet_release_prepared(&p->core);
  for (auto& kv : p->core.w)
    if (kv.second.p) (void)hipFree(kv.second.p);
  delete p;
}

extern "C" int dv_penc_set_weight(dv_penc* p, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim) {
  if (!p) return dv_fail(DV_ERR_INVALID, "dv_penc_set_weight: null handle");
  return dv_unet_set_weight(&p->core, name, dev_ptr, shape, ndim);
}

extern "C" int dv_penc_prepare(dv_penc* p, int32_t B, int32_t L, int32_t precision) {
  if (!p) return dv_fail(DV_ERR_INVALID, "dv_penc_prepare: null handle");
  if (B <= 0 || L <= 0) return dv_fail(DV_ERR_INVALID, "dv_penc_prepare: B, L must be positive");
  if (precision != DV_PREC_BF16X3 && precision != DV_PREC_BF16) return dv_fail(DV_ERR_INVALID, "unknown precision %d", precision);
  dv_unet* u = &p->core;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(gemm_init());
  gemm_env_refresh();
  HIPCHK(attn_init());
  unet_release_prepared(u);
  u->B = B; u->T = L; u->L = L; u->precision = precision; u->force_up = 0;
  const char* keep = getenv("DVITS_KEEP_INTERMEDIATES");
  u->keep_intermediates = keep && keep[0] == '1';
  size_t need = 0;
  {
    Builder b{};
    b.u = u; b.dry = true; b.B = B; b.T = L; b.L = L; b.prec = precision;
    b.arena.reuse = !u->keep_intermediates;
    int rc = b.build_penc();
    if (rc != DV_OK) return rc;
    need = b.arena.high;
  }
  HIPCHK(hipMalloc((void**)&u->slab, need + 256));
  u->slab_bytes = need;
  {
    Builder b{};
    b.u = u; b.dry = false; b.B = B; b.T = L; b.L = L; b.prec = precision;
    b.arena.reuse = !u->keep_intermediates;
    int rc = b.build_penc();
    if (rc != DV_OK) { unet_release_prepared(u); return rc; }
  }
  HIPCHK(hipDeviceSynchronize());
  u->prepared = true;
  u->weights_dirty = false;
  u->generation = ++g_generation;
  return DV_OK;
}

// (DVITS_SYNC_OPS=1, development: every operation is announced on stderr and waited for - a device fault names its launch)
static int run_ops(const std::vector<OpFn>& ops, hipStream_t st, const char* what, const std::vector<dv_unet::OpMeta>* meta = nullptr) {
  static const bool sync_ops = [] { const char* e = getenv("DVITS_SYNC_OPS"); return e && e[0] == '1'; }();
  int i = 0;
  for (const OpFn& f : ops) {
    if (sync_ops) fprintf(stderr, "[dvits] %s op %d %s %s\n", what, i, meta && i < (int)meta->size() ? (*meta)[i].kind : "", meta && i < (int)meta->size() ? (*meta)[i].desc.c_str() : "");
    hipError_t e = f(st);
    if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "%s: op %d failed to launch: %s", what, i, hipGetErrorString(e));
    if (sync_ops && (e = hipStreamSynchronize(st)) != hipSuccess) return dv_fail(DV_ERR_HIP, "%s: op %d failed: %s", what, i, hipGetErrorString(e));
    ++i;
  }
  return DV_OK;
}

extern "C" int dv_penc_forward(dv_penc* p, const float* prompt, const float* keep, float* out, void* stream) {
  if (!p || !prompt || !keep || !out) return dv_fail(DV_ERR_INVALID, "dv_penc_forward: null argument");
  dv_unet* u = &p->core;
  if (!u->prepared) return dv_fail(DV_ERR_STATE, "dv_penc_forward before dv_penc_prepare");
  u->io.x = prompt; u->io.mask = keep; u->io.y = out;
  return run_ops(u->step_ops, (hipStream_t)stream, "prompt encoder");
}

extern "C" int dv_penc_stats(dv_penc* p, int64_t* n_launch, double* flops) {
  if (!p || !p->core.prepared) return dv_fail(DV_ERR_STATE, "dv_penc_stats before prepare");
  if (n_launch) *n_launch = (int64_t)p->core.step_ops.size();
  if (flops) *flops = p->core.flops;
  return DV_OK;
}

extern "C" int dv_unet_set_cond(dv_unet* u, const float* enc, const float* mask_bias, void* stream) {
  if (!u || !enc) return dv_fail(DV_ERR_INVALID, "dv_unet_set_cond: null argument");
  if (!u->prepared) return dv_fail(DV_ERR_STATE, "dv_unet_set_cond before dv_unet_prepare");
  u->io.enc = enc; u->io.mask = mask_bias;
  int rc = run_ops(u->cond_ops, (hipStream_t)stream, "set_cond");
  if (rc == DV_OK) u->cond_set = true;
  return rc;
}

// In-kernel hand-over health (GnxParams): *n_ops = GEMMs of the schedule that finish their consumer's GroupNorm in the
// epilogue; *timed_out = 1 if any such launch since prepare gave up waiting (its results are invalid).  No device
// synchronisation: the flag lives in host memory the kernels write through; complete after the stream has drained.
extern "C" int dv_unet_set_exclusive(dv_unet* u, int32_t exclusive) {
  if (!u) return dv_fail(DV_ERR_INVALID, "dv_unet_set_exclusive: null handle");
  if (u->exclusive != (exclusive != 0)) { u->exclusive = exclusive != 0; u->prepared = false; }   // takes effect at the next prepare
  return DV_OK;
}

// Clears the time-out flag (after the caller has dealt with the lost run: engine.py re-plans the handle without the
// in-launch hand-over and repeats the run); the exchange words are reset at the head of every forward anyway.
extern "C" int dv_unet_handover_reset(dv_unet* u) {
  if (!u) return dv_fail(DV_ERR_INVALID, "dv_unet_handover_reset: null handle");
  HIPCHK(hipDeviceSynchronize());
  if (u->gnx_status) { for (int i = 0; i < 8; ++i) ((volatile unsigned*)u->gnx_status)[i] = 0; }
  return DV_OK;
}
extern "C" int dv_unet_handover_status(dv_unet* u, int32_t* n_ops, int32_t* timed_out) {
  if (!u || !u->prepared) return dv_fail(DV_ERR_STATE, "dv_unet_handover_status before prepare");
  if (n_ops) *n_ops = u->gnx_ops;
  if (timed_out) *timed_out = u->gnx_status ? (int32_t)*(volatile unsigned*)u->gnx_status : 0;
  return DV_OK;
}
int dv_unet_health(const dv_unet* u) {
  if (u->gnx_status && *(volatile unsigned*)u->gnx_status) {
    const volatile unsigned* st = u->gnx_status;
    const GemmParams* hit = nullptr;
    for (const auto& gp : u->gemm_store)
      if (gp->gnx.xchg && (unsigned)(size_t)gp->gnx.xchg == st[1]) hit = gp.get();
    return dv_fail(DV_ERR_HIP, "an earlier launch's in-kernel GroupNorm hand-over timed out (its results are invalid): GEMM M=%d N=%d "
                               "T_out=%d taps=%d split-K=%d, workgroup %u, group %u, %u entries missing; DVITS_GNX=0 runs GroupNorm "
                               "as separate launches",
                   hit ? hit->M : -1, hit ? hit->N : -1, hit ? hit->T_out : -1, hit ? hit->seg[0].taps : -1,
                   hit ? (hit->sk_buf ? hit->sk_split : 0) : -1, st[2], st[3], st[4]);
  }
  return DV_OK;
}

// internal (sampler): the time-embedding chain of n_evals evaluations (t_all [n_evals, B]) in one pass; returns 1 if this
// schedule cannot batch it (then the per-step chain runs as usual), 0 when the table is enqueued, < 0 on error
int dv_unet_temb_all(dv_unet* u, const float* t_all, int n_evals, hipStream_t st) {
  if (!u->prepared || !u->cond_set) return dv_fail(DV_ERR_STATE, "dv_unet_temb_all before prepare / set_cond");
  static const bool off = [] { const char* e = getenv("DVITS_TEMB_BATCH"); return e && e[0] == '0'; }();
  dv_unet::TembCtx& c = u->temb;
  if (off || !c.ok || n_evals < 1 || u->persist_on || u->keep_intermediates) return 1;
  const int R = n_evals * u->B;
  if (R > c.rows_cap) {
    // A larger table (a plan with more evaluations).  The old buffers are NOT freed: captured graphs of other plans
    // (sampler.hip keys them on handle / generation / x / cond, not on this table) have their addresses baked in - the
    // head-of-loop chain writes them and every GroupNorm of that graph reads them - and stay valid as long as they live.
    // They move to the schedule's owned list (released with it); growth is geometric, so at most a few generations exist.
    for (float* b : {c.tsin, c.h1, c.emb, c.tproj_all}) if (b) u->owned.push_back(b);
    c.tsin = c.h1 = c.emb = c.tproj_all = nullptr;
    const int cap = std::max(R, c.rows_cap + c.rows_cap / 2);
    c.rows_cap = 0;
    HIPCHK(hipMalloc((void**)&c.tsin, (size_t)cap * c.C0 * 4));
    HIPCHK(hipMalloc((void**)&c.h1, (size_t)cap * c.E * 4));
    HIPCHK(hipMalloc((void**)&c.emb, (size_t)cap * c.E * 4));
    HIPCHK(hipMalloc((void**)&c.tproj_all, (size_t)cap * c.tt * 4));
    c.rows_cap = cap;
  }
  HIPCHK(launch_timestep_sincos(t_all, c.tsin, R, c.C0, st));
  HIPCHK(launch_small_linear_t(c.tsin, c.C0, c.w1T, c.b1, nullptr, c.h1, c.E, R, c.C0, c.E, 0, 1, st));
  HIPCHK(launch_small_linear_t(c.h1, c.E, c.w2T, c.b2, c.aug_emb, c.emb, c.E, R, c.E, c.E, 0, 0, st, u->B));
  HIPCHK(launch_small_linear_t(c.emb, c.E, c.WtT, c.bt, nullptr, c.tproj_all, c.tt, R, c.E, c.tt, 1, 0, st));
  return 0;
}

// internal: enqueue one forward (used by dv_unet_forward and the sampler).  eval_idx >= 0: the time-embedding rows of
// that evaluation were computed by dv_unet_temb_all - the per-step chain is skipped
int dv_unet_enqueue(dv_unet* u, const float* x, int cx, const float* cond, const float* t, float* y, hipStream_t st, int eval_idx) {
  if (!u->prepared) return dv_fail(DV_ERR_STATE, "forward before dv_unet_prepare");
  if (int hrc = dv_unet_health(u)) return hrc;
  const bool batched = eval_idx >= 0 && u->temb.ok && u->temb.tproj_all && !u->persist_on;
  u->io.tp_base = batched ? u->temb.tproj_all + (size_t)eval_idx * u->B * u->temb.tt : nullptr;
  struct TpReset { dv_unet* u; ~TpReset() { u->io.tp_base = nullptr; } } tp_reset{u};
  if (!u->cond_set) return dv_fail(DV_ERR_STATE, "forward before dv_unet_set_cond");
  if (cx <= 0 || cx > u->cfg.in_channels || (cx < u->cfg.in_channels && !cond))
    return dv_fail(DV_ERR_INVALID, "forward: cx=%d inconsistent with in_channels=%d / cond", cx, u->cfg.in_channels);
  u->io.x = x; u->io.cx = cx; u->io.cond = cond; u->io.t = t; u->io.y = y;
  if (batched) {
    for (int i = 0; i < (int)u->step_ops.size(); ++i) {
      if (i >= u->temb.begin && i < u->temb.end) continue;
      hipError_t e = u->step_ops[i](st);
      if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "forward: op %d failed to launch: %s", i, hipGetErrorString(e));
    }
    return DV_OK;
  }
  if (!u->persist_on) return run_ops(u->step_ops, st, "forward", &u->step_meta);
  for (int i = 0; i < u->p_begin; ++i) {
    hipError_t e = u->step_ops[i](st);
    if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "forward: op %d failed to launch: %s", i, hipGetErrorString(e));
  }
  {
    hipError_t e = launch_persist(u->pops_dev, u->p_end - u->p_begin, u->psync, u->B, st);
    if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "forward: persistent launch failed: %s", hipGetErrorString(e));
  }
  for (int i = u->p_end; i < (int)u->step_ops.size(); ++i) {
    hipError_t e = u->step_ops[i](st);
    if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "forward: op %d failed to launch: %s", i, hipGetErrorString(e));
  }
  return DV_OK;
}

extern "C" int dv_unet_persist_ticks(dv_unet* u, int32_t* first_op, uint64_t* ticks, int32_t capacity) {
  if (!u || !u->prepared || !u->persist_on || !ticks) return dv_fail(DV_ERR_STATE, "dv_unet_persist_ticks: persistent schedule is off");
  PersistSync h;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(&h, u->psync, sizeof(h), hipMemcpyDeviceToHost));
  const int n = std::min(capacity, std::min(1024, u->p_end - u->p_begin + 1));
  for (int i = 0; i < n; ++i) ticks[i] = h.ticks[i];
  if (first_op) *first_op = u->p_begin;
  return n;
}

extern "C" int dv_unet_persist_status(dv_unet* u, int32_t* n_ops, int32_t* error_flag) {
  if (!u || !u->prepared) return dv_fail(DV_ERR_STATE, "dv_unet_persist_status before prepare");
  if (n_ops) *n_ops = u->persist_on ? u->p_end - u->p_begin : 0;
  if (error_flag) {
    *error_flag = 0;
    if (u->persist_on) {
      PersistSync h;
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(&h, u->psync, sizeof(h), hipMemcpyDeviceToHost));
      *error_flag = (int32_t)h.error;
    }
  }
  return DV_OK;
}

extern "C" int dv_unet_forward(dv_unet* u, const float* x, int32_t cx, const float* cond, const float* t, float* y, void* stream) {
  if (!u || !x || !t || !y) return dv_fail(DV_ERR_INVALID, "dv_unet_forward: null argument");
  return dv_unet_enqueue(u, x, cx, cond, t, y, (hipStream_t)stream, -1);
}

extern "C" int dv_unet_forward_timed(dv_unet* u, const float* x, int32_t cx, const float* cond, const float* t, float* y,
                                     void* stream, float* ms_per_op, int32_t capacity) {
  if (!u || !x || !t || !y || !ms_per_op) return dv_fail(DV_ERR_INVALID, "dv_unet_forward_timed: null argument");
  if (!u->prepared || !u->cond_set) return dv_fail(DV_ERR_STATE, "forward before prepare/set_cond");
  const int n = (int)u->step_ops.size();
  if (capacity < n) return dv_fail(DV_ERR_INVALID, "ms_per_op capacity %d < %d ops", capacity, n);
  hipStream_t st = (hipStream_t)stream;
  u->io.x = x; u->io.cx = cx; u->io.cond = cond; u->io.t = t; u->io.y = y;
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) HIPCHK(hipEventCreate(&e));
  HIPCHK(hipEventRecord(ev[0], st));
  for (int i = 0; i < n; ++i) {
    hipError_t e = u->step_ops[i](st);
    if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "op %d failed: %s", i, hipGetErrorString(e));
    HIPCHK(hipEventRecord(ev[i + 1], st));
  }
  HIPCHK(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) HIPCHK(hipEventElapsedTime(&ms_per_op[i], ev[i], ev[i + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  return DV_OK;
}

extern "C" int dv_unet_time_family(dv_unet* u, const char* kind, int32_t reps, void* stream, float* ms_total,
                                   int32_t* launches) {
  if (!u || !kind || !ms_total || !launches || reps < 1) return dv_fail(DV_ERR_INVALID, "dv_unet_time_family: bad argument");
  if (!u->prepared || !u->cond_set || !u->io.x) return dv_fail(DV_ERR_STATE, "dv_unet_time_family needs a completed forward");
  hipStream_t st = (hipStream_t)stream;
  std::vector<int> idx;
  int per_rep = 0;
  for (int i = 0; i < (int)u->step_ops.size(); ++i)
    if (strcmp(u->step_meta[i].kind, kind) == 0) { idx.push_back(i); per_rep += u->step_meta[i].launches(); }
  if (idx.empty()) return dv_fail(DV_ERR_INVALID, "no launch of family '%s' in the schedule", kind);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  // (GEMMs that finish a GroupNorm in their epilogue wait for their exchange words, which the first kernel of a real
  // forward resets: every replay of the family starts with that reset, one extra small launch inside the timed region)
  auto reset_xchg = [&]() { if (u->gnx_pool && u->gnx_words) (void)hipMemsetAsync(u->gnx_pool, 0xff, u->gnx_words * 8, st); };
  reset_xchg();
  for (int i : idx) {               // untimed pass: code and arguments warm
    hipError_t e = u->step_ops[i](st);
    if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "op %d failed: %s", i, hipGetErrorString(e));
  }
  HIPCHK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) {
    reset_xchg();
    for (int i : idx) {
      hipError_t e = u->step_ops[i](st);
      if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "op %d failed: %s", i, hipGetErrorString(e));
    }
  }
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  HIPCHK(hipEventElapsedTime(ms_total, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *launches = (int32_t)per_rep * reps;
  return DV_OK;
}

extern "C" int dv_unet_op_info(dv_unet* u, int32_t index, char* kind16, double* flops, char* desc128) {
  if (!u || !u->prepared || index < 0 || index >= (int)u->step_meta.size()) return dv_fail(DV_ERR_INVALID, "dv_unet_op_info: bad index");
  if (kind16) { strncpy(kind16, u->step_meta[index].kind, 15); kind16[15] = 0; }
  if (desc128) {
    std::string d = u->step_meta[index].desc;
    if (const GemmParams* gp = u->step_meta[index].gp) {
      static const char* names[] = {"auto", "128x128", "128x64", "64x64k2", "64x64", "64x64g", "64x32", "32x32", "32x64g"};
      const int ft = gp->force_tile & 0xff;
      d += std::string(" tile=") + (ft < 9 ? names[ft] : "?") + ((gp->force_tile & GT_BK64) ? "x64" : "");
      if (gp->sk_buf && gp->sk_split >= 2) d += " splitk=" + std::to_string(gp->sk_split) + (gp->sk_ticket && gp->sk_split == 2 ? "f" : "");
    }
    strncpy(desc128, d.c_str(), 127); desc128[127] = 0;
  }
  if (flops) *flops = u->step_meta[index].flops;
  return DV_OK;
}

extern "C" int dv_unet_op_count(dv_unet* u, int32_t* n_ops) {
  if (!u || !u->prepared || !n_ops) return dv_fail(DV_ERR_STATE, "dv_unet_op_count before prepare");
  *n_ops = (int32_t)u->step_ops.size();
  return DV_OK;
}

extern "C" int dv_unet_stats(dv_unet* u, int64_t* n_launch, double* flops) {
  if (!u || !u->prepared) return dv_fail(DV_ERR_STATE, "dv_unet_stats before prepare");
  if (n_launch) {
    *n_launch = 0;
    for (const auto& m : u->step_meta) *n_launch += m.launches();
  }
  if (flops) *flops = u->flops;
  return DV_OK;
}

// accessors for the sampler translation unit
int dv_unet_dims(const dv_unet* u, int* B, int* T, int* cin, int* cout, int64_t* gen) {
  *B = u->B; *T = u->T; *cin = u->cfg.in_channels; *cout = u->cfg.out_channels; *gen = u->generation;
  return u->prepared && u->cond_set;
}

extern "C" int dv_unet_probe(dv_unet* u, const char* name, float* host_out, int64_t capacity, int64_t* dims) {
  if (!u || !name) return dv_fail(DV_ERR_INVALID, "dv_unet_probe: null argument");
  for (const Probe& p : u->probes) {
    if (p.name == name) {
      const int64_t n = (int64_t)u->B * p.T * p.C;
      if (dims) { dims[0] = u->B; dims[1] = p.T; dims[2] = p.C; }
      if (!host_out) return DV_OK;
      if (capacity < n) return dv_fail(DV_ERR_INVALID, "probe buffer too small");
      HIPCHK(hipDeviceSynchronize());
      // (rows of a padded row space - Builder::pitch - are skipped: B pieces of T frames, Tp frames apart)
      HIPCHK(hipMemcpy2D(host_out, (size_t)p.T * p.C * sizeof(float), p.p, (size_t)p.Tp * p.C * sizeof(float),
                         (size_t)p.T * p.C * sizeof(float), (size_t)u->B, hipMemcpyDeviceToHost));
      return DV_OK;
    }
  }
  return dv_fail(DV_ERR_INVALID, "no probe named %s (prepare with DVITS_KEEP_INTERMEDIATES=1)", name);
}

// ----------------------------------------------------------------------------- single-operator entry points
// scratch for the single-operator entry points
struct OpScratch {
  std::vector<void*> ptrs;
  ~OpScratch() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T> T* get(size_t bytes, hipStream_t st, bool zero) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    if (zero) (void)hipMemsetAsync(p, 0, bytes ? bytes : 16, st);
    return reinterpret_cast<T*>(p);
  }
};

extern "C" int dv_op_conv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin, int32_t T,
                            int32_t Cout, int32_t k, int32_t stride, int32_t up_T, int32_t precision, void* stream) {
  if (!x || !w || !y || (k != 1 && k != 3) || (stride != 1 && stride != 2)) return dv_fail(DV_ERR_INVALID, "dv_op_conv1d: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(gemm_init());
  gemm_env_refresh();
  HIPCHK(attn_init());
  const bool x3 = precision == DV_PREC_BF16X3;
  const int cpad = rup(Cin, 32), Kp = k * cpad, Npad = rup(Cout, 128);
  if (2 * (size_t)cpad + 256 > DV_ZERO_PAGE_BYTES)
    return dv_fail(DV_ERR_INVALID, "dv_op_conv1d: Cin = %d exceeds the %d channels one source tensor may have", Cin, (DV_ZERO_PAGE_BYTES - 256) / 2);
  OpScratch sc;
  bf16_t* xh = sc.get<bf16_t>((size_t)B * T * cpad * 2, st, false);
  bf16_t* xl = x3 ? sc.get<bf16_t>((size_t)B * T * cpad * 2, st, false) : nullptr;
  bf16_t* hi = sc.get<bf16_t>((size_t)Npad * Kp * 2, st, true);
  bf16_t* lo = x3 ? sc.get<bf16_t>((size_t)Npad * Kp * 2, st, true) : nullptr;
  bf16_t* zp = sc.get<bf16_t>(DV_ZERO_PAGE_BYTES, st, true);
  if (!xh || !hi || !zp || (x3 && (!xl || !lo))) return dv_fail(DV_ERR_HIP, "dv_op_conv1d: hipMalloc failed");
  HIPCHK(launch_pack_input(x, Cin, nullptr, 0, xh, xl, cpad, B, T, st));
  PackSpec s{};
  s.src = w; s.N = Cout; s.kind = 1; s.C = Cin; s.taps = k; s.c_pad = cpad; s.k_off = 0; s.n_off = 0;
  HIPCHK(launch_pack_weight(s, hi, lo, Kp, st));
  GemmParams g{};
  g.seg[0].a0_hi = xh; g.seg[0].a0_lo = xl; g.seg[0].c0 = cpad; g.seg[0].taps = k; g.seg[0].pad = (k - 1) / 2;
  g.nseg = 1; g.B = B; g.T_in = T;
  g.T_virt = up_T > 0 ? up_T : T;
  g.up_mode = up_T > 0 ? UP_SIZE : UP_NONE;
  g.up_scale = up_T > 0 ? (float)T / (float)up_T : 1.f;
  g.stride = stride;
  g.T_out = (g.T_virt + 2 * g.seg[0].pad - k) / stride + 1;
  g.w_hi = hi; g.w_lo = lo; g.Kp = Kp; g.N_pad = Npad; g.bias = bias;
  g.M = B * g.T_out; g.N = Cout; g.epi = EPI_STORE_NCT; g.out = y; g.ldo = Cout; g.zero_page = zp;
  HIPCHK(launch_gemm(g, precision, st));
  HIPCHK(hipStreamSynchronize(st));
  return DV_OK;
}

// y = x W^T + bias through k_gemm; output as fp32 [M, ldo] (y) and / or split bf16 planes [M, ldo] (y_hi, y_lo);
// geglu: W = [value rows | gate rows] (N = 2 x N_out, reference activations GEGLU: hidden * gelu(gate)), output N / 2 columns
static int op_linear(const float* x, const float* w, const float* bias, float* y, bf16_t* y_hi, bf16_t* y_lo, int M, int K, int N, int ldo,
                     int geglu, int precision, hipStream_t st, const char* what) {
  if (!x || !w || (!y && !y_hi) || K % 32 != 0) return dv_fail(DV_ERR_INVALID, "%s: K must be a multiple of 32 (and an output is needed)", what);
  if (geglu && (N % 64 != 0 || y)) return dv_fail(DV_ERR_INVALID, "%s: GEGLU needs N %% 64 == 0 and writes planes only", what);
  const int n_out = geglu ? N / 2 : N;
  if (ldo < n_out) return dv_fail(DV_ERR_INVALID, "%s: ldo %d < %d output columns", what, ldo, n_out);
  if (2 * (size_t)K + 256 > DV_ZERO_PAGE_BYTES) return dv_fail(DV_ERR_INVALID, "%s: K = %d exceeds the %d channels one source tensor may have", what, K, (DV_ZERO_PAGE_BYTES - 256) / 2);
  HIPCHK(gemm_init());
  gemm_env_refresh();
  HIPCHK(attn_init());
  const bool x3 = precision == DV_PREC_BF16X3;
  const int Npad = rup(N, 128);
  OpScratch sc;
  bf16_t* xh = sc.get<bf16_t>((size_t)M * K * 2, st, false);
  bf16_t* xl = x3 ? sc.get<bf16_t>((size_t)M * K * 2, st, false) : nullptr;
  bf16_t* hi = sc.get<bf16_t>((size_t)Npad * K * 2, st, true);
  bf16_t* lo = x3 ? sc.get<bf16_t>((size_t)Npad * K * 2, st, true) : nullptr;
  bf16_t* zp = sc.get<bf16_t>(DV_ZERO_PAGE_BYTES, st, true);
  float* pb = (geglu && bias) ? sc.get<float>((size_t)Npad * 4, st, true) : nullptr;
  if (!xh || !hi || !zp || (x3 && (!xl || !lo)) || (geglu && bias && !pb)) return dv_fail(DV_ERR_HIP, "%s: hipMalloc failed", what);
  HIPCHK(launch_split(x, xh, xl, (int64_t)M * K, st));
  PackSpec s{};
  s.src = w; s.N = N; s.kind = 0; s.C = K; s.taps = 1; s.c_pad = K; s.geglu = geglu;
  HIPCHK(launch_pack_weight(s, hi, lo, K, st));
  if (pb) HIPCHK(launch_fold_bias(nullptr, bias, nullptr, pb, N, K, 0, 1, st));   // bias in the packed [32 a | 32 gate] column order
  GemmParams g{};
  g.seg[0].a0_hi = xh; g.seg[0].a0_lo = xl; g.seg[0].c0 = K; g.seg[0].taps = 1;
  g.nseg = 1; g.B = 1; g.T_in = g.T_out = g.T_virt = M; g.stride = 1;
  g.w_hi = hi; g.w_lo = lo; g.Kp = K; g.N_pad = Npad; g.bias = geglu ? pb : bias;
  g.M = M; g.N = N; g.epi = geglu ? EPI_GEGLU : EPI_STORE; g.out = y; g.out_hi = y_hi; g.out_lo = y_lo; g.ldo = ldo; g.zero_page = zp;
  hipError_t e = launch_gemm(g, precision, st);
  if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  HIPCHK(hipStreamSynchronize(st));
  return DV_OK;
}

extern "C" int dv_op_linear(const float* x, const float* w, const float* bias, float* y, int32_t M, int32_t K, int32_t N,
                            int32_t precision, void* stream) {
  return op_linear(x, w, bias, y, nullptr, nullptr, M, K, N, N, 0, precision, (hipStream_t)stream, "dv_op_linear");
}

extern "C" int dv_op_linear_planes(const float* x, const float* w, const float* bias, float* y, uint16_t* y_hi, uint16_t* y_lo, int32_t M,
                                   int32_t K, int32_t N, int32_t ldo, int32_t geglu, int32_t precision, void* stream) {
  if (!y_hi || (precision == DV_PREC_BF16X3 && !y_lo)) return dv_fail(DV_ERR_INVALID, "dv_op_linear_planes: plane outputs are required");
  return op_linear(x, w, bias, y, y_hi, precision == DV_PREC_BF16X3 ? y_lo : nullptr, M, K, N, ldo, geglu, precision, (hipStream_t)stream,
                   "dv_op_linear_planes");
}

extern "C" int dv_op_group_stats(const float* x, float* mean, float* rstd, int32_t B, int32_t T, int32_t C, int32_t groups,
                                 float eps, void* stream) {
  if (!x || !mean || !rstd || C % groups != 0 || (C / groups) % 4 != 0 || groups > 64)
    return dv_fail(DV_ERR_INVALID, "dv_op_group_stats: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = std::max(1, std::min(64, (T + 63) / 64));
  double* part = nullptr;
  HIPCHK(hipMalloc((void**)&part, (size_t)B * nchunk * groups * 2 * sizeof(double)));
  HIPCHK(launch_gn_partial(x, C, nullptr, 0, part, B, T, groups, nchunk, st));
  HIPCHK(launch_gn_finalize(part, nchunk, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, mean, rstd, B, T, C, groups, eps, st));
  HIPCHK(hipStreamSynchronize(st));
  (void)hipFree(part);
  return DV_OK;
}

extern "C" int dv_op_attention(const float* q, const float* k, const float* v, const float* bias, float* o, int32_t B,
                               int32_t H, int32_t Tq, int32_t Tk, int32_t d, void* stream) {
  if (!q || !k || !v || !o) return dv_fail(DV_ERR_INVALID, "dv_op_attention: null argument");
  AttnParams a{};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.o = o; a.o_hi = nullptr; a.o_lo = nullptr;
  a.ldq = a.ldk = a.ldv = a.ldo = H * d;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.d = d; a.scale = 1.0f / sqrtf((float)d); a.nsplit = 3;
  HIPCHK(attn_init());
  hipError_t e = launch_attention(a, (hipStream_t)stream);
  if (e != hipSuccess) return dv_fail(DV_ERR_HIP, "attention launch failed: %s (d must be a multiple of 4, <= 64)", hipGetErrorString(e));
  return DV_OK;
}
