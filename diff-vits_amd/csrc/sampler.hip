// Sampler plans and loops behind dv_sampler_* (include/dvits_hip.h).
//
// The reference evaluates every schedule scalar with ~35 tiny tensor ops per look-up inside
// the loop (sampler/dpm_solver.py:1253-1292 interpolate_fn, called ~12x per step).  Nothing in
// those scalars depends on tensor data, so here the whole multistep loop is compiled ONCE on
// the host in fp64 into a list of events
//     EVAL : m[slot] = model(x or x_pred, t_input)
//     COMB : x or x_pred = c0*x + sum_k c_k * m[slot_k]
// (DPM-Solver++ orders 1-3: dpm_solver.py:547-580, 796-831, 854-889; UniPC bh1/bh2/vary_coeff, orders 1-8:
// uni_pc.py:368-588; loops dpm_solver.py:1171-1213 and uni_pc.py:606-658), and executed as
// steps x (UNet schedule + one fused lincomb kernel), captured into a hipGraph.
//
// Deviation from the reference, documented in DESIGN.md section 5: the x0 -> noise -> x0 round trip of
// model_wrapper/data_prediction_fn (dpm_solver.py:290-292, 433-442) is algebraically the
// identity and is not replayed.
#include "../../include/dvits_hip.h"
#include "dv_common.h"

#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int dv_fail(int code, const char* fmt, ...);
struct dv_unet;
int dv_unet_enqueue(dv_unet* u, const float* x, int cx, const float* cond, const float* t, float* y, hipStream_t st, int eval_idx);
int dv_unet_temb_all(dv_unet* u, const float* t_all, int n_evals, hipStream_t st);
int dv_unet_dims(const dv_unet* u, int* B, int* T, int* cin, int* cout, int64_t* gen);
int dv_unet_health(const dv_unet* u);

#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return dv_fail(DV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

namespace {

// NoiseScheduleVP('discrete', betas) — dpm_solver.py:6-167 / uni_pc.py:6-152.  The arrays are
// built in float32 exactly as the reference stores them, then evaluated in fp64.
// kind 1 / 2: the continuous-time VP schedules 'linear' (dpm_solver.py:108-111,133-134,160-163; uni_pc.py the same) and
// 'cosine' (uni_pc.py:73-100 only): closed forms, total_N = 1000, T = 1 / 0.9946; the model then sees t itself
// (get_model_input_time, dpm_solver.py:271-280).
struct Schedule {
  std::vector<double> t_arr, la_arr;   // keypoints (float32 values widened)
  int total_N = 0;
  int kind = 0;                        // 0 discrete, 1 linear, 2 cosine
  double beta0 = 0.1, beta1 = 20.0, T = 1.0;
  static constexpr double COS_S = 0.008;
  double cos_la0 = 0.0;
  void init_continuous(int k, double b0, double b1) {
    kind = k; beta0 = b0; beta1 = b1; total_N = 1000;
    T = k == 2 ? 0.9946 : 1.0;
    cos_la0 = log(cos(COS_S / (1.0 + COS_S) * M_PI / 2.0));
  }
  void init(const float* betas, int n, bool clip) {
    std::vector<float> la(n);
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
      acc += logf(1.0f - betas[i]);       // torch.log(1 - betas).cumsum(0), float32
      la[i] = 0.5f * acc;
    }
    if (clip) {                            // numerical_clip_alpha, dpm_solver.py:114-125
      // lambdas are decreasing in i; count trailing entries with lambda < -5.1
      int idx = 0;
      for (int i = n - 1; i >= 0; --i) {
        const float ls = 0.5f * logf(1.0f - expf(2.0f * la[i]));
        if (la[i] - ls < -5.1f) ++idx; else break;
      }
      if (idx > 0) la.resize(n - idx);
    }
    total_N = (int)la.size();
    la_arr.assign(la.begin(), la.end());
    t_arr.resize(total_N);
    // torch.linspace(0, 1, N+1)[1:], float32 (symmetric fill as ATen does)
    const int pts = total_N + 1;
    const float step = 1.0f / (float)(pts - 1);
    for (int i = 1; i < pts; ++i) {
      const float v = (i < pts / 2) ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(pts - i - 1), 1.0f);
      t_arr[i - 1] = (double)v;
    }
  }
  static double interp(double x, const std::vector<double>& xp, const std::vector<double>& yp, bool flipped) {
    // piecewise linear with outermost-segment extrapolation (interpolate_fn)
    const int K = (int)xp.size();
    auto X = [&](int i) { return flipped ? xp[K - 1 - i] : xp[i]; };
    auto Y = [&](int i) { return flipped ? yp[K - 1 - i] : yp[i]; };
    int lo = 0, hi = K;                    // first index with X(i) >= x
    while (lo < hi) { int mid = (lo + hi) / 2; if (X(mid) < x) lo = mid + 1; else hi = mid; }
    int i = lo - 1;
    if (i < 0) i = 0;
    if (i > K - 2) i = K - 2;
    return Y(i) + (x - X(i)) * (Y(i + 1) - Y(i)) / (X(i + 1) - X(i));
  }
  double log_alpha(double t) const {
    if (kind == 1) return -0.25 * t * t * (beta1 - beta0) - 0.5 * t * beta0;
    if (kind == 2) return log(cos((t + COS_S) / (1.0 + COS_S) * M_PI / 2.0)) - cos_la0;
    return interp(t, t_arr, la_arr, false);
  }
  double alpha(double t) const { return exp(log_alpha(t)); }
  double sigma(double t) const { return sqrt(1.0 - exp(2.0 * log_alpha(t))); }
  double lambda(double t) const { const double la = log_alpha(t); return la - 0.5 * log(1.0 - exp(2.0 * la)); }
  double inverse_lambda(double lamb) const {
    // log_alpha = -0.5 * logaddexp(0, -2 lamb); interpolate t on the flipped arrays
    const double a = -2.0 * lamb;
    const double lse = a > 0 ? a + log1p(exp(-a)) : log1p(exp(a));      // logaddexp(-2 lamb, 0)
    if (kind == 1) {
      const double tmp = 2.0 * (beta1 - beta0) * lse, delta = beta0 * beta0 + tmp;
      return tmp / (sqrt(delta) + beta0) / (beta1 - beta0);
    }
    if (kind == 2) return acos(exp(-0.5 * lse + cos_la0)) * 2.0 * (1.0 + COS_S) / M_PI - COS_S;
    const double lae = -0.5 * lse;
    return interp(lae, la_arr, t_arr, true);
  }
};

struct Event {
  int type;          // 0 EVAL, 1 COMB
  int src;           // EVAL: input, COMB: base of the c0 term; 0 = x, 1 = x_pred
  int eval_idx;      // EVAL: index into t_input
  int dst;           // EVAL: history slot;  COMB: 0 = x, 1 = x_pred, 2 + s = history slot s (in place: x0 -> noise, DV_SOLVER_DPM)
  int coef;          // COMB: row of the coefficient table
  int slots[4];      // COMB: history slots of the m terms (-1 = unused)
};

// solve A x = b (n <= MAXO), Gaussian elimination with partial pivoting (torch.linalg.solve / inv of the reference)
constexpr int MAXO = 8;               // highest multistep order compiled (history slots: order + 1)
bool solve_small(int n, double A[MAXO][MAXO], double b[MAXO], double x[MAXO]) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (fabs(A[piv][c]) < 1e-300) return false;
    if (piv != c) { for (int k = 0; k < n; ++k) std::swap(A[c][k], A[piv][k]); std::swap(b[c], b[piv]); }
    for (int r = c + 1; r < n; ++r) {
      const double f = A[r][c] / A[c][c];
      for (int k = c; k < n; ++k) A[r][k] -= f * A[c][k];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[r][k] * x[k];
    x[r] = s / A[r][r];
  }
  return true;
}
// inverse of an n x n matrix (columns = solutions for the unit vectors)
bool invert_small(int n, const double A[MAXO][MAXO], double inv[MAXO][MAXO]) {
  for (int c = 0; c < n; ++c) {
    double M[MAXO][MAXO], e[MAXO], col[MAXO];
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) M[i][j] = A[i][j]; e[i] = i == c ? 1.0 : 0.0; }
    if (!solve_small(n, M, e, col)) return false;
    for (int i = 0; i < n; ++i) inv[i][c] = col[i];
  }
  return true;
}

}  // namespace

struct dv_plan {
  int solver = 0, steps = 0, order = 0, skip = 0, lof = 1;
  double t_start = -1.0, t_end = -1.0;   // <= 0: the defaults T and 1/N
  int denoise_to_zero = 0;
  Schedule ns;
  std::vector<double> timesteps;      // steps + 1
  std::vector<double> t_input;        // per EVAL
  std::vector<double> eval_time;      // per EVAL: its continuous time
  int method = 0;                     // 0 multistep, 1 singlestep ("DPM-Solver-fast"), 2 singlestep_fixed (DPM-Solver(++) only)
  std::vector<Event> ev;
  std::vector<std::array<float, 8>> coefs;
  int n_slots = 0;
  // device state
  float* d_coefs = nullptr;
  float* d_tin = nullptr; int tin_B = 0;
  float* xp = nullptr; std::vector<float*> m; int64_t buf_numel = 0;
  // graph cache
  hipGraphExec_t exec = nullptr; hipStream_t cap_stream = nullptr;
  struct { dv_unet* u = nullptr; int64_t gen = -1; float* x = nullptr; const float* cond = nullptr; } key;
};

static int build_plan(dv_plan* p) {
  const Schedule& ns = p->ns;
  const int N = p->steps, order = p->order;
  // dpm_solver.py:1157-1158 / uni_pc.py:596-597: t_0 = 1/N unless t_end is given, t_T = T unless t_start is given
  const double t_0 = p->t_end > 0 ? p->t_end : 1.0 / ns.total_N, t_T = p->t_start > 0 ? p->t_start : ns.T;
  // ---- time grid (get_time_steps, dpm_solver.py:453-480), float32 as the reference stores it
  auto time_grid = [&](double tT, double t0, int n, std::vector<double>& out) -> bool {   // n steps: n + 1 points from tT to t0
    out.assign(n + 1, 0.0);
    auto linspace32 = [&](float a, float b, std::vector<double>& o) {
      const int pts = n + 1;
      const float step = (b - a) / (float)(pts - 1);
      for (int i = 0; i < pts; ++i)   // ATen's vectorised fill rounds once per element (fma)
        o[i] = (double)((i < pts / 2) ? fmaf(step, (float)i, a) : fmaf(-step, (float)(pts - i - 1), b));
    };
    if (p->skip == DV_SKIP_TIME_UNIFORM) linspace32((float)tT, (float)t0, out);
    else if (p->skip == DV_SKIP_TIME_QUADRATIC) {
      linspace32((float)sqrt(tT), (float)sqrt(t0), out);
      for (auto& v : out) { const float f = (float)v; v = (double)(f * f); }
    } else if (p->skip == DV_SKIP_LOGSNR) {
      std::vector<double> lam(n + 1);
      linspace32((float)ns.lambda(tT), (float)ns.lambda(t0), lam);
      for (int i = 0; i <= n; ++i) out[i] = (double)(float)ns.inverse_lambda(lam[i]);
    } else return false;
    return true;
  };
  if (!time_grid(t_T, t_0, N, p->timesteps)) return dv_fail(DV_ERR_INVALID, "Unsupported skip_type %d", p->skip);

  auto t_in = [&](double t) {   // get_model_input_time, float32 arithmetic (dpm_solver.py:271-280)
    const float tf = (float)t;
    if (ns.kind != 0) return (double)tf;            // continuous-time schedules: the model's time is t
    return (double)((tf - (float)(1.0 / ns.total_N)) * (float)ns.total_N);
  };
  auto add_eval = [&](int src, double t, int slot) {
    Event e{}; e.type = 0; e.src = src; e.eval_idx = (int)p->t_input.size(); e.dst = slot;
    p->t_input.push_back(t_in(t));
    p->eval_time.push_back(t);
    p->ev.push_back(e);
  };
  // COMB: dst = c0 * x + sum_k c_k * m[slot_k].  The update kernel takes four m terms: longer sums (orders >= 4) are
  // chained through x_pred (dst' = 1 * x_pred + next four terms) - x_pred is free whenever the chain is needed: a
  // predictor writes it anyway, and the corrector's EVAL has consumed it by the time the corrector sum is formed.
  auto add_comb = [&](int dst, double c0, const std::vector<std::pair<int, double>>& terms) {
    const int n = (int)terms.size();
    int done = 0, src = 0;
    do {
      const int take = std::min(4, n - done), last = done + take >= n;
      Event e{}; e.type = 1; e.src = src; e.dst = last ? dst : 1; e.coef = (int)p->coefs.size();
      std::array<float, 8> row{};
      row[0] = done == 0 ? (float)c0 : 1.0f;
      for (int k = 0; k < 4; ++k) e.slots[k] = -1;
      for (int k = 0; k < take; ++k) { e.slots[k] = terms[done + k].first; row[1 + k] = (float)terms[done + k].second; }
      p->coefs.push_back(row);
      p->ev.push_back(e);
      done += take; src = 1;
    } while (done < n);
  };
  const std::vector<double>& ts = p->timesteps;
  std::vector<int> hist;            // history slots, newest first
  std::vector<double> htime;        // their times
  const bool taylor = p->solver == DV_SOLVER_DPMPP_TAYLOR || p->solver == DV_SOLVER_DPM_TAYLOR;   // solver_type='taylor' (second order only)
  const bool noise = p->solver == DV_SOLVER_DPM || p->solver == DV_SOLVER_DPM_TAYLOR;   // multistep updates on the noise prediction (algorithm_type='dpmsolver')
  const bool unoise = p->solver >= DV_SOLVER_UNIPC_BH1_NOISE && p->solver <= DV_SOLVER_UNIPC_VARY_NOISE;   // UniPC on the noise prediction
  const bool unipc = (p->solver >= DV_SOLVER_UNIPC_BH1 && p->solver <= DV_SOLVER_UNIPC_VARY) || unoise;
  p->n_slots = unipc ? order + 1 : order;
  auto free_slot = [&]() {
    for (int s = 0; s < p->n_slots; ++s) {
      bool used = false;
      for (int h : hist) used |= (h == s);
      if (!used) return s;
    }
    return -1;
  };
  auto push_hist = [&](int slot, double t) {
    hist.insert(hist.begin(), slot);
    htime.insert(htime.begin(), t);
    if ((int)hist.size() > order) { hist.pop_back(); htime.pop_back(); }
  };

  if (!unipc) {
    // ---------------- DPM-Solver++ / DPM-Solver multistep ----------------
    // (noise form: the network predicts x0; its output becomes eps = (x - alpha x0) / sigma in place, with the x it was
    // evaluated on - model_wrapper's 'x_start' branch, dpm_solver.py:290-292)
    auto add_eval_m = [&](double t, int slot, int src = 0) {
      add_eval(src, t, slot);
      if (!noise) return;
      Event e{}; e.type = 1; e.src = src; e.dst = 2 + slot; e.coef = (int)p->coefs.size();
      std::array<float, 8> row{};
      // (x - alpha m) / sigma with the subtraction BEFORE the scaling, as the reference computes it (dpm_solver.py:290-292) -
      // as c0 x + c1 m with two float32-rounded coefficients 1 / sigma and -alpha / sigma the cancellation is amplified by
      // 1 / sigma at the low-noise end (singlestep order 3 with 3-4 steps: 7e-4 from the reference).  row[7] = 1
      // marks the form for k_lincomb / Plan.run_python: row[0] = sigma, row[1] = alpha.
      row[0] = (float)ns.sigma(t); row[1] = (float)ns.alpha(t); row[7] = 1.0f;
      e.slots[0] = slot; e.slots[1] = e.slots[2] = e.slots[3] = -1;
      p->coefs.push_back(row);
      p->ev.push_back(e);
    };
    auto update_noise = [&](double t, int ord) {      // dpm_solver.py:581-592 (first), 841-847 (second), 895-904 (third)
      const double t0 = htime[0];
      const double lam0 = ns.lambda(t0), lam_t = ns.lambda(t);
      const double h = lam_t - lam0, phi_1 = expm1(h);
      const double c0 = exp(ns.log_alpha(t) - ns.log_alpha(t0)), s = ns.sigma(t);
      if (ord == 1) { add_comb(0, c0, {{hist[0], -s * phi_1}}); return; }
      if (ord == 2) {
        const double r0 = (lam0 - ns.lambda(htime[1])) / h;
        const double d = taylor ? s * (phi_1 / h - 1.0) : 0.5 * s * phi_1;      // coefficient of D1_0 = (m0 - m1) / r0 (:841-851)
        add_comb(0, c0, {{hist[0], -s * phi_1 - d / r0}, {hist[1], d / r0}});
        return;
      }
      const double lam1 = ns.lambda(htime[1]), lam2 = ns.lambda(htime[2]);
      const double r0 = (lam0 - lam1) / h, r1 = (lam1 - lam2) / h;
      const double a0 = 1.0 / r0, a1 = 1.0 / r1, g = r0 / (r0 + r1), e = 1.0 / (r0 + r1);
      const double phi_2 = phi_1 / h - 1.0, phi_3 = phi_2 / h - 0.5;
      const double S2 = s * phi_2, S3 = s * phi_3;
      add_comb(0, c0, {{hist[0], -s * phi_1 - S2 * (1 + g) * a0 - S3 * e * a0},
                       {hist[1], S2 * (1 + g) * a0 + S2 * g * a1 + S3 * e * a0 + S3 * e * a1},
                       {hist[2], -S2 * g * a1 - S3 * e * a1}});
    };
    auto update = [&](double t, int ord) {
      if (noise) { update_noise(t, ord); return; }
      const double t0 = htime[0];
      const double lam0 = ns.lambda(t0), lam_t = ns.lambda(t);
      const double h = lam_t - lam0, phi_1 = expm1(-h);
      const double c0 = ns.sigma(t) / ns.sigma(t0), a = ns.alpha(t);
      if (ord == 1) { add_comb(0, c0, {{hist[0], -a * phi_1}}); return; }
      if (ord == 2) {
        const double r0 = (lam0 - ns.lambda(htime[1])) / h;
        if (taylor) {                                   // x_t = c0 x - a phi_1 m0 + a (phi_1 / h + 1) D1_0 (:825-829)
          const double d = a * (phi_1 / h + 1.0);
          add_comb(0, c0, {{hist[0], -a * phi_1 + d / r0}, {hist[1], -d / r0}});
          return;
        }
        add_comb(0, c0, {{hist[0], -a * phi_1 * (1.0 + 0.5 / r0)}, {hist[1], 0.5 * a * phi_1 / r0}});
        return;
      }
      const double lam1 = ns.lambda(htime[1]), lam2 = ns.lambda(htime[2]);
      const double r0 = (lam0 - lam1) / h, r1 = (lam1 - lam2) / h;
      const double a0 = 1.0 / r0, a1 = 1.0 / r1, g = r0 / (r0 + r1), e = 1.0 / (r0 + r1);
      const double phi_2 = phi_1 / h + 1.0, phi_3 = phi_2 / h - 0.5;
      const double P2 = a * phi_2, P3 = a * phi_3;
      add_comb(0, c0, {{hist[0], -a * phi_1 + P2 * (1 + g) * a0 - P3 * e * a0},
                       {hist[1], -P2 * (1 + g) * a0 - P2 * g * a1 + P3 * e * a0 + P3 * e * a1},
                       {hist[2], P2 * g * a1 - P3 * e * a1}});
    };
    if (p->method != 0) {
      // ---------------- singlestep DPM-Solver(++) ("DPM-Solver-fast", dpm_solver.py:482-539, 594-794, 1214-1232) ----------------
      // outer grid + the order of every outer step (NFE = steps); inside a step of order k the evaluations sit at
      // lambda_s + r_i h with r_i from the step's own time grid of k pieces.  History slots: 0 model_s, 1 model_s1, 2 model_s2.
      std::vector<int> orders;
      int K = 0;
      if (p->method == 2) { K = N / order; orders.assign(K, order); }
      else if (order == 3) {
        K = N / 3 + 1;
        if (N % 3 == 0) { orders.assign(std::max(K - 2, 0), 3); orders.push_back(2); orders.push_back(1); }
        else if (N % 3 == 1) { orders.assign(K - 1, 3); orders.push_back(1); }
        else { orders.assign(K - 1, 3); orders.push_back(2); }
      } else if (order == 2) {
        if (N % 2 == 0) { K = N / 2; orders.assign(K, 2); } else { K = N / 2 + 1; orders.assign(K - 1, 2); orders.push_back(1); }
      } else { K = 1; orders.assign(N, 1); }
      std::vector<double> outer;
      if (p->method == 2 || p->skip == DV_SKIP_LOGSNR) { if (!time_grid(t_T, t_0, K, outer)) return dv_fail(DV_ERR_INVALID, "Unsupported skip_type %d", p->skip); }
      else {                                         // the multistep grid at the cumulative orders (:533-534)
        outer.push_back(ts[0]);
        int c = 0;
        for (int o : orders) { c += o; outer.push_back(ts[c]); }
      }
      if (outer.size() != orders.size() + 1)         // (order 1 with skip_type 'logSNR': the reference indexes past its K = 1 grid)
        return dv_fail(DV_ERR_INVALID, "singlestep: %d steps of order %d need %d grid points, the '%s' grid has %d (reference: IndexError)",
                       N, order, (int)orders.size() + 1, "logSNR", (int)outer.size());
      p->n_slots = 3;
      p->timesteps = outer;
      for (size_t st = 0; st < orders.size(); ++st) {
        const int k = orders[st];
        const double s_ = outer[st], t = outer[st + 1];
        std::vector<double> inner;
        time_grid(s_, t, k, inner);
        const double lam_s = ns.lambda(s_), lam_t = ns.lambda(t), h = lam_t - lam_s;
        // (r from the float32 lambdas of the inner grid, as the reference computes them: :1224-1227)
        const double hi = (double)((float)ns.lambda(inner[k]) - (float)ns.lambda(inner[0]));
        const double r1 = k >= 2 ? (double)(((float)ns.lambda(inner[1]) - (float)ns.lambda(inner[0])) / (float)hi) : 0.0;
        const double r2 = k >= 3 ? (double)(((float)ns.lambda(inner[2]) - (float)ns.lambda(inner[0])) / (float)hi) : 0.0;
        const double la_s = ns.log_alpha(s_), sg_s = ns.sigma(s_);
        // x at time u from (x, model_s): first-order piece shared by every stage; pp: coefficient of x, of model_s
        auto first = [&](double u, double hu, double& cx, double& cm) {
          if (noise) { cx = exp(ns.log_alpha(u) - la_s); cm = -ns.sigma(u) * expm1(hu); }
          else { cx = ns.sigma(u) / sg_s; cm = -ns.alpha(u) * expm1(-hu); }
        };
        double cx, cm;
        add_eval_m(s_, 0);
        if (k == 1) { first(t, h, cx, cm); add_comb(0, cx, {{0, cm}}); continue; }
        const double s1 = (double)(float)ns.inverse_lambda(lam_s + r1 * h);
        first(s1, r1 * h, cx, cm);
        add_comb(1, cx, {{0, cm}});
        add_eval_m(s1, 1, 1);
        const double sgn = noise ? 1.0 : -1.0;                 // h enters the phi functions as -h in the data form
        const double amp_t = noise ? -ns.sigma(t) : ns.alpha(t);   // x_t = cx x + amp_t * (...): alpha_t for x0, -sigma_t for noise
        const double phi_1 = expm1(sgn * h);
        first(t, h, cx, cm);
        if (k == 2) {
          // dpmsolver: x_t = cx x + cm m0 - (0.5 / r1) (a phi_1) (m1 - m0)   [noise: - (0.5 / r1) (sigma phi_1) (m1 - m0)]
          // taylor   : x_t = cx x + cm m0 + (1 / r1) a (phi_1 / h + 1) (m1 - m0)   [noise: - (1 / r1) sigma (phi_1 / h - 1) (m1 - m0)]
          double d;
          if (!taylor) d = noise ? -(0.5 / r1) * ns.sigma(t) * phi_1 : -(0.5 / r1) * ns.alpha(t) * phi_1;
          else d = noise ? -(1.0 / r1) * ns.sigma(t) * (phi_1 / h - 1.0) : (1.0 / r1) * ns.alpha(t) * (phi_1 / h + 1.0);
          add_comb(0, cx, {{0, cm - d}, {1, d}});
          continue;
        }
        // ---- third order ----
        const double s2 = (double)(float)ns.inverse_lambda(lam_s + r2 * h);
        const double phi_22 = noise ? expm1(r2 * h) / (r2 * h) - 1.0 : expm1(-r2 * h) / (r2 * h) + 1.0;
        const double phi_2 = noise ? phi_1 / h - 1.0 : phi_1 / h + 1.0;
        const double phi_3 = phi_2 / h - 0.5;
        {   // x_s2 = cx x + cm m0 (+|-) (r2 / r1) (a_s2 | sigma_s2) phi_22 (m1 - m0)
          double c2x, c2m;
          first(s2, r2 * h, c2x, c2m);
          const double d = noise ? -(r2 / r1) * ns.sigma(s2) * phi_22 : (r2 / r1) * ns.alpha(s2) * phi_22;
          add_comb(1, c2x, {{0, c2m - d}, {1, d}});
        }
        add_eval_m(s2, 2, 1);
        if (!taylor) {                                         // x_t = cx x + cm m0 (+|-) (1 / r2) (a | sigma) phi_2 (m2 - m0)
          const double d = noise ? -(1.0 / r2) * ns.sigma(t) * phi_2 : (1.0 / r2) * ns.alpha(t) * phi_2;
          add_comb(0, cx, {{0, cm - d}, {2, d}});
        } else {
          // D1_0 = (m1 - m0) / r1, D1_1 = (m2 - m0) / r2, D1 = (r2 D1_0 - r1 D1_1) / (r2 - r1), D2 = 2 (D1_1 - D1_0) / (r2 - r1)
          // x_t = cx x + cm m0 + A2 D1 - A3 D2 with (A2, A3) = (a phi_2, a phi_3) [noise: (-sigma phi_2, +sigma phi_3)]
          const double A2 = noise ? -ns.sigma(t) * phi_2 : ns.alpha(t) * phi_2;
          const double A3 = noise ? ns.sigma(t) * phi_3 : ns.alpha(t) * phi_3;
          const double q = 1.0 / (r2 - r1);
          const double w1 = A2 * q * r2 / r1 + A3 * 2.0 * q / r1;          // coefficient of (m1 - m0)
          const double w2 = -A2 * q * r1 / r2 - A3 * 2.0 * q / r2;         // coefficient of (m2 - m0)
          add_comb(0, cx, {{0, cm - w1 - w2}, {1, w1}, {2, w2}});
        }
        (void)amp_t;
      }
      if (p->denoise_to_zero) {
        add_eval(0, t_0, 0);
        add_comb(0, 0.0, {{0, 1.0}});
      }
      return DV_OK;
    }
    add_eval_m(ts[0], 0);
    push_hist(0, ts[0]);
    for (int step = 1; step < order; ++step) {
      update(ts[step], step);
      int s = free_slot();
      add_eval_m(ts[step], s);
      push_hist(s, ts[step]);
    }
    for (int step = order; step <= N; ++step) {
      const int so = (p->lof && N < 10) ? std::min(order, N + 1 - step) : order;
      update(ts[step], so);
      if (step < N) {
        // the new evaluation overwrites the oldest history entry (no longer needed)
        int s;
        if ((int)hist.size() == order) { s = hist.back(); hist.pop_back(); htime.pop_back(); }
        else s = free_slot();
        add_eval_m(ts[step], s);
        push_hist(s, ts[step]);
      }
    }
  } else {
    // ---------------- UniPC multistep, x0-prediction: B(h) variants (uni_pc.py:471-588) and 'vary_coeff'
    // (multistep_uni_pc_vary_update, uni_pc.py:368-469); any order <= MAXO (the reference solves the order x order
    // systems with torch.linalg.solve / inv, :545-560, :410-420) ----------------
    const bool bh1 = p->solver == DV_SOLVER_UNIPC_BH1 || p->solver == DV_SOLVER_UNIPC_BH1_NOISE;
    const bool vary = p->solver == DV_SOLVER_UNIPC_VARY || p->solver == DV_SOLVER_UNIPC_VARY_NOISE;
    // (noise form: the network predicts x0; its output becomes eps = (x_in - alpha x0) / sigma in place - see DV_SOLVER_DPM)
    auto add_eval_u = [&](int src, double t, int slot) {
      add_eval(src, t, slot);
      if (!unoise) return;
      Event e{}; e.type = 1; e.src = src; e.dst = 2 + slot; e.coef = (int)p->coefs.size();
      std::array<float, 8> row{};
      // (x - alpha m) / sigma with the subtraction BEFORE the scaling, as the reference computes it (dpm_solver.py:290-292) -
      // as c0 x + c1 m with two float32-rounded coefficients 1 / sigma and -alpha / sigma the cancellation is amplified by
      // 1 / sigma at the low-noise end (singlestep order 3 with 3-4 steps: 7e-4 from the reference).  row[7] = 1
      // marks the form for k_lincomb / Plan.run_python: row[0] = sigma, row[1] = alpha.
      row[0] = (float)ns.sigma(t); row[1] = (float)ns.alpha(t); row[7] = 1.0f;
      e.slots[0] = slot; e.slots[1] = e.slots[2] = e.slots[3] = -1;
      p->coefs.push_back(row);
      p->ev.push_back(e);
    };
    auto do_step = [&](double t, int ord, bool corr) -> int {
      const double t0 = htime[0];
      const double lam0 = ns.lambda(t0), lam_t = ns.lambda(t);
      // data form: hh = -h, amplitude alpha_t, x coefficient sigma_t / sigma_0; noise form (uni_pc.py:448-468, 569-587): hh = h,
      // amplitude sigma_t, x coefficient exp(log alpha_t - log alpha_0) - the same algebra otherwise
      const double h = lam_t - lam0, hh = unoise ? h : -h;
      const double a = unoise ? ns.sigma(t) : ns.alpha(t);
      const double c0 = unoise ? exp(ns.log_alpha(t) - ns.log_alpha(t0)) : ns.sigma(t) / ns.sigma(t0);
      const double h_phi_1 = expm1(hh);
      double rks[MAXO];
      for (int i = 1; i < ord; ++i) rks[i - 1] = (ns.lambda(htime[i]) - lam0) / h;
      rks[ord - 1] = 1.0;
      // Both variants reduce to  x_t = c0 x - a h_phi_1 m0 - a * sum_j w_j D1_j  (- a * w_t (m_t - m0) in the
      // corrector), D1_j = (m_j - m0) / rk_j for the older evaluations j = 1 .. ord-1: wp[] / wc[] are the weights.
      double wp[MAXO] = {0}, wc[MAXO] = {0}, wt = 0.0;
      if (!vary) {
        const double B_h = bh1 ? hh : expm1(hh);
        double R[MAXO][MAXO], b[MAXO];
        {
          double h_phi_k = h_phi_1 / hh - 1.0, fact = 1.0;
          for (int i = 1; i <= ord; ++i) {
            for (int j = 0; j < ord; ++j) R[i - 1][j] = pow(rks[j], i - 1);
            b[i - 1] = h_phi_k * fact / B_h;
            fact *= (i + 1);
            h_phi_k = h_phi_k / hh - 1.0 / fact;
          }
        }
        double rho_p[MAXO] = {0}, rho_c[MAXO] = {0};
        if (ord == 2) rho_p[0] = 0.5;                       // the reference's simplified order-2 predictor
        else if (ord >= 3) {
          double A2[MAXO][MAXO], b2[MAXO];
          for (int i = 0; i < ord - 1; ++i) { for (int j = 0; j < ord - 1; ++j) A2[i][j] = R[i][j]; b2[i] = b[i]; }
          if (!solve_small(ord - 1, A2, b2, rho_p)) return dv_fail(DV_ERR_INVALID, "UniPC predictor system is singular");
        }
        if (corr) {
          if (ord == 1) rho_c[0] = 0.5;                     // simplified order-1 corrector
          else {
            double A3[MAXO][MAXO], b3[MAXO];
            for (int i = 0; i < ord; ++i) { for (int j = 0; j < ord; ++j) A3[i][j] = R[i][j]; b3[i] = b[i]; }
            if (!solve_small(ord, A3, b3, rho_c)) return dv_fail(DV_ERR_INVALID, "UniPC corrector system is singular");
          }
        }
        for (int j = 0; j < ord - 1; ++j) { wp[j] = B_h * rho_p[j]; wc[j] = B_h * rho_c[j]; }
        wt = B_h * rho_c[ord - 1];
      } else {
        const int K = ord;
        double Cm[MAXO][MAXO];                              // C[i][k] = rks[i]^k / (k+1)!   (uni_pc.py:399-405)
        for (int i = 0; i < K; ++i) {
          double col = 1.0;
          for (int k = 1; k <= K; ++k) { Cm[i][k - 1] = col; col = col * rks[i] / (k + 1); }
        }
        double h_phi_ks[MAXO + 2];                          // uni_pc.py:419-426
        {
          double h_phi_k = h_phi_1, fact = 1.0;
          for (int k = 1; k <= K + 1; ++k) { h_phi_ks[k - 1] = h_phi_k; h_phi_k = h_phi_k / hh - 1.0 / fact; fact *= (k + 1); }
        }
        if (K > 1) {
          double Ap[MAXO][MAXO];
          if (!invert_small(K - 1, Cm, Ap)) return dv_fail(DV_ERR_INVALID, "UniPC vary_coeff predictor matrix is singular");
          for (int k = 0; k < K - 1; ++k)
            for (int j = 0; j < K - 1; ++j) wp[j] += h_phi_ks[k + 1] * Ap[k][j];
        }
        if (corr) {
          double Ac[MAXO][MAXO];
          if (!invert_small(K, Cm, Ac)) return dv_fail(DV_ERR_INVALID, "UniPC vary_coeff corrector matrix is singular");
          for (int k = 0; k < K - 1; ++k)
            for (int j = 0; j < K - 1; ++j) wc[j] += h_phi_ks[k + 1] * Ac[k][j];
          // the reference indexes A_c with the residual loop's LAST k (uni_pc.py:444-447: `k` leaks out of the loop;
          // 0 when the loop body never ran) - reproduced as written
          const int kq = K >= 2 ? K - 2 : 0;
          wt = h_phi_ks[K] * Ac[kq][K - 1];
        }
      }
      // predictor
      {
        std::vector<std::pair<int, double>> terms;
        double cm0 = -a * h_phi_1;
        for (int k = 1; k < ord; ++k) cm0 += a * wp[k - 1] / rks[k - 1];
        terms.push_back({hist[0], cm0});
        for (int k = 1; k < ord; ++k) terms.push_back({hist[k], -a * wp[k - 1] / rks[k - 1]});
        add_comb(corr ? 1 : 0, c0, terms);
      }
      if (corr) {
        const int s = free_slot();
        add_eval_u(1, t, s);
        std::vector<std::pair<int, double>> terms;
        double cm0 = -a * h_phi_1 + a * wt;
        for (int k = 1; k < ord; ++k) cm0 += a * wc[k - 1] / rks[k - 1];
        terms.push_back({hist[0], cm0});
        for (int k = 1; k < ord; ++k) terms.push_back({hist[k], -a * wc[k - 1] / rks[k - 1]});
        terms.push_back({s, -a * wt});
        add_comb(0, c0, terms);
        push_hist(s, t);
      }
      return DV_OK;
    };
    add_eval_u(0, ts[0], 0);
    push_hist(0, ts[0]);
    for (int step = 1; step < order; ++step) {
      int rc = do_step(ts[step], step, true);
      if (rc != DV_OK) return rc;
    }
    for (int step = order; step <= N; ++step) {
      const int so = p->lof ? std::min(order, N + 1 - step) : order;
      int rc = do_step(ts[step], so, step != N);
      if (rc != DV_OK) return rc;
    }
  }
  if (p->denoise_to_zero) {   // dpm_solver.py:1234-1240, uni_pc.py:660-666: x <- data prediction at t_0 (one more evaluation)
    add_eval(0, t_0, 0);
    add_comb(0, 0.0, {{0, 1.0}});
  }
  return DV_OK;
}

extern "C" int dv_sampler_plan(int32_t solver, const float* betas, int32_t n_betas, int32_t steps, int32_t order,
                               int32_t skip_type, int32_t lower_order_final, dv_plan** out) {
  return dv_sampler_plan_ex(solver, betas, n_betas, steps, order, skip_type, lower_order_final, -1.0, -1.0, 0, out);
}

extern "C" int dv_sampler_plan_ex(int32_t solver, const float* betas, int32_t n_betas, int32_t steps, int32_t order,
                                  int32_t skip_type, int32_t lower_order_final, double t_start, double t_end,
                                  int32_t denoise_to_zero, dv_plan** out) {
  return dv_sampler_plan_sched(solver, DV_SCHEDULE_DISCRETE, betas, n_betas, 0.0, 0.0, steps, order, skip_type, lower_order_final,
                               t_start, t_end, denoise_to_zero, out);
}

extern "C" int dv_sampler_plan_sched(int32_t solver, int32_t schedule, const float* betas, int32_t n_betas, double beta_0,
                                     double beta_1, int32_t steps, int32_t order, int32_t skip_type,
                                     int32_t lower_order_final, double t_start, double t_end, int32_t denoise_to_zero,
                                     dv_plan** out) {
  return dv_sampler_plan_method(solver, schedule, betas, n_betas, beta_0, beta_1, DV_METHOD_MULTISTEP, steps, order, skip_type,
                                lower_order_final, t_start, t_end, denoise_to_zero, out);
}

extern "C" int dv_sampler_plan_method(int32_t solver, int32_t schedule, const float* betas, int32_t n_betas, double beta_0,
                                      double beta_1, int32_t method, int32_t steps, int32_t order, int32_t skip_type,
                                      int32_t lower_order_final, double t_start, double t_end, int32_t denoise_to_zero,
                                      dv_plan** out) {
  if (!out) return dv_fail(DV_ERR_INVALID, "dv_sampler_plan: bad argument");
  if (method < DV_METHOD_MULTISTEP || method > DV_METHOD_SINGLESTEP_FIXED) return dv_fail(DV_ERR_INVALID, "unknown method %d", method);
  const bool dpm_family = solver == DV_SOLVER_DPMPP || (solver >= DV_SOLVER_DPM && solver <= DV_SOLVER_DPM_TAYLOR);   // DPM-Solver(++) (the others: UniPC variants)
  if (schedule < DV_SCHEDULE_DISCRETE || schedule > DV_SCHEDULE_COSINE) return dv_fail(DV_ERR_INVALID, "unknown noise schedule %d", schedule);
  if (schedule == DV_SCHEDULE_DISCRETE && (!betas || n_betas < 2)) return dv_fail(DV_ERR_INVALID, "dv_sampler_plan: bad argument");
  if (schedule == DV_SCHEDULE_COSINE && dpm_family)   // (dpm_solver.py:94: 'discrete' or 'linear')
    return dv_fail(DV_ERR_INVALID, "the 'cosine' schedule exists for the UniPC solvers only");
  if (schedule == DV_SCHEDULE_LINEAR && !(beta_1 > beta_0 && beta_0 >= 0.0)) return dv_fail(DV_ERR_INVALID, "linear schedule: need 0 <= beta_0 < beta_1");
  if (solver < DV_SOLVER_DPMPP || solver > DV_SOLVER_UNIPC_VARY_NOISE) return dv_fail(DV_ERR_INVALID, "unknown solver %d", solver);
  if (dpm_family && (order < 1 || order > 3)) return dv_fail(DV_ERR_INVALID, "Solver order must be 1 or 2 or 3, got %d", order);
  if (order < 1 || order > MAXO) return dv_fail(DV_ERR_INVALID, "UniPC order must be 1..%d, got %d", MAXO, order);
  if (steps < order) return dv_fail(DV_ERR_INVALID, "steps (%d) must be >= order (%d)", steps, order);
  if (method != DV_METHOD_MULTISTEP && !dpm_family) return dv_fail(DV_ERR_INVALID, "the singlestep methods exist for DPM-Solver(++) only");
  dv_plan* p = new dv_plan();
  p->method = method;
  p->solver = solver; p->steps = steps; p->order = order; p->skip = skip_type; p->lof = lower_order_final;
  p->t_start = t_start; p->t_end = t_end; p->denoise_to_zero = denoise_to_zero ? 1 : 0;
  if (schedule == DV_SCHEDULE_DISCRETE) p->ns.init(betas, n_betas, dpm_family);
  else p->ns.init_continuous(schedule, beta_0, beta_1);
  int rc = build_plan(p);
  if (rc != DV_OK) { delete p; return rc; }
  *out = p;
  return DV_OK;
}

static void plan_drop_graph(dv_plan* p) {
  if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  p->key.u = nullptr; p->key.gen = -1; p->key.x = nullptr; p->key.cond = nullptr;
}

extern "C" void dv_plan_destroy(dv_plan* p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  plan_drop_graph(p);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  if (p->d_coefs) (void)hipFree(p->d_coefs);
  if (p->d_tin) (void)hipFree(p->d_tin);
  if (p->xp) (void)hipFree(p->xp);
  for (float* b : p->m) (void)hipFree(b);
  delete p;
}

extern "C" int dv_plan_info(const dv_plan* p, int32_t* nfe, double* t_input, double* timesteps) {
  if (!p) return dv_fail(DV_ERR_INVALID, "dv_plan_info: null plan");
  if (nfe) *nfe = (int32_t)p->t_input.size();
  if (t_input) memcpy(t_input, p->t_input.data(), p->t_input.size() * sizeof(double));
  if (timesteps) memcpy(timesteps, p->timesteps.data(), p->timesteps.size() * sizeof(double));
  return DV_OK;
}

// ... and: the number of grid points (steps + 1 for the multistep loops; outer steps + 1 for the singlestep ones) and the
// continuous time of every model evaluation (eval_times[nfe])
extern "C" int dv_plan_times(const dv_plan* p, int32_t* n_timesteps, double* eval_times) {
  if (!p) return dv_fail(DV_ERR_INVALID, "dv_plan_times: null plan");
  if (n_timesteps) *n_timesteps = (int32_t)p->timesteps.size();
  if (eval_times) memcpy(eval_times, p->eval_time.data(), p->eval_time.size() * sizeof(double));
  return DV_OK;
}

// number of events / coefficient rows, for tests
extern "C" int dv_plan_coefs(const dv_plan* p, int32_t* n_rows, float* rows8) {
  if (!p) return dv_fail(DV_ERR_INVALID, "dv_plan_coefs: null plan");
  if (n_rows) *n_rows = (int32_t)p->coefs.size();
  if (rows8) memcpy(rows8, p->coefs.data(), p->coefs.size() * 8 * sizeof(float));
  return DV_OK;
}

// events as rows of 9 int32: type, src, eval_idx, dst, coef, slot0, slot1, slot2, slot3
extern "C" int dv_plan_events(const dv_plan* p, int32_t* n_events, int32_t* ev9, int32_t* n_slots) {
  if (!p) return dv_fail(DV_ERR_INVALID, "dv_plan_events: null plan");
  if (n_events) *n_events = (int32_t)p->ev.size();
  if (n_slots) *n_slots = p->n_slots;
  if (ev9) {
    for (size_t i = 0; i < p->ev.size(); ++i) {
      const Event& e = p->ev[i];
      int32_t* r = ev9 + i * 9;
      r[0] = e.type; r[1] = e.src; r[2] = e.type == 0 ? e.eval_idx : -1; r[3] = e.dst; r[4] = e.coef;
      for (int k = 0; k < 4; ++k) r[5 + k] = e.type == 1 ? e.slots[k] : -1;
    }
  }
  return DV_OK;
}

static int plan_buffers(dv_plan* p, int64_t numel, int B) {
  if (!p->d_coefs) {
    HIPCHK(hipMalloc((void**)&p->d_coefs, p->coefs.size() * 8 * sizeof(float)));
    HIPCHK(hipMemcpy(p->d_coefs, p->coefs.data(), p->coefs.size() * 8 * sizeof(float), hipMemcpyHostToDevice));
  }
  if (p->buf_numel != numel) {
    plan_drop_graph(p);
    HIPCHK(hipDeviceSynchronize());
    if (p->xp) (void)hipFree(p->xp);
    for (float* b : p->m) (void)hipFree(b);
    p->m.clear(); p->xp = nullptr;
    HIPCHK(hipMalloc((void**)&p->xp, numel * sizeof(float)));
    for (int s = 0; s < p->n_slots; ++s) {
      float* b = nullptr;
      HIPCHK(hipMalloc((void**)&b, numel * sizeof(float)));
      p->m.push_back(b);
    }
    p->buf_numel = numel;
  }
  if (B > 0 && p->tin_B != B) {
    plan_drop_graph(p);
    HIPCHK(hipDeviceSynchronize());
    if (p->d_tin) (void)hipFree(p->d_tin);
    const size_t nfe = p->t_input.size();
    std::vector<float> h(nfe * B);
    for (size_t e = 0; e < nfe; ++e)
      for (int b = 0; b < B; ++b) h[e * B + b] = (float)p->t_input[e];
    HIPCHK(hipMalloc((void**)&p->d_tin, h.size() * sizeof(float)));
    HIPCHK(hipMemcpy(p->d_tin, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    p->tin_B = B;
  }
  return DV_OK;
}

template <typename EvalFn>
static int run_events(dv_plan* p, float* x, int64_t numel, EvalFn eval, hipStream_t st) {
  for (const Event& e : p->ev) {
    if (e.type == 0) {
      int rc = eval(e.src == 0 ? x : p->xp, e.eval_idx, p->m[e.dst]);
      if (rc != DV_OK) return rc;
    } else {
      const float* ms[4];
      for (int k = 0; k < 4; ++k) ms[k] = e.slots[k] >= 0 ? p->m[e.slots[k]] : nullptr;
      float* const dst = e.dst == 0 ? x : (e.dst == 1 ? p->xp : p->m[e.dst - 2]);   // (a history slot: in place, elementwise)
      hipError_t he = launch_lincomb(dst, e.src == 0 ? x : p->xp, ms[0], ms[1], ms[2], ms[3], p->d_coefs + (size_t)e.coef * 8,
                                     numel, st);
      if (he != hipSuccess) return dv_fail(DV_ERR_HIP, "lincomb launch failed: %s", hipGetErrorString(he));
    }
  }
  return DV_OK;
}

extern "C" int dv_sampler_run(dv_plan* p, dv_unet* u, float* x_inout, const float* cond, void* stream) {
  if (!p || !u || !x_inout) return dv_fail(DV_ERR_INVALID, "dv_sampler_run: null argument");
  int B, T, cin, cout; int64_t gen;
  if (!dv_unet_dims(u, &B, &T, &cin, &cout, &gen)) return dv_fail(DV_ERR_STATE, "dv_sampler_run: unet not prepared / cond not set");
  if (cin > cout && !cond) return dv_fail(DV_ERR_INVALID, "dv_sampler_run: cond is required (in_channels > out_channels)");
  if (int hrc = dv_unet_health(u)) return hrc;
  const int64_t numel = (int64_t)B * cout * T;
  int rc = plan_buffers(p, numel, B);
  if (rc != DV_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  // the time-embedding chain of every evaluation runs once, at the head of the loop (the timesteps of the plan are known)
  const int nfe = (int)p->t_input.size();
  auto eval = [&](hipStream_t s, bool batched) {
    return [=](const float* src, int idx, float* dst) {
      return dv_unet_enqueue(u, src, cout, cond, p->d_tin + (size_t)idx * B, dst, s, batched ? idx : -1);
    };
  };
  const char* ng = getenv("DVITS_NO_GRAPH");
  if (ng && ng[0] == '1') {
    const int tb = dv_unet_temb_all(u, p->d_tin, nfe, st);
    if (tb < 0) return tb;
    return run_events(p, x_inout, numel, eval(st, tb == 0), st);
  }

  if (!(p->exec && p->key.u == u && p->key.gen == gen && p->key.x == x_inout && p->key.cond == cond)) {
    plan_drop_graph(p);
    if (!p->cap_stream) HIPCHK(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    {   // (buffers of the batched chain are sized outside the capture)
      const int tb0 = dv_unet_temb_all(u, p->d_tin, nfe, st);
      if (tb0 < 0) return tb0;
      HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
    const int tb = dv_unet_temb_all(u, p->d_tin, nfe, p->cap_stream);
    rc = tb < 0 ? tb : run_events(p, x_inout, numel, eval(p->cap_stream, tb == 0), p->cap_stream);
    hipError_t ce = hipStreamEndCapture(p->cap_stream, &graph);
    if (rc != DV_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) return dv_fail(DV_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    hipError_t ie = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { p->exec = nullptr; return dv_fail(DV_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
    p->key.u = u; p->key.gen = gen; p->key.x = x_inout; p->key.cond = cond;
  }
  HIPCHK(hipGraphLaunch(p->exec, st));
  return DV_OK;
}

extern "C" int dv_sampler_run_custom(dv_plan* p, dv_model_fn fn, void* user, float* x_inout, int64_t numel, void* stream) {
  if (!p || !fn || !x_inout || numel <= 0) return dv_fail(DV_ERR_INVALID, "dv_sampler_run_custom: bad argument");
  int rc = plan_buffers(p, numel, 0);
  if (rc != DV_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  auto eval = [&](const float* src, int idx, float* dst) {
    int r = fn(user, src, p->t_input[idx], dst, (void*)st);
    return r == 0 ? DV_OK : dv_fail(DV_ERR_INVALID, "model callback failed with %d", r);
  };
  return run_events(p, x_inout, numel, eval, st);
}
