// K0: per-column precompute for the batched canopy-RT solvers (gfx950).
//
// Everything the reference computes once per `solve_*` call before its band loop -- K_b, mu_bar,
// the G integrals, tau_d of every layer/level, exp(-K_b lai) -- is band-independent, so it is
// computed once per column here (one 128-thread workgroup per column) and written as a small
// record into the caller's workspace; the solve kernels stage it into LDS.
//
// The reference evaluates its integrals with adaptive QUADPACK (scipy.integrate.quad) over
// arbitrary Python callables.  On device the integrals use FIXED nodes:
//   * tau_d(L) = 2 int_0^{pi/2} exp(-K_b(psi) L) sin cos dpsi   (common.py:30-37)
//     mu_bar   =   int_0^{pi/2} cos sin / G(psi) dpsi           (_solve_2s.py:32)
//       6 panels x 16-point Gauss-Legendre on psi in [0, pi/2], panel edges at decades of pi/2 - psi toward
//       pi/2 where exp(-G L / cos psi) has its essential singularity: 1 - tau_d <= 2.3e-12 relative for
//       L in [3e-4, 20], mu_bar <= 3e-13 (the reference's own quad error is up to ~3e-8 in tau_d).
//   * G_int_1 = int_0^{mu_s} G(acos m) dm,  G_int_2 = int_{mu_s}^1   (_solve_4s.py:148-149)
//       16-point Gauss-Legendre each, in psi (dm = -sin psi dpsi).
//   * '9sky': the reference's own 9 fixed angles                 (common.py:40-53)
// A CRT_G_TABLE column brings G sampled at exactly these nodes (crt_hip_quad_nodes).
#include <math.h>

#include <mutex>

#include "crt_internal.hpp"

namespace crt {

namespace {

constexpr int NQT = CRT_NQ_TAU;
constexpr int NQG = CRT_NQ_G4;
constexpr int NPAN = 6;
constexpr int NGL = 16;
// Panel edges in t = pi/2 - psi, as fractions of pi/2: decades towards psi = pi/2, where e^{-G L / cos psi} has its boundary layer (at
// cos psi ~ G L), and a split of the wide end, where 1/G of an ellipsoidal distribution with small x has its own structure at psi -> 0
// (mu_bar).  Chosen by a scan over edge sets against 30-digit quadrature, ellipsoidal-approx G with x in {0.2, 0.3, 0.96, 3}
// (DESIGN 3.2): 1 - tau_d(L) <= 2.3e-12 relative for every L in [3e-4, 20] (1.3e-10 at 1e-4), mu_bar <= 3e-13.  The geometric 4x grading
// of rounds 1-2 gave 1e-8 at L = 3e-4 -- n79 divides 1 - tau_d(dlai) by dlai, and its per-leaf-area absorption on fine ragged grids was
// the one output that saw it -- and 8e-10 in mu_bar at x = 0.2.
constexpr double PAN_EDGE[NPAN + 1] = {0.0, 1e-4, 1e-3, 1e-2, 0.1, 0.6, 1.0};
static_assert(NPAN * NGL == NQT, "tau_d rule size");
static_assert(2 * NGL == NQG, "4s rule size");

struct QuadConst {
  double psi[NQT], cs[NQT], sn[NQT];
  double w[NQT];     // plain weights in psi
  double w2sc[NQT];  // 2 w sin cos  (tau_d weights)
  double gx[NGL], gw[NGL];
  double cs9[CRT_NQ_9SKY], sn9[CRT_NQ_9SKY], sc9[CRT_NQ_9SKY];
};

__constant__ QuadConst qc;
QuadConst h_qc;
std::once_flag h_once;
bool dev_inited[64] = {};
std::mutex dev_mu;

// Gauss-Legendre nodes/weights on [-1, 1] by Newton iteration on P_n
void gauss_legendre(int n, double* x, double* w) {
  for (int i = 0; i < n; ++i) {
    double z = cos(M_PI * (i + 0.75) / (n + 0.5));
    double pp = 0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 1; j <= n; ++j) {
        double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      double dz = p1 / pp;
      z -= dz;
      if (fabs(dz) < 1e-16) break;
    }
    // ascending order
    x[n - 1 - i] = z;
    w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
}

void build_host_tables() {
  gauss_legendre(NGL, h_qc.gx, h_qc.gw);
  const double T = M_PI / 2;
  // panel edges in t = pi/2 - psi
  double edge[NPAN + 1];
  for (int k = 0; k <= NPAN; ++k) edge[k] = T * PAN_EDGE[k];
  int q = 0;
  for (int k = 0; k < NPAN; ++k) {
    const double a = edge[k], b = edge[k + 1];
    for (int i = 0; i < NGL; ++i, ++q) {
      const double t = a + (h_qc.gx[i] + 1.0) * (b - a) / 2;
      const double psi = T - t;
      h_qc.psi[q] = psi;
      // cos(pi/2 - t) = sin t: keeps full relative accuracy next to pi/2
      h_qc.cs[q] = sin(t);
      h_qc.sn[q] = cos(t);
      h_qc.w[q] = h_qc.gw[i] * (b - a) / 2;
      h_qc.w2sc[q] = 2.0 * h_qc.w[q] * h_qc.sn[q] * h_qc.cs[q];
    }
  }
  for (int i = 0; i < CRT_NQ_9SKY; ++i) {
    const double psi = (5.0 + 10.0 * i) * (M_PI / 180.0);  // math.radians(sza), common.py:46-48
    h_qc.cs9[i] = cos(psi);
    h_qc.sn9[i] = sin(psi);
    h_qc.sc9[i] = sin(psi) * cos(psi);
  }
}

// 4s nodes: interval 0 = psi in [acos(mu_s), pi/2] (m in [0, mu_s]), interval 1 = [0, acos(mu_s)]
__host__ __device__ inline void g4_interval(double mu_s, int iv, double& lo, double& hi) {
  const double ps = acos(mu_s);
  if (iv == 0) {
    lo = ps;
    hi = 1.57079632679489661923;
  } else {
    lo = 0.0;
    hi = ps;
  }
}

constexpr int K0_BLOCK = 128;
constexpr int BL_UNIF_MAX_NZ = 512;
constexpr int BL_CHUNK = 16;
static_assert(K0_BLOCK == 8 * BL_CHUNK && CRT_NQ_TAU % 8 == 0, "bl chunk reduction layout");

__device__ inline double tau_d_quad(const double* kq, double L) {
  double s = 0.0;
  for (int q = 0; q < NQT; ++q) s += qc.w2sc[q] * fexp(-kq[q] * L);
  return s;
}

// 1 - tau_d(L) = 2 int (1 - e^{-K_b(psi) L}) sin cos dpsi  (the weights 2 w sin cos integrate to one): every term by expm1, so that the
// result keeps its RELATIVE accuracy at small L, where tau_d -> 1.  n79 divides (1 - tau_d(dlai)) by dlai afterwards (_solve_n79.py:146,
// 154-155): formed as 1 - tau_d the ~3e-13 by which two quadrature rules differ in tau_d became ~2e-8 at dlai ~ 3e-4 (round 2).
__device__ inline double one_minus_tau_d_quad(const double* kq, double L) {
  double s = 0.0;
  for (int q = 0; q < NQT; ++q) s -= qc.w2sc[q] * fexpm1(-kq[q] * L);
  return s;
}

__device__ inline double tau_d_9sky(const double* k9, double L) {
  double s = 0.0;
  for (int i = 0; i < CRT_NQ_9SKY; ++i) s += fexp(-k9[i] * L) * qc.sc9[i];
  return s * (2.0 * 0.17453292519943295);  // * 2 radians(10), common.py:51
}

__device__ inline double wave_sum64(double v) {  // fixed tree order -> bitwise reproducible
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}

__global__ __launch_bounds__(K0_BLOCK) void k_colpre(ColArgs a) {
  __shared__ double kq[NQT];    // K_b(psi_q) = G/cos at the tau_d nodes
  __shared__ double pmb[NQT];   // mu_bar terms, later tau_d(dlai_mean) terms
  __shared__ double pg[NQG];    // G-integral terms
  __shared__ double k9[CRT_NQ_9SKY];
  __shared__ double sh_kb, sh_dlm, sh_dl, sh_tdu;
  __shared__ int sh_unif;
  __shared__ double xis[104];  // zq_pa: cumulative LAI of the computational interfaces
  // bl only (dynamic, so that the other schemes keep 16 workgroups per CU): the 96 quadrature terms of 16 levels, the sums
  // over groups of 12 nodes, tau_d of every level
  extern __shared__ double bl_lds[];
  double* bl_terms = bl_lds;
  double* bl_gsum = bl_terms + NQT * (BL_CHUNK + 1);
  double* bl_td = bl_gsum + 8 * BL_CHUNK;

  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nz = a.nz;
  const int kind = a.g_kind[c];
  const double param = a.g_param ? a.g_param[c] : 0.0;
  const double* tab = (kind == CRT_G_TABLE) ? a.g_table + (long long)c * CRT_NQ : nullptr;
  const double* lai = a.lai + (long long)c * nz;
  const int reclen = rec_len(a.scheme, nz);
  double* rec = a.ws + (long long)c * reclen;

  const double gden = tab ? 0.0 : G_den(kind, param);  // angle-independent part of G, once per thread
  for (int q = tid; q < NQT; q += K0_BLOCK) {
    const double g = tab ? tab[q] : G_eval(kind, param, gden, qc.cs[q], qc.sn[q]);
    kq[q] = g / qc.cs[q];
    pmb[q] = qc.w[q] * qc.cs[q] * qc.sn[q] / g;
  }
  if (a.scheme == CRT_SCHEME_4S && tid < NQG) {
    const int iv = tid / NGL, i = tid % NGL;
    double lo, hi;
    g4_interval(a.mu_s, iv, lo, hi);
    const double p = lo + (qc.gx[i] + 1.0) * (hi - lo) / 2;
    const double g = tab ? tab[NQT + tid] : G_eval(kind, param, gden, cos(p), sin(p));
    pg[tid] = qc.gw[i] * (hi - lo) / 2 * g * sin(p);
  }
  if (tid < CRT_NQ_9SKY) {
    const double g = tab ? tab[NQT + NQG + tid] : G_eval(kind, param, gden, qc.cs9[tid], qc.sn9[tid]);
    k9[tid] = g / qc.cs9[tid];
  }
  __syncthreads();

  // ---- header: wave 0, reductions by shuffles ----
  if (wave == 0) {
    const double psi = a.psi[c];
    const double cs = cos(psi), sn = sin(psi);
    const double G = tab ? a.g_at_psi[c] : G_eval(kind, param, gden, cs, sn);
    const double Kb = G / cs;
    // uniform-dlai detection (lets the solve kernels advance exponentials by recurrence)
    const double dl = (lai[0] - lai[nz - 1]) / (nz - 1);
    const double tol = 4.0 * 2.220446049250313e-16 * fabs(lai[0]);
    bool ok = dl > 0.0;
    double dsum = 0.0;  // zq: sum and count of the non-zero diff(lai)   _solve_zq.py:30,50
    double dcnt = 0.0;
    for (int j = lane; j + 1 < nz; j += 64) {
      const double d = lai[j] - lai[j + 1];
      ok = ok && fabs(d - dl) <= tol;
      if (d != 0.0) {
        dsum -= d;  // diff(lai) = lai[j+1] - lai[j]
        dcnt += 1.0;
      }
    }
    const bool unif = __all(ok);
    double mubar = 0.0, g1 = 0.0, g2 = 0.0, dlm = 0.0;
    if (a.scheme == CRT_SCHEME_2S) mubar = wave_sum64(pmb[lane] + (lane < NQT - 64 ? pmb[64 + lane] : 0.0));
    if (a.scheme == CRT_SCHEME_4S) {
      g1 = wave_sum64(lane < NGL ? pg[lane] : 0.0);
      g2 = wave_sum64(lane < NGL ? pg[NGL + lane] : 0.0);
    }
    if (a.scheme == CRT_SCHEME_ZQ) dlm = fabs(wave_sum64(dsum) / wave_sum64(dcnt));
    if (lane == 0) {
      rec[S_KB] = Kb;
      rec[S_MU] = cs;
      rec[S_G] = G;
      rec[S_MUBAR] = mubar;
      rec[S_GINT1] = g1;
      rec[S_GINT2] = g2;
      rec[S_DLM] = dlm;
      rec[S_TAUI] = 0.0;
      rec[S_TPSI] = a.scheme == CRT_SCHEME_ZQ ? fexp(-Kb * dlm) : 0.0;  // _solve_zq.py:52
      double cos2 = 0.0;
      if (a.scheme == CRT_SCHEME_2S) {
        const double cm = cos(a.mla[c] * (M_PI / 180.0));
        cos2 = cm * cm;
      }
      rec[S_COS2] = cos2;
      rec[S_LT] = lai[0];
      rec[S_INVMU] = 1.0 / cs;
      rec[S_UNIF] = unif ? 1.0 : 0.0;
      rec[S_DL] = dl;
      rec[S_M] = (double)zqpa_M(nz);
      rec[15] = 0.0;
      sh_kb = Kb;
      sh_dlm = dlm;
      sh_dl = dl;
      sh_unif = unif ? 1 : 0;
    }
  }
  __syncthreads();
  const double Kb = sh_kb;
  if (a.scheme == CRT_SCHEME_ZQ_PA) {
    // computational grid of M equal layers (_solve_zq_pa.py:94-100): tau_d(LAI/M) (:176), exp(-K_b LAI/M) (:174)
    const int M = zqpa_M(nz);
    const double Lm = lai[0] / M;
    for (int q = tid; q < NQT; q += K0_BLOCK) pmb[q] = qc.w2sc[q] * fexp(-kq[q] * Lm);
    __syncthreads();
    if (wave == 0) {
      const double t = wave_sum64(pmb[lane] + (lane < NQT - 64 ? pmb[64 + lane] : 0.0));
      if (lane == 0) {
        rec[S_TAUI] = t;
        rec[S_TPSI] = fexp(-Kb * Lm);
        // xi[i] = cumulative LAI of computational interface i from the top, built by repeated addition as np.cumsum
        // does (_solve_zq_pa.py:159): xi[0] = 0, xi[i] = xi[i-1] + Lm
        double sacc = 0.0;
        xis[0] = 0.0;
        for (int i = 1; i <= M; ++i) {
          sacc += Lm;
          xis[i] = sacc;
        }
      }
    }
    __syncthreads();
  }
  if (a.scheme == CRT_SCHEME_ZQ) {
    // tau_i = tau_d(dlai_mean), always 'quad' (_solve_zq.py:51): one node per thread, then the same tree reduction
    const double dlm = sh_dlm;
    for (int q = tid; q < NQT; q += K0_BLOCK) pmb[q] = qc.w2sc[q] * fexp(-kq[q] * dlm);
    __syncthreads();
    if (wave == 0) {
      const double t = wave_sum64(pmb[lane] + (lane < NQT - 64 ? pmb[64 + lane] : 0.0));
      if (lane == 0) rec[S_TAUI] = t;
    }
  }

  // n79 on an equal-dLAI column (S_UNIF): one layer transmittance pair (tb, td) serves every layer, so tau_d is
  // evaluated once, one node per thread, instead of once per level; the solve kernels rely on the three layer
  // vectors (tb, td, 1/(1-td)) being constant when S_UNIF is set (TriN79U, tri_schemes.hpp).
  const bool n79u = a.scheme == CRT_SCHEME_N79 && sh_unif != 0 && nz >= 3;
  if (n79u) {
    const double dl = sh_dl;
    // (sh_tdu holds 1 - tau_d: see one_minus_tau_d_quad; '9sky' has no such form -- its nine weights do not sum to one -- and is 1 - tau_d
    //  as the reference computes it)
    if (a.tau_d_method == CRT_TAU_D_9SKY) {
      if (tid == 0) sh_tdu = 1.0 - tau_d_9sky(k9, dl);
    } else {
      for (int q = tid; q < NQT; q += K0_BLOCK) pmb[q] = -qc.w2sc[q] * fexpm1(-kq[q] * dl);
      __syncthreads();
      if (wave == 0) {
        const double t = wave_sum64(pmb[lane] + (lane < NQT - 64 ? pmb[64 + lane] : 0.0));
        if (lane == 0) sh_tdu = t;
      }
    }
    __syncthreads();
  }
  // bl on an equal-dLAI column: tau_d is needed at every level L_j = (nz-1-j) dl.  Instead of 96 exponentials per level
  // (the kernel's whole cost: 5760 fp64 exps per column), thread q carries E_q^m = exp(-K_q dl)^m down the levels by one
  // multiplication per level (drift <= nz ulp).  The 96 terms of a level are summed through LDS, 16 levels at a time:
  // thread q writes its 16 terms (row stride 17: conflict-free both ways), thread (g, jj) adds the 12 nodes of group g for
  // level jj, the first 16 threads add the 8 group sums.  (A wave reduction per level -- 60 x DPP -- cost as much as the
  // exponentials it replaced.)
  const bool blu = a.scheme == CRT_SCHEME_BL && sh_unif != 0 && nz <= BL_UNIF_MAX_NZ;
  if (blu) {
    const double dl = sh_dl;
    const bool on = tid < NQT;
    const double E = on ? fexp(-kq[on ? tid : 0] * dl) : 0.0;
    const double w = on ? qc.w2sc[tid] : 0.0;
    double P = 1.0;  // level nz-1: L = 0
    for (int jtop = nz - 1; jtop >= 0; jtop -= BL_CHUNK) {  // levels jtop, jtop-1, ... (chunk position jj <-> level jtop - jj)
      if (on) {
#pragma unroll
        for (int jj = 0; jj < BL_CHUNK; ++jj) {
          bl_terms[tid * (BL_CHUNK + 1) + jj] = w * P;
          P *= E;
        }
      }
      __syncthreads();
      {
        const int jj = tid & (BL_CHUNK - 1), g = tid / BL_CHUNK;  // 128 threads = 8 groups x 16 levels
        double acc = 0.0;
        for (int q = g * (NQT / 8); q < (g + 1) * (NQT / 8); ++q) acc += bl_terms[q * (BL_CHUNK + 1) + jj];
        bl_gsum[g * BL_CHUNK + jj] = acc;
      }
      __syncthreads();
      if (tid < BL_CHUNK && jtop - tid >= 0) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += bl_gsum[g * BL_CHUNK + tid];
        bl_td[jtop - tid] = t;
      }
      // (bl_terms / bl_gsum are rewritten only after the next chunk's first barrier resp. after its second one)
      __syncthreads();
    }
  }
  double* v = rec + REC_HDR;
  for (int j = tid; j < nz; j += K0_BLOCK) {
    const double L = lai[j];
    const double ekl = fexp(-Kb * L);
    switch (a.scheme) {
      case CRT_SCHEME_ZQ:
        v[j] = ekl;
        break;
      case CRT_SCHEME_ZQ_PA: {
        const int M = zqpa_M(nz);
        auto xi = [&](int i) { return xis[i]; };
        // beam fraction seen by computational layer li = j+1 (:164-168): f_sl[li] = exp(-K_b xi[M+1-li])
        v[j] = (j < M) ? fexp(-Kb * xi(M - j)) : 0.0;
        v[nz + j] = ekl;
        // linear interpolation of the interface fluxes back to lai[j] (:357-362, np.interp semantics)
        int lo = 0;
        for (int i = 1; i < M; ++i)
          if (xi(i) <= L) lo = i;
        const double x0 = xi(lo), x1 = xi(lo + 1);
        double w = (L - x0) / (x1 - x0);
        if (L >= x1) w = 1.0;  // at or beyond the last node: np.interp returns fp[-1]
        v[2 * nz + j] = (double)(M - lo);  // interface index (0 = ground) of the node above lai[j]
        v[3 * nz + j] = w;
        break;
      }
      case CRT_SCHEME_BL:
        v[j] = L;
        v[nz + j] = ekl;
        v[2 * nz + j] = blu ? bl_td[j] : tau_d_quad(kq, L);  // _solve_bl.py:35-37
        break;
      case CRT_SCHEME_N79: {
        v[j] = ekl;  // tbcum  _solve_n79.py:46
        double omtb = 1, omtd = 1, fs = 0, isl = 0, ish = 0;  // 1 - tb, 1 - td
        if (j + 1 < nz) {
          const double Ln = lai[j + 1];
          const double dl = L - Ln;                                             // :40
          if (n79u) {
            omtb = -fexpm1(-Kb * sh_dl);
            omtd = sh_tdu;
          } else {
            omtb = -fexpm1(-Kb * dl);                                            // 1 - tb, :45
            omtd = (a.tau_d_method == CRT_TAU_D_9SKY) ? 1.0 - tau_d_9sky(k9, dl) : one_minus_tau_d_quad(kq, dl);  // 1 - td, :53
          }
          fs = fexp(-Kb * ((L + Ln) / 2));                                       // :57-58
          isl = 1.0 / (fs * dl);                                                // :154
          ish = 1.0 / ((1.0 - fs) * dl);                                        // :155
        }
        // what the level loops multiply by, formed once per column (tri_schemes.hpp, TriN79): 1 - tb, 1 - td, fs / (fs dlai), 1 / (fs dlai),
        // (1 - fs) / ((1 - fs) dlai), 1 / (1 - td)
        v[nz + j] = omtb;
        v[2 * nz + j] = omtd;
        v[3 * nz + j] = fs * isl;
        v[4 * nz + j] = isl;
        v[5 * nz + j] = (1.0 - fs) * ish;
        v[6 * nz + j] = 1.0 / omtd;  // refld = (1 - td) rho  ->  1/refld = (1/rho) * this
        break;
      }
      default:  // 2s, 4s, g77, bf
        v[j] = L;
        v[nz + j] = ekl;
        break;
    }
  }
}

// tau_d for arbitrary LAI values from K_b sampled at the library's nodes (common.py:30-87 `tau_df_fn`)
__global__ __launch_bounds__(256) void k_tau_d(const double* __restrict__ kb_nodes, const double* __restrict__ L, long long n, int method,
                                                double* __restrict__ out) {
  __shared__ double kq[NQT];
  __shared__ double k9[CRT_NQ_9SKY];
  for (int q = threadIdx.x; q < NQT; q += blockDim.x) kq[q] = kb_nodes[q];
  if (threadIdx.x < CRT_NQ_9SKY) k9[threadIdx.x] = kb_nodes[NQT + NQG + threadIdx.x];
  __syncthreads();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = method == CRT_TAU_D_9SKY ? tau_d_9sky(k9, L[i]) : tau_d_quad(kq, L[i]);
}

}  // namespace

int launch_tau_d(const double* kb_nodes, const double* L, long long n, int method, double* out, hipStream_t s) {
  const int st = init_quadrature(s);
  if (st != CRT_OK) return st;
  const long long nblk = (n + 255) / 256;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_tau_d, dim3((unsigned)nblk), dim3(256), 0, s, kb_nodes, L, n, method, out);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

void host_quad_nodes(double mu_s, double* psi_nodes) {
  std::call_once(h_once, build_host_tables);
  for (int q = 0; q < NQT; ++q) psi_nodes[q] = h_qc.psi[q];
  for (int t = 0; t < NQG; ++t) {
    double lo, hi;
    g4_interval(mu_s, t / NGL, lo, hi);
    psi_nodes[NQT + t] = lo + (h_qc.gx[t % NGL] + 1.0) * (hi - lo) / 2;
  }
  for (int i = 0; i < CRT_NQ_9SKY; ++i) psi_nodes[NQT + NQG + i] = (5.0 + 10.0 * i) * (M_PI / 180.0);
}

int init_quadrature(hipStream_t s) {
  std::call_once(h_once, build_host_tables);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CRT_ERR_LAUNCH;
  if (dev < 0 || dev >= 64) return CRT_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(dev_mu);
  if (!dev_inited[dev]) {
    // first call on this device only; not capturable into a hipGraph (documented in DESIGN.md)
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(qc), &h_qc, sizeof(QuadConst), 0, hipMemcpyHostToDevice, s) != hipSuccess)
      return CRT_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return CRT_ERR_LAUNCH;
    dev_inited[dev] = true;
  }
  return CRT_OK;
}

int launch_colpre(const ColArgs& a, hipStream_t s) {
  int st = init_quadrature(s);
  if (st != CRT_OK) return st;
  const size_t dyn = a.scheme == CRT_SCHEME_BL ? (NQT * (BL_CHUNK + 1) + 8 * BL_CHUNK + BL_UNIF_MAX_NZ) * sizeof(double) : 0;
  hipLaunchKernelGGL(k_colpre, dim3(a.ncol), dim3(K0_BLOCK), dyn, s, a);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

}  // namespace crt
