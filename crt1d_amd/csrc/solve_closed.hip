// Closed-form schemes: 2s, 4s, bl, g77, bf (gfx950).
//
// Every scheme is a small per-lane state object: init() turns one band's optical properties plus the
// column's K0 record into a handful of register-resident coefficients, level(j) evaluates the
// NARR output profiles at canopy level j from the band-independent column vectors (lai_j,
// exp(-K_b lai_j), ...) that sit in LDS (same address for the whole wave -> LDS broadcast).
// One lane owns one (column, band) and sweeps the levels.
//
// The kernels are HBM-WRITE-bound (4-7 fp64 profiles out per 5 fp64 scalars in), so the store
// pattern decides the speed.  Measured on MI355X (tools/store_bw*.hip, profiles/r01_store_patterns.md):
// lanes storing their own band directly write, per level, a 512-B/1-KiB run whose start is only
// 8-B aligned whenever a row of nb doubles is not a multiple of 128 B (nb = 300: 2400 B) --
// 3.0-3.9 TB/s; the same bytes written as 128-B-aligned contiguous chunks reach 5.5-7 TB/s
// (the L2 does not merge partial lines, not even between waves of one CU).  Three kernels share the scheme objects
// (and therefore produce identical bits):
//
//  k_pipe   (default, 4 <= nb <= 1024): ONE column per workgroup with the roles separated -- compute waves (one lane per
//           band) fill a T-level LDS tile laid out exactly like the output run ([ncol][nz][nb] with bands contiguous:
//           T consecutive levels of a column are ONE contiguous run of T*nb elements) while 1-3 STORE WAVES stream the
//           previous tile out with 16-B-per-lane non-temporal stores (whole 128-B lines; double-buffered, one LDS-only
//           barrier per tile).  Odd nb: the generic flush carries the part-line across tiles (whole lines only).
//  k_tile   (tiles too large for two buffers -- g77 / bf at 300 bands -- or several narrow columns per workgroup): the same
//           tile, all waves alternate between the level arithmetic and the flush.
//  k_direct (nb < 4, or no tile fits in LDS): lanes store their own band per level.
#include <algorithm>
#include <type_traits>

#include "crt_internal.hpp"

namespace crt {
namespace {

constexpr double PI = 3.14159265358979323846;

struct BandIn {
  double I_dr0, I_df0, r, t, s;
};

// How a scheme's level() treats its per-level exponentials.  MODE 0: the column has ragged dlai -- evaluate them at every level;
// MODE 1: uniform dlai -- advance by one multiplication per level, exact every 8th (exact_level); MODE 2: decide per lane from the K0
// flag (kernels that mix columns in a workgroup).  k_pipe owns one column per workgroup and takes the decision ONCE, outside its level
// loop: no per-level branch, no recurrence state carried through the ragged loop, and the exponentials of a level interleave freely.
// Same arithmetic in every mode -> same bits.
template <int MODE>
__device__ __forceinline__ bool exact_here(bool unif, int j) {
  if constexpr (MODE == 0) return true;
  if constexpr (MODE == 1) return exact_level(j);
  return !unif || exact_level(j);
}

template <typename TIO>
__device__ inline BandIn load_band(const SolveArgs& a, int c, int b, bool need_soil) {
  const long long i = (long long)c * a.col_stride + b;
  BandIn in;
  in.I_dr0 = ldio<TIO>(a.I_dr0, i);
  in.I_df0 = ldio<TIO>(a.I_df0, i);
  in.r = ldio<TIO>(a.leaf_r, i);
  in.t = ldio<TIO>(a.leaf_t, i);
  in.s = need_soil ? ldio<TIO>(a.soil_r, i) : 0.0;
  return in;
}

// ------------------------------------------------------------------------------------------
// 2s  Dickinson-Sellers two-stream (crt1d/solvers/_solve_2s.py:54-156)
struct Sch2s {
  static constexpr const char* NAME = "2s";
  static constexpr int NARR = 4;
  static constexpr bool SOIL = true;
  static constexpr bool HEAVY_INIT = false;
  double h;                       // diffuse extinction              :84
  double Au, Bu, Cu, Ad, Bd, Cd;  // up/dn = A e^{-KL} + B e^{-hL} + C e^{+hL}   :125-135
  double I0, invmu;
  double em, ep, qp, qm;          // e^{-hL_j}, e^{+hL_j} and their per-level factors (uniform dlai)
  bool unif;

  __device__ inline void init(const double* rec, const BandIn& in, const SolveArgs&) {
    const double K = rec[S_KB], mu = rec[S_MU], mb = rec[S_MUBAR], cos2 = rec[S_COS2], LT = rec[S_LT];
    const double al = in.r, ta = in.t, rs = in.s;
    const double om = al + ta;                                         // :65
    const double beta = 0.5 * (al + ta + (al - ta) * cos2) / om;       // :68
    const double a_s = om / 2 * (1 - mu * log((mu + 1) / mu));         // :73
    const double mbK = mb * K;
    const double beta0 = (1 + mbK) / (om * mbK) * a_s;                 // :76
    const double b = 1 - (1 - beta) * om;                              // :80
    const double c = om * beta;
    const double d = om * mbK * beta0;
    const double f = om * mbK * (1 - beta0);
    h = sqrt(b * b - c * c) / mb;
    const double sig = mbK * mbK + c * c - b * b;                      // :85
    const double u1 = b - c / rs;                                      // :87
    const double u2 = b - c * rs;
    const double u3 = f + c * rs;
    const double S1 = fexp(-h * LT);
    const double S2 = fexp(-K * LT);
    const double mh = mb * h;
    const double p1 = b + mh, p2 = b - mh, p3 = b + mbK, p4 = b - mbK;
    const double iS1 = 1.0 / S1;
    const double D1 = p1 * (u1 - mh) * iS1 - p2 * (u1 + mh) * S1;      // :96
    const double D2 = (u2 + mh) * iS1 - (u2 - mh) * S1;
    const double iD1 = 1.0 / D1, iD2 = 1.0 / D2, isig = 1.0 / sig;
    const double h1 = -d * p4 - c * f;                                 // :99
    const double h1s = h1 * isig;
    const double t1 = d - h1s * p3;
    const double t2 = d - c - h1s * (u1 + mbK);
    const double h2 = iD1 * (t1 * (u1 - mh) * iS1 - p2 * t2 * S2);
    const double h3 = -iD1 * (t1 * (u1 + mh) * S1 - p1 * t2 * S2);
    const double h4 = -f * p3 - c * d;                                 // :108 (Sellers 1996)
    const double h4s = h4 * isig;
    const double t3 = u3 - h4s * (u2 - mbK);
    const double h5 = -iD2 * (h4s * (u2 + mh) * iS1 + t3 * S2);
    const double h6 = iD2 * (h4s * (u2 - mh) * S1 + t3 * S2);
    const double h7 = c * iD1 * (u1 - mh) * iS1;
    const double h8 = -c * iD1 * (u1 + mh) * S1;
    const double h9 = iD2 * (u2 + mh) * iS1;
    const double h10 = -iD2 * (u2 - mh) * S1;                          // :120
    I0 = in.I_dr0;
    invmu = rec[S_INVMU];
    Au = in.I_dr0 * h1s;
    Bu = in.I_dr0 * h2 + in.I_df0 * h7;
    Cu = in.I_dr0 * h3 + in.I_df0 * h8;
    Ad = in.I_dr0 * h4s;
    Bd = in.I_dr0 * h5 + in.I_df0 * h9;
    Cd = in.I_dr0 * h6 + in.I_df0 * h10;
    unif = rec[S_UNIF] != 0.0;
    qp = fexp(h * rec[S_DL]);  // lai decreases with j: e^{-hL} grows by e^{+h dl} per level
    qm = fast_rcp(qp);
    em = ep = 1.0;
  }

  // levels must be visited in ascending order (both kernels sweep j = 0 .. nz-1)
  template <int MODE = 2>
  __device__ inline void level(int j, const double* rec, int nz, double (&o)[NARR]) {
    const double L = rec[REC_HDR + j], eK = rec[REC_HDR + nz + j];
    if (exact_here<MODE>(unif, j)) {
      em = fexp(-h * L);
      ep = fast_rcp(em);
    } else {
      em *= qp;
      ep *= qm;
    }
    // (explicit FMAs: the build runs with -ffp-contract=off so that no kernel variant contracts differently from another)
    const double up = __builtin_fma(Au, eK, __builtin_fma(Bu, em, Cu * ep));
    const double dn = __builtin_fma(Ad, eK, __builtin_fma(Bd, em, Cd * ep));
    const double idr = I0 * eK;                    // :150
    o[0] = idr;
    o[1] = dn;
    o[2] = up;
    o[3] = __builtin_fma(idr, invmu, 2 * (up + dn));  // :156
  }
};

// ------------------------------------------------------------------------------------------
// bl  Beer-Lambert (crt1d/solvers/_solve_bl.py:51-90)
struct SchBl {
  static constexpr const char* NAME = "bl";
  static constexpr int NARR = 4;
  static constexpr bool SOIL = false;
  static constexpr bool HEAVY_INIT = false;
  double Kg, I_dr0, I_df0, invmu, tg, qg;
  bool unif;

  __device__ inline void init(const double* rec, const BandIn& in, const SolveArgs&) {
    Kg = rec[S_KB] * sqrt(1 - (in.t + in.r));  // :58-62
    I_dr0 = in.I_dr0;
    I_df0 = in.I_df0;
    invmu = rec[S_INVMU];
    unif = rec[S_UNIF] != 0.0;
    qg = fexp(Kg * rec[S_DL]);
    tg = 1.0;
  }
  template <int MODE = 2>
  __device__ inline void level(int j, const double* rec, int nz, double (&o)[NARR]) {
    const double L = rec[REC_HDR + j], tb = rec[REC_HDR + nz + j], td = rec[REC_HDR + 2 * nz + j];
    if (exact_here<MODE>(unif, j)) tg = fexp(-Kg * L);             // :65
    else tg *= qg;
    const double idr = I_dr0 * tb;                                // :69
    const double dn = I_df0 * td + 0.5 * (I_dr0 * (tg - tb));     // :70,74,79
    o[0] = idr;
    o[1] = dn;
    o[2] = 0.0;                                                   // :87
    o[3] = __builtin_fma(idr, invmu, 2 * dn);                     // :90
  }
};

// ------------------------------------------------------------------------------------------
// g77 Goudriaan 1977 (crt1d/solvers/_solve_g77.py:48-124) and bf Bodin & Franklin
// (crt1d/solvers/_solve_bf.py:60-140): same inputs/outputs, different scattered-light terms.
template <bool BF>
struct SchG77 {
  static constexpr const char* NAME = BF ? "bf" : "g77";
  static constexpr int NARR = 7;
  static constexpr bool SOIL = true;
  static constexpr bool HEAVY_INIT = false;
  double kb, invmu, LT, kp, kd, omr, cdf, csr, ct, gnd, I_dr0, I_df0, r, t;
  double ed, ex, er;      // e^{-kd L}, second scattered-light exponential (see level()), e^{-kd (LT - L)}
  double qd, qx, qr;      // their per-level factors for uniform dlai
  // One exponential per level instead of three (bf) / two instead of three (g77): with m = LT / 2 and u = e^{-kd (L - m)},
  //   e^{-kd L} = u cm,   e^{-kd (LT - L)} = cm / u,   bf: e^{kd L - (kb + kd) LT} = cx / u,      cm = e^{-kd m}, cx = e^{-(kb + kd/2) LT}.
  // u is centred on the middle of the canopy, so 1/u cannot overflow unless kd LT / 2 > 700 (LAI > 1750: nothing the reference's own
  // exponentials survive either), and the products underflow to zero exactly where the true values do -- no branch, no second code path
  // (a guarded "evaluate each exponential by itself" path next to this one cost k_tile<g77> 32 registers and a quarter of its speed).
  double hm, cm, cx;      // kd LT / 2, e^{-kd LT / 2}, bf: e^{-(kb + kd/2) LT}
  double ikm, ikp;        // bf: 1/(kd - kb), 1/(kd + kb)
  bool unif;

  __device__ inline void init(const double* rec, const BandIn& in, const SolveArgs& a) {
    kb = rec[S_KB];
    invmu = rec[S_INVMU];
    LT = rec[S_LT];
    const double mu = rec[S_MU];
    const double A0 = rec[REC_HDR + a.nz];  // A_sl at the ground level (ekl[0])
    I_dr0 = in.I_dr0;
    I_df0 = in.I_df0;
    r = in.r;
    t = in.t;
    const double sigma = in.r + in.t;                                        // g77:57
    kp = sqrt(1 - sigma);                                                    // :59
    const double rho_c = ((1 - kp) / (1 + kp)) * (2 / (1 + 1.6 * mu));       // :66
    omr = BF ? 1.0 : (1 - rho_c);                                            // bf:84 drops (1 - rho_c)
    kd = 0.8 * sqrt(1 - sigma);                                              // :69
    cdf = kd / kp;
    csr = kd / sqrt(1 - in.r);
    ct = kd / sqrt(1 - in.t);
    // ground-level terms for the soil-reflected stream (g77:95, bf:112)
    const double ed0 = fexp(-kd * LT);
    hm = 0.5 * (kd * LT);
    cm = fexp(-hm);
    cx = BF ? fexp(-(kb + 0.5 * kd) * LT) : 0.0;
    ikm = BF ? 1.0 / (kd - kb) : 0.0;
    ikp = BF ? 1.0 / (kd + kb) : 0.0;
    const double Idf0 = in.I_df0 * omr * ed0;
    double Iscd0;
    if (BF)
      Iscd0 = in.I_dr0 * in.t * ((A0 - ed0) * ikm);                          // bf:95
    else
      Iscd0 = 0.5 * (in.I_dr0 * (1 - rho_c) * fexp(-kp * kb * LT) - in.I_dr0 * (1 - sigma) * A0);
    gnd = in.s * (in.I_dr0 * A0 + Idf0 + Iscd0);
    unif = rec[S_UNIF] != 0.0;
    const double dl = rec[S_DL];
    qd = fexp(kd * dl);                       // lai decreases with j
    qr = fast_rcp(qd);
    qx = BF ? qr : fexp(kp * kb * dl);
    ed = ex = er = 1.0;
  }
  // levels must be visited in ascending order
  template <int MODE = 2>
  __device__ inline void level(int j, const double* rec, int nz, double (&o)[NARR]) {
    const double L = rec[REC_HDR + j], Asl = rec[REC_HDR + nz + j];
    if (exact_here<MODE>(unif, j)) {
      const double u = fexp(hm - kd * L);
      const double iu = fast_rcp(u);
      ed = u * cm;
      er = cm * iu;
      ex = BF ? cx * iu : fexp(-kp * kb * L);
    } else {
      ed *= qd;
      ex *= qx;
      er *= qr;
    }
    const double Idf = I_df0 * omr * ed;                                     // g77:73 / bf:84
    double Iscd, Iscu;
    if (BF) {
      Iscd = I_dr0 * t * ((Asl - ed) * ikm);                                 // bf:95
      Iscu = I_dr0 * r * ((Asl - ex) * ikp);                                 // bf:99-103
    } else {
      const double sigma = r + t;
      const double Isc = I_dr0 * omr * ex - I_dr0 * (1 - sigma) * Asl;       // g77:84-86
      Iscd = 0.5 * Isc;
      Iscu = 0.5 * Isc;
    }
    const double Isr = gnd * er;                                             // g77:95
    const double common = cdf * Idf + csr * Iscu + ct * Iscd;
    const double ash = (1 - Asl) * common;                                   // :99-101
    const double asl = Asl * (common + kb * I_dr0);                          // :106-111
    const double idr = I_dr0 * Asl;                                          // :77
    const double dn = Iscd + Idf;                                            // :115
    const double up = Iscu + Isr;                                            // :116
    o[0] = idr;
    o[1] = dn;
    o[2] = up;
    o[3] = __builtin_fma(idr, invmu, 2 * (up + dn));                         // :122
    o[4] = asl;
    o[5] = ash;
    o[6] = asl + ash;
  }
};

// ------------------------------------------------------------------------------------------
// 4s  Tian et al. (2007) four-stream (crt1d/solvers/_solve_4s.py:161-290).
//
// The reference integrates  y' = A y + g e^{-kappa x}  (y = [R2d, R1d, R1u, R2u], x = cumulative LAI)
// twice per band with scipy.integrate.solve_bvp(tol=1e-6).  A is constant and mirror-symmetric,
// so with yd = [R2d, R1d], yu = [R2u, R1u], p = yd + yu, m = yd - yu:
//     p' = -M Kd m,     m' = M (2S - Kd) p + 2 gd e^{-kappa x}
//     p'' = B p - 2 M Kd gd e^{-kappa x},     B = M Kd M (Kd - 2S)      (2x2)
// with M = diag(1/mu2, 1/mu1), Kd = diag(G_int_2, G_int_1), S = [[alpha, beta], [beta, gamma]].
// B has real eigenvalues l1 > 0 and l2 (l2 < 0 for strongly scattering leaves -> oscillatory
// modes, which the reference's BVP solver integrates just the same).  The direct and diffuse
// problems are linear in their data, so they are solved once, summed.  The four boundary
// conditions (:110-116, :128-138) give a 4x4 linear system solved in registers by Gaussian
// elimination with branch-free partial pivoting.
__device__ inline void solve4(double (&A)[4][5]) {
  // partial pivoting by conditional row swaps (no dynamic register indexing)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int r = k + 1; r < 4; ++r) {
      const bool sw = fabs(A[r][k]) > fabs(A[k][k]);
#pragma unroll
      for (int cidx = k; cidx < 5; ++cidx) {
        const double x = A[k][cidx], y = A[r][cidx];
        A[k][cidx] = sw ? y : x;
        A[r][cidx] = sw ? x : y;
      }
    }
    const double ip = 1.0 / A[k][k];
#pragma unroll
    for (int r = k + 1; r < 4; ++r) {
      const double fct = A[r][k] * ip;
#pragma unroll
      for (int cidx = k + 1; cidx < 5; ++cidx) A[r][cidx] -= fct * A[k][cidx];
    }
  }
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    double sacc = A[k][4];
#pragma unroll
    for (int cidx = k + 1; cidx < 4; ++cidx) sacc -= A[k][cidx] * A[cidx][4];
    A[k][4] = sacc / A[k][k];
  }
}

struct Sch4s {
  static constexpr const char* NAME = "4s";
  static constexpr int NARR = 4;
  static constexpr bool SOIL = true;
  static constexpr bool HEAVY_INIT = true;  // eigen-decomposition + 4x4 solve per band before the first level: wants more workgroups per CU
  double lam1, lam2;  // sqrt(|l1|), sqrt(|l2|)
  bool osc;           // l2 < 0
  double d[5], u[5];  // I_df_d / I_df_u = sum_k coef[k] * phi_k(x), phi = {E1, F1, phi3, phi4, e^{-kappa x}}
  double I0, invmu, LT;
  double p1, p2, p3, p4;  // basis functions at the current level
  double q1p, q1m, q2a, q2b;  // per-level factors for uniform dlai (q2a/q2b: e^{+-lam2 dl}, or cos/sin(lam2 dl))
  double E1L, E2L;            // e^{-lam1 LT}, e^{-lam2 LT}: phi2 = E1L / phi1, phi4 = E2L / phi3 (two exponentials per level, not four)
  bool unif, direct;          // direct: lam LT so large that 1/phi could overflow -> every exponential evaluated by itself

  __device__ inline void init(const double* rec, const BandIn& in, const SolveArgs& a) {
    const double mu_s = a.mu_s;
    const double G = rec[S_G], mu0 = rec[S_MU], kap = rec[S_KB], G1 = rec[S_GINT1], G2 = rec[S_GINT2];
    LT = rec[S_LT];
    invmu = rec[S_INVMU];
    const double om = in.r + in.t, rho = in.s;
    const double R_dr0 = in.I_dr0 / (PI * mu0);  // :169
    const double R_df0 = in.I_df0 / PI;          // :170
    const double mu1 = 0.5 * mu_s * mu_s;        // :188
    const double mu2 = 0.5 * (1 - mu_s * mu_s);
    const double al = 0.5 * om * (1 - mu_s) * G2;  // :190
    const double be = 0.5 * om * (1 - mu_s) * G1;  // :192
    const double ga = 0.5 * om * mu_s * G1;        // :194
    const double e1 = 0.25 * om * R_dr0 * mu_s;        // :196
    const double e2 = 0.25 * om * R_dr0 * (1 - mu_s);  // :198
    // index 0 <-> stream "2" (mu2, G2), index 1 <-> stream "1"
    const double gd0 = G * e2 / mu2, gd1 = G * e1 / mu1;  // forcing of yd' (:83,:87)
    const double w0 = G2 / (mu2 * mu2), w1 = G1 / (mu1 * mu1);
    const double ba = w0 * (G2 - 2 * al), bb = -2 * w0 * be, bc = -2 * w1 * be, bd = w1 * (G1 - 2 * ga);
    const double disc = sqrt((ba - bd) * (ba - bd) + 4 * bb * bc);
    const double l1 = 0.5 * (ba + bd + disc);
    const double l2 = (ba * bd - bb * bc) / l1;
    // eigenvectors, each taken from the better-conditioned row
    double v1x, v1y, v2x, v2y;
    if (disc == 0.0) {
      v1x = 1; v1y = 0; v2x = 0; v2y = 1;
    } else if (ba >= bd) {
      v1x = l1 - bd; v1y = bc;
      v2x = bb;      v2y = l2 - ba;
    } else {
      v1x = bb;      v1y = l1 - ba;
      v2x = l2 - bd; v2y = bc;
    }
    const double n0 = mu2 / G2, n1 = mu1 / G1;  // N = (M Kd)^-1
    // particular solution p = P e^{-kappa x}:  (kappa^2 I - B) P = -2 M Kd gd
    const double k2 = kap * kap;
    const double r0 = -2 * (G2 / mu2) * gd0, r1 = -2 * (G1 / mu1) * gd1;
    const double m00 = k2 - ba, m01 = -bb, m10 = -bc, m11 = k2 - bd;
    const double idet = 1.0 / (m00 * m11 - m01 * m10);
    const double P0 = (r0 * m11 - m01 * r1) * idet, P1 = (m00 * r1 - m10 * r0) * idet;
    const double Pd0 = 0.5 * (P0 + kap * n0 * P0), Pd1 = 0.5 * (P1 + kap * n1 * P1);
    const double Pu0 = 0.5 * (P0 - kap * n0 * P0), Pu1 = 0.5 * (P1 - kap * n1 * P1);

    I0 = in.I_dr0;
    osc = l2 < 0.0;
    lam1 = sqrt(l1);
    lam2 = sqrt(fabs(l2));
    // basis functions: phi1 = e^{-lam1 x}, phi2 = e^{-lam1 (L - x)},
    //   real:  phi3 = e^{-lam2 x}, phi4 = e^{-lam2 (L - x)}      osc: phi3 = cos(lam2 x), phi4 = sin(lam2 x)
    const double U1x = 0.5 * (v1x + lam1 * n0 * v1x), U1y = 0.5 * (v1y + lam1 * n1 * v1y);
    const double W1x = 0.5 * (v1x - lam1 * n0 * v1x), W1y = 0.5 * (v1y - lam1 * n1 * v1y);
    E1L = fexp(-lam1 * LT);
    E2L = 0.0;
    direct = lam1 * LT > 600.0 || (!osc && lam2 * LT > 600.0);
    // yd, yu contribution of each basis function: [basis][x=0 | x=L][yd | yu][component]
    double Y[4][2][2][2];
    Y[0][0][0][0] = U1x;       Y[0][0][0][1] = U1y;       Y[0][0][1][0] = W1x;       Y[0][0][1][1] = W1y;
    Y[0][1][0][0] = U1x * E1L; Y[0][1][0][1] = U1y * E1L; Y[0][1][1][0] = W1x * E1L; Y[0][1][1][1] = W1y * E1L;
    Y[1][0][0][0] = W1x * E1L; Y[1][0][0][1] = W1y * E1L; Y[1][0][1][0] = U1x * E1L; Y[1][0][1][1] = U1y * E1L;
    Y[1][1][0][0] = W1x;       Y[1][1][0][1] = W1y;       Y[1][1][1][0] = U1x;       Y[1][1][1][1] = U1y;
    const double hx = 0.5 * v2x, hy = 0.5 * v2y;
    const double qx = 0.5 * lam2 * n0 * v2x, qy = 0.5 * lam2 * n1 * v2y;  // (lam2/2) N v2
    if (osc) {
      double sL, cL;
      sincos(lam2 * LT, &sL, &cL);
      // phi3 = cos: p = v c -> m = lam N v s : yd = h c + q s, yu = h c - q s
      Y[2][0][0][0] = hx; Y[2][0][0][1] = hy; Y[2][0][1][0] = hx; Y[2][0][1][1] = hy;
      Y[2][1][0][0] = hx * cL + qx * sL; Y[2][1][0][1] = hy * cL + qy * sL;
      Y[2][1][1][0] = hx * cL - qx * sL; Y[2][1][1][1] = hy * cL - qy * sL;
      // phi4 = sin: p = v s -> m = -lam N v c : yd = h s - q c, yu = h s + q c
      Y[3][0][0][0] = -qx; Y[3][0][0][1] = -qy; Y[3][0][1][0] = qx; Y[3][0][1][1] = qy;
      Y[3][1][0][0] = hx * sL - qx * cL; Y[3][1][0][1] = hy * sL - qy * cL;
      Y[3][1][1][0] = hx * sL + qx * cL; Y[3][1][1][1] = hy * sL + qy * cL;
    } else {
      E2L = fexp(-lam2 * LT);
      const double U2x = hx + qx, U2y = hy + qy, W2x = hx - qx, W2y = hy - qy;
      Y[2][0][0][0] = U2x;       Y[2][0][0][1] = U2y;       Y[2][0][1][0] = W2x;       Y[2][0][1][1] = W2y;
      Y[2][1][0][0] = U2x * E2L; Y[2][1][0][1] = U2y * E2L; Y[2][1][1][0] = W2x * E2L; Y[2][1][1][1] = W2y * E2L;
      Y[3][0][0][0] = W2x * E2L; Y[3][0][0][1] = W2y * E2L; Y[3][0][1][0] = U2x * E2L; Y[3][0][1][1] = U2y * E2L;
      Y[3][1][0][0] = W2x;       Y[3][1][0][1] = W2y;       Y[3][1][1][0] = U2x;       Y[3][1][1][1] = U2y;
    }
    // boundary conditions
    //   top    (x = 0): yd = R_df0 [1, 1]                                   (:110-116, both problems summed)
    //   bottom (x = L): yu - rho (2 muv . yd + mu0 R_dr0 e^{-kappa L}) [1, 1] = 0     (:128-138)
    const double eKL = fexp(-kap * LT);
    double A[4][5];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      A[0][q] = Y[q][0][0][0];
      A[1][q] = Y[q][0][0][1];
      const double tq = 2 * rho * (mu2 * Y[q][1][0][0] + mu1 * Y[q][1][0][1]);
      A[2][q] = Y[q][1][1][0] - tq;
      A[3][q] = Y[q][1][1][1] - tq;
    }
    A[0][4] = R_df0 - Pd0;
    A[1][4] = R_df0 - Pd1;
    const double tP = 2 * rho * (mu2 * Pd0 + mu1 * Pd1);
    A[2][4] = eKL * (tP + rho * mu0 * R_dr0 - Pu0);
    A[3][4] = eKL * (tP + rho * mu0 * R_dr0 - Pu1);
    solve4(A);
    // irradiance coefficients: I_df_d = 2 pi muv . yd, I_df_u = 2 pi muv . yu   (:246-249, :280-281)
    const double tp = 2 * PI;
    const double mU1 = tp * (mu2 * U1x + mu1 * U1y), mW1 = tp * (mu2 * W1x + mu1 * W1y);
    d[0] = mU1 * A[0][4];
    u[0] = mW1 * A[0][4];
    d[1] = mW1 * A[1][4];
    u[1] = mU1 * A[1][4];
    const double mh = tp * (mu2 * hx + mu1 * hy), mq = tp * (mu2 * qx + mu1 * qy);
    if (osc) {
      // yd = a3 (h c + q s) + a4 (h s - q c);  yu = a3 (h c - q s) + a4 (h s + q c)
      d[2] = mh * A[2][4] - mq * A[3][4];
      d[3] = mq * A[2][4] + mh * A[3][4];
      u[2] = mh * A[2][4] + mq * A[3][4];
      u[3] = -mq * A[2][4] + mh * A[3][4];
    } else {
      d[2] = (mh + mq) * A[2][4];
      u[2] = (mh - mq) * A[2][4];
      d[3] = (mh - mq) * A[3][4];
      u[3] = (mh + mq) * A[3][4];
    }
    d[4] = tp * (mu2 * Pd0 + mu1 * Pd1);
    u[4] = tp * (mu2 * Pu0 + mu1 * Pu1);
    unif = rec[S_UNIF] != 0.0;
    const double dl = rec[S_DL];
    q1p = fexp(lam1 * dl);  // x = lai decreases with j
    q1m = fast_rcp(q1p);
    if (osc) {
      sincos(lam2 * dl, &q2b, &q2a);  // q2a = cos, q2b = sin
    } else {
      q2a = fexp(lam2 * dl);
      q2b = fast_rcp(q2a);
    }
    p1 = p2 = p3 = p4 = 1.0;
  }

  // levels must be visited in ascending order
  template <int MODE = 2>
  __device__ inline void level(int j, const double* rec, int nz, double (&o)[NARR]) {
    const double x = rec[REC_HDR + j], eK = rec[REC_HDR + nz + j];
    if (exact_here<MODE>(unif, j)) {
      // both exponentials for every lane, side by side (two independent dependency chains); the few oscillatory lanes (0.3 % of the
      // synthetic bands, one wave in eight holds one) overwrite the second with sin / cos afterwards
      p1 = fexp(-lam1 * x);
      p3 = fexp(-lam2 * x);
      if (direct) {
        p2 = fexp(-lam1 * (LT - x));
        p4 = fexp(-lam2 * (LT - x));
      } else {
        p2 = E1L * fast_rcp(p1);
        p4 = E2L * fast_rcp(p3);
      }
      if (osc) fast_sincos(lam2 * x, p4, p3);
    } else {
      p1 *= q1p;
      p2 *= q1m;
      if (osc) {  // rotate (cos, sin)(lam2 x) by -lam2 dl
        const double c = p3 * q2a + p4 * q2b;
        p4 = p4 * q2a - p3 * q2b;
        p3 = c;
      } else {
        p3 *= q2a;
        p4 *= q2b;
      }
    }
    const double dn = __builtin_fma(d[0], p1, __builtin_fma(d[1], p2, __builtin_fma(d[2], p3, __builtin_fma(d[3], p4, d[4] * eK))));
    const double up = __builtin_fma(u[0], p1, __builtin_fma(u[1], p2, __builtin_fma(u[2], p3, __builtin_fma(u[3], p4, u[4] * eK))));
    const double idr = I0 * eK;                      // :284
    o[0] = idr;
    o[1] = dn;
    o[2] = up;
    o[3] = __builtin_fma(idr, invmu, 2 * (up + dn));  // :290
  }
};

// ------------------------------------------------------------------------------------------
// k_direct: lanes store their own band(s) per level (fallback for small nb)
constexpr int DBLOCK = 256;

template <class S, typename TIO, int VEC, bool USE_LDS>
__global__ __launch_bounds__(DBLOCK) void k_direct(SolveArgs a) {
  extern __shared__ double lds[];
  const Item it = locate<DBLOCK, VEC>(a.ncol, a.nb);
  const double* rec = stage_records<DBLOCK, VEC, USE_LDS>(a, it, lds);
  if (!it.active) return;
  S st[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) st[v].init(rec, load_band<TIO>(a, it.c, it.b + v, S::SOIL), a);
  const int nz = a.nz;
  long long o = ((long long)it.c * nz) * a.nb + it.b;
  for (int j = 0; j < nz; ++j, o += a.nb) {
    double val[VEC][S::NARR];
#pragma unroll
    for (int v = 0; v < VEC; ++v) st[v].level(j, rec, nz, val[v]);
#pragma unroll
    for (int k = 0; k < S::NARR; ++k) {
      double pk[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) pk[v] = val[v][k];
      store_stream<TIO, VEC>(outp<TIO>(a.o[k]) + o, pk);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_tile: workgroup = CB whole columns; T levels staged in LDS, flushed as contiguous aligned runs
struct TileCfg {
  int CB;        // columns per workgroup
  int T;         // levels per LDS tile
  int rec_dbl;   // doubles reserved for the staged column records (CB * reclen, rounded up to even)
  int flags;     // bit0: full __syncthreads() barriers (A/B aid)
};


template <class S, typename TIO, int MAXT, bool FUSED>
__global__ __launch_bounds__(MAXT) void k_tile(SolveArgs a, TileCfg cfg) {
  constexpr int VW = 16 / (int)sizeof(TIO);  // elements per 16-B store: 2 doubles / 4 floats
  typedef TIO vt __attribute__((ext_vector_type(VW)));
  extern __shared__ double lds[];
  const int nb = a.nb, nz = a.nz, CB = cfg.CB, T = cfg.T;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c0 = blockIdx.x * CB;
  const int ncb = min(CB, a.ncol - c0);  // columns actually present in this workgroup
  const int cl = tid / nb, b = tid - cl * nb;
  const bool active = cl < ncb;
  BandIn bin = {};  // requested together with the records (as in k_pipe): one round trip before the first level instead of two
  if (active) bin = load_band<TIO>(a, c0 + cl, b, S::SOIL);
  // stage the column records (contiguous in the workspace)
  {
    const double* src = a.ws + (long long)c0 * a.reclen;
    const int n = ncb * a.reclen;
    for (int i = tid; i < n; i += nthr) lds[i] = src[i];
  }
  __syncthreads();
  TIO* tile = reinterpret_cast<TIO*>(lds + cfg.rec_dbl);  // [NARR][CB][T][nb]
  const double* rec = lds + (active ? cl : 0) * a.reclen;
  S st;
  if (active) st.init(rec, bin, a);
  const int colrun = T * nb;  // elements per (array, column) slot of the tile
  // FUSED flush (one column per workgroup, nb a multiple of the 16-B vector width, 16-B aligned outputs): a vector never
  // straddles a row, so thread -> (row offset, vector within the row) is fixed for the whole kernel and all NARR arrays
  // are flushed in one pass with no per-array address set-up.  Rows of a tile are adjacent in memory, so consecutive
  // threads still write consecutive 16-B words (a wave store = 1 KiB, line aligned).
  const int nbv = nb / VW;
  const int f_toff = FUSED ? tid / nbv : 0;
  const int f_p = FUSED ? tid - f_toff * nbv : 0;
  const int f_rpi = FUSED ? nthr / nbv : 1;

  for (int j0 = 0; j0 < nz; j0 += T) {
    const int Tc = min(T, nz - j0);
    if (active) {
      TIO* tl = tile + cl * colrun + b;
      for (int t = 0; t < Tc; ++t) {
        double val[S::NARR];
        st.level(j0 + t, rec, nz, val);
#pragma unroll
        for (int k = 0; k < S::NARR; ++k) tl[k * (CB * colrun) + t * nb] = (TIO)val[k];
      }
    }
    if (cfg.flags & 1) __syncthreads(); else lds_barrier();
    if constexpr (FUSED) {
      if (f_toff < f_rpi) {
        const vt* tv = reinterpret_cast<const vt*>(tile);
        for (int t = f_toff; t < Tc; t += f_rpi) {
          const long long go = ((long long)c0 * nz + j0 + t) * nbv + f_p;
          vt v[S::NARR];
#pragma unroll
          for (int k = 0; k < S::NARR; ++k) v[k] = tv[(k * colrun + t * nb) / VW + f_p];
#pragma unroll
          for (int k = 0; k < S::NARR; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<vt*>(a.o[k]) + go);
        }
      }
    } else {
    // flush: per (array, column) one contiguous run of Tc * nb elements
    const int n = Tc * nb;
    for (int k = 0; k < S::NARR; ++k) {
      for (int q = 0; q < ncb; ++q) {
        TIO* g = outp<TIO>(a.o[k]) + ((long long)(c0 + q) * nz + j0) * nb;
        const TIO* s = tile + k * (CB * colrun) + q * colrun;
        // leading elements up to the next 16-B boundary of the destination
        int mis = (int)(((16 - (reinterpret_cast<uintptr_t>(g) & 15)) & 15) / sizeof(TIO));
        if (mis > n) mis = n;
        const int nvec = (n - mis) / VW;
        const int tail = (n - mis) - nvec * VW;
        vt* gv = reinterpret_cast<vt*>(g + mis);
        if ((reinterpret_cast<uintptr_t>(s + mis) & 15) == 0) {
          const vt* sv = reinterpret_cast<const vt*>(s + mis);
          for (int i = tid; i < nvec; i += nthr) gv[i] = sv[i];
        } else {
          for (int i = tid; i < nvec; i += nthr) {
            vt v;
#pragma unroll
            for (int w = 0; w < VW; ++w) v[w] = s[mis + VW * i + w];
            gv[i] = v;
          }
        }
        if (tid < mis) g[tid] = s[tid];
        if (tid < tail) g[mis + nvec * VW + tid] = s[mis + nvec * VW + tid];
      }
    }
    }
    if (cfg.flags & 1) __syncthreads(); else lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------
// k_pipe: k_tile's fused case (one column per workgroup) with the roles separated.  In k_tile all waves alternate between
// the level arithmetic and the flush, and every workgroup on the chip follows the same schedule, so the store queues run
// dry while the level arithmetic executes (time ~ store time + compute time).  Here `ncomp` compute threads (one per band)
// fill tile g in one LDS buffer while the remaining "store waves" stream tile g-1 out of the other buffer;
// one LDS-only barrier per tile (protocol: see k_tri_pipe in tri_tile_impl.hpp).
struct PipeTileCfg {
  int ncomp;    // compute threads (multiple of 64)
  int T;        // levels per tile
  int rec_dbl;  // doubles reserved for the staged column record
};

template <class S, typename TIO, int MAXT, bool FUSED>
__global__ __launch_bounds__(MAXT) void k_pipe(SolveArgs a, PipeTileCfg cfg) {
  constexpr int VW = 16 / (int)sizeof(TIO);
  typedef TIO vt __attribute__((ext_vector_type(VW)));
  extern __shared__ double lds[];
  const int nb = a.nb, nz = a.nz, T = cfg.T;
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
#ifdef CRT_STAMP
  if ((threadIdx.x & 63) == 0) {  // where the hardware put this wave (SIMD id = HW_ID[5:4]; LDS base / size in 256-B granules)
    double* dbg = const_cast<double*>(a.ws) + (long long)c * a.reclen + 100 + 3 * (threadIdx.x >> 6);
    dbg[0] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, all 32 bits
    dbg[1] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 6);   // HW_REG_LDS_ALLOC
    dbg[2] = (double)(tid >> 6);
  }
#endif
  // the band's spectra are requested together with the column record (not after the barrier that publishes it): one round trip
  // instead of two before the first level can be formed
  BandIn bin = {};
  if (tid < cfg.ncomp) bin = load_band<TIO>(a, c, tid < nb ? tid : 0, S::SOIL);
  {
    const double* src = a.ws + (long long)c * a.reclen;
    for (int i = threadIdx.x; i < a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  const double* rec = lds;
  // [2][NARR][HEAD + T * nb]; HEAD = one 128-B line of head-room in front of every array's tile (generic flush only, see below)
  constexpr int HEAD = FUSED ? 0 : 128 / (int)sizeof(TIO);
  TIO* tile = reinterpret_cast<TIO*>(lds + cfg.rec_dbl) + HEAD;
  const int colrun = HEAD + T * nb, bufrun = S::NARR * colrun;
  if (tid < cfg.ncomp) {
    // ---- compute role ----
    const bool active = tid < nb;
    const int b = active ? tid : 0;
#ifdef CRT_STAMP
    // diagnostic build only (tools/stamp_timeline.py): wall_clock64 stamps of compute wave 0 / store wave 0 of every workgroup, written
    // over the column's own K0 record in the workspace (already staged in LDS; K0 rewrites it) -- never into an output
    double* stamp = const_cast<double*>(a.ws) + (long long)c * a.reclen;
    int nstamp = 0;
    const bool stamper = tid == 0;
#define STAMP() do { if (stamper && nstamp < 60) stamp[nstamp++] = (double)wall_clock64(); } while (0)
    STAMP();
#else
#define STAMP() do {} while (0)
#endif
    S st;
    st.init(rec, bin, a);
    STAMP();
    // (the K0 flag is the same for every lane of the workgroup: read through a scalar register, so that the branch is a scalar one)
    const bool unif = __builtin_amdgcn_readfirstlane((int)(rec[S_UNIF] != 0.0)) != 0;
    auto sweep = [&](auto mode) {
      constexpr int MODE = decltype(mode)::value;
      int buf = 0;
      for (int j0 = 0; j0 < nz; j0 += T) {
        const int Tc = min(T, nz - j0);
        TIO* tl = tile + buf * bufrun + b;
        for (int t = 0; t < Tc; ++t) {
          double val[S::NARR];
          st.template level<MODE>(j0 + t, rec, nz, val);
          if (active) {
#pragma unroll
            for (int k = 0; k < S::NARR; ++k) tl[k * colrun + t * nb] = (TIO)val[k];
          }
        }
        STAMP();
        lds_barrier();  // tile complete: hand it to the store waves
        STAMP();
        buf ^= 1;
      }
    };
    if (unif)
      sweep(std::integral_constant<int, 1>{});
    else
      sweep(std::integral_constant<int, 0>{});
  } else if constexpr (!FUSED) {
    // ---- store role, any nb / alignment.  A column is one flat run of nz * nb elements per array, so a round need not stop at a
    // level boundary: it writes every 128-B line that is complete so far.  The part-line behind the last complete line is
    // copied into the head-room in front of the OTHER buffer's tile (memory the compute waves never touch), where the next
    // round finds it directly in front of its own data: every round flushes one contiguous LDS range into whole lines, and
    // only the first and the last line of a column can be partial (2 of ~400 at nb = 107, instead of 2 per tile).
    constexpr int LINE = 128 / (int)sizeof(TIO);
    const int sid = tid - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
    const long long col0 = (long long)c * nz * nb;
    int F[S::NARR];  // elements of this column already flushed, per array (the arrays may be aligned differently)
#pragma unroll
    for (int k = 0; k < S::NARR; ++k) F[k] = 0;
    int buf = 0;
    for (int j0 = 0; j0 < nz; j0 += T) {
      const int Tc = min(T, nz - j0);
      lds_barrier();  // tile `buf` is complete
      const int t0e = j0 * nb;          // first element of this tile
      const int avail = t0e + Tc * nb;  // elements of the column computed so far
      const bool last = j0 + Tc >= nz;
#pragma unroll
      for (int k = 0; k < S::NARR; ++k) {
        TIO* gk = outp<TIO>(a.o[k]) + col0;
        const int f = F[k];
        const TIO* src = tile + buf * bufrun + k * colrun - (t0e - f);  // element e of the column sits at src[e - f]
        int end = avail;
        if (!last) end -= (int)((reinterpret_cast<uintptr_t>(gk + avail) / sizeof(TIO)) % LINE);  // end of the last complete line
        if (end < f) end = f;
        const int n = end - f;
        TIO* g = gk + f;
        int mis = (int)(((16 - (reinterpret_cast<uintptr_t>(g) & 15)) & 15) / sizeof(TIO));  // non-zero at the column start only
        if (mis > n) mis = n;
        const int nvec = (n - mis) / VW;
        const int tail = (n - mis) - nvec * VW;  // non-zero at the column end only
        vt* gv = reinterpret_cast<vt*>(g + mis);
        for (int i = sid; i < nvec; i += nst) {
          vt v;
#pragma unroll
          for (int w = 0; w < VW; ++w) v[w] = src[mis + VW * i + w];
          __builtin_nontemporal_store(v, gv + i);
        }
        if (sid < mis) __builtin_nontemporal_store(src[sid], g + sid);
        if (sid < tail) __builtin_nontemporal_store(src[mis + nvec * VW + sid], g + mis + nvec * VW + sid);
        // park the part-line [end, avail) in front of the other buffer's tile of this array
        const int npark = avail - end;
        if (sid < npark) tile[(buf ^ 1) * bufrun + k * colrun - npark + sid] = src[n + sid];
        F[k] = end;
      }
      buf ^= 1;
    }
  } else {
    // ---- store role: thread -> (row, 16-B vector) walk over the T x nbv vectors of a tile ----
    const int nbv = nb / VW;
    const int sid = tid - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
    const int dt = nst / nbv, dp = nst - dt * nbv;
    const int t0 = sid / nbv, p0 = sid - t0 * nbv;
#ifdef CRT_STAMP
    double* sstamp = const_cast<double*>(a.ws) + (long long)c * a.reclen + 64;
    int nsstamp = 0;
#define SSTAMP() do { if (sid == 0 && nsstamp < 60) sstamp[nsstamp++] = (double)wall_clock64(); } while (0)
#else
#define SSTAMP() do {} while (0)
#endif
    SSTAMP();
    int buf = 0;
    for (int j0 = 0; j0 < nz; j0 += T) {
      const int Tc = min(T, nz - j0);
      lds_barrier();  // tile `buf` is complete
      SSTAMP();
      const vt* tv = reinterpret_cast<const vt*>(tile + buf * bufrun);
      int t = t0, p = p0;
      while (t < Tc) {
        const long long go = ((long long)c * nz + j0 + t) * nbv + p;
        vt v[S::NARR];
#pragma unroll
        for (int k = 0; k < S::NARR; ++k) v[k] = tv[(k * colrun + t * nb) / VW + p];
#pragma unroll
        for (int k = 0; k < S::NARR; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<vt*>(a.o[k]) + go);
        p += dp;
        t += dt;
        if (p >= nbv) {
          p -= nbv;
          ++t;
        }
      }
      SSTAMP();
      buf ^= 1;
    }
#ifdef CRT_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SSTAMP();  // every store of the column acknowledged
#endif
  }
}

// ------------------------------------------------------------------------------------------
// k_pipe_pack: k_pipe for spectra so narrow that one column leaves most of a compute wave idle (nb <= 32: 12 bands use 12 of 64 lanes).
// The compute threads (ncw waves) own cpw = 64 ncw / nb consecutive columns, thread -> (column in pack, band) = (tid / nb, tid % nb) -- a
// column may straddle two waves, nothing in the compute role is wave-level; every lane reads its own column's record (LDS, a few distinct
// addresses per wave) and decides uniform / ragged for itself (level<2>).  The tile is [buffer][column][array][T x nb]: the
// T-level run of one (column, array) is contiguous in the output, so the store wave walks (column, 16-B vector) with an incremental
// index and issues one store per array.  Round 3: 2s at 4e5 x 12 x 60 -- see DESIGN 8a.9.
struct PackCfg {
  int ncw;  // compute waves
  int cpw;  // columns per workgroup = 64 ncw / nb
  int T;    // levels per tile
};

template <class S, typename TIO, int MAXT>
__global__ __launch_bounds__(MAXT) void k_pipe_pack(SolveArgs a, PackCfg cfg) {
  constexpr int VW = 16 / (int)sizeof(TIO);
  typedef TIO vt __attribute__((ext_vector_type(VW)));
  extern __shared__ double lds[];
  const int nb = a.nb, nz = a.nz, T = cfg.T;
  const int tid = threadIdx.x;
  const int cpw = cfg.cpw;
  const int c0 = blockIdx.x * cpw;
  const int ncol_here = min(cpw, a.ncol - c0);
  const int ncomp = 64 * cfg.ncw;
  const int cl = tid / nb;  // column of the pack
  const bool active = tid < ncomp && cl < ncol_here;  // (ncol_here <= cpw)
  const int b = active ? tid - cl * nb : 0;
  BandIn bin = {};
  if (tid < ncomp) bin = load_band<TIO>(a, c0 + (active ? cl : 0), b, S::SOIL);
  {  // the records of consecutive columns are consecutive in the workspace
    const double* src = a.ws + (long long)c0 * a.reclen;
    for (int i = tid; i < ncol_here * a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  const int rec_dbl = (cpw * a.reclen + 1) & ~1;
  TIO* tile = reinterpret_cast<TIO*>(lds + rec_dbl);
  const int arrrun = T * nb, colrun = S::NARR * arrrun, bufrun = cpw * colrun;
  if (tid < ncomp) {
    const double* rec = lds + (active ? cl : 0) * a.reclen;
    S st;
    st.init(rec, bin, a);
    int buf = 0;
    for (int j0 = 0; j0 < nz; j0 += T) {
      const int Tc = min(T, nz - j0);
      TIO* tl = tile + buf * bufrun + (active ? cl : 0) * colrun + b;
      for (int t = 0; t < Tc; ++t) {
        double val[S::NARR];
        st.template level<2>(j0 + t, rec, nz, val);
        if (active) {
#pragma unroll
          for (int k = 0; k < S::NARR; ++k) tl[k * arrrun + t * nb] = (TIO)val[k];
        }
      }
      lds_barrier();  // tile complete: hand it to the store waves
      buf ^= 1;
    }
  } else {
    const int sid = tid - ncomp, nst = blockDim.x - ncomp;
    const int nbv = nb / VW, arrv = arrrun / VW;
    int buf = 0;
    for (int j0 = 0; j0 < nz; j0 += T) {
      const int Tc = min(T, nz - j0);
      lds_barrier();  // tile `buf` is complete
      const int per = Tc * nbv;  // 16-B vectors of one (column, array) run
      const int total = ncol_here * per;
      int c = sid / per, v = sid - c * per;
      const int dc = nst / per, dv = nst - dc * per;
      for (int idx = sid; idx < total; idx += nst) {
        const vt* src = reinterpret_cast<const vt*>(tile + buf * bufrun + c * colrun) + v;
        const long long go = ((long long)(c0 + c) * nz + j0) * nbv + v;
        vt x[S::NARR];
#pragma unroll
        for (int k = 0; k < S::NARR; ++k) x[k] = src[k * arrv];
#pragma unroll
        for (int k = 0; k < S::NARR; ++k) __builtin_nontemporal_store(x[k], reinterpret_cast<vt*>(a.o[k]) + go);
        v += dv;
        c += dc;
        if (v >= per) {
          v -= per;
          ++c;
        }
      }
      buf ^= 1;
    }
  }
}

// ------------------------------------------------------------------------------------------
constexpr int MAX_DIRECT_LDS = 64 * 1024;
// tunables (crt_options.tune, per call): [0] LDS bytes a tile may take per workgroup, [1] force T (0 = automatic),
// [2] TileCfg.flags (bit0: __syncthreads barriers, bit1: generic instead of fused flush).  Measured on MI355X, 2s at 1e4 x 300 x 60 (tools/ab_tile.py, interleaved rounds):
//   T=2 (not line aligned) 1.66 ms | T=4, 4 WG/CU 1.20 ms | T=8, 2 WG/CU 1.03 ms | T=12, 1 WG/CU 1.25 ms
// -> take the longest line-aligned run that still leaves two workgroups resident per CU (160 KB LDS).
constexpr int DEFAULT_TILE_LDS = 78 * 1024;
// Narrow spectra (the 36-38-band shards of an 8-rank band partition): one compute wave + store waves per column beats the
// direct-store kernel 1.3-1.6x down to 16 bands (tools/ab_narrow.py, 2e5 x 38 x 60: k_direct 4.36 ms, k_pipe 2.67-2.74 ms;
// 4s: 2.25 -> 1.71 ms with one store wave).  Even at 4-15 bands the pipeline with ONE store wave beats the direct kernel, whose 8-byte
// stores leave partial lines (PMC traffic 1.2 x algorithmic): 4e5 x 12 x 60 3.90 -> 2.77 ms, 6e5 x 6 x 60 6.32 -> 3.54 ms.
constexpr int MIN_TILE_NB = 4;

int gcd(int x, int y) { return y ? gcd(y, x % y) : x; }

template <class S, typename TIO>
int launch_tile(const SolveArgs& a, hipStream_t s, bool& done) {
  done = false;
  const int nb = a.nb;
  int g_tune[8];  // this call's overrides (crt_options.tune); [0] = 0 means the default LDS budget
  for (int i = 0; i < 8; ++i) g_tune[i] = a.tune[i];
  if (g_tune[0] <= 0) g_tune[0] = DEFAULT_TILE_LDS;
  const int min_nb = a.tune[12] > 0 ? a.tune[12] : MIN_TILE_NB;
  if (nb < min_nb || nb > 1024) return CRT_OK;
  // very narrow spectra: several columns per compute wave (tune key 5 = 1 keeps one column per workgroup: A/B)
  if ((nb <= 32 || a.tune[5] == 2) && nb <= 128 && a.tune[5] != 1) {
    constexpr int VWp = 16 / (int)sizeof(TIO);
    bool ok = nb % VWp == 0 && (a.nz * nb) % VWp == 0;
    for (int i = 0; i < S::NARR && ok; ++i)
      if (reinterpret_cast<uintptr_t>(a.o[i]) & 15) ok = false;
    if (ok) {
      PackCfg pc;
      pc.ncw = a.tune[6] > 0 ? a.tune[6] : (nb <= 32 ? 1 : 2);
      if (64 * pc.ncw < nb) pc.ncw = (nb + 63) / 64;
      pc.cpw = 64 * pc.ncw / nb;
      const int linep = 128 / (int)sizeof(TIO);
      const int Tap = linep / gcd(nb, linep);  // levels per line-aligned run
      int Tp = a.tune[4] > 0 ? a.tune[4] : std::max(4, Tap);
      if (Tp > a.nz) Tp = a.nz;
      pc.T = Tp;
      const int cpw = pc.cpw;
      const int nswp = a.tune[3] > 0 ? a.tune[3] : 1;
      const size_t shp = (size_t)((cpw * a.reclen + 1) & ~1) * sizeof(double) + (size_t)2 * cpw * S::NARR * Tp * nb * sizeof(TIO);
      const int thr = 64 * (pc.ncw + nswp);
      if (shp <= 160 * 1024 && thr <= 512 && (Tp * nb) % VWp == 0) {
        auto kern = k_pipe_pack<S, TIO, 512>;
        if (shp > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp) != hipSuccess)
          return (int)CRT_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3((a.ncol + cpw - 1) / cpw), dim3(thr), shp, s, a, pc);
        if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
        note_kernel("k_pipe_pack<%s,%s> columns=%d compute_waves=%d T=%d store_waves=%d lds=%zu", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", pc.cpw,
                    pc.ncw, Tp, nswp, shp);
        done = true;
        return (int)CRT_OK;
      }
    }
  }
  const int CB = nb <= 256 ? 256 / nb : 1;
  const int nthr = CB > 1 ? 256 : ((nb + 63) / 64) * 64;
  const int line = 128 / (int)sizeof(TIO);  // elements per 128-B line
  const int Ta = line / gcd(nb, line);      // levels per line-aligned run
  const size_t per_level = (size_t)S::NARR * CB * nb * sizeof(TIO);
  const size_t target = (size_t)g_tune[0];
  int T = Ta;
  if (per_level * Ta > target) {
    // aligned runs do not fit: fall back to the largest run that does (only its two end lines are partial)
    T = (int)(target / per_level);
    if (T < 2) return CRT_OK;
  } else {
    while (per_level * (T + Ta) <= target && T + Ta <= a.nz) T += Ta;
  }
  if (g_tune[1] > 0) T = g_tune[1];
  if (T > a.nz) T = a.nz;
  TileCfg cfg;
  cfg.CB = CB;
  cfg.T = T;
  cfg.rec_dbl = (CB * a.reclen + 1) & ~1;
  cfg.flags = g_tune[2];
  const size_t sh = cfg.rec_dbl * sizeof(double) + (size_t)S::NARR * CB * T * nb * sizeof(TIO);
  if (sh > 160 * 1024) return CRT_OK;
  constexpr int VW = 16 / (int)sizeof(TIO);
  bool fused = CB == 1 && nb % VW == 0 && !(g_tune[2] & 2);
  for (int i = 0; i < S::NARR && fused; ++i)
    if (reinterpret_cast<uintptr_t>(a.o[i]) & 15) fused = false;
  const int grid = (a.ncol + CB - 1) / CB;
  // wave-specialised pipeline, always one column per workgroup (tune [2] bit2 disables it, bit3 disables its generic-flush
  // form [odd nb, several columns per k_tile workgroup]; [3] = store waves, [4] = its T)
  bool pfused = nb % VW == 0 && !(g_tune[2] & 2);  // the pipeline always has one column per workgroup
  for (int i = 0; i < S::NARR && pfused; ++i)
    if (reinterpret_cast<uintptr_t>(a.o[i]) & 15) pfused = false;
  if (!(g_tune[2] & 4) && (pfused || !(g_tune[2] & 8))) {
    const bool fused = pfused;  // (shadows k_tile's flag inside this block)
    const int pcomp = ((nb + 63) / 64) * 64;
    const size_t plevel = (size_t)S::NARR * nb * sizeof(TIO);
    // one compute wave (nb <= 64): one store wave keeps up with it and leaves more columns resident (tools/ab_2s_narrow.py, streaming
    // stores: 2e5 x 38 x 60 2s 2.53 -> 2.47 ms, g77 4.53 -> 4.27; 4e5 x 16 x 60 2s 3.39 -> 2.87, g77 4.52 -> 4.20; 4s already had one)
    // (4s -- heavy per-band set-up -- also takes one store wave with two compute waves and the whole-line generic flush:
    // tools/ab_closed_107.py, 3e4 x 107 x 60 1.09 -> 1.03 ms; not with the fused flush: 25000 x 128 x 60 0.95 -> 0.99)
    int nsw = g_tune[3] > 0 ? g_tune[3] : (pcomp <= 64 || (S::HEAVY_INIT && pcomp <= 128 && !fused) ? 1 : pcomp <= 128 ? 2 : 3);
    if (pcomp + 64 * nsw > 1024) nsw = (1024 - pcomp) / 64;
    // Two tile buffers of up to 8 levels, line-aligned runs when they fit, at least two workgroups per CU -- and about four
    // for a scheme with a heavy per-band set-up when the spectrum is narrow enough to allow it.  Measured (tools/ab_shapes.py,
    // profiles/r01/ab_shapes.txt), k_tile -> k_pipe: 2s nb=107 1.325 -> 1.20 ms (T=8), nb=64 1.148 -> 0.97 (T=8), nb=128 1.173 -> 1.154,
    // nb=200 no change; 4s nb=107 1.363 -> 1.24..1.29 (T=2..6; T=8 1.46), nb=64 1.231 -> 0.946; g77 nb=107 2.477 -> 2.24 (T=6).
    // Longer tiles lose: T=11 at nb=128 leaves one workgroup per CU, 1.5 ms.
    size_t budget = target;
    if (S::HEAVY_INIT && 2 * plevel * 4 <= 40 * 1024) budget = 40 * 1024;
    // f32 storage halves the stores but not the per-band set-up: for 2s three resident workgroups (tiles of 4 levels) cover it better than
    // two with 8 (tools/ab_f32_T.py: 1e4 x 300 x 60 0.597 -> 0.510 ms, 6000 x 300 x 100 0.566 -> 0.494, 3334 x 300 x 60 0.235 -> 0.187;
    // 4s indifferent, bl 2 % the other way: left at 8)
    if (sizeof(TIO) == 4 && std::is_same<S, Sch2s>::value && 2 * plevel * 4 <= 40 * 1024) budget = 40 * 1024;
    // generic flush: whole lines whatever T is, so narrow spectra take short tiles and four or five workgroups per CU
    // (3e4 x 107 x 60, k_tile -> T=8 -> T=4: 2s 1.268 -> 1.285 -> 1.085 ms, bl 1.174 -> 1.136 -> 1.066, g77 2.174 -> 1.957 -> 1.861)
    if (!fused && pcomp <= 128 && budget > 32 * 1024) budget = 32 * 1024;
    int Tmax = (int)std::min<size_t>(8, budget / (2 * plevel));
    if (fused && pcomp <= 64 && nb >= 32 && Tmax > 4) Tmax = 4;  // 32..64 bands: tiles of 4 levels (same measurement: 38 bands, T = 8 -> 4: 2s 2.52 -> 2.47, g77 4.41 -> 4.27)
    if (Tmax < 4 && 2 * plevel * 4 <= target) Tmax = 4;  // not below 4 levels while two workgroups still fit (g77 at nb = 107: T=2 1.96 ms, T=4 1.87)
    int Tp = Ta <= Tmax ? (Tmax / Ta) * Ta : Tmax;
    if (g_tune[4] > 0) Tp = g_tune[4];
    if (Tp > a.nz) Tp = a.nz;
    PipeTileCfg pc;
    pc.ncomp = pcomp;
    pc.T = Tp;
    pc.rec_dbl = (a.reclen + 1) & ~1;
    // two tile buffers (the generic flush adds one line of head-room per array and buffer)
    const size_t psh = pc.rec_dbl * sizeof(double) + 2 * plevel * Tp + (fused ? 0 : 2 * S::NARR * 128);
    if (nsw >= 1 && Tp >= 2 && psh <= 160 * 1024 && (fused ? (2 * plevel * Ta <= target || g_tune[4] > 0) : true)) {
      const int pthr = pcomp + 64 * nsw;
      auto gop = [&](auto kern) {
        if (psh > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)psh) != hipSuccess)
          return (int)CRT_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(pthr), psh, s, a, pc);
        if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
        note_kernel("k_pipe<%s,%s>%s T=%d store_waves=%d lds=%zu", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", fused ? "" : " generic-flush", Tp, nsw, psh);  // (only a launch that succeeded is reported)
        return (int)CRT_OK;
      };
      int st;
      if (fused) st = pthr <= 512 ? gop(k_pipe<S, TIO, 512, true>) : gop(k_pipe<S, TIO, 1024, true>);
      else st = pthr <= 512 ? gop(k_pipe<S, TIO, 512, false>) : gop(k_pipe<S, TIO, 1024, false>);
      done = st == CRT_OK;
      return st;
    }
  }
  auto go = [&](auto kern) {
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return (int)CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), sh, s, a, cfg);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_tile<%s,%s>%s CB=%d T=%d lds=%zu", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", fused ? "" : " generic-flush", CB, T, sh);  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  int st;
  if (fused) {
    if (nthr <= 256) st = go(k_tile<S, TIO, 256, true>);
    else if (nthr <= 512) st = go(k_tile<S, TIO, 512, true>);
    else st = go(k_tile<S, TIO, 1024, true>);
  } else {
    if (nthr <= 256) st = go(k_tile<S, TIO, 256, false>);
    else if (nthr <= 512) st = go(k_tile<S, TIO, 512, false>);
    else st = go(k_tile<S, TIO, 1024, false>);
  }
  done = st == CRT_OK;
  return st;
}

template <class S, typename TIO, int VEC, bool USE_LDS>
int launch_direct_v(const SolveArgs& a, size_t lds_bytes, hipStream_t s) {
  const long long items = (long long)a.ncol * (a.nb / VEC);
  const long long nblk = (items + DBLOCK - 1) / DBLOCK;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((k_direct<S, TIO, VEC, USE_LDS>), dim3((unsigned)nblk), dim3(DBLOCK), USE_LDS ? lds_bytes : 0, s, a);
  if (hipGetLastError() != hipSuccess) return CRT_ERR_LAUNCH;
  note_kernel("k_direct<%s,%s> VEC=%d", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", VEC);  // (only a launch that succeeded is reported)
  return CRT_OK;
}

template <class S, typename TIO>
int launch_scheme(const SolveArgs& a, hipStream_t s, int force) {
  if (force != 1) {
    bool done;
    int st = launch_tile<S, TIO>(a, s, done);
    if (st != CRT_OK || done) return st;
  }
  // two bands per lane when rows keep the alignment of a 2-element vector, else one
  bool vec2 = (a.nb % 2 == 0) && (a.col_stride % 2 == 0);
  for (int i = 0; i < S::NARR && vec2; ++i)
    if (reinterpret_cast<uintptr_t>(a.o[i]) & (2 * sizeof(TIO) - 1)) vec2 = false;
  const int nbv = a.nb / (vec2 ? 2 : 1);
  const long long cols_per_block = (DBLOCK - 1) / nbv + 2;
  const size_t lds_bytes = (size_t)cols_per_block * a.reclen * sizeof(double);
  const bool use_lds = lds_bytes <= (size_t)MAX_DIRECT_LDS;
  if (vec2)
    return use_lds ? launch_direct_v<S, TIO, 2, true>(a, lds_bytes, s) : launch_direct_v<S, TIO, 2, false>(a, lds_bytes, s);
  return use_lds ? launch_direct_v<S, TIO, 1, true>(a, lds_bytes, s) : launch_direct_v<S, TIO, 1, false>(a, lds_bytes, s);
}

template <class S>
int launch_io(const SolveArgs& a, hipStream_t s, int force) {
  return a.f32 ? launch_scheme<S, float>(a, s, force) : launch_scheme<S, double>(a, s, force);
}

// ------------------------------------------------------------------------------------------
// k_int: integrated outputs only (see IntArgs in crt_internal.hpp).  One workgroup per column, one lane per band; the
// level values never leave the registers, the only cross-lane traffic is ngroup shuffle reductions per level.
template <class S, typename TIO, int MAXT, bool PROF>
__global__ __launch_bounds__(MAXT) void k_int(SolveArgs a, IntArgs ia, int rec_dbl) {
  extern __shared__ double lds[];
  const int nb = a.nb, nz = a.nz, ng = ia.ngroup;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const int c = blockIdx.x;
  {
    const double* src = a.ws + (long long)c * a.reclen;
    for (int i = tid; i < a.reclen; i += nthr) lds[i] = src[i];
  }
  __syncthreads();
  const double* rec = lds;
  const IntLds L = int_lds_carve(lds + rec_dbl, nz, nwave, PROF);
  const bool active = tid < nb;
  const int b = active ? tid : 0;
  const BandIn in = load_band<TIO>(a, c, b, S::SOIL);
  S st;
  st.init(rec, in, a);
  double w[INT_MAXG];
#pragma unroll
  for (int g = 0; g < INT_MAXG; ++g) w[g] = (g < ng && active) ? ia.band_w[(long long)g * nb + b] : 0.0;
#pragma unroll
  for (int g = 0; g < INT_MAXG; ++g)
    if (g < ng) {
      const double t = wave_sum_all(w[g] * (1 - (in.r + in.t)) * in.I_dr0);
      if (lane == 0) L.pdr[wave * INT_MAXG + g] = t;
      if constexpr (PROF) {
        const double t0 = wave_sum_all(w[g] * in.I_dr0);
        if (lane == 0) L.pi0[wave * INT_MAXG + g] = t0;
      }
    }
  for (int j = 0; j < nz; ++j) {
    double val[S::NARR];
    st.level(j, rec, nz, val);  // val[0..2] = I_dr, I_df_d, I_df_u for every scheme
    int_accumulate<false, PROF>(L, nwave, wave, lane, nz, j, ng, w, active, val[0], val[1], val[2]);
  }
  int_finish<false, PROF>(L, ia, nwave, nz, c, rec[S_KB], rec[S_INVMU]);
}

template <class S, typename TIO>
int launch_int(const SolveArgs& a, const IntArgs& ia, hipStream_t s) {
  if (a.nb > 1024) return CRT_ERR_UNSUPPORTED;
  const int nthr = ((a.nb + 63) / 64) * 64;
  const int rec_dbl = (a.reclen + 1) & ~1;
  // wave totals of all band groups with one wave_sum4 per level (round 2; per-row partials and __shfl_xor butterflies before that:
  // 2s 1e4 x 300 x 60 0.80 / 0.625 / 1.59 ms)
  const bool prof = ia.L_dr != nullptr;
  const size_t sh = (rec_dbl + int_lds_doubles(a.nz, nthr / 64, false, prof)) * sizeof(double);
  if (sh > 160 * 1024) return CRT_ERR_UNSUPPORTED;
  auto go = [&](auto kern) {
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return (int)CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, a, ia, rec_dbl);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_int<%s>%s", S::NAME, prof ? " + level profiles" : " wave totals");  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  if (prof) {
    if (nthr <= 256) return go(k_int<S, TIO, 256, true>);
    if (nthr <= 512) return go(k_int<S, TIO, 512, true>);
    return go(k_int<S, TIO, 1024, true>);
  }
  if (nthr <= 256) return go(k_int<S, TIO, 256, false>);
  if (nthr <= 512) return go(k_int<S, TIO, 512, false>);
  return go(k_int<S, TIO, 1024, false>);
}

template <class S>
int launch_int_io(const SolveArgs& a, const IntArgs& ia, hipStream_t s) {
  return a.f32 ? launch_int<S, float>(a, ia, s) : launch_int<S, double>(a, ia, s);
}

}  // namespace

int launch_closed_int(int scheme, const SolveArgs& a, const IntArgs& ia, hipStream_t s) {
  switch (scheme) {
    case CRT_SCHEME_2S: return launch_int_io<Sch2s>(a, ia, s);
    case CRT_SCHEME_4S: return launch_int_io<Sch4s>(a, ia, s);
    case CRT_SCHEME_BL: return launch_int_io<SchBl>(a, ia, s);
    case CRT_SCHEME_G77: return launch_int_io<SchG77<false>>(a, ia, s);
    case CRT_SCHEME_BF: return launch_int_io<SchG77<true>>(a, ia, s);
    default: return CRT_ERR_BAD_ARG;
  }
}

// force: 0 = pick (tile when it applies), 1 = direct-store kernel (kept selectable for A/B measurements)
int launch_closed(int scheme, const SolveArgs& a, hipStream_t s, int force) {
  switch (scheme) {
    case CRT_SCHEME_2S: return launch_io<Sch2s>(a, s, force);
    case CRT_SCHEME_4S: return launch_io<Sch4s>(a, s, force);
    case CRT_SCHEME_BL: return launch_io<SchBl>(a, s, force);
    case CRT_SCHEME_G77: return launch_io<SchG77<false>>(a, s, force);
    case CRT_SCHEME_BF: return launch_io<SchG77<true>>(a, s, force);
    default: return CRT_ERR_BAD_ARG;
  }
}

}  // namespace crt
