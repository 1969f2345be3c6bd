// Closed-form schemes: 2s, 4s, bl, g77, bf (gfx950).
//
// Mapping (all kernels): one lane owns VEC adjacent bands of one column, computes the per-band
// coefficients once in registers, then sweeps the nz canopy levels.  Per level it reads the
// band-independent column vectors (lai_j, exp(-K_b lai_j), ...) from the K0 record staged in LDS
// (same address for the whole wave -> LDS broadcast, no vmem traffic in the loop besides stores)
// and issues one streaming store per output array; a wave's store is one contiguous 512-B
// (VEC=1) / 1-KiB (VEC=2) run of the band-contiguous [ncol][nz][nb] output.
// The kernels are HBM-write-bound: 4-7 fp64 profiles out per 5 fp64 scalars in.
#include "crt_internal.hpp"

namespace crt {
namespace {

constexpr int BLOCK = 256;
constexpr double PI = 3.14159265358979323846;

struct BandIn {
  double I_dr0, I_df0, r, t, s;
};

template <int VEC>
__device__ inline void load_bands(const SolveArgs& a, const Item& it, bool need_soil, BandIn (&in)[VEC]) {
  const long long base = (long long)it.c * a.col_stride + it.b;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    in[v].I_dr0 = a.I_dr0[base + v];
    in[v].I_df0 = a.I_df0[base + v];
    in[v].r = a.leaf_r[base + v];
    in[v].t = a.leaf_t[base + v];
    in[v].s = need_soil ? a.soil_r[base + v] : 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// 2s  Dickinson-Sellers two-stream (crt1d/solvers/_solve_2s.py:54-156)
struct Coef2s {
  double h;                       // diffuse extinction              :84
  double Au, Bu, Cu, Ad, Bd, Cd;  // up/dn = A e^{-KL} + B e^{-hL} + C e^{+hL}   :125-135
  double I0;
};

__device__ inline Coef2s coef_2s(const double* rec, const BandIn& in) {
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
  const double h = sqrt(b * b - c * c) / mb;
  const double sig = mbK * mbK + c * c - b * b;                      // :85
  const double u1 = b - c / rs;                                      // :87
  const double u2 = b - c * rs;
  const double u3 = f + c * rs;
  const double S1 = exp(-h * LT);
  const double S2 = exp(-K * LT);
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
  Coef2s k;
  k.h = h;
  k.I0 = in.I_dr0;
  k.Au = in.I_dr0 * h1s;
  k.Bu = in.I_dr0 * h2 + in.I_df0 * h7;
  k.Cu = in.I_dr0 * h3 + in.I_df0 * h8;
  k.Ad = in.I_dr0 * h4s;
  k.Bd = in.I_dr0 * h5 + in.I_df0 * h9;
  k.Cd = in.I_dr0 * h6 + in.I_df0 * h10;
  return k;
}

template <int VEC, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void k_2s(SolveArgs a) {
  extern __shared__ double lds[];
  const Item it = locate<BLOCK, VEC>(a.ncol, a.nb);
  const double* rec = stage_records<BLOCK, VEC, USE_LDS>(a, it, lds);
  if (!it.active) return;
  BandIn in[VEC];
  load_bands<VEC>(a, it, true, in);
  Coef2s k[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) k[v] = coef_2s(rec, in[v]);
  const double invmu = rec[S_INVMU];
  const int nz = a.nz;
  const double* lai = rec + REC_HDR;
  const double* ekl = lai + nz;
  long long o = ((long long)it.c * nz) * a.nb + it.b;
  for (int j = 0; j < nz; ++j, o += a.nb) {
    const double L = lai[j], eK = ekl[j];
    double idr[VEC], dn[VEC], up[VEC], F[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double em = exp(-k[v].h * L);
      const double ep = 1.0 / em;
      up[v] = k[v].Au * eK + k[v].Bu * em + k[v].Cu * ep;
      dn[v] = k[v].Ad * eK + k[v].Bd * em + k[v].Cd * ep;
      idr[v] = k[v].I0 * eK;                              // :150
      F[v] = idr[v] * invmu + 2 * up[v] + 2 * dn[v];      // :156
    }
    store_stream<VEC>(a.o[0] + o, idr);
    store_stream<VEC>(a.o[1] + o, dn);
    store_stream<VEC>(a.o[2] + o, up);
    store_stream<VEC>(a.o[3] + o, F);
  }
}

// ------------------------------------------------------------------------------------------
// bl  Beer-Lambert (crt1d/solvers/_solve_bl.py:51-90)
template <int VEC, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void k_bl(SolveArgs a) {
  extern __shared__ double lds[];
  const Item it = locate<BLOCK, VEC>(a.ncol, a.nb);
  const double* rec = stage_records<BLOCK, VEC, USE_LDS>(a, it, lds);
  if (!it.active) return;
  BandIn in[VEC];
  load_bands<VEC>(a, it, false, in);
  const double Kb = rec[S_KB], invmu = rec[S_INVMU];
  double Kg[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) Kg[v] = Kb * sqrt(1 - (in[v].t + in[v].r));  // :58-62
  const int nz = a.nz;
  const double* lai = rec + REC_HDR;
  const double* ekl = lai + nz;
  const double* tdf = ekl + nz;
  long long o = ((long long)it.c * nz) * a.nb + it.b;
  for (int j = 0; j < nz; ++j, o += a.nb) {
    const double L = lai[j], tb = ekl[j], td = tdf[j];
    double idr[VEC], dn[VEC], up[VEC], F[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double tg = exp(-Kg[v] * L);                                   // :65
      idr[v] = in[v].I_dr0 * tb;                                           // :69
      dn[v] = in[v].I_df0 * td + 0.5 * (in[v].I_dr0 * (tg - tb));          // :70,74,79
      up[v] = 0.0;                                                         // :87
      F[v] = idr[v] * invmu + 2 * dn[v];                                   // :90
    }
    store_stream<VEC>(a.o[0] + o, idr);
    store_stream<VEC>(a.o[1] + o, dn);
    store_stream<VEC>(a.o[2] + o, up);
    store_stream<VEC>(a.o[3] + o, F);
  }
}

// ------------------------------------------------------------------------------------------
// g77 Goudriaan 1977 (crt1d/solvers/_solve_g77.py:48-124) and bf Bodin & Franklin
// (crt1d/solvers/_solve_bf.py:60-140): same inputs/outputs, different scattered-light terms.
template <int VEC, bool USE_LDS, bool BF>
__global__ __launch_bounds__(BLOCK) void k_g77(SolveArgs a) {
  extern __shared__ double lds[];
  const Item it = locate<BLOCK, VEC>(a.ncol, a.nb);
  const double* rec = stage_records<BLOCK, VEC, USE_LDS>(a, it, lds);
  if (!it.active) return;
  BandIn in[VEC];
  load_bands<VEC>(a, it, true, in);
  const double kb = rec[S_KB], mu = rec[S_MU], invmu = rec[S_INVMU], LT = rec[S_LT];
  const int nz = a.nz;
  const double* lai = rec + REC_HDR;
  const double* ekl = lai + nz;
  const double A0 = ekl[0];  // A_sl at the ground level
  double kp[VEC], kd[VEC], omr[VEC], cdf[VEC], csr[VEC], ct[VEC], gnd[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const double sigma = in[v].r + in[v].t;                                       // g77:57
    kp[v] = sqrt(1 - sigma);                                                      // :59
    const double rho_c = ((1 - kp[v]) / (1 + kp[v])) * (2 / (1 + 1.6 * mu));      // :66
    omr[v] = BF ? 1.0 : (1 - rho_c);                                              // bf:84 drops (1 - rho_c)
    kd[v] = 0.8 * sqrt(1 - sigma);                                                // :69
    cdf[v] = kd[v] / kp[v];
    csr[v] = kd[v] / sqrt(1 - in[v].r);
    ct[v] = kd[v] / sqrt(1 - in[v].t);
    // ground-level terms for the soil-reflected stream (g77:95, bf:112)
    const double ed0 = exp(-kd[v] * LT);
    const double Idf0 = in[v].I_df0 * omr[v] * ed0;
    double Iscd0;
    if (BF)
      Iscd0 = in[v].I_dr0 * in[v].t * ((A0 - ed0) / (kd[v] - kb));                // bf:95
    else
      Iscd0 = 0.5 * (in[v].I_dr0 * (1 - rho_c) * exp(-kp[v] * kb * LT) - in[v].I_dr0 * (1 - sigma) * A0);
    gnd[v] = in[v].s * (in[v].I_dr0 * A0 + Idf0 + Iscd0);
  }
  long long o = ((long long)it.c * nz) * a.nb + it.b;
  for (int j = 0; j < nz; ++j, o += a.nb) {
    const double L = lai[j], Asl = ekl[j];
    double idr[VEC], dn[VEC], up[VEC], F[VEC], asl[VEC], ash[VEC], al[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double ed = exp(-kd[v] * L);
      const double Idf = in[v].I_df0 * omr[v] * ed;                               // g77:73 / bf:84
      double Iscd, Iscu;
      if (BF) {
        Iscd = in[v].I_dr0 * in[v].t * ((Asl - ed) / (kd[v] - kb));               // bf:95
        Iscu = in[v].I_dr0 * in[v].r * ((Asl - exp(kd[v] * L - (kb + kd[v]) * LT)) / (kd[v] + kb));  // bf:99-103
      } else {
        const double sigma = in[v].r + in[v].t;
        const double Isc = in[v].I_dr0 * omr[v] * exp(-kp[v] * kb * L) - in[v].I_dr0 * (1 - sigma) * Asl;  // g77:84-86
        Iscd = 0.5 * Isc;
        Iscu = 0.5 * Isc;
      }
      const double Isr = gnd[v] * exp(-kd[v] * (LT - L));                         // g77:95
      const double common = cdf[v] * Idf + csr[v] * Iscu + ct[v] * Iscd;
      ash[v] = (1 - Asl) * common;                                                // :99-101
      asl[v] = Asl * (common + kb * in[v].I_dr0);                                 // :106-111
      al[v] = asl[v] + ash[v];
      idr[v] = in[v].I_dr0 * Asl;                                                 // :77
      dn[v] = Iscd + Idf;                                                         // :115
      up[v] = Iscu + Isr;                                                         // :116
      F[v] = idr[v] * invmu + 2 * up[v] + 2 * dn[v];                              // :122
    }
    store_stream<VEC>(a.o[0] + o, idr);
    store_stream<VEC>(a.o[1] + o, dn);
    store_stream<VEC>(a.o[2] + o, up);
    store_stream<VEC>(a.o[3] + o, F);
    store_stream<VEC>(a.o[4] + o, asl);
    store_stream<VEC>(a.o[5] + o, ash);
    store_stream<VEC>(a.o[6] + o, al);
  }
}

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
struct Coef4s {
  double lam1, lam2;  // sqrt(|l1|), sqrt(|l2|)
  bool osc;           // l2 < 0
  double d[5], u[5];  // I_df_d / I_df_u = sum_k coef[k] * phi_k(x), phi = {E1, F1, phi3, phi4, e^{-kappa x}}
  double I0;
};

__device__ inline void solve4(double (&A)[4][5]) {
  // Gaussian elimination, partial pivoting by conditional row swaps (no dynamic register indexing)
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

__device__ inline Coef4s coef_4s(const double* rec, const BandIn& in, double mu_s) {
  const double G = rec[S_G], mu0 = rec[S_MU], kap = rec[S_KB], G1 = rec[S_GINT1], G2 = rec[S_GINT2], LT = rec[S_LT];
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

  Coef4s k;
  k.I0 = in.I_dr0;
  k.osc = l2 < 0.0;
  const double lam1 = sqrt(l1), lam2 = sqrt(fabs(l2));
  k.lam1 = lam1;
  k.lam2 = lam2;
  // basis functions: phi1 = e^{-lam1 x}, phi2 = e^{-lam1 (L - x)},
  //   real:  phi3 = e^{-lam2 x}, phi4 = e^{-lam2 (L - x)}      osc: phi3 = cos(lam2 x), phi4 = sin(lam2 x)
  // yd, yu contribution vectors of each basis function at x = 0 and x = L
  const double U1x = 0.5 * (v1x + lam1 * n0 * v1x), U1y = 0.5 * (v1y + lam1 * n1 * v1y);
  const double W1x = 0.5 * (v1x - lam1 * n0 * v1x), W1y = 0.5 * (v1y - lam1 * n1 * v1y);
  const double E1L = exp(-lam1 * LT);
  // [basis][x=0 | x=L][yd | yu][component]
  double Y[4][2][2][2];
  // phi1: yd = U1 e^{-lam1 x}, yu = W1 e^{-lam1 x}
  Y[0][0][0][0] = U1x;       Y[0][0][0][1] = U1y;       Y[0][0][1][0] = W1x;       Y[0][0][1][1] = W1y;
  Y[0][1][0][0] = U1x * E1L; Y[0][1][0][1] = U1y * E1L; Y[0][1][1][0] = W1x * E1L; Y[0][1][1][1] = W1y * E1L;
  // phi2: yd = W1 e^{-lam1 (L-x)}, yu = U1 e^{-lam1 (L-x)}
  Y[1][0][0][0] = W1x * E1L; Y[1][0][0][1] = W1y * E1L; Y[1][0][1][0] = U1x * E1L; Y[1][0][1][1] = U1y * E1L;
  Y[1][1][0][0] = W1x;       Y[1][1][0][1] = W1y;       Y[1][1][1][0] = U1x;       Y[1][1][1][1] = U1y;
  // second pair
  const double hx = 0.5 * v2x, hy = 0.5 * v2y;
  const double qx = 0.5 * lam2 * n0 * v2x, qy = 0.5 * lam2 * n1 * v2y;  // (lam2/2) N v2
  double cL = 0, sL = 0, E2L = 0;
  if (k.osc) {
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
    E2L = exp(-lam2 * LT);
    const double U2x = hx + qx, U2y = hy + qy, W2x = hx - qx, W2y = hy - qy;
    Y[2][0][0][0] = U2x;       Y[2][0][0][1] = U2y;       Y[2][0][1][0] = W2x;       Y[2][0][1][1] = W2y;
    Y[2][1][0][0] = U2x * E2L; Y[2][1][0][1] = U2y * E2L; Y[2][1][1][0] = W2x * E2L; Y[2][1][1][1] = W2y * E2L;
    Y[3][0][0][0] = W2x * E2L; Y[3][0][0][1] = W2y * E2L; Y[3][0][1][0] = U2x * E2L; Y[3][0][1][1] = U2y * E2L;
    Y[3][1][0][0] = W2x;       Y[3][1][0][1] = W2y;       Y[3][1][1][0] = U2x;       Y[3][1][1][1] = U2y;
  }
  // boundary conditions
  //   top    (x = 0): yd = R_df0 [1, 1]                                   (:110-116, both problems summed)
  //   bottom (x = L): yu - rho (2 muv . yd + mu0 R_dr0 e^{-kappa L}) [1, 1] = 0     (:128-138)
  const double eKL = exp(-kap * LT);
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
  k.d[0] = mU1 * A[0][4];
  k.u[0] = mW1 * A[0][4];
  k.d[1] = mW1 * A[1][4];
  k.u[1] = mU1 * A[1][4];
  const double mh = tp * (mu2 * hx + mu1 * hy), mq = tp * (mu2 * qx + mu1 * qy);
  if (k.osc) {
    // yd = a3 (h c + q s) + a4 (h s - q c);  yu = a3 (h c - q s) + a4 (h s + q c)
    k.d[2] = mh * A[2][4] - mq * A[3][4];
    k.d[3] = mq * A[2][4] + mh * A[3][4];
    k.u[2] = mh * A[2][4] + mq * A[3][4];
    k.u[3] = -mq * A[2][4] + mh * A[3][4];
  } else {
    k.d[2] = (mh + mq) * A[2][4];
    k.u[2] = (mh - mq) * A[2][4];
    k.d[3] = (mh - mq) * A[3][4];
    k.u[3] = (mh + mq) * A[3][4];
  }
  k.d[4] = tp * (mu2 * Pd0 + mu1 * Pd1);
  k.u[4] = tp * (mu2 * Pu0 + mu1 * Pu1);
  return k;
}

template <int VEC, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void k_4s(SolveArgs a) {
  extern __shared__ double lds[];
  const Item it = locate<BLOCK, VEC>(a.ncol, a.nb);
  const double* rec = stage_records<BLOCK, VEC, USE_LDS>(a, it, lds);
  if (!it.active) return;
  BandIn in[VEC];
  load_bands<VEC>(a, it, true, in);
  Coef4s k[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) k[v] = coef_4s(rec, in[v], a.mu_s);
  const double invmu = rec[S_INVMU], LT = rec[S_LT];
  const int nz = a.nz;
  const double* lai = rec + REC_HDR;
  const double* ekl = lai + nz;
  long long o = ((long long)it.c * nz) * a.nb + it.b;
  for (int j = 0; j < nz; ++j, o += a.nb) {
    const double x = lai[j], eK = ekl[j];
    double idr[VEC], dn[VEC], up[VEC], F[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double p1 = exp(-k[v].lam1 * x);
      const double p2 = exp(-k[v].lam1 * (LT - x));
      double p3, p4;
      if (k[v].osc) {
        sincos(k[v].lam2 * x, &p4, &p3);
      } else {
        p3 = exp(-k[v].lam2 * x);
        p4 = exp(-k[v].lam2 * (LT - x));
      }
      dn[v] = k[v].d[0] * p1 + k[v].d[1] * p2 + k[v].d[2] * p3 + k[v].d[3] * p4 + k[v].d[4] * eK;
      up[v] = k[v].u[0] * p1 + k[v].u[1] * p2 + k[v].u[2] * p3 + k[v].u[3] * p4 + k[v].u[4] * eK;
      idr[v] = k[v].I0 * eK;                               // :284
      F[v] = idr[v] * invmu + 2 * up[v] + 2 * dn[v];       // :290
    }
    store_stream<VEC>(a.o[0] + o, idr);
    store_stream<VEC>(a.o[1] + o, dn);
    store_stream<VEC>(a.o[2] + o, up);
    store_stream<VEC>(a.o[3] + o, F);
  }
}

// ------------------------------------------------------------------------------------------
constexpr int MAX_LDS_BYTES = 64 * 1024;  // keep >= 2 workgroups per CU resident

template <int VEC, bool USE_LDS>
int launch_vec(int scheme, const SolveArgs& a, size_t lds_bytes, hipStream_t s) {
  const long long items = (long long)a.ncol * (a.nb / VEC);
  const long long nblk = (items + BLOCK - 1) / BLOCK;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nblk), block(BLOCK);
  const size_t sh = USE_LDS ? lds_bytes : 0;
  switch (scheme) {
    case CRT_SCHEME_2S: hipLaunchKernelGGL((k_2s<VEC, USE_LDS>), grid, block, sh, s, a); break;
    case CRT_SCHEME_4S: hipLaunchKernelGGL((k_4s<VEC, USE_LDS>), grid, block, sh, s, a); break;
    case CRT_SCHEME_BL: hipLaunchKernelGGL((k_bl<VEC, USE_LDS>), grid, block, sh, s, a); break;
    case CRT_SCHEME_G77: hipLaunchKernelGGL((k_g77<VEC, USE_LDS, false>), grid, block, sh, s, a); break;
    case CRT_SCHEME_BF: hipLaunchKernelGGL((k_g77<VEC, USE_LDS, true>), grid, block, sh, s, a); break;
    default: return CRT_ERR_BAD_ARG;
  }
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

}  // namespace

int launch_closed(int scheme, const SolveArgs& a, hipStream_t s) {
  // two bands per lane (16-B stores) when rows keep 16-B alignment, else one
  bool vec2 = (a.nb % 2 == 0) && (a.col_stride % 2 == 0);
  for (int i = 0; i < 7 && vec2; ++i)
    if (a.o[i] && (reinterpret_cast<uintptr_t>(a.o[i]) & 15)) vec2 = false;
  const int vec = vec2 ? 2 : 1;
  const int nbv = a.nb / vec;
  const long long cols_per_block = (BLOCK - 1) / nbv + 2;
  const size_t lds_bytes = (size_t)cols_per_block * a.reclen * sizeof(double);
  const bool use_lds = lds_bytes <= (size_t)MAX_LDS_BYTES;
  if (vec2) return use_lds ? launch_vec<2, true>(scheme, a, lds_bytes, s) : launch_vec<2, false>(scheme, a, lds_bytes, s);
  return use_lds ? launch_vec<1, true>(scheme, a, lds_bytes, s) : launch_vec<1, false>(scheme, a, lds_bytes, s);
}

}  // namespace crt
