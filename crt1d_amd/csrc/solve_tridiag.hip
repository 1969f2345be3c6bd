// Tridiagonal schemes: n79 (Norman 1979) and zq (Zhao & Qualls 2005) on gfx950.
//
// Both reference solvers assemble, per band, a tridiagonal system over the interleaved unknowns
// x = [up_0, dn_0, up_1, dn_1, ...] and solve it (n79: its own Python Thomas routine,
// _solve_n79.py:167-200; zq: scipy.sparse spsolve, _solve_zq.py:158-164).  Here one lane owns one
// (column, band) system and runs the Thomas recurrence serially over the levels; 64 lanes = 64
// bands advance in lockstep so every per-level store is one contiguous 512-B run.
//
// LDS budget.  The textbook algorithm keeps the forward-sweep pair (e_i, f_i) of all 2 nz rows
// for the back substitution: 32 nz bytes per lane = 123 KB per wave at nz = 60, 205 KB at nz = 100
// (more than a CU has).  Only the pairs of the EVEN rows ("up_k = f - e dn_k": the reflectance of
// everything below level k) are kept; the odd unknowns are recovered in the back sweep from the
// ORIGINAL even-row equation one level up, which is a contraction (coefficients < 1), so the
// sweep stays as stable as Thomas and the result differs from it by rounding only.  That halves
// the footprint: 16 nz bytes per lane -> 2 waves/CU at nz = 60, 1 wave/CU at nz = 100.
#include "crt_internal.hpp"

namespace crt {
namespace {

constexpr int TB = 64;  // one wavefront per workgroup

struct EF {
  double* base;  // lane-private column of the [row][TB] LDS array
  __device__ inline void put(int k, double e, double f) {
    base[(2 * k) * TB] = e;
    base[(2 * k + 1) * TB] = f;
  }
  __device__ inline void get(int k, double& e, double& f) const {
    e = base[(2 * k) * TB];
    f = base[(2 * k + 1) * TB];
  }
};

// ------------------------------------------------------------------------------------------
// n79: crt1d/solvers/_solve_n79.py:70-155
template <typename TIO, bool USE_LDS>
__global__ __launch_bounds__(TB) void k_n79(SolveArgs a, int rec_lds_doubles) {
  extern __shared__ double lds[];
  const Item it = locate<TB, 1>(a.ncol, a.nb);
  const double* rec = stage_records<TB, 1, USE_LDS>(a, it, lds);
  if (!it.active) return;
  EF ef{lds + rec_lds_doubles + threadIdx.x};

  const int nz = a.nz, nb = a.nb;
  const long long ib = (long long)it.c * a.col_stride + it.b;
  const double swb = ldio<TIO>(a.I_dr0, ib), swd = ldio<TIO>(a.I_df0, ib), rho = ldio<TIO>(a.leaf_r, ib), tau = ldio<TIO>(a.leaf_t, ib), alb = ldio<TIO>(a.soil_r, ib);
  const double invmu = rec[S_INVMU];
  const double* tbcum = rec + REC_HDR;
  const double* tb = tbcum + nz;
  const double* td = tb + nz;
  const double* fsun = td + nz;
  const double* isl = fsun + nz;
  const double* ish = isl + nz;

  // layer scattering coefficients (:85-88 / :102-105): r = trand/refld, s = refld - trand^2/refld
  auto layer = [&](int j, double& r, double& s) {
    const double t = td[j];
    const double refld = (1 - t) * rho;
    const double trand = (1 - t) * tau + t;
    const double inv = fast_rcp(refld);
    r = trand * inv;
    s = refld - trand * trand * inv;
  };

  // ---- forward elimination (b_i = 1 for every row) ----
  double e = -alb;                   // row 0: soil, upward (:79-82)
  double f = swb * tbcum[0] * alb;
  ef.put(0, e, f);
  double r, s;
  {
    // row 1: first downward equation uses layer index 1 of td/tb/tbcum (:85-92), as the reference
    layer(1, r, s);
    const double d = swb * tbcum[1] * (1 - tb[1]) * (tau - rho * r);
    const double iden = fast_rcp(1 + s * e);
    e = -r * iden;
    f = (d + s * f) * iden;
  }
  layer(0, r, s);
  for (int k = 1; k < nz; ++k) {
    {  // row 2k: upward flux at level k, layer k-1 (:102-109, :122-129)
      const double d = swb * tbcum[k] * (1 - tb[k - 1]) * (rho - tau * r);
      const double iden = fast_rcp(1 + r * e);
      e = -s * iden;
      f = (d + r * f) * iden;
      ef.put(k, e, f);
    }
    if (k <= nz - 2) {  // row 2k+1: downward flux at level k, layer k (:112-119)
      layer(k, r, s);
      const double d = swb * tbcum[k + 1] * (1 - tb[k]) * (tau - rho * r);
      const double iden = fast_rcp(1 + s * e);
      e = -r * iden;
      f = (d + s * f) * iden;
    }
  }

  // ---- back substitution, top -> ground, writing the outputs as it goes ----
  const double oma = 1 - (rho + tau);  // 1 - omega (:56,145)
  double dn = swd;                     // last row: dn_top = sky diffuse (:132-135)
  double up;
  {
    double ee, ff;
    ef.get(nz - 1, ee, ff);
    up = ff - ee * dn;
  }
  long long o = ((long long)it.c * nz + (nz - 1)) * nb + it.b;
  long long om = ((long long)it.c * (nz - 1) + (nz - 2)) * nb + it.b;
  {
    const double idr = swb * tbcum[nz - 1];
    __builtin_nontemporal_store((TIO)(idr), outp<TIO>(a.o[0]) + o);
    __builtin_nontemporal_store((TIO)(dn), outp<TIO>(a.o[1]) + o);
    __builtin_nontemporal_store((TIO)(up), outp<TIO>(a.o[2]) + o);
    __builtin_nontemporal_store((TIO)(idr * invmu + 2 * dn + 2 * up), outp<TIO>(a.o[3]) + o);
  }
  for (int k = nz - 2; k >= 0; --k) {
    o -= nb;
    const double dn1 = dn;
    // dn_k from the upward equation of level k+1 (layer k):  -r dn_k + up_{k+1} - s dn_{k+1} = d
    const double t = td[k];
    const double refld = (1 - t) * rho;
    const double trand = (1 - t) * tau + t;
    const double src = swb * tbcum[k + 1] * (1 - tb[k]);
    dn = (refld * up + (trand * trand - refld * refld) * dn1 - src * (rho * refld - tau * trand)) * fast_rcp(trand);
    double ee, ff;
    ef.get(k, ee, ff);
    up = ff - ee * dn;
    // absorbed by sunlit / shaded leaves of layer k, per unit leaf area (:145-155)
    const double direct = src * oma;
    const double diffuse = (dn1 + up) * (1 - t) * oma;
    const double fs = fsun[k];
    __builtin_nontemporal_store((TIO)((diffuse * fs + direct) * isl[k]), outp<TIO>(a.o[4]) + om);
    __builtin_nontemporal_store((TIO)((diffuse * (1 - fs)) * ish[k]), outp<TIO>(a.o[5]) + om);
    om -= nb;
    const double idr = swb * tbcum[k];
    __builtin_nontemporal_store((TIO)(idr), outp<TIO>(a.o[0]) + o);
    __builtin_nontemporal_store((TIO)(dn), outp<TIO>(a.o[1]) + o);
    __builtin_nontemporal_store((TIO)(up), outp<TIO>(a.o[2]) + o);
    __builtin_nontemporal_store((TIO)(idr * invmu + 2 * dn + 2 * up), outp<TIO>(a.o[3]) + o);
  }
}

// ------------------------------------------------------------------------------------------
// zq: crt1d/solvers/_solve_zq.py:74-219
template <typename TIO, bool USE_LDS>
__global__ __launch_bounds__(TB) void k_zq(SolveArgs a, int rec_lds_doubles) {
  extern __shared__ double lds[];
  const Item it = locate<TB, 1>(a.ncol, a.nb);
  const double* rec = stage_records<TB, 1, USE_LDS>(a, it, lds);
  if (!it.active) return;
  EF ef{lds + rec_lds_doubles + threadIdx.x};

  const int m = a.nz, nb = a.nb;
  const long long ib = (long long)it.c * a.col_stride + it.b;
  const double I_dr0 = ldio<TIO>(a.I_dr0, ib), I_df0 = ldio<TIO>(a.I_df0, ib), bL = ldio<TIO>(a.leaf_r, ib), tL = ldio<TIO>(a.leaf_t, ib), rho = ldio<TIO>(a.soil_r, ib);
  const double mu = rec[S_MU], invmu = rec[S_INVMU], t = rec[S_TAUI], t_psi = rec[S_TPSI];
  const double* ekl = rec + REC_HDR;

  const double aL = 1 - (bL + tL);                                          // :87
  const double r_i = 2.0 / 3 * (bL / (bL + tL)) + 1.0 / 3 * (tL / (bL + tL));  // eq. 23 :40-43
  const double r_psi = 0.5 + 0.3334 * ((bL - tL) / (bL + tL)) * mu;         // eq. 22 :35-38
  const double fwd = t + (1 - t) * (1 - aL) * (1 - r_i);                    // :116
  const double q = r_i * (1 - aL) * (1 - t);        // r (1-a) (1-t) of an interior layer
  const double q0 = 1.0 * (1 - (1 - rho)) * (1 - 0.0);  // ground "layer": r=1, t=0, a=1-rho (:106-108)
  const double cu = r_psi * (1 - t_psi) * (1 - aL);        // :139
  const double cd = (1 - t_psi) * (1 - aL) * (1 - r_psi);  // :142

  // ---- forward elimination over rows 0 .. 2m+1; (e, f) kept for the even rows ----
  double e = 0.0;                       // row 0: x0 = rho S_0 (:115,136)
  double f = rho * (I_dr0 * ekl[0]);
  ef.put(0, e, f);
  for (int li = 1; li <= m; ++li) {
    const double S = I_dr0 * ekl[li - 1];  // :130
    const double qlo = (li == 1) ? q0 : q;
    const double qhi = (li == m) ? 0.0 : q;
    const double dlo = 1 - qlo * q;        // 1 - r r (1-a)(1-t)(1-a)(1-t)  (:118)
    const double dhi = 1 - q * qhi;        // (:119)
    {  // row 2li-1: sub = -fwd, dia = -qlo fwd, sup = dlo (:116-118), rhs :137-139
      const double iden = fast_rcp(-qlo * fwd + fwd * e);
      const double C = dlo * cu * S;
      e = dlo * iden;
      f = (C + fwd * f) * iden;
    }
    {  // row 2li: sub = dhi, dia = -qhi fwd, sup = -fwd (:119-121), rhs :140-142
      const double iden = fast_rcp(-qhi * fwd - dhi * e);
      const double C = dhi * cd * S;
      e = -fwd * iden;
      f = (C - dhi * f) * iden;
      ef.put(li, e, f);
    }
  }
  // row 2m+1: x = I_df0 (:122,143)
  double xd = I_df0;                 // SWd0[li]
  double xu;                         // SWu0[li]
  {
    double ee, ff;
    ef.get(m, ee, ff);
    xu = ff - ee * xd;
  }
  long long o = ((long long)it.c * m + (m - 1)) * nb + it.b;
  for (int li = m; li >= 1; --li, o -= nb) {
    const int z = li - 1;
    const double S = I_dr0 * ekl[z];
    const double qlo = (li == 1) ? q0 : q;
    const double qhi = (li == m) ? 0.0 : q;
    const double dhi = 1 - q * qhi;
    const double dlo = 1 - qlo * q;
    // SWd0[li-1] from the original row 2li:  dhi x_{2li-1} - qhi fwd x_{2li} - fwd x_{2li+1} = C
    const double xdl = (dhi * cd * S + qhi * fwd * xu + fwd * xd) * fast_rcp(dhi);
    double ee, ff;
    ef.get(z, ee, ff);
    const double xul = ff - ee * xdl;  // SWu0[li-1]
    // multiple-scattering correction, eqs. 24/25 (:180-187), at output level z = li-1
    const double iden = fast_rcp(dlo);
    const double dn = (xd + q * xul) * iden;
    const double up = (xul + qlo * xd) * iden;
    const double Fss = S * invmu + 2 * xul + 2 * xd;
    const double F = S * invmu + 2 * up + 2 * dn;
    __builtin_nontemporal_store((TIO)(S), outp<TIO>(a.o[0]) + o);     // :219
    __builtin_nontemporal_store((TIO)(dn), outp<TIO>(a.o[1]) + o);    // :198
    __builtin_nontemporal_store((TIO)(up), outp<TIO>(a.o[2]) + o);    // :200
    __builtin_nontemporal_store((TIO)(F), outp<TIO>(a.o[3]) + o);     // :202
    __builtin_nontemporal_store((TIO)(xd), outp<TIO>(a.o[4]) + o);    // I_df_d_ss :197
    __builtin_nontemporal_store((TIO)(xul), outp<TIO>(a.o[5]) + o);   // I_df_u_ss :199
    __builtin_nontemporal_store((TIO)(Fss), outp<TIO>(a.o[6]) + o);   // :201
    xd = xdl;
    xu = xul;
  }
}

constexpr size_t MAX_WG_LDS = 160 * 1024;

template <typename K>
int set_lds_limit(K kern, size_t bytes) {
  if (bytes <= 64 * 1024) return CRT_OK;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
                 hipSuccess
             ? CRT_OK
             : CRT_ERR_LAUNCH;
}

}  // namespace

template <typename TIO>
int launch_tridiag_io(int scheme, const SolveArgs& a, hipStream_t s, int force) {
  if (force != 1) {  // column-tile kernel (solve_tridiag_tile.hip) when it applies; force = 1 keeps the per-wave kernels
    bool done = false;
    const int st = launch_tridiag_tile(scheme, a, s, done);
    if (st != CRT_OK || done) return st;
  }
  const long long items = (long long)a.ncol * a.nb;
  const long long nblk = (items + TB - 1) / TB;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  const int pairs = (scheme == CRT_SCHEME_ZQ) ? a.nz + 1 : a.nz;
  const size_t ef_bytes = (size_t)pairs * 2 * TB * sizeof(double);
  const long long cols_per_block = (TB - 1) / a.nb + 2;
  size_t rec_doubles = (size_t)cols_per_block * a.reclen;
  bool use_lds = ef_bytes + rec_doubles * sizeof(double) <= MAX_WG_LDS && rec_doubles * sizeof(double) <= 32 * 1024;
  if (!use_lds) rec_doubles = 0;
  const size_t sh = ef_bytes + rec_doubles * sizeof(double);
  if (sh > MAX_WG_LDS) return CRT_ERR_UNSUPPORTED;  // nz beyond what one wave's sweep state fits in a CU's LDS
  dim3 grid((unsigned)nblk), block(TB);
  int st;
  if (scheme == CRT_SCHEME_N79) {
    if (use_lds) {
      if ((st = set_lds_limit(k_n79<TIO, true>, sh)) != CRT_OK) return st;
      hipLaunchKernelGGL((k_n79<TIO, true>), grid, block, sh, s, a, (int)rec_doubles);
    } else {
      if ((st = set_lds_limit(k_n79<TIO, false>, sh)) != CRT_OK) return st;
      hipLaunchKernelGGL((k_n79<TIO, false>), grid, block, sh, s, a, 0);
    }
  } else if (scheme == CRT_SCHEME_ZQ) {
    if (use_lds) {
      if ((st = set_lds_limit(k_zq<TIO, true>, sh)) != CRT_OK) return st;
      hipLaunchKernelGGL((k_zq<TIO, true>), grid, block, sh, s, a, (int)rec_doubles);
    } else {
      if ((st = set_lds_limit(k_zq<TIO, false>, sh)) != CRT_OK) return st;
      hipLaunchKernelGGL((k_zq<TIO, false>), grid, block, sh, s, a, 0);
    }
  } else {
    return CRT_ERR_BAD_ARG;
  }
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int launch_tridiag(int scheme, const SolveArgs& a, hipStream_t s, int force) {
  return a.f32 ? launch_tridiag_io<float>(scheme, a, s, force) : launch_tridiag_io<double>(scheme, a, s, force);
}

}  // namespace crt
