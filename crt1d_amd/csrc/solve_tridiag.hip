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
#include "tri_schemes.hpp"

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
// n79: crt1d/solvers/_solve_n79.py:70-155 (scheme arithmetic in tri_schemes.hpp, shared with the column-tile kernel)
template <typename TIO, bool USE_LDS>
__global__ __launch_bounds__(TB) void k_n79(SolveArgs a, int rec_lds_doubles) {
  extern __shared__ double lds[];
  const Item it = locate<TB, 1>(a.ncol, a.nb);
  const double* rec = stage_records<TB, 1, USE_LDS>(a, it, lds);
  if (!it.active) return;
  EF ef{lds + rec_lds_doubles + threadIdx.x};
  const int nz = a.nz, nb = a.nb;
  TriN79 st;
  st.template init<TIO>(rec, a, it.c, it.b);
  const double bc = st.band_const(), invmu = rec[S_INVMU];

  double e, f;
  st.first(rec, nz, e, f);
  ef.put(0, e, f);
  for (int k = 0; k + 1 < nz; ++k) {
    st.advance(k, rec, nz, e, f);
    ef.put(k + 1, e, f);
  }
  long long o = ((long long)it.c * nz + (nz - 1)) * nb + it.b;
  long long om = ((long long)it.c * (nz - 1) + (nz - 2)) * nb + it.b;
  for (int k = nz - 1; k >= 0; --k, o -= nb) {
    double v[TriN79::NST];
    ef.get(k, e, f);
    if (k == nz - 1) {
      st.top(rec, nz, e, f, v);
    } else {
      st.back(k, rec, nz, e, f, v);
      __builtin_nontemporal_store((TIO)v[2], outp<TIO>(a.o[4]) + om);
      __builtin_nontemporal_store((TIO)v[3], outp<TIO>(a.o[5]) + om);
      om -= nb;
    }
    const double idr = bc * rec[REC_HDR + k];
    __builtin_nontemporal_store((TIO)idr, outp<TIO>(a.o[0]) + o);
    __builtin_nontemporal_store((TIO)v[0], outp<TIO>(a.o[1]) + o);
    __builtin_nontemporal_store((TIO)v[1], outp<TIO>(a.o[2]) + o);
    __builtin_nontemporal_store((TIO)(idr * invmu + 2 * v[0] + 2 * v[1]), outp<TIO>(a.o[3]) + o);
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
  TriZq st;
  st.template init<TIO>(rec, a, it.c, it.b);
  const double bc = st.band_const(), invmu = rec[S_INVMU];

  double e, f;
  st.first(rec, m, e, f);
  ef.put(0, e, f);
  for (int k = 0; k < m; ++k) {
    st.advance(k, rec, m, e, f);
    ef.put(k + 1, e, f);
  }
  double v[TriZq::NST];
  ef.get(m, e, f);
  st.top(rec, m, e, f, v);
  long long o = ((long long)it.c * m + (m - 1)) * nb + it.b;
  for (int k = m - 1; k >= 0; --k, o -= nb) {
    ef.get(k, e, f);
    st.back(k, rec, m, e, f, v);
    const double S = bc * rec[REC_HDR + k];
    __builtin_nontemporal_store((TIO)S, outp<TIO>(a.o[0]) + o);                                   // :219
    __builtin_nontemporal_store((TIO)v[0], outp<TIO>(a.o[1]) + o);                                // :198
    __builtin_nontemporal_store((TIO)v[1], outp<TIO>(a.o[2]) + o);                                // :200
    __builtin_nontemporal_store((TIO)(S * invmu + 2 * v[1] + 2 * v[0]), outp<TIO>(a.o[3]) + o);   // :202
    __builtin_nontemporal_store((TIO)v[2], outp<TIO>(a.o[4]) + o);                                // :197
    __builtin_nontemporal_store((TIO)v[3], outp<TIO>(a.o[5]) + o);                                // :199
    __builtin_nontemporal_store((TIO)(S * invmu + 2 * v[3] + 2 * v[2]), outp<TIO>(a.o[6]) + o);   // :201
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
