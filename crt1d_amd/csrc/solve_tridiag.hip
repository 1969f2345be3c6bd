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
#include <type_traits>

#include "crt_internal.hpp"
#include "tri_schemes.hpp"

namespace crt {
namespace {

constexpr int TB = 64;  // one wavefront per workgroup

// Forward-sweep pairs of one lane.  EfLds: a lane-private column of a [row][TB] LDS array.
struct EfLds {
  double* base;
  template <typename A>
  __device__ inline EfLds(const A&, const Item&, double* lds_ef) : base(lds_ef + threadIdx.x) {}
  __device__ inline void put(int k, double e, double f) {
    base[(2 * k) * TB] = e;
    base[(2 * k + 1) * TB] = f;
  }
  __device__ inline void get(int k, double& e, double& f) const {
    e = base[(2 * k) * TB];
    f = base[(2 * k + 1) * TB];
  }
};

// EfOut: nz beyond what a CU's LDS holds (16 nz bytes per lane).  Pair k is parked in row k of this lane's own output
// elements, which the back sweep overwrites with the results of level k right after reading the pair back, so no
// extra memory is needed.  fp64 outputs: e -> I_df_d, f -> I_df_u.  f32 storage: the two 32-bit halves of e -> I_df_d, I_df_u and
// of f -> I_dr, F (bit-exact, so the f32 variant stays the rounded fp64 result).
template <typename TIO>
struct EfOut;
template <>
struct EfOut<double> {
  double *pe, *pf;
  long long nb;
  __device__ inline EfOut(const SolveArgs& a, const Item& it, double*) : nb(a.nb) {
    const long long o = (long long)it.c * a.nz * a.nb + it.b;
    pe = static_cast<double*>(a.o[1]) + o;
    pf = static_cast<double*>(a.o[2]) + o;
  }
  __device__ inline void put(int k, double e, double f) {
    pe[k * nb] = e;
    pf[k * nb] = f;
  }
  __device__ inline void get(int k, double& e, double& f) const {
    e = pe[k * nb];
    f = pf[k * nb];
  }
};
template <>
struct EfOut<float> {
  int* p[4];
  long long nb;
  __device__ inline EfOut(const SolveArgs& a, const Item& it, double*) : nb(a.nb) {
    const long long o = (long long)it.c * a.nz * a.nb + it.b;
    for (int i = 0; i < 4; ++i) p[i] = static_cast<int*>(a.o[i]) + o;
  }
  __device__ inline void put(int k, double e, double f) {
    p[1][k * nb] = __double2hiint(e);
    p[2][k * nb] = __double2loint(e);
    p[0][k * nb] = __double2hiint(f);
    p[3][k * nb] = __double2loint(f);
  }
  __device__ inline void get(int k, double& e, double& f) const {
    e = __hiloint2double(p[1][k * nb], p[2][k * nb]);
    f = __hiloint2double(p[0][k * nb], p[3][k * nb]);
  }
};

// all NOUT outputs of level k for this lane's band (same expressions as the column-tile kernel: S::emit)
template <class S, typename TIO>
__device__ inline void emit_level(const SolveArgs& a, const double* rec, const Item& it, int k, double bc, double invmu,
                                  const double (&v)[S::NST]) {
  d2 st2[S::NST], o2[S::NOUT];
#pragma unroll
  for (int q = 0; q < S::NST; ++q) st2[q] = d2{v[q], v[q]};
  S::emit(rec, a.nz, k, d2{bc, bc}, invmu, st2, o2);
#pragma unroll
  for (int i = 0; i < S::NOUT; ++i) {
    const int rows = S::out_rows(i, a.nz);
    if (k < rows) __builtin_nontemporal_store((TIO)o2[i].x, outp<TIO>(a.o[i]) + ((long long)it.c * rows + k) * a.nb + it.b);
  }
}

// ------------------------------------------------------------------------------------------
// One lane = one (column, band) system; scheme arithmetic in tri_schemes.hpp (shared with the column-tile kernel).
//   n79: crt1d/solvers/_solve_n79.py:70-155     zq: crt1d/solvers/_solve_zq.py:74-219
// The pair of the last even row stays in registers; pairs 0 .. K-2 go through EF.
template <class S, typename TIO, class EF>
__device__ __forceinline__ void tri_wave_body(const SolveArgs& a, const Item& it, const double* rec, double* lds_ef) {
  EF ef(a, it, lds_ef);
  const int nz = a.nz, K = S::rows(nz);
  S st;
  st.template init<TIO>(rec, a, it.c, it.b);
  const double bc = st.band_const(), invmu = rec[S_INVMU];

  double e, f;
  typename S::St fs;
  st.first(rec, nz, fs);
  st.pair(fs, e, f);
  for (int k = 0; k + 1 < K; ++k) {
    ef.put(k, e, f);
    tri_step(st, k, rec, nz, fs, e, f);  // the same operations, re-seeding schedule included, as the column-tile kernels
  }
  double v[S::NST];
  st.top(rec, nz, e, f, v);
  emit_level<S, TIO>(a, rec, it, K - 1, bc, invmu, v);
  for (int k = K - 2; k >= 0; --k) {
    ef.get(k, e, f);
    st.back(k, rec, nz, e, f, v);
    emit_level<S, TIO>(a, rec, it, k, bc, invmu, v);
  }
}

// a wave can span two columns (nb < 64): the uniform-dLAI object is chosen per lane (divergent only in such waves)
template <class S, typename TIO, bool USE_LDS, class EF>
__global__ __launch_bounds__(TB) void k_tri_wave(SolveArgs a, int rec_lds_doubles) {
  extern __shared__ double lds[];
  const Item it = locate<TB, 1>(a.ncol, a.nb);
  const double* rec = stage_records<TB, 1, USE_LDS>(a, it, lds);
  if (!it.active) return;
  typedef typename UniformOf<S>::type SU;
  if constexpr (!std::is_same<S, SU>::value) {
    if (rec[S_UNIF] != 0.0) {
      tri_wave_body<SU, TIO, EF>(a, it, rec, lds + rec_lds_doubles);
      return;
    }
  }
  tri_wave_body<S, TIO, EF>(a, it, rec, lds + rec_lds_doubles);
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

template <class S, typename TIO>
int launch_wave(const SolveArgs& a, hipStream_t s) {
  const long long items = (long long)a.ncol * a.nb;
  const long long nblk = (items + TB - 1) / TB;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  const size_t ef_lds = (size_t)(S::rows(a.nz) - 1) * 2 * TB * sizeof(double);
  const long long cols_per_block = (TB - 1) / a.nb + 2;
  const size_t rec_bytes = (size_t)cols_per_block * a.reclen * sizeof(double);
  const bool ef_in_lds = ef_lds + (rec_bytes <= 32 * 1024 ? rec_bytes : 0) <= MAX_WG_LDS;
  const bool use_lds = rec_bytes <= (ef_in_lds ? 32 * 1024 : 64 * 1024);  // records in LDS, else read through the cache
  const size_t rec_doubles = use_lds ? rec_bytes / sizeof(double) : 0;
  const size_t sh = (ef_in_lds ? ef_lds : 0) + rec_doubles * sizeof(double);
  if (!ef_in_lds && (S::NOUT < 4 || a.nz < S::rows(a.nz) - 1)) return CRT_ERR_UNSUPPORTED;  // nowhere to park the pairs
  auto go = [&](auto kern) {
    const int st = set_lds_limit(kern, sh);
    if (st != CRT_OK) return st;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(TB), sh, s, a, (int)rec_doubles);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_tri_wave<%s,%s> pairs in %s", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", ef_in_lds ? "LDS" : "output rows");  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  if (ef_in_lds) return use_lds ? go(k_tri_wave<S, TIO, true, EfLds>) : go(k_tri_wave<S, TIO, false, EfLds>);
  return use_lds ? go(k_tri_wave<S, TIO, true, EfOut<TIO>>) : go(k_tri_wave<S, TIO, false, EfOut<TIO>>);
}

template <typename TIO>
int launch_tridiag_io(int scheme, const SolveArgs& a, hipStream_t s, int force) {
  if (force != 1) {  // column-tile / pipeline kernels (tri_tile_impl.hpp) when they apply; force = 1 keeps the per-wave kernels
    bool done = false;
    const int st = launch_tridiag_tile(scheme, a, s, done);
    if (st != CRT_OK || done) return st;
  }
  if (scheme == CRT_SCHEME_N79) return launch_wave<TriN79, TIO>(a, s);
  if (scheme == CRT_SCHEME_ZQ) return launch_wave<TriZq, TIO>(a, s);
  return CRT_ERR_BAD_ARG;
}

// zq_pa computational-grid solve when the column-tile kernel does not apply (nb > 1024)
int launch_zqpa_wave(const SolveArgs& g, hipStream_t s) { return launch_wave<TriZqPa, double>(g, s); }

int launch_tridiag(int scheme, const SolveArgs& a, hipStream_t s, int force) {
  return a.f32 ? launch_tridiag_io<float>(scheme, a, s, force) : launch_tridiag_io<double>(scheme, a, s, force);
}

}  // namespace crt
