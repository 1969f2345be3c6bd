// Internal declarations shared by the gfx950 kernels of libcrt1d_hip.so.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crt1d_hip.h"

namespace crt {

// ------------------------------------------------------------------------------------------
// Per-column record written by the column-precompute kernel (K0) into the caller's workspace
// and staged into LDS by every solve kernel: a 16-double header + nvec vectors of nz doubles.
//
//   header: band-independent scalars the reference computes before its band loop
//   vectors (by scheme):
//     2s, 4s, g77, bf : lai, ekl = exp(-K_b lai)
//     bl              : lai, ekl, tau_d(lai_j)                       (_solve_bl.py:31-37)
//     n79             : tbcum = ekl, 1 - tb, 1 - td, fracsun/(fracsun dlai), 1/(fracsun dlai), fracsha/(fracsha dlai), 1/(1 - td)
//                                                                     (_solve_n79.py:40-59)
//     zq              : ekl                                           (_solve_zq.py:130)
//     zq_pa           : beam fraction on the M computational layers, ekl, interpolation index, weight
//                                                                     (_solve_zq_pa.py:159-168,357-362)
enum RecScalar {
  S_KB = 0,     // K_b = G(psi)/cos(psi)              model.py:291-293
  S_MU = 1,     // cos(psi)
  S_G = 2,      // G(psi)
  S_MUBAR = 3,  // int cos sin / G                     _solve_2s.py:32
  S_GINT1 = 4,  // int_0^{mu_s} G(acos m) dm           _solve_4s.py:148
  S_GINT2 = 5,  // int_{mu_s}^1                        _solve_4s.py:149
  S_DLM = 6,    // |mean(dlai[dlai != 0])|             _solve_zq.py:50
  S_TAUI = 7,   // tau_d(dlai_mean)                    _solve_zq.py:51
  S_TPSI = 8,   // exp(-K_b dlai_mean)                 _solve_zq.py:52
  S_COS2 = 9,   // cos^2(radians(mla))                 _solve_2s.py:28,68
  S_LT = 10,    // lai[0], total LAI
  S_INVMU = 11, // 1/cos(psi)
  S_UNIF = 12,  // 1.0 when every dlai_j equals S_DL to within 4 ulp of LAI (all reference LAI generators), else 0.0
  S_DL = 13,    // (lai[0] - lai[nz-1]) / (nz - 1)
  S_M = 14,     // zq_pa: number of computational layers min(100, nz)   _solve_zq_pa.py:95
  REC_HDR = 16
};

__host__ __device__ inline int rec_nvec(int scheme) {
  switch (scheme) {
    case CRT_SCHEME_BL: return 3;
    case CRT_SCHEME_N79: return 7;
    case CRT_SCHEME_ZQ_PA: return 4;
    case CRT_SCHEME_ZQ: return 1;
    default: return 2;
  }
}
__host__ __device__ inline int rec_len(int scheme, int nz) { return REC_HDR + rec_nvec(scheme) * nz; }

// ------------------------------------------------------------------------------------------
// kernel argument blocks (passed by value)
struct ColArgs {
  int ncol, nz, scheme, tau_d_method;
  double mu_s;
  const double* psi;
  const double* lai;
  const double* mla;
  const int32_t* g_kind;
  const double* g_param;
  const double* g_at_psi;
  const double* g_table;
  double* ws;  // [ncol][rec_len]
};

// Spectra and output profiles are stored as TIO = double (crt_hip_*_f64) or float (crt_hip_*_f32); the per-column
// geometry, the K0 records and ALL arithmetic are fp64 in both cases (the f32 entry points halve the HBM bytes, they
// do not lower the precision of the solve: results are the fp64 results rounded once to fp32).
struct SolveArgs {
  int ncol, nb, nz, reclen;
  long long col_stride;
  const double* ws;  // K0 records
  const void* I_dr0;
  const void* I_df0;
  const void* leaf_r;
  const void* leaf_t;
  const void* soil_r;
  void* o[7];  // I_dr, I_df_d, I_df_u, F, x0, x1, x2
  double mu_s;
  int f32;     // 0: TIO = double, 1: TIO = float
  // kernel-selection overrides of THIS call (crt_options.tune; 0 = automatic; validated by solve_impl, api.hip: out-of-range values and
  // non-zero reserved keys are CRT_ERR_BAD_ARG): [0] LDS bytes a closed-form tile may take (<= 160 KB), [1] force T of k_tile (<= 64),
  // [2] flags (bit0 __syncthreads barriers, bit1 generic flush, bit2 no pipeline, bit3 no generic-flush pipeline, value 16: four-pair
  // store role for narrow tridiagonal pipelines), [3] store waves of k_pipe (<= 12), [4] T of k_pipe (<= 32); [8] M (8 / 12 / 16),
  // [9] T (4 / 8 / 12), [10] kernel family of the tridiagonal kernels (1 no pipeline, 2 double-buffered, 3 register-staged, 4 generic
  // pipeline; zq_pa: 1 two-kernel path, 5 round-2 fused kernel, 6 / 7 double-buffered / register-staged k_zqpa_pipe2), [11] their store
  // waves (<= 12) (tri_tile_impl.hpp, tri_zqpa.hip); [12] smallest nb that takes the tile / pipeline kernels (0 = default);
  // [13] 1 = no flat fused flush for odd nb (per-array generic flush instead), 2 / 3 = its part-line / whole-line form (0: whole lines for
  // zq and zq_pa, part-lines for n79); [5] column packing of narrow spectra (k_pipe_pack, packed k_tri_pipe): 1 = off, 2 = closed forms also
  // above 32 bands, [6] compute waves of a pack (<= 4); [7], [14], [15] reserved (zero)
  int tune[CRT_NTUNE];
};

// name of the solve kernel the last launch_* call of this thread chose (crt_hip_last_kernel; reporting only)
void note_kernel(const char* fmt, ...);
const char* last_kernel();

template <typename TIO>
__device__ inline double ldio(const void* p, long long i) {
  return (double)reinterpret_cast<const TIO*>(p)[i];
}
template <typename TIO>
__device__ inline TIO* outp(void* p) {
  return reinterpret_cast<TIO*>(p);
}

// ------------------------------------------------------------------------------------------
// G(psi) closed forms on device (crt1d/leaf_angle.py:118-202); `cs`, `sn` = cos/sin(psi).
// Ellipsoidal forms use sqrt(x^2 cos^2 + sin^2)/p2 == sqrt(x^2 + tan^2)/p2 * cos, finite at pi/2.
__device__ inline double ellipsoidal_p2(double x) {
  if (x > 1.0) {
    double e = sqrt(1.0 - 1.0 / (x * x));
    return x + log((1.0 + e) / (1.0 - e)) / (2.0 * e * x);
  }
  double e = sqrt(1.0 - x * x);
  return x + asin(e) / e;
}

// G(psi) = G_eval(kind, param, G_den(kind, param), cos psi, sin psi).  G_den is the part that does not depend on the angle
// (the ellipsoidal normalisations); K0 evaluates it once per thread and G_eval at each of its quadrature nodes.
// x^-0.733 as exp(-0.733 log x): ~3 ulp instead of pow's < 1 ulp, a third of its instructions (K0 is bound by exactly these).
__device__ inline double G_den(int kind, double param) {
  switch (kind) {
    case CRT_G_ELLIPSOIDAL: return param == 1.0 ? 0.0 : ellipsoidal_p2(param);
    case CRT_G_ELLIPSOIDAL_APPROX: return param + 1.774 * exp(-0.733 * log(param + 1.182));
    default: return 0.0;
  }
}

__device__ inline double G_eval(int kind, double param, double den, double cs, double sn) {
  switch (kind) {
    case CRT_G_HORIZONTAL: return cs;
    case CRT_G_SPHERICAL: return 0.5;
    case CRT_G_VERTICAL: return 0.63661977236758134308 * sn;  // 2/pi
    case CRT_G_ELLIPSOIDAL: {
      if (param == 1.0) return 0.5;
      return sqrt(param * param * cs * cs + sn * sn) / den;
    }
    case CRT_G_ELLIPSOIDAL_APPROX: return sqrt(param * param * cs * cs + sn * sn) / den;
    case CRT_G_ELLIPSOIDAL_APPROX_BONAN: {
      double chil = fmin(fmax(param, -0.4), 0.6);
      double phi1 = 0.5 - 0.633 * chil - 0.330 * chil * chil;
      double phi2 = 0.877 * (1.0 - 2.0 * phi1);
      return phi1 + phi2 * cs;
    }
    default: return 0.0 / 0.0;
  }
}

__device__ inline double G_closed(int kind, double param, double cs, double sn) {
  return G_eval(kind, param, G_den(kind, param), cs, sn);
}

// ------------------------------------------------------------------------------------------
// Work decomposition of every solve kernel: one lane owns VEC adjacent bands of one column and
// sweeps the canopy levels; consecutive lanes own consecutive bands, so each per-level store of a
// wave is one contiguous 512-B (VEC=1) or 1-KiB (VEC=2) segment of the band-contiguous output.
// A block covers BLOCK consecutive (column, band-group) items, i.e. at most
// (BLOCK-1)/nbv + 2 columns, whose K0 records are contiguous in the workspace and are staged
// into LDS with one coalesced copy.
struct Item {
  int c;          // column
  int b;          // first band
  int c_first;    // first column of this block
  bool active;
};

template <int BLOCK, int VEC>
__device__ inline Item locate(int ncol, int nb) {
  const int nbv = nb / VEC;
  const long long first = (long long)blockIdx.x * BLOCK;
  const long long item = first + threadIdx.x;
  Item it;
  it.c_first = (int)(first / nbv);
  it.active = item < (long long)ncol * nbv;
  const long long itc = it.active ? item : first;
  it.c = (int)(itc / nbv);
  it.b = (int)(itc - (long long)it.c * nbv) * VEC;
  return it;
}

// copy the block's column records global -> LDS (USE_LDS) and return this lane's record base
template <int BLOCK, int VEC, bool USE_LDS>
__device__ inline const double* stage_records(const SolveArgs& a, const Item& it, double* lds) {
  if constexpr (USE_LDS) {
    const int nbv = a.nb / VEC;
    const long long first = (long long)blockIdx.x * BLOCK;
    long long last = first + BLOCK - 1;
    const long long total = (long long)a.ncol * nbv;
    if (last >= total) last = total - 1;
    const int c_last = (int)(last / nbv);
    const int n = (c_last - it.c_first + 1) * a.reclen;
    const double* src = a.ws + (long long)it.c_first * a.reclen;
    for (int i = threadIdx.x; i < n; i += BLOCK) lds[i] = src[i];
    __syncthreads();
    return lds + (it.c - it.c_first) * a.reclen;
  } else {
    return a.ws + (long long)it.c * a.reclen;
  }
}

// streaming (write-once) stores: outputs are never re-read by the kernel
template <typename TIO, int VEC>
__device__ inline void store_stream(TIO* p, const double (&v)[VEC]) {
  if constexpr (VEC == 1) {
    __builtin_nontemporal_store((TIO)v[0], p);
  } else {
    typedef TIO vt __attribute__((ext_vector_type(2)));
    vt t;
    t.x = (TIO)v[0];
    t.y = (TIO)v[1];
    __builtin_nontemporal_store(t, reinterpret_cast<vt*>(p));
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits s_waitcnt vmcnt(0), i.e.
// every wave would drain its outstanding global stores (a full HBM write round trip) twice per tile;
// the tile protocol only needs the LDS writes/reads of the other waves to have completed.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 1/x for normal, positive-or-negative x well inside the exponent range: hardware seed + two Newton steps
// (<= 1 ulp; skips the scaling / fix-up of the IEEE division sequence, ~5 instructions instead of ~10)
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// ------------------------------------------------------------------------------------------
// Polynomial steps with the coefficient in a SCALAR register.  hipcc (ROCm 7.2) turns a Horner step fma(acc, r, C) whose constant it keeps
// in a VGPR into v_mov_b64 tmp, C + v_fmac_f64 tmp, acc, r -- two issue slots per step (ISA of k_pipe<4s>, round 3: ocml's exp was 35
// VALU instructions, 11 of them such copies).  v_fma_f64 takes one SGPR pair as a source, so the step is ONE instruction when the
// constant is pinned to scalar registers; the asm is not volatile, the compiler still schedules and interleaves it freely.
__device__ __forceinline__ double fma_vvs(double a, double b, double c_uniform) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
  return d;
}
__device__ __forceinline__ double fma_vsv(double a, double c_uniform, double b) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(c_uniform), "v"(b));
  return d;
}
__device__ __forceinline__ double mul_vs(double a, double c_uniform) {
  double d;
  asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c_uniform));
  return d;
}

// e^x in 18 VALU instructions, <= 1 ulp: k = rint(x log2 e), r = x - k ln2 (two-term Cody-Waite, exact first product for |k| < 2^20),
// e^r = 1 + r + r^2 q(r) with q the degree-9 interpolant of (e^r - 1 - r) / r^2 at the Chebyshev nodes of [-ln2/2, ln2/2] (coefficients
// computed for this file with mpmath at 60 digits and rounded to double: 1.6e-17 relative in exact arithmetic), scaled by v_ldexp_f64 --
// which underflows to 0 and overflows to inf by itself, so no range checks.  NaN in, NaN out.
__device__ __forceinline__ double fexp(double x) {
  const double k = __builtin_rint(mul_vs(x, 1.44269504088896338700e+00));
  double r = fma_vsv(k, -6.93147180369123816490e-01, x);
  r = fma_vsv(k, -1.90821492927058770002e-10, r);
  double q = fma_vvs(r, 0x1.af389ecfc4b9cp-26, 0x1.28917c89a43a7p-22);
  q = fma_vvs(q, r, 0x1.71de0db2f6b19p-19);
  q = fma_vvs(q, r, 0x1.a019b9149a41cp-16);
  q = fma_vvs(q, r, 0x1.a01a01a7c2efep-13);
  q = fma_vvs(q, r, 0x1.6c16c17889ef1p-10);
  q = fma_vvs(q, r, 0x1.11111111109b5p-7);
  q = fma_vvs(q, r, 0x1.5555555553d68p-5);
  q = fma_vvs(q, r, 0x1.5555555555556p-3);
  q = fma_vvs(q, r, 0x1.0000000000001p-1);
  q = __builtin_fma(q, r, 1.0);
  q = __builtin_fma(q, r, 1.0);
  return __builtin_ldexp(q, (int)k);
}

// e^x - 1 without the cancellation of fexp(x) - 1 at small |x|: inside the reduced range (|x| <= ln2 / 2, k = 0) e^x - 1 = x + x^2 q(x) with the
// SAME polynomial q as fexp -- relative error ~1e-16 down to x -> 0 -- and fexp(x) - 1 beyond, where e^x is <= 0.71 or >= 1.41.
// Used where a quantity of the form 1 - e^{-k L} is divided by L afterwards (n79's per-leaf-area absorption at small dlai).
__device__ __forceinline__ double fexpm1(double x) {
  if (__builtin_fabs(x) > 0.34657359027997264) return fexp(x) - 1.0;
  double q = fma_vvs(x, 0x1.af389ecfc4b9cp-26, 0x1.28917c89a43a7p-22);
  q = fma_vvs(q, x, 0x1.71de0db2f6b19p-19);
  q = fma_vvs(q, x, 0x1.a019b9149a41cp-16);
  q = fma_vvs(q, x, 0x1.a01a01a7c2efep-13);
  q = fma_vvs(q, x, 0x1.6c16c17889ef1p-10);
  q = fma_vvs(q, x, 0x1.11111111109b5p-7);
  q = fma_vvs(q, x, 0x1.5555555553d68p-5);
  q = fma_vvs(q, x, 0x1.5555555555556p-3);
  q = fma_vvs(q, x, 0x1.0000000000001p-1);
  return __builtin_fma(x * x, q, x);
}

// sin and cos of a moderate argument (|th| < ~1e6) in ~35 instructions: two-term Cody-Waite reduction by pi/2 (the first product is
// exact in the FMA for |th| < 2^20, the reduced argument is good to ~1.2e-16 absolute) and the minimax polynomials of fdlibm's
// __kernel_sin / __kernel_cos on [-pi/4, pi/4].  Absolute error <= ~2e-16: the oscillatory modes of the four-stream scheme need
// nothing better (ocml's sincos carries a Payne-Hanek path and double-double reduction: ~200 instructions).
__device__ inline void fast_sincos(double th, double& sn, double& cs) {
  const double k = __builtin_rint(mul_vs(th, 0.63661977236758134308));
  double r = fma_vsv(k, -1.57079632679489655800e+00, th);
  r = fma_vsv(k, -6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = fma_vvs(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma_vvs(z, ps, 2.75573137070700676789e-06);
  ps = fma_vvs(z, ps, -1.98412698298579493134e-04);
  ps = fma_vvs(z, ps, 8.33333333332248946124e-03);
  ps = fma_vvs(z, ps, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(z * r, ps, r);
  double pc = fma_vvs(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma_vvs(z, pc, -2.75573143513906633035e-07);
  pc = fma_vvs(z, pc, 2.48015872894767294178e-05);
  pc = fma_vvs(z, pc, -1.38888888888741095749e-03);
  pc = fma_vvs(z, pc, 4.16666666666666019037e-02);
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int n = (int)k;
  const bool swap = (n & 1) != 0;
  const double ss = swap ? cr : sr, cc = swap ? sr : cr;
  sn = (n & 2) ? -ss : ss;
  cs = ((n + 1) & 2) ? -cc : cc;
}

// Levels at which the per-level exponentials are re-evaluated exactly when a column has uniform dlai; in between
// they advance by one multiplication (e^{-h(L - dl)} = e^{-hL} e^{h dl}).  At most 7 products since the last exact
// value -> <= 8 ulp; the rule depends on j only, so every kernel variant produces the same bits.
__device__ inline bool exact_level(int j) { return (j & 7) == 0; }

// ------------------------------------------------------------------------------------------
// Integrated-output path (no profiles written): per column and band group g (weights w_g[b], diagnostics.py:81)
//   Phi_g(j) = sum_b w_g[b] (I_dr + I_df_d - I_df_u)(j, b)            net downward flux at level j
//   aI_g(k)  = Phi_g(k+1) - Phi_g(k)                                  == sum_b w_g[b] aI(k, b)      (model.py:609)
//   aI_dr    = (1 - e^{-K_b dlai_k}) e^{-K_b lai_{k+1}} sum_b w_g[b] (1 - r - t)[b] I_dr0[b]       (model.py:617-621;
//              I_dr factorises into band x level, so this needs ONE reduction per column)
//   aI_sl = (aI - aI_dr) f_sl(k) + aI_dr,  aI_sh = (aI - aI_dr)(1 - f_sl(k))                       (model.py:628-634)
// so only ngroup values per level are reduced across the bands (wave shuffles + one LDS pass per column).
constexpr int INT_MAXG = 4;

struct IntArgs {
  const double* lai;     // [ncol][nz]
  const double* band_w;  // [ngroup][nb]
  int ngroup;
  double* aI;            // [ncol][nz-1][ngroup]
  double* aI_sl;
  double* aI_sh;
  double* totals;        // [ncol][ngroup][4]: incoming, reflected, transmitted, soil-reflected (may be NULL)
  // optional, all or none (crt_bandsum_out): direct-beam part of the absorption [ncol][nz-1][ngroup] and the band-integrated level
  // profiles [ncol][nz][ngroup] of I_dr, I_df_d, I_df_u, F, I_d (diagnostics.py:84-91)
  double* aI_dr;
  double* L_dr;
  double* L_dn;
  double* L_up;
  double* L_F;
  double* L_Id;
};

// Sum over each 16-lane DPP row, result in every lane of the row: 4 steps of (2 x v_mov_b32 dpp + v_add_f64), pure VALU
// (a __shfl_xor butterfly over the whole wave goes through ds_bpermute: LDS-crossbar latency on every step, measured
// 1.6 ms per 3e6 solves for the integrated 2s kernel against 1.0 ms for the kernel that writes full profiles).
template <int CTRL>
__device__ inline double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ inline double row_sum16(double v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}
// ... plus row_bcast15 / row_bcast31: the wave total ends up in lane 63 (only that lane is valid)
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add_rows(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ inline double wave_sum_lane63(double v) {
  v = row_sum16(v);
  v = dpp_add_rows<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v = dpp_add_rows<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return v;
}
// FOUR per-lane values -> their four wave totals in ONE register: the lanes of DPP row 0 end up with sum(a), row 1 with sum(c),
// row 2 with sum(b), row 3 with sum(d).  v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even rows between
// two registers, so each swap + add halves the number of registers while summing across the halves; one row_sum16 finishes all
// four at once: 2 x 3 + 3 + 12 = 21 VALU instructions for four totals, against 4 x 18 for four wave_sum_lane63.
__device__ inline double swap_add32(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ inline double swap_add16(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ inline double wave_sum4(double a, double b, double c, double d) {
  return row_sum16(swap_add16(swap_add32(a, b), swap_add32(c, d)));
}
// index (0..3 = a, b, c, d) of the value whose total a lane of DPP row `row` holds after wave_sum4
__device__ inline int wave_sum4_slot(int row) { return ((row & 1) << 1) | (row >> 1); }
// wave totals of v[0..N) written to dst[0..nvalid) (LDS or global) by the first lane of each row, four values per pass
template <int N>
__device__ inline void wave_sum_store(const double (&v)[N], double* dst, int nvalid, int lane) {
  const int slot = wave_sum4_slot(lane >> 4);
#pragma unroll
  for (int j = 0; j < (N + 3) / 4; ++j) {
    const double z = wave_sum4(v[4 * j], 4 * j + 1 < N ? v[4 * j + 1] : 0.0, 4 * j + 2 < N ? v[4 * j + 2] : 0.0, 4 * j + 3 < N ? v[4 * j + 3] : 0.0);
    const int idx = 4 * j + slot;
    if ((lane & 15) == 0 && idx < nvalid) dst[idx] = z;
  }
}

// TWO per-lane values -> their totals over each HALF of the wave (lanes 0-31 / 32-63), for kernels that give a column to each
// half: lanes of DPP row 0 end up with sum(x) over the lower half, row 1 with sum(y) over the lower half, rows 2 / 3 the same for the
// upper half.  3 + 12 = 15 VALU instructions for two values of two columns.
__device__ inline double half_sum2(double x, double y) { return row_sum16(swap_add16(x, y)); }
template <int N>
__device__ inline void half_sum_store(const double (&v)[N], double* dst, int nvalid, int lane) {  // dst: this lane's half's area
  const int which = (lane >> 4) & 1;
#pragma unroll
  for (int j = 0; j < (N + 1) / 2; ++j) {
    const double z = half_sum2(v[2 * j], 2 * j + 1 < N ? v[2 * j + 1] : 0.0);
    const int idx = 2 * j + which;
    if ((lane & 15) == 0 && idx < nvalid) dst[idx] = z;
  }
}

__device__ inline double wave_sum_all(double v) {  // sum over the 64 lanes (used once per column only)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LDS layout of the partial sums: nslot = waves (ROWS = false: wave totals, 6 DPP steps per value) or 4 * waves
// (ROWS = true: one partial per 16-lane row, 4 DPP steps per value, 4x the LDS):
//   part[(j * nslot + slot) * INT_MAXG + g];  ends[((e * nslot + slot) * 2 + q) * INT_MAXG + g] with e = 0 ground / 1 top,
//   q = 0 (I_dr + I_df_d) / 1 (I_df_u);  pdr[wave * INT_MAXG + g]
struct IntLds {
  double* part;
  double* ends;
  double* pdr;
  double* lev;   // PROF: [nz][nwave][2][INT_MAXG] wave sums of w I_df_d, w I_df_u per level
  double* pi0;   // PROF: [nwave][INT_MAXG] wave sums of w I_dr0 (I_dr factorises into band x level)
};
// carve the integrated kernels' LDS area (after the record) into its parts
__device__ inline IntLds int_lds_carve(double* base, int nz, int nwave, bool prof) {
  IntLds L;
  L.part = base;
  L.ends = L.part + (size_t)nz * nwave * INT_MAXG;
  L.pdr = L.ends + (size_t)2 * nwave * 2 * INT_MAXG;
  L.lev = L.pdr + (size_t)nwave * INT_MAXG;
  L.pi0 = L.lev + (prof ? (size_t)nz * nwave * 2 * INT_MAXG : 0);
  return L;
}

template <bool ROWS, bool PROF = false>
__device__ inline void int_accumulate(const IntLds& L, int nwave, int wave, int lane, int nz, int j, int ng, const double (&w)[INT_MAXG],
                                      bool active, double idr, double dn, double up) {
  if constexpr (!ROWS) {
    if constexpr (PROF) {  // the level sums of the two diffuse streams (the direct beam's follow from one sum per column)
      const int g2 = wave_sum4_slot(lane >> 4);
      const double dd = active ? dn : 0.0, uu = active ? up : 0.0;
      const double zd = wave_sum4(w[0] * dd, w[1] * dd, w[2] * dd, w[3] * dd);
      const double zu = wave_sum4(w[0] * uu, w[1] * uu, w[2] * uu, w[3] * uu);
      if ((lane & 15) == 0 && g2 < ng) {
        L.lev[((j * nwave + wave) * 2 + 0) * INT_MAXG + g2] = zd;
        L.lev[((j * nwave + wave) * 2 + 1) * INT_MAXG + g2] = zu;
      }
    }
    // wave totals of all (up to four) band groups with ONE wave_sum4: 21 VALU instructions per level instead of 18 per group
    // (w[g] = 0 for g >= ngroup and for inactive lanes)
    const int g = wave_sum4_slot(lane >> 4);
    const bool writer = (lane & 15) == 0 && g < ng;
    const double net = active ? idr + dn - up : 0.0;
    const double z = wave_sum4(w[0] * net, w[1] * net, w[2] * net, w[3] * net);
    if (writer) L.part[(j * nwave + wave) * INT_MAXG + g] = z;
    if (j == 0 || j == nz - 1) {
      const int e = j == 0 ? 0 : 1;
      const double dd = active ? idr + dn : 0.0, uu = active ? up : 0.0;
      const double z0 = wave_sum4(w[0] * dd, w[1] * dd, w[2] * dd, w[3] * dd);
      const double z1 = wave_sum4(w[0] * uu, w[1] * uu, w[2] * uu, w[3] * uu);
      if (writer) {
        L.ends[((e * nwave + wave) * 2 + 0) * INT_MAXG + g] = z0;
        L.ends[((e * nwave + wave) * 2 + 1) * INT_MAXG + g] = z1;
      }
    }
    return;
  }
  const int nslot = ROWS ? nwave * 4 : nwave;
  const int slot = ROWS ? wave * 4 + (lane >> 4) : wave;
  const bool writer = ROWS ? (lane & 15) == 0 : lane == 63;
  auto red = [](double v) { return ROWS ? row_sum16(v) : wave_sum_lane63(v); };
  const double net = active ? idr + dn - up : 0.0;
#pragma unroll
  for (int g = 0; g < INT_MAXG; ++g)
    if (g < ng) {
      const double t = red(w[g] * net);
      if (writer) L.part[(j * nslot + slot) * INT_MAXG + g] = t;
    }
  if (j == 0 || j == nz - 1) {
    const int e = j == 0 ? 0 : 1;
#pragma unroll
    for (int g = 0; g < INT_MAXG; ++g)
      if (g < ng) {
        const double t0 = red(active ? w[g] * (idr + dn) : 0.0);
        const double t1 = red(active ? w[g] * up : 0.0);
        if (writer) {
          L.ends[((e * nslot + slot) * 2 + 0) * INT_MAXG + g] = t0;
          L.ends[((e * nslot + slot) * 2 + 1) * INT_MAXG + g] = t1;
        }
      }
  }
}

// after the sweep (call with all threads of the workgroup): combine the partials and write the column's outputs
template <bool ROWS, bool PROF = false>
__device__ inline void int_finish(const IntLds& L, const IntArgs& ia, int nwave, int nz, int c, double Kb, double invmu = 0.0) {
  __syncthreads();
  if constexpr (PROF) {
    const int ngp = ia.ngroup;
    for (int i = threadIdx.x; i < nz * ngp; i += blockDim.x) {
      const int j = i / ngp, g = i - j * ngp;
      double p0 = 0.0, sD = 0.0, sU = 0.0;
      for (int wv = 0; wv < nwave; ++wv) {
        p0 += L.pi0[wv * INT_MAXG + g];
        sD += L.lev[((j * nwave + wv) * 2 + 0) * INT_MAXG + g];
        sU += L.lev[((j * nwave + wv) * 2 + 1) * INT_MAXG + g];
      }
      const double sR = exp(-Kb * ia.lai[(long long)c * nz + j]) * p0;  // sum_b w I_dr0[b] e^{-K_b lai_j}
      const long long o = ((long long)c * nz + j) * ngp + g;
      ia.L_dr[o] = sR;
      ia.L_dn[o] = sD;
      ia.L_up[o] = sU;
      ia.L_F[o] = sR * invmu + 2 * (sU + sD);
      ia.L_Id[o] = sR + sD;
    }
  }
  const int ng = ia.ngroup, nrow = ROWS ? nwave * 4 : nwave;
  const double* lai = ia.lai + (long long)c * nz;
  for (int i = threadIdx.x; i < (nz - 1) * ng; i += blockDim.x) {
    const int k = i / ng, g = i - k * ng;
    double p0 = 0.0, p1 = 0.0, pd = 0.0;
    for (int r = 0; r < nrow; ++r) {
      p0 += L.part[(k * nrow + r) * INT_MAXG + g];
      p1 += L.part[((k + 1) * nrow + r) * INT_MAXG + g];
    }
    for (int wv = 0; wv < nwave; ++wv) pd += L.pdr[wv * INT_MAXG + g];
    const double dl = lai[k] - lai[k + 1];
    const double fsl = exp(-Kb * ((lai[k] + lai[k + 1]) / 2));       // model.py:601-602
    const double adr = (1 - exp(-Kb * dl)) * exp(-Kb * lai[k + 1]) * pd;  // :617-621
    const double a = p1 - p0;                                          // :609
    const double adf = a - adr;
    const long long o = ((long long)c * (nz - 1) + k) * ng + g;
    ia.aI[o] = a;
    ia.aI_sl[o] = adf * fsl + adr;
    ia.aI_sh[o] = adf * (1 - fsl);
    if constexpr (PROF) ia.aI_dr[o] = adr;
  }
  if (ia.totals) {
    for (int i = threadIdx.x; i < ng * 4; i += blockDim.x) {
      const int g = i >> 2, q = i & 3;  // incoming (top, I_d), reflected (top, up), transmitted (ground, I_d), soil-reflected
      const int e = q < 2 ? 1 : 0, which = q & 1;
      double t = 0.0;
      for (int r = 0; r < nrow; ++r) t += L.ends[((e * nrow + r) * 2 + which) * INT_MAXG + g];
      ia.totals[((long long)c * ng + g) * 4 + q] = t;
    }
  }
}

__host__ __device__ inline size_t int_lds_doubles(int nz, int nwave, bool rows = false, bool prof = false) {
  const size_t nslot = rows ? (size_t)nwave * 4 : (size_t)nwave;
  return (size_t)nz * nslot * INT_MAXG + 2 * nslot * 2 * INT_MAXG + (size_t)nwave * INT_MAXG +
         (prof ? (size_t)nz * nwave * 2 * INT_MAXG + (size_t)nwave * INT_MAXG : 0);
}

int launch_closed_int(int scheme, const SolveArgs& a, const IntArgs& ia, hipStream_t s);
int launch_tridiag_int(int scheme, const SolveArgs& a, const IntArgs& ia, hipStream_t s);

// launchers implemented in the .hip files
int launch_colpre(const ColArgs& a, hipStream_t s);
int launch_closed(int scheme, const SolveArgs& a, hipStream_t s, int force);
int launch_tridiag(int scheme, const SolveArgs& a, hipStream_t s, int force);
int init_quadrature(hipStream_t s);
int launch_tau_d(const double* kb_nodes, const double* L, long long n, int method, double* out, hipStream_t s);
int launch_tridiag_tile(int scheme, const SolveArgs& a, hipStream_t s, bool& done);
// per-(scheme, storage type) instantiation units: tri_inst.hip compiled four times
int launch_tri_tile_n79_f64(const SolveArgs& a, hipStream_t s, bool& done);
int launch_tri_tile_n79_f32(const SolveArgs& a, hipStream_t s, bool& done);
int launch_tri_tile_zq_f64(const SolveArgs& a, hipStream_t s, bool& done);
int launch_tri_tile_zq_f32(const SolveArgs& a, hipStream_t s, bool& done);
int launch_tri_int_n79_f64(const SolveArgs& a, const IntArgs& ia, hipStream_t s);
int launch_tri_int_n79_f32(const SolveArgs& a, const IntArgs& ia, hipStream_t s);
int launch_tri_int_zq_f64(const SolveArgs& a, const IntArgs& ia, hipStream_t s);
int launch_tri_int_zq_f32(const SolveArgs& a, const IntArgs& ia, hipStream_t s);
int launch_zqpa(const SolveArgs& a, double* scratch, hipStream_t s);
int launch_zqpa_wave(const SolveArgs& g, hipStream_t s);  // per-wave fallback of the computational-grid solve
__host__ __device__ inline int zqpa_M(int nz) { return nz < 100 ? nz : 100; }
void host_quad_nodes(double mu_s, double* psi_nodes);

}  // namespace crt
