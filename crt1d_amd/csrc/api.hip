// C-ABI entry points of libcrt1d_hip.so (see include/crt1d_hip.h), argument validation,
// the absorption + band-integral epilogue kernel and the HBM bandwidth probes.
#include <math.h>

#include "crt_internal.hpp"

namespace crt {
namespace {

// ------------------------------------------------------------------------------------------
// Epilogue: model.py:573-647 (_calc_absorption) fused with diagnostics.py:39-108 (band sums).
// One wave per (column, layer): lanes stride over the bands (coalesced 512-B row reads), accumulate
// the weighted sums for up to MAXG band groups, then reduce across the wave with DPP shuffles.
constexpr int MAXG = 4;
constexpr int EB = 256;

struct EpiArgs {
  int ncol, nb, nz, ngroup;
  long long col_stride;
  const double* psi;
  const double* lai;
  const int32_t* g_kind;
  const double* g_param;
  const double* g_at_psi;
  const double* leaf_r;
  const double* leaf_t;
  const double* I_dr;
  const double* I_df_d;
  const double* I_df_u;
  const double* band_w;
  double* aI;
  double* aI_sl;
  double* aI_sh;
  double* totals;
};

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(EB) void k_absorb_bandsum(EpiArgs a) {
  const int c = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = EB >> 6;
  const int nz = a.nz, nb = a.nb, ng = a.ngroup;
  const double psi = a.psi[c];
  const int kind = a.g_kind[c];
  const double G = (kind == CRT_G_TABLE) ? a.g_at_psi[c] : G_closed(kind, a.g_param ? a.g_param[c] : 0.0, cos(psi), sin(psi));
  const double Kb = G / cos(psi);
  const double* lai = a.lai + (long long)c * nz;
  const long long cb = (long long)c * nz * nb;
  const double* lr = a.leaf_r + (long long)c * a.col_stride;
  const double* lt = a.leaf_t + (long long)c * a.col_stride;

  for (int k = wave; k < nz - 1; k += nwave) {
    const double dl = lai[k] - lai[k + 1];                    // model.py:248
    const double fsl = exp(-Kb * ((lai[k] + lai[k + 1]) / 2)); // :601-602
    const double absd = 1 - exp(-Kb * dl);                     // :619
    const double* r0 = a.I_dr + cb + (long long)k * nb;
    const double* d0 = a.I_df_d + cb + (long long)k * nb;
    const double* u0 = a.I_df_u + cb + (long long)k * nb;
    double sa[MAXG] = {0, 0, 0, 0}, ssl[MAXG] = {0, 0, 0, 0}, ssh[MAXG] = {0, 0, 0, 0};
    for (int b = lane; b < nb; b += 64) {
      const double idr1 = r0[nb + b];
      const double av = idr1 - r0[b] + d0[nb + b] - d0[b] + u0[b] - u0[nb + b];  // :609
      const double adr = idr1 * absd * (1 - (lr[b] + lt[b]));                    // :617-621
      const double adf = av - adr;
      const double asl = adf * fsl + adr;                                        // :631-633
      const double ash = adf * (1 - fsl);
#pragma unroll
      for (int g = 0; g < MAXG; ++g)
        if (g < ng) {
          const double w = a.band_w[(long long)g * nb + b];
          sa[g] += w * av;
          ssl[g] += w * asl;
          ssh[g] += w * ash;
        }
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < ng) {
        const double ta = wave_sum(sa[g]), tsl = wave_sum(ssl[g]), tsh = wave_sum(ssh[g]);
        if (lane == 0) {
          const long long o = ((long long)c * (nz - 1) + k) * ng + g;
          a.aI[o] = ta;
          a.aI_sl[o] = tsl;
          a.aI_sh[o] = tsh;
        }
      }
  }
  if (wave == 0 && a.totals) {
    // energy-balance terms of diagnostics.py:476-530: incoming, reflected, transmitted, soil-reflected
    const long long top = cb + (long long)(nz - 1) * nb;
    double s[MAXG][4] = {};
    for (int b = lane; b < nb; b += 64) {
#pragma unroll
      for (int g = 0; g < MAXG; ++g)
        if (g < ng) {
          const double w = a.band_w[(long long)g * nb + b];
          s[g][0] += w * (a.I_dr[top + b] + a.I_df_d[top + b]);
          s[g][1] += w * a.I_df_u[top + b];
          s[g][2] += w * (a.I_dr[cb + b] + a.I_df_d[cb + b]);
          s[g][3] += w * a.I_df_u[cb + b];
        }
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < ng) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double t = wave_sum(s[g][i]);
          if (lane == 0) a.totals[((long long)c * ng + g) * 4 + i] = t;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------
// Per-band layer absorption, model.py:573-647: the seven (nz-1, nb) arrays + laim, f_slm of the reference's
// `Model.absorption` dict.  One workgroup per column, lanes over bands, previous level kept in registers.
struct AbsArgs {
  int ncol, nb, nz;
  long long col_stride;
  const double* psi;
  const double* lai;
  const int32_t* g_kind;
  const double* g_param;
  const double* g_at_psi;
  const double* leaf_r;
  const double* leaf_t;
  const double* I_dr;
  const double* I_df_d;
  const double* I_df_u;
  double* o[7];  // aI, aI_df, aI_dr, aI_sh, aI_sl, aI_df_sl, aI_df_sh
  double* laim;
  double* f_slm;
};

__global__ __launch_bounds__(256) void k_absorb(AbsArgs a) {
  const int c = blockIdx.x;
  const int nz = a.nz, nb = a.nb;
  const double psi = a.psi[c];
  const int kind = a.g_kind[c];
  const double G = (kind == CRT_G_TABLE) ? a.g_at_psi[c] : G_closed(kind, a.g_param ? a.g_param[c] : 0.0, cos(psi), sin(psi));
  const double Kb = G / cos(psi);
  const double* lai = a.lai + (long long)c * nz;
  const long long cb = (long long)c * nz * nb;
  const long long cm = (long long)c * (nz - 1) * nb;
  for (int b = threadIdx.x; b < nb; b += 256) {
    const double leaf_a = 1 - (a.leaf_r[(long long)c * a.col_stride + b] + a.leaf_t[(long long)c * a.col_stride + b]);  // :584
    double r0 = a.I_dr[cb + b], d0 = a.I_df_d[cb + b], u0 = a.I_df_u[cb + b];
    for (int k = 0; k < nz - 1; ++k) {
      const long long i1 = cb + (long long)(k + 1) * nb + b;
      const double r1 = a.I_dr[i1], d1 = a.I_df_d[i1], u1 = a.I_df_u[i1];
      const double dl = lai[k] - lai[k + 1];
      const double fsl = exp(-Kb * ((lai[k] + lai[k + 1]) / 2));   // :601-602
      const double av = r1 - r0 + d1 - d0 + u0 - u1;               // :609
      const double adr = r1 * (1 - exp(-Kb * dl)) * leaf_a;        // :617-621
      const double adf = av - adr;                                 // :628
      const double adfsl = adf * fsl, adfsh = adf * (1 - fsl);     // :631-632
      const long long o = cm + (long long)k * nb + b;
      a.o[0][o] = av;
      a.o[1][o] = adf;
      a.o[2][o] = adr;
      a.o[3][o] = adfsh;
      a.o[4][o] = adfsl + adr;
      a.o[5][o] = adfsl;
      a.o[6][o] = adfsh;
      if (b == 0) {
        a.laim[(long long)c * (nz - 1) + k] = (lai[k] + lai[k + 1]) / 2;
        a.f_slm[(long long)c * (nz - 1) + k] = fsl;
      }
      r0 = r1;
      d0 = d1;
      u0 = u1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// bandwidth probes: plain 16-B-per-lane streaming fill / copy, grid-stride
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fill(d2* dst, size_t n2, double v) {
  d2 t;
  t.x = v;
  t.y = v;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(t, dst + i);
}

__global__ __launch_bounds__(256) void k_copy(d2* dst, const d2* src, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

bool scheme_ok(int s) { return s >= 0 && s < CRT_NUM_SCHEMES; }

}  // namespace
}  // namespace crt

using namespace crt;

extern "C" {

int crt_hip_abi_version(void) { return CRT_ABI_VERSION; }

const char* crt_hip_strerror(int st) {
  switch (st) {
    case CRT_OK: return "ok";
    case CRT_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or invalid option)";
    case CRT_ERR_WORKSPACE: return "workspace too small (see crt_hip_workspace_bytes)";
    case CRT_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels (nz too large for LDS, or grid overflow)";
    case CRT_ERR_LAUNCH: return "HIP runtime / launch error";
    case CRT_ERR_SHAPE: return "shape violates a reference assertion";
    default: return "unknown status";
  }
}

size_t crt_hip_workspace_bytes(int scheme, int32_t ncol, int32_t nz) {
  if (!scheme_ok(scheme) || ncol <= 0 || nz <= 0) return 0;
  return (size_t)ncol * (size_t)rec_len(scheme, nz) * sizeof(double);
}

size_t crt_hip_workspace_bytes_nb(int scheme, int32_t ncol, int32_t nz, int32_t nb) {
  size_t n = crt_hip_workspace_bytes(scheme, ncol, nz);
  if (n == 0 || nb <= 0) return 0;
  if (scheme == CRT_SCHEME_ZQ_PA) n += 2 * (size_t)ncol * (size_t)zqpa_M(nz) * (size_t)nb * sizeof(double);
  return n;
}

int crt_hip_quad_nodes(double mu_s, double* psi_nodes) {
  if (!psi_nodes || !(mu_s > 0.0 && mu_s < 1.0)) return CRT_ERR_BAD_ARG;
  host_quad_nodes(mu_s, psi_nodes);
  return CRT_OK;
}

static int solve_impl(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                      const crt_outputs* out, void* workspace, size_t workspace_bytes, crt_stream_t stream, int f32,
                      const IntArgs* integ = nullptr) {
  if (!scheme_ok(scheme) || !cols || !bands || !out) return CRT_ERR_BAD_ARG;
  const int ncol = cols->ncol, nz = cols->nz, nb = bands->nb;
  if (ncol <= 0 || nz <= 0 || nb <= 0) return CRT_ERR_BAD_ARG;
  if (!cols->psi || !cols->lai || !cols->g_kind) return CRT_ERR_BAD_ARG;
  if (scheme == CRT_SCHEME_2S && !cols->mla) return CRT_ERR_BAD_ARG;
  if (!bands->I_dr0 || !bands->I_df0 || !bands->leaf_r || !bands->leaf_t) return CRT_ERR_BAD_ARG;
  if (scheme != CRT_SCHEME_BL && !bands->soil_r) return CRT_ERR_BAD_ARG;
  if (bands->col_stride != 0 && bands->col_stride < nb) return CRT_ERR_BAD_ARG;
  if (!integ && (!out->I_dr || !out->I_df_d || !out->I_df_u || !out->F)) return CRT_ERR_BAD_ARG;
  const bool tri = scheme == CRT_SCHEME_N79 || scheme == CRT_SCHEME_ZQ;
  const int nextra = scheme == CRT_SCHEME_N79 ? 2 : (scheme == CRT_SCHEME_ZQ || scheme == CRT_SCHEME_G77 || scheme == CRT_SCHEME_BF) ? 3 : 0;
  if (!integ) {
    if (nextra >= 1 && !out->x0) return CRT_ERR_BAD_ARG;
    if (nextra >= 2 && !out->x1) return CRT_ERR_BAD_ARG;
    if (nextra >= 3 && !out->x2) return CRT_ERR_BAD_ARG;
  }
  if (nz < 2) return CRT_ERR_SHAPE;
  if (scheme == CRT_SCHEME_N79 && nz < 3) return CRT_ERR_SHAPE;  // td[1]/tb[1] of _solve_n79.py:85-92
  double mu_s = 0.501;
  int method = CRT_TAU_D_QUAD, flags = 0;
  int tune[CRT_NTUNE] = {};
  if (opts) {
    mu_s = opts->mu_s;
    method = opts->tau_d_method;
    flags = opts->flags;
    for (int i = 0; i < CRT_NTUNE; ++i) tune[i] = opts->tune[i];
  }
  if (scheme == CRT_SCHEME_4S && !(mu_s > 0.0 && mu_s < 1.0)) return CRT_ERR_BAD_ARG;
  if (method != CRT_TAU_D_QUAD && method != CRT_TAU_D_9SKY) return CRT_ERR_BAD_ARG;  // ValueError, common.py:78
  const size_t need = crt_hip_workspace_bytes_nb(scheme, ncol, nz, nb);
  if (!workspace || workspace_bytes < need) return CRT_ERR_WORKSPACE;

  hipStream_t s = static_cast<hipStream_t>(stream);
  ColArgs ca;
  ca.ncol = ncol;
  ca.nz = nz;
  ca.scheme = scheme;
  ca.tau_d_method = method;
  ca.mu_s = mu_s;
  ca.psi = cols->psi;
  ca.lai = cols->lai;
  ca.mla = cols->mla;
  ca.g_kind = cols->g_kind;
  ca.g_param = cols->g_param;
  ca.g_at_psi = cols->g_at_psi;
  ca.g_table = cols->g_table;
  ca.ws = static_cast<double*>(workspace);
  if (!(flags & CRT_FLAG_SKIP_PRECOMPUTE)) {
    int st = launch_colpre(ca, s);
    if (st != CRT_OK) return st;
  }
  if (flags & CRT_FLAG_PRECOMPUTE_ONLY) return CRT_OK;

  SolveArgs sa;
  sa.ncol = ncol;
  sa.nb = nb;
  sa.nz = nz;
  sa.reclen = rec_len(scheme, nz);
  sa.col_stride = bands->col_stride;
  sa.ws = static_cast<const double*>(workspace);
  sa.I_dr0 = bands->I_dr0;
  sa.I_df0 = bands->I_df0;
  sa.leaf_r = bands->leaf_r;
  sa.leaf_t = bands->leaf_t;
  sa.soil_r = bands->soil_r;
  sa.o[0] = out->I_dr;
  sa.o[1] = out->I_df_d;
  sa.o[2] = out->I_df_u;
  sa.o[3] = out->F;
  sa.o[4] = out->x0;
  sa.o[5] = out->x1;
  sa.o[6] = out->x2;
  sa.mu_s = mu_s;
  sa.f32 = f32;
  for (int i = 0; i < CRT_NTUNE; ++i) sa.tune[i] = tune[i];
  if (integ) {
    if (scheme == CRT_SCHEME_ZQ_PA) return CRT_ERR_UNSUPPORTED;
    return tri ? launch_tridiag_int(scheme, sa, *integ, s) : launch_closed_int(scheme, sa, *integ, s);
  }
  if (scheme == CRT_SCHEME_ZQ_PA)
    return launch_zqpa(sa, static_cast<double*>(workspace) + (size_t)ncol * sa.reclen, s);
  const int force = (flags & CRT_FLAG_DIRECT_STORES) ? 1 : 0;
  return tri ? launch_tridiag(scheme, sa, s, force) : launch_closed(scheme, sa, s, force);
}


int crt_hip_solve_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                      const crt_outputs* out, void* workspace, size_t workspace_bytes, crt_stream_t stream) {
  return solve_impl(scheme, cols, bands, opts, out, workspace, workspace_bytes, stream, 0);
}

// the f32 structs have the layout of the f64 ones (pointers + sizes); only the element type behind the pointers differs
int crt_hip_solve_f32(int scheme, const crt_columns* cols, const crt_bands_f32* bands, const crt_options* opts,
                      const crt_outputs_f32* out, void* workspace, size_t workspace_bytes, crt_stream_t stream) {
  static_assert(sizeof(crt_bands_f32) == sizeof(crt_bands) && sizeof(crt_outputs_f32) == sizeof(crt_outputs), "layout");
  return solve_impl(scheme, cols, reinterpret_cast<const crt_bands*>(bands), opts, reinterpret_cast<const crt_outputs*>(out), workspace,
                    workspace_bytes, stream, 1);
}

#define CRT_ENTRY(name, id)                                                                                        \
  int name(const crt_columns* c, const crt_bands* b, const crt_options* o, const crt_outputs* out, void* ws,       \
           size_t wsb, crt_stream_t s) {                                                                           \
    return crt_hip_solve_f64(id, c, b, o, out, ws, wsb, s);                                                        \
  }
CRT_ENTRY(crt_hip_2s_f64, CRT_SCHEME_2S)
CRT_ENTRY(crt_hip_4s_f64, CRT_SCHEME_4S)
CRT_ENTRY(crt_hip_n79_f64, CRT_SCHEME_N79)
CRT_ENTRY(crt_hip_zq_f64, CRT_SCHEME_ZQ)
CRT_ENTRY(crt_hip_bl_f64, CRT_SCHEME_BL)
CRT_ENTRY(crt_hip_g77_f64, CRT_SCHEME_G77)
CRT_ENTRY(crt_hip_bf_f64, CRT_SCHEME_BF)
CRT_ENTRY(crt_hip_zq_pa_f64, CRT_SCHEME_ZQ_PA)
#undef CRT_ENTRY

#define CRT_ENTRY32(name, id)                                                                                      \
  int name(const crt_columns* c, const crt_bands_f32* b, const crt_options* o, const crt_outputs_f32* out, void* ws, \
           size_t wsb, crt_stream_t s) {                                                                           \
    return crt_hip_solve_f32(id, c, b, o, out, ws, wsb, s);                                                        \
  }
CRT_ENTRY32(crt_hip_2s_f32, CRT_SCHEME_2S)
CRT_ENTRY32(crt_hip_4s_f32, CRT_SCHEME_4S)
CRT_ENTRY32(crt_hip_n79_f32, CRT_SCHEME_N79)
CRT_ENTRY32(crt_hip_zq_f32, CRT_SCHEME_ZQ)
CRT_ENTRY32(crt_hip_bl_f32, CRT_SCHEME_BL)
CRT_ENTRY32(crt_hip_g77_f32, CRT_SCHEME_G77)
CRT_ENTRY32(crt_hip_bf_f32, CRT_SCHEME_BF)
CRT_ENTRY32(crt_hip_zq_pa_f32, CRT_SCHEME_ZQ_PA)
#undef CRT_ENTRY32

int crt_hip_absorb_bandsum_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                               const double* I_df_u, const double* band_w, int32_t ngroup, double* aI, double* aI_sl,
                               double* aI_sh, double* totals, crt_stream_t stream) {
  if (!cols || !bands || !I_dr || !I_df_d || !I_df_u || !band_w || !aI || !aI_sl || !aI_sh) return CRT_ERR_BAD_ARG;
  if (cols->ncol <= 0 || cols->nz < 2 || bands->nb <= 0 || ngroup <= 0 || ngroup > MAXG) return CRT_ERR_BAD_ARG;
  if (!cols->psi || !cols->lai || !cols->g_kind || !bands->leaf_r || !bands->leaf_t) return CRT_ERR_BAD_ARG;
  EpiArgs a;
  a.ncol = cols->ncol;
  a.nb = bands->nb;
  a.nz = cols->nz;
  a.ngroup = ngroup;
  a.col_stride = bands->col_stride;
  a.psi = cols->psi;
  a.lai = cols->lai;
  a.g_kind = cols->g_kind;
  a.g_param = cols->g_param;
  a.g_at_psi = cols->g_at_psi;
  a.leaf_r = bands->leaf_r;
  a.leaf_t = bands->leaf_t;
  a.I_dr = I_dr;
  a.I_df_d = I_df_d;
  a.I_df_u = I_df_u;
  a.band_w = band_w;
  a.aI = aI;
  a.aI_sl = aI_sl;
  a.aI_sh = aI_sh;
  a.totals = totals;
  hipLaunchKernelGGL(k_absorb_bandsum, dim3(a.ncol), dim3(EB), 0, static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_integrated_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                           const double* band_w, int32_t ngroup, double* aI, double* aI_sl, double* aI_sh, double* totals,
                           void* workspace, size_t workspace_bytes, crt_stream_t stream) {
  if (!band_w || !aI || !aI_sl || !aI_sh || ngroup <= 0 || ngroup > INT_MAXG || !cols) return CRT_ERR_BAD_ARG;
  IntArgs ia;
  ia.lai = cols->lai;
  ia.band_w = band_w;
  ia.ngroup = ngroup;
  ia.aI = aI;
  ia.aI_sl = aI_sl;
  ia.aI_sh = aI_sh;
  ia.totals = totals;
  crt_outputs none = {};
  return solve_impl(scheme, cols, bands, opts, &none, workspace, workspace_bytes, stream, 0, &ia);
}

int crt_hip_absorb_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                       const double* I_df_u, double* const* out7, double* laim, double* f_slm, crt_stream_t stream) {
  if (!cols || !bands || !I_dr || !I_df_d || !I_df_u || !out7 || !laim || !f_slm) return CRT_ERR_BAD_ARG;
  if (cols->ncol <= 0 || cols->nz < 2 || bands->nb <= 0) return CRT_ERR_BAD_ARG;
  if (!cols->psi || !cols->lai || !cols->g_kind || !bands->leaf_r || !bands->leaf_t) return CRT_ERR_BAD_ARG;
  AbsArgs a;
  a.ncol = cols->ncol;
  a.nb = bands->nb;
  a.nz = cols->nz;
  a.col_stride = bands->col_stride;
  a.psi = cols->psi;
  a.lai = cols->lai;
  a.g_kind = cols->g_kind;
  a.g_param = cols->g_param;
  a.g_at_psi = cols->g_at_psi;
  a.leaf_r = bands->leaf_r;
  a.leaf_t = bands->leaf_t;
  a.I_dr = I_dr;
  a.I_df_d = I_df_d;
  a.I_df_u = I_df_u;
  for (int i = 0; i < 7; ++i) {
    if (!out7[i]) return CRT_ERR_BAD_ARG;
    a.o[i] = out7[i];
  }
  a.laim = laim;
  a.f_slm = f_slm;
  hipLaunchKernelGGL(k_absorb, dim3(a.ncol), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_tau_d_f64(const double* kb_nodes, const double* L, int64_t n, int32_t method, double* out, crt_stream_t stream) {
  if (!kb_nodes || !L || !out || n < 0) return CRT_ERR_BAD_ARG;
  if (method != CRT_TAU_D_QUAD && method != CRT_TAU_D_9SKY) return CRT_ERR_BAD_ARG;  // ValueError, common.py:78
  if (n == 0) return CRT_OK;
  return crt::launch_tau_d(kb_nodes, L, (long long)n, method, out, static_cast<hipStream_t>(stream));
}

const char* crt_hip_last_kernel(void) { return last_kernel(); }

int crt_hip_probe_fill_f64(double* dst, size_t n, double value, crt_stream_t stream) {
  if (!dst || n == 0 || (n & 1) || (reinterpret_cast<uintptr_t>(dst) & 15)) return CRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<d2*>(dst), n / 2, value);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_probe_copy_f64(double* dst, const double* src, size_t n, crt_stream_t stream) {
  if (!dst || !src || n == 0 || (n & 1) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15))
    return CRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<d2*>(dst),
                     reinterpret_cast<const d2*>(src), n / 2);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

}  // extern "C"
