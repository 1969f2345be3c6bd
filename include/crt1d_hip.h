/*
 * crt1d_hip.h -- C ABI of the MI355X (gfx950) batched 1-D canopy radiative-transfer path.
 *
 * One call = one scheme over a batch of (column x band) solves.  The entry points replace,
 * for the batched case, the reference's keyword-only plugin functions
 *
 *   solve_2s   crt1d/solvers/_solve_2s.py:11-163     ->  crt_hip_2s_f64
 *   solve_4s   crt1d/solvers/_solve_4s.py:8-293      ->  crt_hip_4s_f64
 *   solve_n79  crt1d/solvers/_solve_n79.py:11-164    ->  crt_hip_n79_f64
 *   solve_zq   crt1d/solvers/_solve_zq.py:13-229     ->  crt_hip_zq_f64
 *   solve_bl   crt1d/solvers/_solve_bl.py:9-93       ->  crt_hip_bl_f64
 *   solve_g77  crt1d/solvers/_solve_g77.py:7-135     ->  crt_hip_g77_f64
 *   solve_bf   crt1d/solvers/_solve_bf.py:7-154      ->  crt_hip_bf_f64
 *   solve_zq_pa crt1d/solvers/_solve_zq_pa.py:24-418 ->  crt_hip_zq_pa_f64
 *
 * which `Model.run` dispatches to at crt1d/model.py:305-310.  The reference has no FFI of
 * its own (pure Python); INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - plain pointers + sizes; all data pointers are DEVICE pointers (HBM), fp64, caller-owned.
 *    Nothing is allocated or freed by the library; launches are asynchronous on `stream`.
 *  - `lai` is cumulative leaf-area index from the canopy top: index 0 = ground (total LAI),
 *    index nz-1 = canopy top (0), strictly as crt1d/variables.yml:73-86 / model.py:244-246.
 *  - per-(column, band) inputs: element (c, b) at p[c * col_stride + b]; col_stride = 0
 *    broadcasts one spectrum over all columns.
 *  - outputs are [ncol][nz or nz-1][nb], bands contiguous (the reference's (nz, nb) C-order
 *    arrays, stacked over columns).
 *  - return value: 0 on success, negative crt_status otherwise (the Python wrapper maps these
 *    to the reference's exception types: AssertionError / ValueError).
 *  - thread-safety: re-entrant -- no mutable process-global state on the solve path (kernel-selection overrides travel in
 *    crt_options.tune; the quadrature tables are uploaded once per device under a lock); concurrent calls must use distinct
 *    streams + workspaces.  The buffer allocator (crt_hip_buffer_*) serialises on one mutex.
 */
#ifndef CRT1D_HIP_H
#define CRT1D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_ABI_VERSION 3

/* leaf-angle G(psi) kinds: crt1d/leaf_angle.py:118-202 */
enum crt_g_kind {
  CRT_G_HORIZONTAL = 0,              /* cos(psi)                       :118-120 */
  CRT_G_SPHERICAL = 1,               /* 0.5                            :123-125 */
  CRT_G_VERTICAL = 2,                /* 2/pi sin(psi)                  :128-130 */
  CRT_G_ELLIPSOIDAL = 3,             /* Campbell 1986, param = x       :133-165 */
  CRT_G_ELLIPSOIDAL_APPROX = 4,      /* Campbell 1990, param = x       :168-180 */
  CRT_G_ELLIPSOIDAL_APPROX_BONAN = 5,/* Ross-Goudriaan, param = chi_l  :183-202 */
  CRT_G_TABLE = 6                    /* caller-sampled callable (K_b_fn / G_fn of the plugin API) */
};

/* fixed quadrature nodes at which a CRT_G_TABLE column supplies G(psi) */
#define CRT_NQ_TAU 96  /* graded composite Gauss-Legendre on [0, pi/2]: tau_d (common.py:30-37), mu_bar (_solve_2s.py:32) */
#define CRT_NQ_G4 32   /* 2 x 16 Gauss-Legendre in psi for G_int_1/2 (_solve_4s.py:148-149); depend on mu_s */
#define CRT_NQ_9SKY 9  /* 5,15,...,85 deg (common.py:46) */
#define CRT_NQ (CRT_NQ_TAU + CRT_NQ_G4 + CRT_NQ_9SKY)

enum crt_scheme {
  CRT_SCHEME_2S = 0, CRT_SCHEME_4S = 1, CRT_SCHEME_N79 = 2, CRT_SCHEME_ZQ = 3,
  CRT_SCHEME_BL = 4, CRT_SCHEME_G77 = 5, CRT_SCHEME_BF = 6, CRT_SCHEME_ZQ_PA = 7, CRT_NUM_SCHEMES = 8
};

enum crt_tau_d_method { CRT_TAU_D_QUAD = 0, CRT_TAU_D_9SKY = 1 }; /* _solve_n79.py:19, common.py:72-78 */

enum crt_status {
  CRT_OK = 0,
  CRT_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, bad enum  -> ValueError      */
  CRT_ERR_WORKSPACE = -2,    /* workspace smaller than crt_hip_workspace_bytes()              */
  CRT_ERR_UNSUPPORTED = -3,  /* shape outside what the kernels handle (e.g. nz too large for LDS) */
  CRT_ERR_LAUNCH = -4,       /* HIP launch / runtime error                                    */
  CRT_ERR_SHAPE = -5         /* shape violates a reference assertion (e.g. nz < 3 for n79) -> AssertionError */
};

typedef void* crt_stream_t; /* hipStream_t */

/* per-column canopy geometry (what one reference `Model` instance holds; model.py:68-116) */
typedef struct crt_columns {
  int32_t ncol;
  int32_t nz;             /* interface levels ("nlayers" of the reference) */
  const double* psi;      /* [ncol] solar zenith angle, radians */
  const double* lai;      /* [ncol][nz] cumulative LAI, index 0 = ground */
  const double* mla;      /* [ncol] mean leaf angle, degrees; 2s only (may be NULL otherwise) */
  const int32_t* g_kind;  /* [ncol] crt_g_kind */
  const double* g_param;  /* [ncol] x or chi_l (ignored for parameter-free kinds) */
  const double* g_at_psi; /* [ncol] G(psi[c]); read only for CRT_G_TABLE columns (may be NULL otherwise) */
  const double* g_table;  /* [ncol][CRT_NQ] G at crt_hip_quad_nodes(); CRT_G_TABLE columns only (may be NULL) */
} crt_columns;

/* per-(column, band) spectra: the (n_wl,) plugin inputs I_dr0_all, I_df0_all, leaf_r, leaf_t, soil_r */
typedef struct crt_bands {
  int32_t nb;
  int64_t col_stride;     /* nb (or larger) for per-column spectra, 0 = same spectrum for every column */
  const double* I_dr0;
  const double* I_df0;
  const double* leaf_r;
  const double* leaf_t;
  const double* soil_r;   /* not read by bl */
} crt_bands;

/* crt_options.flags */
#define CRT_FLAG_SKIP_PRECOMPUTE 1 /* workspace already holds the column records of an earlier call with the
                                      same scheme/columns/options (only the spectra changed): skip kernel K0 */
#define CRT_FLAG_PRECOMPUTE_ONLY 2 /* run K0 only (fills the workspace), no solve kernel */
#define CRT_FLAG_DIRECT_STORES 4   /* measurement aid: use the direct-store solve kernel even where the LDS-tiled,
                                      line-aligned one applies (same results, different store pattern) */

#define CRT_NTUNE 16
typedef struct crt_options {
  double mu_s;            /* 4s: cosine of the dividing angle, default 0.501 (_solve_4s.py:9) */
  int32_t tau_d_method;   /* n79: crt_tau_d_method, default CRT_TAU_D_QUAD (_solve_n79.py:19) */
  int32_t flags;          /* CRT_FLAG_* */
  int32_t tune[CRT_NTUNE];/* measurement aid, all zero in production: per-CALL overrides of the kernel-selection heuristics
                             (tile height, store waves, kernel family; keys in csrc/crt_internal.hpp).  Part of the call's
                             arguments, so there is no process-global tuning state: calls with different settings may run
                             concurrently */
} crt_options;

/* outputs; each [ncol][nz][nb] unless noted.  Unused slots may be NULL. */
typedef struct crt_outputs {
  double* I_dr;
  double* I_df_d;
  double* I_df_u;
  double* F;
  double* x0; /* n79: aI_lsl [ncol][nz-1][nb] | zq: I_df_d_ss | g77, bf: aI_lsl */
  double* x1; /* n79: aI_lsh [ncol][nz-1][nb] | zq: I_df_u_ss | g77, bf: aI_lsh */
  double* x2; /*                                zq: F_ss      | g77, bf: aI_l   */
} crt_outputs;

/* f32 storage variants: spectra read and profiles written as float (half the HBM bytes).  Geometry (crt_columns), the
 * workspace and ALL arithmetic stay fp64: results are the fp64 results rounded once to fp32. */
typedef struct crt_bands_f32 {
  int32_t nb;
  int64_t col_stride;
  const float* I_dr0;
  const float* I_df0;
  const float* leaf_r;
  const float* leaf_t;
  const float* soil_r;
} crt_bands_f32;

typedef struct crt_outputs_f32 {
  float* I_dr;
  float* I_df_d;
  float* I_df_u;
  float* F;
  float* x0;
  float* x1;
  float* x2;
} crt_outputs_f32;

int crt_hip_abi_version(void);
const char* crt_hip_strerror(int status);

/* bytes of device workspace a solve of `scheme` needs for (ncol, nz) */
size_t crt_hip_workspace_bytes(int scheme, int32_t ncol, int32_t nz);

/* same, for schemes whose workspace also depends on the number of bands (zq_pa keeps its computational-grid fluxes
 * there: 2 * ncol * min(100, nz) * nb doubles); equals crt_hip_workspace_bytes for every other scheme */
size_t crt_hip_workspace_bytes_nb(int scheme, int32_t ncol, int32_t nz, int32_t nb);

/* host: fill psi_nodes[CRT_NQ] with the zenith angles (radians) at which g_table is sampled */
int crt_hip_quad_nodes(double mu_s, double* psi_nodes);

/* generic entry (scheme = crt_scheme) and the per-scheme entry points */
int crt_hip_solve_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                      const crt_outputs* out, void* workspace, size_t workspace_bytes, crt_stream_t stream);
int crt_hip_2s_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_4s_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_n79_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_zq_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_bl_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_g77_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_bf_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);
int crt_hip_zq_pa_f64(const crt_columns*, const crt_bands*, const crt_options*, const crt_outputs*, void*, size_t, crt_stream_t);

int crt_hip_solve_f32(int scheme, const crt_columns* cols, const crt_bands_f32* bands, const crt_options* opts,
                      const crt_outputs_f32* out, void* workspace, size_t workspace_bytes, crt_stream_t stream);
int crt_hip_2s_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
int crt_hip_4s_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
int crt_hip_n79_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
int crt_hip_zq_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
int crt_hip_bl_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
int crt_hip_g77_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
int crt_hip_bf_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);
/* zq_pa with f32 storage runs in the single-kernel form only: 16 <= nb <= 832 (even or odd: the reference's default 107 bands are
 * covered), else CRT_ERR_UNSUPPORTED */
int crt_hip_zq_pa_f32(const crt_columns*, const crt_bands_f32*, const crt_options*, const crt_outputs_f32*, void*, size_t, crt_stream_t);

/*
 * Epilogue (model.py:573-647 `_calc_absorption` + diagnostics.py:39-108 `band`): layer absorption
 * from the three irradiance profiles, reduced over bands with `ngroup` weight vectors
 * (spectra.py:71-126 `_x_frac_in_bounds`).  Outputs, each [ncol][nz-1][ngroup]:
 *   aI (total), aI_sl (sunlit), aI_sh (shaded); and totals [ncol][ngroup][4]:
 *   incoming I_d[top], reflected I_df_u[top], transmitted I_d[ground], soil-reflected I_df_u[ground]
 *   (the terms of diagnostics.py:476-530 `compare_ebal`).  band_w is [ngroup][nb].
 */
int crt_hip_absorb_bandsum_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                               const double* I_df_u, const double* band_w, int32_t ngroup, double* aI, double* aI_sl,
                               double* aI_sh, double* totals, crt_stream_t stream);

/*
 * The complete output of diagnostics.band (crt1d/diagnostics.py:39-108) for up to four band groups at once: besides the layer
 * absorption sums above, the band-integrated LEVEL profiles of every irradiance variable band() reduces (:84-91; the variables of
 * Model.to_xr, model.py:421-426): I_dr, I_df_d, I_df_u, F and I_d = I_dr + I_df_d, each [ncol][nz][ngroup]
 *     X_g(z) = sum_b band_w[g][b] X(z, b)                                            (diagnostics.py:81)
 * and aI_dr [ncol][nz-1][ngroup], the direct-beam part of the absorbed irradiance, from which the remaining entries of the
 * reference's absorption dict follow exactly: aI_df = aI - aI_dr, aI_df_sl = aI_sl - aI_dr, aI_df_sh = aI_sh (model.py:628-634).
 * calc_PFD=True (:92-104, _E_to_PFD_da :19-36) is a choice of WEIGHTS: band_w[g][b] / e_wl_umol(wl[b]) gives the photon-flux variants.
 * The six optional pointers are given all together or all NULL (then this is crt_hip_absorb_bandsum_f64); totals may be NULL.
 */
typedef struct crt_bandsum_out {
  double* aI;      /* [ncol][nz-1][ngroup] */
  double* aI_sl;
  double* aI_sh;
  double* totals;  /* [ncol][ngroup][4] or NULL */
  double* aI_dr;   /* [ncol][nz-1][ngroup] or NULL */
  double* I_dr;    /* [ncol][nz][ngroup] or NULL */
  double* I_df_d;
  double* I_df_u;
  double* F;
  double* I_d;
} crt_bandsum_out;
int crt_hip_absorb_bandsum2_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                                const double* I_df_u, const double* band_w, int32_t ngroup, const crt_bandsum_out* out, crt_stream_t stream);
/* ... and the same complete output from the fused solve + epilogue (crt_hip_integrated_f64): no profile is written to memory, the
 * level sums the kernel forms anyway are kept instead of thrown away. */
int crt_hip_integrated2_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts, const double* band_w,
                            int32_t ngroup, const crt_bandsum_out* out, void* workspace, size_t workspace_bytes, crt_stream_t stream);

/*
 * Fused solve + epilogue: the integrated outputs of crt_hip_absorb_bandsum_f64 (same shapes and meaning) straight from
 * the inputs, WITHOUT writing any profile to memory (bytes per solve drop from ~2 kB to ~40 B; the variant SURVEY.md
 * section 8(d) asks to report separately).  Schemes: 2s, 4s, bl, g77, bf, n79, zq; nb <= 1024.
 */
int crt_hip_integrated_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                           const double* band_w, int32_t ngroup, double* aI, double* aI_sl, double* aI_sh, double* totals,
                           void* workspace, size_t workspace_bytes, crt_stream_t stream);

/*
 * diagnostics.band's reduction for ANY variable with a trailing wavelength axis (diagnostics.py:81, `(da * w).sum(dim="wl")`):
 * out[row][g] = sum_b band_w[g][b] X[row][b], X = [nrow][nb], band_w = [ngroup <= 4][nb], out = [nrow][ngroup].  Used by
 * crt1d_amd.diagnostics.band for the variables of a single Model's dataset (incl. the schemes' own aI_*_scheme outputs); the batched
 * path uses crt_hip_absorb_bandsum2_f64 / crt_hip_integrated2_f64.
 */
int crt_hip_band_reduce_f64(const double* X, int64_t nrow, int32_t nb, const double* band_w, int32_t ngroup, double* out, crt_stream_t stream);

/*
 * Per-band layer absorption (model.py:573-647 `_calc_absorption`): out7 = {aI, aI_df, aI_dr, aI_sh, aI_sl, aI_df_sl,
 * aI_df_sh}, each [ncol][nz-1][nb]; laim, f_slm [ncol][nz-1].
 */
int crt_hip_absorb_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                       const double* I_df_u, double* const* out7, double* laim, double* f_slm, crt_stream_t stream);

/*
 * tau_d(L) = 2 int_0^{pi/2} exp(-K_b(psi) L) sin(psi) cos(psi) dpsi for n values of L: crt1d/solvers/common.py:56-87 `tau_df_fn`
 * (`_tau_df_fn_scalar` :30-37 with the library's fixed 96-node rule instead of QUADPACK, `_tau_df_fn_scalar_9sky` :40-53
 * unchanged).  kb_nodes[CRT_NQ] = K_b at the angles of crt_hip_quad_nodes() (device memory, like L and out).  The solvers get the
 * same numbers from their K0 kernel; this entry point serves callers of `tau_df_fn` / `K_df_fn` (:90-95).
 */
int crt_hip_tau_d_f64(const double* kb_nodes, const double* L, int64_t n, int32_t method, double* out, crt_stream_t stream);

/*
 * Input side, batched.
 *
 * crt_hip_smear_tuv_f64 replaces crt1d/spectra.py:260-300 `smear_tuv(x, y, bins)` (per-bin kernel `_smear_tuv_1`,
 * spectra.py:221-257): out[s][i] = trapezoidal integral of y_s(x) over [bins[i], bins[i+1]] (clipped to the x range)
 * divided by the bin width.  x: increasing grid, [nx] shared by all spectra (x_stride = 0) or [nspec][x_stride];
 * y: [nspec][nx]; bins: [nbins + 1]; out: [nspec][nbins].  Same floating-point operations, same order as the reference.
 *
 * crt_hip_lai_beta_f64 replaces crt1d/leaf_area.py:42-93 `distribute_lai_beta(h_c, LAI, n, h_min=0.5)` for ncol canopies:
 * lai, z (and lad unless NULL) are [ncol][nz]; h_c, LAI, h_min are [ncol] (h_min NULL = 0.5 everywhere).  lai is
 * bit-identical to the reference (numpy.linspace arithmetic); z and lad agree to ~1e-14 (Beta ppf by Newton iteration on
 * the closed form of I_x(a, 3) instead of scipy's inverse).
 */
int crt_hip_smear_tuv_f64(const double* x, int64_t x_stride, int32_t nx, const double* y, int32_t nspec, const double* bins,
                          int32_t nbins, double* out, crt_stream_t stream);
int crt_hip_lai_beta_f64(const double* h_c, const double* LAI, const double* h_min, int32_t ncol, int32_t nz, double* lai, double* z,
                         double* lad, crt_stream_t stream);

/*
 * Device buffers for the output profiles, with a deterministic placement in HBM (optional; any device pointer works with the
 * solve entry points).  On MI355X the store pattern of the solve kernels runs at ~5.5 TB/s when all the memory written at one
 * time lies in one "class" of physical memory and at ~7 TB/s when it is spread over two or three (csrc/buffers.hip explains the
 * measurement).  crt_hip_buffer_alloc_set allocates the n arrays of one output set from 512 MB physical chunks (HIP
 * virtual-memory API), classifies every new chunk with a ~1 ms probe, and interleaves the classes across the arrays so that a
 * kernel sweeping all n arrays in step always writes a balanced mix.  ptrs[a] receives a device pointer to at least bytes[a]
 * bytes (2 MB aligned).  Buffers belong to the current device; free each with crt_hip_buffer_free after all work that uses it
 * has completed.  Its chunks return to a per-device pool that keeps at most a RETENTION CAP (8 GB by default; environment
 * CRT1D_POOL_RETAIN_MB, or crt_hip_buffer_set_retain(bytes), which also applies the new cap at once): what exceeds the cap goes back
 * to the driver inside crt_hip_buffer_free, so freeing a large set makes its memory available to every other allocator again;
 * crt_hip_buffer_trim hands back the rest.  Looking for chunks of other classes ("exploration") transiently takes at most half of the
 * memory that stays free beside the request (<= 96 GB), and is abandoned for good on a device that shows a single class.
 * Synchronous (like hipMalloc): runs short probe kernels on a private stream and waits for them -- not capturable into a hipGraph,
 * so allocate (and construct plans) outside stream capture.  No counterpart in the reference (NumPy owns its arrays).
 * crt_hip_buffer_alloc = a set of one.
 * crt_hip_buffer_describe writes one letter per chunk of a buffer (X / Y / Z = class, ? = ambiguous) into buf;
 * crt_hip_buffer_stats fills {chunks created, chunks released, probes run, probe microseconds, free chunks, classes seen}.
 */
int crt_hip_buffer_alloc_set(int32_t n, const size_t* bytes, void** ptrs);
int crt_hip_buffer_alloc(size_t bytes, void** ptr);
int crt_hip_buffer_free(void* ptr);
int crt_hip_buffer_trim(void);
int crt_hip_buffer_set_retain(size_t bytes);
int crt_hip_buffer_describe(const void* ptr, char* buf, size_t n);
int crt_hip_buffer_stats(int64_t* out6);

/* name and configuration of the solve kernel chosen by the calling thread's most recent solve / integrated call
 * (e.g. "k_pipe<2s,f64> T=8 store_waves=3 lds=81536"); reporting only, valid until the thread's next call */
const char* crt_hip_last_kernel(void);

/* bandwidth probes used by bench.py to report a measured HBM ceiling next to the 8 TB/s spec */
int crt_hip_probe_fill_f64(double* dst, size_t n, double value, crt_stream_t stream);
int crt_hip_probe_copy_f64(double* dst, const double* src, size_t n, crt_stream_t stream);
/* The device math the solve kernels use in place of libm: e[i] = exp(x[i]) (18 instructions, <= 1 ulp), sn / cs[i] = sin / cos(x[i])
 * (two-term Cody-Waite reduction, absolute error ~2e-16 for |x| < 1e6; the oscillatory modes of solve_4s).  Diagnostic: lets a test
 * bound their error against the host's libm, which is what the reference's numpy.exp calls (_solve_2s.py:125-131). */
int crt_hip_probe_math_f64(const double* x, size_t n, double* e, double* sn, double* cs, crt_stream_t stream);
/* The flush of the solve kernels alone: workgroup c writes column c (col_doubles doubles) of all `narrays` arrays in step, run_doubles at a
 * time (T levels x nb bands), with 16-byte streaming stores.  On a set from crt_hip_buffer_alloc_set it measures the store rate that
 * placement allows; it OVERWRITES the arrays with `value`.  No counterpart in the reference. */
int crt_hip_probe_store_set_f64(double* const* arrays, int32_t narrays, int64_t ncol, int64_t col_doubles, int32_t run_doubles, double value,
                                crt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CRT1D_HIP_H */
