#include "tri_tile_impl.hpp"

// ------------------------------------------------------------------------------------------
// zq_pa (crt1d/solvers/_solve_zq_pa.py:24-418): the zq system on a computational grid of M = min(100, nz) equal
// layers, then linear interpolation of the interface fluxes back to the caller's levels.
//   step 1: the zq column-tile kernel above runs with nz := M on the K0 record of scheme zq_pa (whose first vector is
//           the beam fraction per computational layer and whose tau_i / tau_psi are those of LAI/M) and writes only
//           I_df_d[z] = SWd[z+1], I_df_u[z] = SWu[z], z = 0..M-1, into workspace scratch;
//   step 2: k_zqpa_interp forms SWd[0] := SWd[1], SWu[M] := SWu[M-1] (:310,335), interpolates and writes the outputs.
namespace crt {
namespace {

struct InterpArgs {
  int ncol, nb, nz, M, reclen;
  long long col_stride;
  const double* ws;
  const double* dnz;  // [ncol][M][nb]  SWd[z+1]
  const double* upz;  // [ncol][M][nb]  SWu[z]
  const void* I_dr0;
  void* o[4];
};

template <typename TIO>
__global__ __launch_bounds__(256) void k_zqpa_interp(InterpArgs a) {
  extern __shared__ double lds[];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int nb = a.nb, nz = a.nz, M = a.M;
  const double* src = a.ws + (long long)c * a.reclen;
  for (int i = tid; i < a.reclen; i += 256) lds[i] = src[i];
  __syncthreads();
  const double* rec = lds;
  const double invmu = rec[S_INVMU];
  const double* ekl = rec + REC_HDR + nz;
  const double* kidx = ekl + nz;
  const double* wgt = kidx + nz;
  const long long sb = (long long)c * M * nb;
  // flat sweep over the column's nz * nb outputs: consecutive threads -> consecutive addresses
  for (int i = tid; i < nz * nb; i += 256) {
    const int j = i / nb, b = i - j * nb;
    const int ka = (int)kidx[j], kb = ka - 1;  // interfaces above / below lai[j] (0 = ground)
    const double w = wgt[j];
    // SWd[k] = dnz[max(k,1)-1],  SWu[k] = upz[min(k, M-1)]
    const double da = a.dnz[sb + (long long)(max(ka, 1) - 1) * nb + b], db = a.dnz[sb + (long long)(max(kb, 1) - 1) * nb + b];
    const double ua = a.upz[sb + (long long)min(ka, M - 1) * nb + b], ub = a.upz[sb + (long long)min(kb, M - 1) * nb + b];
    const double dn = da + (db - da) * w;  // :360
    const double up = ua + (ub - ua) * w;  // :361
    const double idr = ldio<TIO>(a.I_dr0, (long long)c * a.col_stride + b) * ekl[j];  // :354-355
    const long long o = (long long)c * nz * nb + i;
    outp<TIO>(a.o[0])[o] = (TIO)idr;
    outp<TIO>(a.o[1])[o] = (TIO)dn;
    outp<TIO>(a.o[2])[o] = (TIO)up;
    outp<TIO>(a.o[3])[o] = (TIO)(idr * invmu + 2 * up + 2 * dn);  // :412
  }
}

}  // namespace

int launch_zqpa(const SolveArgs& a, double* scratch, hipStream_t s) {
  if (a.f32) return CRT_ERR_UNSUPPORTED;  // the computational-grid scratch is fp64; f32 storage not wired for zq_pa yet
  const int M = zqpa_M(a.nz);
  SolveArgs g = a;  // computational-grid solve: nz := M, outputs := scratch
  g.nz = M;
  g.o[0] = scratch;
  g.o[1] = scratch + (size_t)a.ncol * M * a.nb;
  for (int i = 2; i < 7; ++i) g.o[i] = nullptr;
  bool done = false;
  int st = launch_scheme<TriZqPa, double>(g, s, done, 1);
  if (st != CRT_OK) return st;
  if (!done && (st = launch_zqpa_wave(g, s)) != CRT_OK) return st;  // nb > 1024: per-wave kernel (solve_tridiag.hip)
  InterpArgs ia;
  ia.ncol = a.ncol;
  ia.nb = a.nb;
  ia.nz = a.nz;
  ia.M = M;
  ia.reclen = a.reclen;
  ia.col_stride = a.col_stride;
  ia.ws = a.ws;
  ia.dnz = scratch;
  ia.upz = scratch + (size_t)a.ncol * M * a.nb;
  ia.I_dr0 = a.I_dr0;
  for (int i = 0; i < 4; ++i) ia.o[i] = a.o[i];
  const size_t sh = a.reclen * sizeof(double);
  if (sh > MAX_WG_LDS) return CRT_ERR_UNSUPPORTED;
  if (sh > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_zqpa_interp<double>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
    return CRT_ERR_LAUNCH;
  hipLaunchKernelGGL((k_zqpa_interp<double>), dim3(a.ncol), dim3(256), sh, s, ia);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

}  // namespace crt
