// Input side of the hot path: the two per-canopy preparations the reference does on
// the host before a solve, batched so that a million-column run never stages its inputs through NumPy.
//
//   crt_hip_smear_tuv_f64   spectral re-binning   crt1d/spectra.py:221-300  (`_smear_tuv_1`, `smear_tuv`)
//   crt_hip_lai_beta_f64    leaf-area profiles    crt1d/leaf_area.py:42-93  (`distribute_lai_beta`)
//
// Both are tiny next to the solve (bytes: a few hundred per column); they are written for exactness first: the
// re-binning performs the reference's floating-point operations in the reference's order.
#include <hip/hip_runtime.h>

#include "crt1d_hip.h"

namespace crt {
namespace {

// One thread = one (spectrum, bin).  `_smear_tuv_1` walks every trapezoid [x_k, x_k+1] of the original grid, skips
// those left of the bin, stops at the first one right of it, and adds the clipped trapezoid areas in index order
// (spectra.py:239-252).  The skip is a binary search here (x is increasing, spectra.py:229-230); the additions are the same.
__global__ __launch_bounds__(256) void k_smear_tuv(const double* __restrict__ x, long long x_stride, int nx, const double* __restrict__ y,
                                                   int nspec, const double* __restrict__ bins, int nbins, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nspec * nbins) return;
  const int s = (int)(i / nbins), b = (int)(i - (long long)s * nbins);
  const double* xs = x + (long long)s * x_stride;
  const double* ys = y + (long long)s * nx;
  const double xl = bins[b], xu = bins[b + 1];
  // first k in [0, nx-1) with x[k+1] >= xl
  int lo = 0, hi = nx - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (xs[mid + 1] < xl)
      lo = mid + 1;
    else
      hi = mid;
  }
  double area = 0.0;
  for (int k = lo; k < nx - 1; ++k) {
    const double x0 = xs[k], x1 = xs[k + 1];
    if (x1 < xl) continue;  // only reachable when x is not increasing; kept for identical behaviour
    if (x0 > xu) break;
    const double a1 = fmax(x0, xl);
    const double a2 = fmin(x1, xu);
    const double y0 = ys[k];
    const double slope = (ys[k + 1] - y0) / (x1 - x0);
    const double b1 = y0 + slope * (a1 - x0);
    const double b2 = y0 + slope * (a2 - x0);
    area = area + (a2 - a1) * (b2 + b1) / 2;
  }
  out[i] = area / (xu - xl);
}

// Regularised incomplete beta function for b = 3:  I_x(a, 3) = x^a (1 + a (1-x) + a (a+1)/2 (1-x)^2)
__device__ inline double ibeta_b3(double a, double x) {
  const double u = 1.0 - x;
  return pow(x, a) * (1.0 + a * u + 0.5 * a * (a + 1.0) * u * u);
}

// One thread = one (column, level).  leaf_area.py:66-91:
//   d = (h_c - 0.7 h_c) / h_c;  b = 3;  a = -((b-2) d + 1) / (d - 1);   frac = linspace(1, 0, n)
//   z = (h_c - h_min) (1 - Beta(a, b).ppf(frac)) + h_min;   lai = frac LAI
//   lad = LAI / (h_c - h_min) Beta(b, a).pdf((z - h_min) / (h_c - h_min))
// ppf: safeguarded Newton on the closed form above (scipy evaluates the same inverse to ~1e-15).
__global__ __launch_bounds__(256) void k_lai_beta(const double* __restrict__ h_c, const double* __restrict__ LAI,
                                                  const double* __restrict__ h_min, int ncol, int nz, double* __restrict__ lai,
                                                  double* __restrict__ z, double* __restrict__ lad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)ncol * nz) return;
  const int c = (int)(i / nz), j = (int)(i - (long long)c * nz);
  const double hc = h_c[c], L = LAI[c], hm = h_min ? h_min[c] : 0.5;
  const double h_max_lad = 0.7 * hc;
  const double d_max_lad = hc - h_max_lad;
  const double d = d_max_lad / hc;
  const double b = 3.0;
  const double a = -((b - 2.0) * d + 1.0) / (d - 1.0);
  // numpy.linspace(1.0, 0, n): arange(n) * step + start, last element forced to stop
  const double step = (0.0 - 1.0) / (double)(nz - 1);
  const double frac = (j == nz - 1) ? 0.0 : (double)j * step + 1.0;
  double xq;
  if (frac <= 0.0) {
    xq = 0.0;
  } else if (frac >= 1.0) {
    xq = 1.0;
  } else {
    const double cden = 0.5 * a * (a + 1.0) * (a + 2.0);  // 1 / B(a, 3)
    double lo = 0.0, hi = 1.0;
    xq = pow(frac, 1.0 / a) * 0.5 + 0.25;  // any interior start works with the bracket
    xq = fmin(fmax(xq, 1e-3), 1.0 - 1e-3);
    for (int it = 0; it < 100; ++it) {
      const double fx = ibeta_b3(a, xq) - frac;
      if (fx > 0.0)
        hi = xq;
      else
        lo = xq;
      const double u = 1.0 - xq;
      const double pdf = cden * pow(xq, a - 1.0) * u * u;
      double xn = xq - fx / pdf;
      if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);  // Newton left the bracket (or pdf = 0): bisect
      const double dx = fabs(xn - xq);
      xq = xn;
      if (dx <= 2.220446049250313e-16 * xq) break;
    }
  }
  const double depth = hc - hm;
  const double zj = depth * (1.0 - xq) + hm;
  lai[i] = frac * L;
  z[i] = zj;
  if (lad) {
    const double zrel = (zj - hm) / depth;
    const double cden = 0.5 * a * (a + 1.0) * (a + 2.0);
    const double pdf = (zrel <= 0.0 || zrel >= 1.0) ? 0.0 : cden * zrel * zrel * pow(1.0 - zrel, a - 1.0);
    lad[i] = L / depth * pdf;
  }
}

}  // namespace
}  // namespace crt

extern "C" {

int crt_hip_smear_tuv_f64(const double* x, int64_t x_stride, int32_t nx, const double* y, int32_t nspec, const double* bins,
                          int32_t nbins, double* out, crt_stream_t stream) {
  if (!x || !y || !bins || !out) return CRT_ERR_BAD_ARG;
  if (nx < 1 || nspec < 0 || nbins < 0 || (x_stride != 0 && x_stride < nx)) return CRT_ERR_BAD_ARG;
  const long long n = (long long)nspec * nbins;
  if (n == 0) return CRT_OK;
  const long long nblk = (n + 255) / 256;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(crt::k_smear_tuv, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)x_stride, nx, y,
                     nspec, bins, nbins, out);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_lai_beta_f64(const double* h_c, const double* LAI, const double* h_min, int32_t ncol, int32_t nz, double* lai, double* z,
                         double* lad, crt_stream_t stream) {
  if (!h_c || !LAI || !lai || !z) return CRT_ERR_BAD_ARG;
  if (ncol < 0 || nz < 2) return CRT_ERR_BAD_ARG;
  const long long n = (long long)ncol * nz;
  if (n == 0) return CRT_OK;
  const long long nblk = (n + 255) / 256;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(crt::k_lai_beta, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream), h_c, LAI, h_min, ncol, nz, lai, z,
                     lad);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

}  // extern "C"
