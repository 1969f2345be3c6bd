// How accurate is v_rcp_f64 on gfx950, and how many Newton steps does fast_rcp (crt_internal.hpp) need?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/rcp_probe.hip -o tools/rcp_probe.bin && tools/rcp_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* rd, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
  r1[i] = r;
  r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
  r2[i] = r;
  rd[i] = 1.0 / v;  // IEEE division (div_scale / div_fmas / div_fixup)
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    const double m = 1.0 + (double)rand() / RAND_MAX + 1e-9 * rand() / RAND_MAX;  // mantissas over [1, 2)
    const int e = (i % 5 == 0) ? (rand() % 600 - 300) : (rand() % 40 - 20);
    x[i] = std::ldexp(m, e) * ((i & 1) ? -1.0 : 1.0);
  }
  double *dx, *d0, *d1, *d2, *dd;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&dd, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, dd, n);
  std::vector<double> r0(n), r1(n), r2(n), rd(n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(rd.data(), dd, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, ed = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / (long double)x[i];
    const double ulp = std::ldexp(1.0, std::ilogb((double)ex) - 52);
    e0 = std::fmax(e0, std::fabs((double)((long double)r0[i] - ex)) / ulp);
    e1 = std::fmax(e1, std::fabs((double)((long double)r1[i] - ex)) / ulp);
    e2 = std::fmax(e2, std::fabs((double)((long double)r2[i] - ex)) / ulp);
    ed = std::fmax(ed, std::fabs((double)((long double)rd[i] - ex)) / ulp);
  }
  printf("max error in ulp over %d arguments: v_rcp_f64 %.3g | + 1 Newton step %.3g | + 2 steps %.3g | IEEE division %.3g\n", n, e0, e1, e2, ed);
  return 0;
}
