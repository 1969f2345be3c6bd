// Follow-up to store_bw.hip: is the nb=300 slowdown caused by 128-B misaligned row segments?
//  (A) lane-per-band pattern for several nb;  (B) block-per-column flat aligned flush (what an LDS-staged
//  writer would do): each block writes T levels x nb doubles (contiguous) per array per step with 16-B stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <bool NT, int VEC>
__global__ __launch_bounds__(256) void pattern(double* o0, double* o1, double* o2, double* o3, int ncol, int nb, int nz) {
  const int nbv = nb / VEC;
  long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long long)ncol * nbv) return;
  int c = item / nbv; int b = (item - (long long)c * nbv) * VEC;
  long long o = ((long long)c * nz) * nb + b;
  double* arr[4] = {o0, o1, o2, o3};
  for (int j = 0; j < nz; ++j, o += nb) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (VEC == 2) { d2 v; v.x = j; v.y = a; if (NT) __builtin_nontemporal_store(v, (d2*)(arr[a] + o)); else *(d2*)(arr[a] + o) = v; }
      else { double v = j + a; if (NT) __builtin_nontemporal_store(v, arr[a] + o); else arr[a][o] = v; }
    }
  }
}
// (B) one block per column; per step flush T*nb doubles per array, contiguous, 16-B per lane
template <bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void flat(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T) {
  const long long base = (long long)blockIdx.x * nz * nb;
  double* arr[4] = {o0 + base, o1 + base, o2 + base, o3 + base};
  const int chunk2 = T * nb / 2;  // d2 units per flush
  for (int j0 = 0; j0 < nz; j0 += T) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      d2* p = (d2*)(arr[a] + (long long)j0 * nb);
      for (int i = threadIdx.x; i < chunk2; i += BLOCK) {
        d2 v; v.x = i; v.y = a;
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
      }
    }
  }
}
template <typename F> float timeit(F f, int rep = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
int main() {
  const size_t bytes = (size_t)7 << 30;
  void* buf; CK(hipMalloc(&buf, bytes));
  const int ncol = 10000, nz = 60;
  printf("== (A) lane-per-band pattern, 4 arrays, ncol=%d nz=%d\n", ncol, nz);
  for (int nb : {256, 288, 296, 298, 300, 304, 312, 320, 107, 112}) {
    size_t per = (size_t)ncol * nz * nb;
    double* o0 = (double*)buf; double* o1 = o0 + per; double* o2 = o1 + per; double* o3 = o2 + per;
    double tot = 4.0 * per * 8;
    printf("nb %3d (row %4d B, mod128=%3d):", nb, nb * 8, (nb * 8) % 128);
    if (nb % 2 == 0) {
      int grid = (ncol * (nb / 2) + 255) / 256;
      float t = timeit([&] { hipLaunchKernelGGL((pattern<true, 2>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
      printf("  vec2 nt %7.1f", tot / t / 1e6);
      t = timeit([&] { hipLaunchKernelGGL((pattern<false, 2>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
      printf("  vec2 plain %7.1f", tot / t / 1e6);
    }
    int grid = (ncol * nb + 255) / 256;
    float t = timeit([&] { hipLaunchKernelGGL((pattern<true, 1>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
    printf("  vec1 nt %7.1f", tot / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((pattern<false, 1>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
    printf("  vec1 plain %7.1f GB/s\n", tot / t / 1e6);
  }
  printf("== (B) block-per-column flat flush, nb=300, 4 arrays\n");
  {
    const int nb = 300;
    size_t per = (size_t)ncol * nz * nb;
    double* o0 = (double*)buf; double* o1 = o0 + per; double* o2 = o1 + per; double* o3 = o2 + per;
    double tot = 4.0 * per * 8;
    for (int T : {1, 2, 4, 12, 60}) {
      float t = timeit([&] { hipLaunchKernelGGL((flat<true, 256>), dim3(ncol), dim3(256), 0, 0, o0, o1, o2, o3, nb, nz, T); });
      printf("T %2d block256 nt %7.1f", T, tot / t / 1e6);
      t = timeit([&] { hipLaunchKernelGGL((flat<false, 256>), dim3(ncol), dim3(256), 0, 0, o0, o1, o2, o3, nb, nz, T); });
      printf("  plain %7.1f", tot / t / 1e6);
      t = timeit([&] { hipLaunchKernelGGL((flat<true, 320>), dim3(ncol), dim3(320), 0, 0, o0, o1, o2, o3, nb, nz, T); });
      printf("  block320 nt %7.1f", tot / t / 1e6);
      t = timeit([&] { hipLaunchKernelGGL((flat<false, 320>), dim3(ncol), dim3(320), 0, 0, o0, o1, o2, o3, nb, nz, T); });
      printf("  plain %7.1f GB/s\n", tot / t / 1e6);
    }
  }
  return 0;
}
