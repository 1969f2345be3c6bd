// Third store-pattern probe (nb = 300 rows, 4 arrays): can partial 128-B lines merge in L2 without LDS staging?
//  (C) block-per-column, lane-per-band direct stores (partial lines only between waves of one block / consecutive rows)
//  (D) flattened lane-per-band pattern with XCD-aware block remap (neighbouring band chunks on the same XCD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <bool NT, int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void percol(double* o0, double* o1, double* o2, double* o3, int nb, int nz) {
  const int b = threadIdx.x * VEC;
  if (b >= nb) return;
  long long o = ((long long)blockIdx.x * nz) * nb + b;
  double* arr[4] = {o0, o1, o2, o3};
  for (int j = 0; j < nz; ++j, o += nb) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (VEC == 2) { d2 v; v.x = j; v.y = a; if (NT) __builtin_nontemporal_store(v, (d2*)(arr[a] + o)); else *(d2*)(arr[a] + o) = v; }
      else { double v = j + a; if (NT) __builtin_nontemporal_store(v, arr[a] + o); else arr[a][o] = v; }
    }
  }
}
template <bool NT, int VEC, bool REMAP>
__global__ __launch_bounds__(256) void pattern(double* o0, double* o1, double* o2, double* o3, int ncol, int nb, int nz) {
  const int nbv = nb / VEC;
  unsigned bid = blockIdx.x;
  if (REMAP) { unsigned per = gridDim.x / 8; if (bid < per * 8) bid = (bid % 8) * per + bid / 8; }
  long long item = (long long)bid * 256 + threadIdx.x;
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
template <typename F> float timeit(F f, int rep = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
int main() {
  const size_t bytes = (size_t)7 << 30;
  void* buf; CK(hipMalloc(&buf, bytes));
  const int ncol = 10000, nz = 60, nb = 300;
  size_t per = (size_t)ncol * nz * nb;
  double* o0 = (double*)buf; double* o1 = o0 + per; double* o2 = o1 + per; double* o3 = o2 + per;
  double tot = 4.0 * per * 8;
  float t;
  t = timeit([&] { hipLaunchKernelGGL((percol<true, 1, 320>), dim3(ncol), dim3(320), 0, 0, o0, o1, o2, o3, nb, nz); });
  printf("(C) per-column vec1 block320 nt    %7.1f GB/s\n", tot / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((percol<false, 1, 320>), dim3(ncol), dim3(320), 0, 0, o0, o1, o2, o3, nb, nz); });
  printf("(C) per-column vec1 block320 plain %7.1f GB/s\n", tot / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((percol<true, 2, 192>), dim3(ncol), dim3(192), 0, 0, o0, o1, o2, o3, nb, nz); });
  printf("(C) per-column vec2 block192 nt    %7.1f GB/s\n", tot / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((percol<false, 2, 192>), dim3(ncol), dim3(192), 0, 0, o0, o1, o2, o3, nb, nz); });
  printf("(C) per-column vec2 block192 plain %7.1f GB/s\n", tot / t / 1e6);
  int grid = (ncol * (nb / 2) + 255) / 256;
  t = timeit([&] { hipLaunchKernelGGL((pattern<true, 2, false>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
  printf("(D) flattened vec2 nt              %7.1f GB/s\n", tot / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((pattern<true, 2, true>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
  printf("(D) flattened vec2 nt    XCD-remap %7.1f GB/s\n", tot / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((pattern<false, 2, true>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
  printf("(D) flattened vec2 plain XCD-remap %7.1f GB/s\n", tot / t / 1e6);
  return 0;
}
