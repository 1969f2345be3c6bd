// Store-bandwidth exploration for the store-bound solve kernels (run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/store_bw.hip -o /tmp/store_bw && /tmp/store_bw
// Variants: plain vs nontemporal, 8 vs 16 B per lane, grid size, and the solve kernel's actual
// pattern (4 arrays, rows of nb doubles, each wave writing a 512 B / 1 KiB segment per row).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT, typename T>
__global__ __launch_bounds__(256) void fill(T* dst, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}
// each block writes one contiguous chunk (block-contiguous instead of grid-strided)
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunk(d2* dst, size_t per_block, d2 v) {
  d2* p = dst + (size_t)blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i < per_block; i += 256) {
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}
// solve-kernel pattern: item = (column, band pair); loop over nz levels; NARR arrays
template <bool NT, int VEC, int NARR>
__global__ __launch_bounds__(256) void pattern(double* o0, double* o1, double* o2, double* o3, int ncol, int nb, int nz) {
  const int nbv = nb / VEC;
  long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long long)ncol * nbv) return;
  int c = item / nbv; int b = (item - (long long)c * nbv) * VEC;
  long long o = ((long long)c * nz) * nb + b;
  double* arr[4] = {o0, o1, o2, o3};
  for (int j = 0; j < nz; ++j, o += nb) {
#pragma unroll
    for (int a = 0; a < NARR; ++a) {
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
  const size_t bytes = (size_t)6 << 30;  // 6 GiB
  void* buf; CK(hipMalloc(&buf, bytes));
  d2 v2; v2.x = 1; v2.y = 2;
  printf("== linear fill of %zu MiB\n", bytes >> 20);
  for (int grid : {256, 1024, 2048, 4096, 16384, 65536}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL((fill<true, d2>), dim3(grid), dim3(256), 0, 0, (d2*)buf, bytes / 16, v2); });
    printf("grid %6d  nt 16B: %7.1f GB/s", grid, bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((fill<false, d2>), dim3(grid), dim3(256), 0, 0, (d2*)buf, bytes / 16, v2); });
    printf("   plain 16B: %7.1f GB/s", bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((fill<true, double>), dim3(grid), dim3(256), 0, 0, (double*)buf, bytes / 8, 1.0); });
    printf("   nt 8B: %7.1f", bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((fill<false, double>), dim3(grid), dim3(256), 0, 0, (double*)buf, bytes / 8, 1.0); });
    printf("   plain 8B: %7.1f", bytes / t / 1e6);
    size_t per_block = bytes / 16 / grid;
    t = timeit([&] { hipLaunchKernelGGL((fill_chunk<true>), dim3(grid), dim3(256), 0, 0, (d2*)buf, per_block, v2); });
    printf("   chunk nt: %7.1f", bytes / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((fill_chunk<false>), dim3(grid), dim3(256), 0, 0, (d2*)buf, per_block, v2); });
    printf("   chunk plain: %7.1f\n", bytes / t / 1e6);
  }
  CK(hipMemsetAsync(buf, 0, bytes, 0));
  float tm = timeit([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); });
  printf("hipMemsetAsync: %7.1f GB/s\n", bytes / tm / 1e6);
  printf("== solve-kernel store pattern, 4 arrays [ncol][nz=60][nb]\n");
  for (int nb : {300, 256, 320}) {
    const int ncol = 10000, nz = 60;
    size_t per = (size_t)ncol * nz * nb;  // doubles per array
    double* o0 = (double*)buf; double* o1 = o0 + per; double* o2 = o1 + per; double* o3 = o2 + per;
    double tot = 4.0 * per * 8;
    {
      int grid = (ncol * (nb / 2) + 255) / 256;
      float t = timeit([&] { hipLaunchKernelGGL((pattern<true, 2, 4>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
      printf("nb %d vec2 nt: %7.1f GB/s (%.3f ms)", nb, tot / t / 1e6, t);
      t = timeit([&] { hipLaunchKernelGGL((pattern<false, 2, 4>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
      printf("   vec2 plain: %7.1f", tot / t / 1e6);
    }
    {
      int grid = (ncol * nb + 255) / 256;
      float t = timeit([&] { hipLaunchKernelGGL((pattern<true, 1, 4>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
      printf("   vec1 nt: %7.1f", tot / t / 1e6);
      t = timeit([&] { hipLaunchKernelGGL((pattern<false, 1, 4>), dim3(grid), dim3(256), 0, 0, o0, o1, o2, o3, ncol, nb, nz); });
      printf("   vec1 plain: %7.1f\n", tot / t / 1e6);
    }
  }
  return 0;
}
