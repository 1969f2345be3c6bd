// Does occupancy cap the flat (tile) flush?  Pure stores, nb=300, 4 arrays, block-per-column flat flush of T levels,
// dynamic LDS used only to limit resident workgroups per CU.  Also: two columns per workgroup, and a variant that
// interleaves the 4 arrays per iteration instead of array-by-array.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int BLOCK, bool INTER>
__global__ __launch_bounds__(BLOCK) void flat(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = 1.0;  // keep the allocation
  const long long base = (long long)blockIdx.x * nz * nb;
  double* arr[4] = {o0 + base, o1 + base, o2 + base, o3 + base};
  const int chunk2 = T * nb / 2;
  for (int j0 = 0; j0 < nz; j0 += T) {
    if (INTER) {
      for (int i = threadIdx.x; i < chunk2; i += BLOCK) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { d2 v; v.x = i; v.y = a; ((d2*)(arr[a] + (long long)j0 * nb))[i] = v; }
      }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        d2* p = (d2*)(arr[a] + (long long)j0 * nb);
        for (int i = threadIdx.x; i < chunk2; i += BLOCK) { d2 v; v.x = i; v.y = a; p[i] = v; }
      }
    }
    __builtin_amdgcn_s_barrier();
  }
}
template <typename F> float timeit(F f, int rep = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
int main() {
  const int ncol = 10000, nz = 60, nb = 300;
  size_t per = (size_t)ncol * nz * nb;
  void* buf; CK(hipMalloc(&buf, 4 * per * 8));
  double* o0 = (double*)buf; double* o1 = o0 + per; double* o2 = o1 + per; double* o3 = o2 + per;
  double tot = 4.0 * per * 8;
  CK(hipFuncSetAttribute((const void*)flat<320, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)flat<320, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)flat<640, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int T : {4, 8, 20}) {
    for (int lds_kb : {1, 20, 39, 52, 78, 156}) {
      float t = timeit([&] { hipLaunchKernelGGL((flat<320, false>), dim3(ncol), dim3(320), lds_kb * 1024, 0, o0, o1, o2, o3, nb, nz, T); });
      float t2 = timeit([&] { hipLaunchKernelGGL((flat<320, true>), dim3(ncol), dim3(320), lds_kb * 1024, 0, o0, o1, o2, o3, nb, nz, T); });
      float t3 = timeit([&] { hipLaunchKernelGGL((flat<640, false>), dim3(ncol), dim3(640), lds_kb * 1024, 0, o0, o1, o2, o3, nb, nz, T); });
      printf("T %2d  LDS %3d KB (<= %d WG/CU): block320 %7.1f GB/s  interleaved %7.1f  block640 %7.1f\n", T, lds_kb, 160 / lds_kb, tot / t / 1e6, tot / t2 / 1e6, tot / t3 / 1e6);
    }
  }
  return 0;
}
