// How many storing waves does a CU need?  Pure stores, nb=300, NARR arrays, one workgroup per column (1 WG/CU forced by a
// 150 KB LDS allocation, or 2 WG/CU with 75 KB), flat flush of T levels, W waves per workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/store_bw5.hip -o /tmp/store_bw5 && /tmp/store_bw5
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int NARR>
__global__ __launch_bounds__(1024) void flat(double* base0, size_t per, int nb, int nz, int T) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = 1.0;
  const long long base = (long long)blockIdx.x * nz * nb;
  const int chunk2 = T * nb / 2;
  for (int j0 = 0; j0 < nz; j0 += T) {
    for (int i = threadIdx.x; i < chunk2; i += blockDim.x) {
#pragma unroll
      for (int a = 0; a < NARR; ++a) { d2 v; v.x = i; v.y = a; ((d2*)(base0 + a * per + base + (long long)j0 * nb))[i] = v; }
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
  const int ncol = 10000, nz = 60, nb = 300;
  size_t per = (size_t)ncol * nz * nb;
  void* buf; CK(hipMalloc(&buf, 7 * per * 8));
  double* o0 = (double*)buf;
  CK(hipFuncSetAttribute((const void*)flat<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)flat<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)flat<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int lds_kb : {150, 75}) {
    for (int T : {4, 12}) {
      for (int W : {1, 2, 3, 4, 5, 8, 12, 16}) {
        float t4 = timeit([&] { hipLaunchKernelGGL((flat<4>), dim3(ncol), dim3(64 * W), lds_kb * 1024, 0, o0, per, nb, nz, T); });
        float t6 = timeit([&] { hipLaunchKernelGGL((flat<6>), dim3(ncol), dim3(64 * W), lds_kb * 1024, 0, o0, per, nb, nz, T); });
        float t7 = timeit([&] { hipLaunchKernelGGL((flat<7>), dim3(ncol), dim3(64 * W), lds_kb * 1024, 0, o0, per, nb, nz, T); });
        printf("LDS %3d KB (%d WG/CU) T %2d  W %2d waves/WG: 4 arrays %7.1f GB/s   6 arrays %7.1f   7 arrays %7.1f\n", lds_kb, 160 / lds_kb, T, W,
               4.0 * per * 8 / t4 / 1e6, 6.0 * per * 8 / t6 / 1e6, 7.0 * per * 8 / t7 / 1e6);
      }
    }
  }
  return 0;
}
