// Follow-up to vmm_bw.hip: in the fast mode, does the column-pattern store rate depend on how many waves of the workgroup store?
// 512-thread workgroups, 2 per CU (78 KB LDS), arrays from hipMalloc (several sets kept alive; the fastest set is used).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(512) void flat4(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T, int first_store_thread, int lds_reads) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = 1.0;
  if ((int)threadIdx.x < first_store_thread) return;
  const int sid = threadIdx.x - first_store_thread, nst = blockDim.x - first_store_thread;
  const long long base = (long long)blockIdx.x * nz * nb;
  const int chunk2 = T * nb / 2;
  const d2* l2 = (const d2*)lds;
  for (int j0 = 0; j0 < nz; j0 += T)
    for (int i = sid; i < chunk2; i += nst) {
      d2 v; v.x = i; v.y = j0;
      d2 a = v, b = v, c = v, d = v;
      if (lds_reads) { a = l2[i]; b = l2[i + chunk2]; c = l2[i + 2 * chunk2]; d = l2[i + 3 * chunk2]; }
      ((d2*)(o0 + base + (long long)j0 * nb))[i] = a;
      ((d2*)(o1 + base + (long long)j0 * nb))[i] = b;
      ((d2*)(o2 + base + (long long)j0 * nb))[i] = c;
      ((d2*)(o3 + base + (long long)j0 * nb))[i] = d;
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
  const size_t per = (size_t)ncol * nz * nb * 8;
  CK(hipFuncSetAttribute((const void*)flat4, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024));
  double* best[4] = {0, 0, 0, 0}; double best_rate = 0;
  for (int trial = 0; trial < 8; ++trial) {
    double* o[4];
    void* pad; CK(hipMalloc(&pad, (size_t)(37 + 61 * trial) << 20));
    for (int k = 0; k < 4; ++k) CK(hipMalloc((void**)&o[k], per));
    float t = timeit([&] { hipLaunchKernelGGL(flat4, dim3(ncol), dim3(512), 78 * 1024, 0, o[0], o[1], o[2], o[3], nb, nz, 4, 0, 0); }, 5);
    double r = 4.0 * per / t / 1e6;
    printf("set %d: %.0f GB/s\n", trial, r);
    if (r > best_rate) { best_rate = r; for (int k = 0; k < 4; ++k) best[k] = o[k]; }
  }
  printf("using the fastest set (%.0f GB/s)\n", best_rate);
  for (int lds_reads = 0; lds_reads < 2; ++lds_reads)
    for (int first : {0, 128, 256, 320, 384, 448}) {
      float t = timeit([&] { hipLaunchKernelGGL(flat4, dim3(ncol), dim3(512), 78 * 1024, 0, best[0], best[1], best[2], best[3], nb, nz, 4, first, lds_reads); });
      printf("storing waves %d of 8, LDS reads %d: %.0f GB/s\n", (512 - first) / 64, lds_reads, 4.0 * per / t / 1e6);
    }
  return 0;
}
