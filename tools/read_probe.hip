// Streaming-read probe (tools; not product): ncol columns of `colbytes` per array, NA arrays; one wave per column, the wave reads its
// column in bursts of `burst` bytes per array (16 B per lane per load), waits for the burst (optionally one burst ahead) and goes on.
// Parameters explored: waves per workgroup, dynamic LDS per workgroup (occupancy), burst size, number of arrays, nt loads, prefetch.
// build: hipcc -O3 --offload-arch=gfx950 tools/read_probe.hip -o tools/read_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NA, int NLD, bool NT, bool PF>
__global__ __launch_bounds__(256) void k_read(const d2* a0, const d2* a1, const d2* a2, long long col_d2, int ncol, int nburst, double* sink) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + wave;
  if (c >= ncol) return;
  const d2* p[3] = {a0 + c * col_d2, a1 + c * col_d2, a2 + c * col_d2};
  d2 acc = {0.0, 0.0};
  d2 v[2][NA][NLD];
  auto fetch = [&](int s, int q) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const d2* src = p[a] + (long long)s * NLD * 64 + i * 64 + lane;
        v[q][a][i] = NT ? __builtin_nontemporal_load(src) : *src;
      }
  };
  auto use = [&](int q) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int i = 0; i < NLD; ++i) acc += v[q][a][i];
  };
  if (PF) {
    fetch(0, 0);
    for (int s = 0; s < nburst; s += 2) {
      if (s + 1 < nburst) fetch(s + 1, 1);
      use(0);
      if (s + 1 < nburst) {
        if (s + 2 < nburst) fetch(s + 2, 0);
        use(1);
      }
    }
  } else {
    for (int s = 0; s < nburst; ++s) {
      fetch(s, 0);
      use(0);
    }
  }
  if (acc.x + acc.y == 1.2345e300) sink[0] = acc.x;
  if (lds[0] == 1.2345e300) sink[1] = lds[0];
}

template <int NA, int NLD, bool NT, bool PF>
double run(const d2* a0, const d2* a1, const d2* a2, long long col_d2, int ncol, int wpb, size_t lds, double* sink) {
  const int nburst = (int)(col_d2 / (NLD * 64));
  auto kern = k_read<NA, NLD, NT, PF>;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 grid((ncol + wpb - 1) / wpb), block(64 * wpb);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, a0, a1, a2, col_d2, ncol, nburst, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, a0, a1, a2, col_d2, ncol, nburst, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)NA * nburst * NLD * 1024.0 * ncol / (ms / 5 * 1e-3) / 1e12;
}

__global__ void k_fill(double* p, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = 1.0 + 1e-9 * (double)((i * 2654435761ull) & 0xffffff);
}
extern "C" int crt_hip_buffer_alloc_set(int n, const size_t* bytes, void** ptrs);
extern "C" int crt_hip_buffer_describe(const void* ptr, char* buf, size_t n);

int main(int argc, char** argv) {
  const int ncol = 100000;
  const bool use_set = argc > 1 && atoi(argv[1]) == 1;  // 1: arrays from the library's class-interleaving set allocator
  const long long col_bytes = argc > 2 ? atoll(argv[2]) : 30720;  // per array per column
  const long long col_d2 = col_bytes / 16;
  d2 *a0, *a1, *a2;
  double* sink;
  if (use_set) {
    size_t bytes[3] = {(size_t)(ncol * col_bytes), (size_t)(ncol * col_bytes), (size_t)(ncol * col_bytes)};
    void* ptrs[3];
    if (crt_hip_buffer_alloc_set(3, bytes, ptrs) != 0) { printf("alloc_set failed\n"); return 1; }
    a0 = (d2*)ptrs[0], a1 = (d2*)ptrs[1], a2 = (d2*)ptrs[2];
    char buf[64];
    for (int i = 0; i < 3; ++i) { crt_hip_buffer_describe(ptrs[i], buf, sizeof buf); printf("array %d classes %s\n", i, buf); }
  } else {
    CK(hipMalloc(&a0, ncol * col_bytes));
    CK(hipMalloc(&a1, ncol * col_bytes));
    CK(hipMalloc(&a2, ncol * col_bytes));
  }
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a0, 0, ncol * col_bytes));
  CK(hipMemset(a1, 0, ncol * col_bytes));
  CK(hipMemset(a2, 0, ncol * col_bytes));
  if (argc > 3 && atoi(argv[3]) == 1) {  // varied data instead of zeros
    for (d2* a : {a0, a1, a2}) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)a, ncol * col_bytes / 8);
    CK(hipDeviceSynchronize());
    printf("arrays hold varied data\n");
  }
  printf("NA burstKB nt pf wpb ldsKB/wave  TB/s\n");
  for (int wpb : {1, 4})
    for (size_t ldsw : {(size_t)4096, (size_t)22528}) {
      const size_t lds = ldsw * wpb;
#define RUN(NA, NLD, NT, PF) printf("%d %5.1f %d %d %d %5.1f   %.2f\n", NA, NLD * 1.0, NT, PF, wpb, ldsw / 1024.0, run<NA, NLD, NT, PF>(a0, a1, a2, col_d2, ncol, wpb, lds, sink)); fflush(stdout);
      RUN(3, 2, true, false) RUN(3, 5, true, false) RUN(3, 5, false, false) RUN(3, 5, true, true) RUN(1, 5, true, false)
    }
  return 0;
}
