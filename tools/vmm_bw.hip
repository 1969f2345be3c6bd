// Does the physical layout of the output arrays decide the speed of the column-pattern stores?
// Pure-store kernel (4 arrays [ncol][nz][nb], one workgroup per column, flat flush of T levels, as the solve kernels do),
// arrays allocated (a) with one hipMalloc each, back to back, (b) through the virtual-memory API from physical chunks of
// CHUNK bytes that are created in one order and mapped in a shuffled order (decorrelates the arrays' physical phases).
//   hipcc -O3 --offload-arch=gfx950 tools/vmm_bw.hip -o tools/vmm_bw.bin && tools/vmm_bw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(512) void flat4(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = 1.0;
  const long long base = (long long)blockIdx.x * nz * nb;
  const int chunk2 = T * nb / 2;
  for (int j0 = 0; j0 < nz; j0 += T)
    for (int i = threadIdx.x; i < chunk2; i += blockDim.x) {
      d2 v; v.x = i; v.y = j0;
      ((d2*)(o0 + base + (long long)j0 * nb))[i] = v;
      ((d2*)(o1 + base + (long long)j0 * nb))[i] = v;
      ((d2*)(o2 + base + (long long)j0 * nb))[i] = v;
      ((d2*)(o3 + base + (long long)j0 * nb))[i] = v;
    }
}
__global__ void fill(d2* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { d2 v; v.x = 1; v.y = 2; p[i] = v; }
}
template <typename F> float timeit(F f, int rep = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
static const int ncol = 10000, nz = 60, nb = 300;
static const size_t per = (size_t)ncol * nz * nb * 8;
double run4(double** o) {
  CK(hipFuncSetAttribute((const void*)flat4, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024));
  float t = timeit([&] { hipLaunchKernelGGL(flat4, dim3(ncol), dim3(512), 78 * 1024, 0, o[0], o[1], o[2], o[3], nb, nz, 4); });
  return 4.0 * per / t / 1e6;
}
struct VmmArr { void* va; size_t size; std::vector<hipMemGenericAllocationHandle_t> h; };
// one handle for the whole array (size rounded to the granularity), or 1 GB handles plus one tail handle
VmmArr vmm_alloc_whole(size_t bytes, bool gb_plus_tail) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  VmmArr a; a.size = ((bytes + gran - 1) / gran) * gran;
  CK(hipMemAddressReserve(&a.va, a.size, 0, nullptr, 0));
  size_t off = 0;
  while (off < a.size) {
    size_t c = gb_plus_tail ? std::min<size_t>(1ull << 30, a.size - off) : a.size;
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, c, &prop, 0));
    CK(hipMemMap((char*)a.va + off, c, 0, h, 0));
    a.h.push_back(h); off += c;
  }
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(a.va, a.size, &acc, 1));
  return a;
}
VmmArr vmm_alloc(size_t bytes, size_t chunk, std::mt19937& rng, bool shuffle) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  chunk = ((chunk + gran - 1) / gran) * gran;
  VmmArr a; a.size = ((bytes + chunk - 1) / chunk) * chunk;
  CK(hipMemAddressReserve(&a.va, a.size, 0, nullptr, 0));
  const size_t n = a.size / chunk;
  a.h.resize(n);
  for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&a.h[i], chunk, &prop, 0));
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  if (shuffle) std::shuffle(order.begin(), order.end(), rng);
  for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)a.va + i * chunk, chunk, 0, a.h[order[i]], 0));
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(a.va, a.size, &acc, 1));
  return a;
}
void vmm_free(VmmArr& a) {
  CK(hipMemUnmap(a.va, a.size));
  for (auto h : a.h) CK(hipMemRelease(h));
  CK(hipMemAddressFree(a.va, a.size));
}
int main() {
  std::mt19937 rng(1234);
  { void* f; CK(hipMalloc(&f, 2ull << 30)); float t = timeit([&] { hipLaunchKernelGGL(fill, dim3(256), dim3(1024), 0, 0, (d2*)f, (2ull << 30) / 16); });
    printf("fill probe %.0f GB/s\n", (2ull << 30) / t / 1e6); CK(hipFree(f)); }
  std::vector<void*> keep;
  printf("hipMalloc per array, back to back:");
  for (int trial = 0; trial < 6; ++trial) {
    double* o[4];
    for (int k = 0; k < 4; ++k) { CK(hipMalloc((void**)&o[k], per)); keep.push_back(o[k]); }
    printf(" %.0f", run4(o)); fflush(stdout);
  }
  printf(" GB/s\n");
  for (void* p : keep) CK(hipFree(p));
  for (size_t chunk : {2ull << 20, 16ull << 20, 128ull << 20, 1024ull << 20}) {
    for (int shuffle = 0; shuffle < 2; ++shuffle) {
      printf("VMM chunks of %4zu MB, %s:", chunk >> 20, shuffle ? "shuffled" : "in order");
      for (int trial = 0; trial < 5; ++trial) {
        VmmArr a[4]; double* o[4];
        for (int k = 0; k < 4; ++k) { a[k] = vmm_alloc(per, chunk, rng, shuffle); o[k] = (double*)a[k].va; }
        printf(" %.0f", run4(o)); fflush(stdout);
        for (int k = 0; k < 4; ++k) vmm_free(a[k]);
      }
      printf(" GB/s\n");
    }
  }
  for (int mode = 0; mode < 2; ++mode) {
    printf("VMM %s:", mode ? "1 GB handles + tail handle" : "one handle per array (1442840576 B)");
    std::vector<VmmArr> held;
    for (int trial = 0; trial < 6; ++trial) {
      VmmArr a[4]; double* o[4];
      for (int k = 0; k < 4; ++k) { a[k] = vmm_alloc_whole(per, mode); o[k] = (double*)a[k].va; held.push_back(a[k]); }
      printf(" %.0f", run4(o)); fflush(stdout);
      if (trial == 5) {  // fill probe on VMM memory
        float t = timeit([&] { hipLaunchKernelGGL(fill, dim3(256), dim3(1024), 0, 0, (d2*)o[0], per / 16); });
        printf(" | fill of one such array %.0f", per / t / 1e6);
      }
    }
    printf(" GB/s\n");
    for (auto& a : held) vmm_free(a);
  }
  { size_t gran = 0; hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)); size_t rec = 0;
    CK(hipMemGetAllocationGranularity(&rec, &prop, hipMemAllocationGranularityRecommended)); printf("granularity min %zu recommended %zu\n", gran, rec); }
  return 0;
}
