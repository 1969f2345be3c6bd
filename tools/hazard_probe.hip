// (1) After hipMemUnmap + hipMemMap of ANOTHER handle at the same virtual address, do kernels write the new physical memory?
// (2) Can one handle be mapped at two virtual addresses at once?
// (3) Can 128 MB / 256 MB handles be classified individually (pair rate against a 1 GB reference)?
//   hipcc -O3 --offload-arch=gfx950 tools/hazard_probe.hip -o tools/hazard_probe.bin && tools/hazard_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static const int nz = 60, nb = 300;
static const size_t COLB = (size_t)nz * nb * 8;
static const size_t GB = 1ull << 30, MB = 1ull << 20;
struct Ptrs { double* o[8]; };
__global__ __launch_bounds__(512) void flatn(Ptrs P, int na, int nb, int nz, int T) {
  const long long base = (long long)blockIdx.x * nz * nb;
  for (int j0 = 0; j0 < nz; j0 += T) {
    const int n2 = min(T, nz - j0) * nb / 2;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
      d2 v; v.x = i; v.y = j0;
#pragma unroll
      for (int a = 0; a < 8; ++a)
        if (a < na) ((d2*)(P.o[a] + base + (long long)j0 * nb))[i] = v;
    }
  }
}
__global__ void setv(double* p, double v) { p[threadIdx.x] = v; }
template <typename F> float timeit(F f, int rep = 4) {
  static hipEvent_t a = nullptr, b = nullptr;
  if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
double rate(const std::vector<char*>& p, int ncol, int rep = 4) {
  Ptrs P = {};
  const int na = (int)p.size();
  for (int a = 0; a < na; ++a) P.o[a] = (double*)p[a];
  float t = timeit([&] { hipLaunchKernelGGL(flatn, dim3(ncol), dim3(512), 0, 0, P, na, nb, nz, 8); }, rep);
  return (double)na * ncol * COLB / t / 1e6;
}
double peek(void* p) { double v; CK(hipMemcpy(&v, p, 8, hipMemcpyDeviceToHost)); return v; }
int main() {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  // ---- (1) remap hazard
  hipMemGenericAllocationHandle_t hA, hB;
  CK(hipMemCreate(&hA, GB, &prop, 0)); CK(hipMemCreate(&hB, GB, &prop, 0));
  void *v1, *v2;
  CK(hipMemAddressReserve(&v1, GB, 0, nullptr, 0)); CK(hipMemAddressReserve(&v2, GB, 0, nullptr, 0));
  CK(hipMemMap(v1, GB, 0, hA, 0)); CK(hipMemSetAccess(v1, GB, &acc, 1));
  CK(hipMemMap(v2, GB, 0, hB, 0)); CK(hipMemSetAccess(v2, GB, &acc, 1));
  hipLaunchKernelGGL(setv, dim3(1), dim3(64), 0, 0, (double*)v1, 1.0);   // A <- 1
  hipLaunchKernelGGL(setv, dim3(1), dim3(64), 0, 0, (double*)v2, 2.0);   // B <- 2
  CK(hipDeviceSynchronize());
  printf("(1) before swap: v1 -> %.0f, v2 -> %.0f\n", peek(v1), peek(v2));
  CK(hipMemUnmap(v1, GB)); CK(hipMemUnmap(v2, GB));
  CK(hipMemMap(v1, GB, 0, hB, 0)); CK(hipMemMap(v2, GB, 0, hA, 0));
  CK(hipMemSetAccess(v1, GB, &acc, 1)); CK(hipMemSetAccess(v2, GB, &acc, 1));
  printf("    after swapping the handles (no kernel yet): v1 -> %.0f (expect 2), v2 -> %.0f (expect 1)\n", peek(v1), peek(v2));
  hipLaunchKernelGGL(setv, dim3(1), dim3(64), 0, 0, (double*)v1, 10.0);  // whatever v1 points at <- 10 (should be B)
  CK(hipDeviceSynchronize());
  printf("    kernel wrote 10 through v1: v1 -> %.0f, v2 -> %.0f (expect 10, 1)\n", peek(v1), peek(v2));
  // ---- (2) double mapping
  void* v3; CK(hipMemAddressReserve(&v3, GB, 0, nullptr, 0));
  hipError_t e = hipMemMap(v3, GB, 0, hA, 0);
  printf("(2) second mapping of a mapped handle: hipMemMap -> %s", hipGetErrorString(e));
  if (e == hipSuccess) {
    e = hipMemSetAccess(v3, GB, &acc, 1);
    printf(", hipMemSetAccess -> %s", hipGetErrorString(e));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(setv, dim3(1), dim3(64), 0, 0, (double*)v3, 33.0);
      CK(hipDeviceSynchronize());
      printf(", wrote 33 through v3: v2 -> %.0f, v3 -> %.0f", peek(v2), peek(v3));
    }
    (void)hipMemUnmap(v3, GB);
  }
  (void)hipGetLastError();
  printf("\n");
  CK(hipMemUnmap(v1, GB)); CK(hipMemUnmap(v2, GB));
  // ---- (3) small handles: 24 chunks of 1 GB classified, then 96 handles of 128 MB and 48 of 256 MB against the references
  const int NCH = 24;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  std::vector<int> cls(NCH, 3), ref = {0};
  cls[0] = 0;
  for (int j = 1; j < NCH; ++j) {
    int c = -1;
    for (size_t r = 0; r < ref.size(); ++r)
      if (rate({(char*)va + ref[r] * GB, (char*)va + j * GB}, 6000) < 6200) { c = (int)r; break; }
    if (c < 0 && ref.size() < 3) { c = (int)ref.size(); ref.push_back(j); }
    cls[j] = c < 0 ? 3 : c;
  }
  printf("(3) 1 GB chunks: ");
  for (int j = 0; j < NCH; ++j) printf("%c", "XYZ?"[cls[j]]);
  printf("   references:");
  for (int r : ref) printf(" %d", r);
  printf("\n");
  for (size_t sz : {128 * MB, 256 * MB}) {
    const int n = (int)(12 * GB / sz), nc = (int)(sz / COLB);
    std::vector<hipMemGenericAllocationHandle_t> s(n);
    void* vs; CK(hipMemAddressReserve(&vs, (size_t)n * sz, 0, nullptr, 0));
    for (int i = 0; i < n; ++i) { CK(hipMemCreate(&s[i], sz, &prop, 0)); CK(hipMemMap((char*)vs + i * sz, sz, 0, s[i], 0)); }
    CK(hipMemSetAccess(vs, (size_t)n * sz, &acc, 1));
    printf("    %zu MB handles (%d columns each), pair rate against each reference chunk's first %d columns:\n", sz / MB, nc, nc);
    for (size_t r = 0; r < ref.size(); ++r) {
      printf("      vs ref %c:", "XYZ"[r]);
      for (int i = 0; i < n; ++i) printf(" %4.0f", rate({(char*)va + ref[r] * GB, (char*)vs + i * sz}, nc, 8) / 10);
      printf("  (x10 GB/s)\n");
    }
    // groups of 512 MB formed from consecutive small handles (they are contiguous in VA): 3700 columns
    printf("      512 MB groups vs refs:");
    for (size_t r = 0; r < ref.size(); ++r) {
      printf("  [%c]", "XYZ"[r]);
      for (size_t g = 0; g + 512 * MB <= (size_t)n * sz; g += 512 * MB) printf(" %4.0f", rate({(char*)va + ref[r] * GB, (char*)vs + g}, 3700, 6) / 10);
    }
    printf("\n");
    CK(hipMemUnmap(vs, (size_t)n * sz));
    for (int i = 0; i < n; ++i) CK(hipMemRelease(s[i]));
  }
  return 0;
}
