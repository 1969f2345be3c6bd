// Does the fast/slow mode of the column-pattern stores depend on WHICH columns are in flight together?
// Same pure-store pattern as tools/chunk_probe.hip (one workgroup per column, flat flush of T levels into NA arrays), but the
// workgroup -> column map is a parameter:
//   mode 0: column = blockIdx (the ~512 workgroups in flight cover a ~74 MB window of every array)
//   mode 1: column = (blockIdx % S) * ceil(ncol / S) + blockIdx / S   (in-flight workgroups spread over the whole array)
//   mode 2: column = (blockIdx * s) mod ncol, s coprime to ncol near 0.618 ncol
//   hipcc -O3 --offload-arch=gfx950 tools/spread_probe.hip -o tools/spread_probe.bin && tools/spread_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <numeric>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static const int nz = 60, nb = 300;
static const size_t COLB = (size_t)nz * nb * 8;
static const size_t GB = 1ull << 30;

template <int NA>
__global__ __launch_bounds__(512) void flat(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T, int ncol, int mode, int S) {
  int c = blockIdx.x;
  if (mode == 1) {
    const int per = (ncol + S - 1) / S;
    c = (blockIdx.x % S) * per + blockIdx.x / S;
    if (c >= ncol) return;
  } else if (mode == 2) {
    c = (int)(((long long)blockIdx.x * S) % ncol);
  }
  const long long base = (long long)c * nz * nb;
  for (int j0 = 0; j0 < nz; j0 += T) {
    const int n2 = min(T, nz - j0) * nb / 2;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
      d2 v; v.x = i; v.y = j0;
      ((d2*)(o0 + base + (long long)j0 * nb))[i] = v;
      if (NA > 1) ((d2*)(o1 + base + (long long)j0 * nb))[i] = v;
      if (NA > 2) ((d2*)(o2 + base + (long long)j0 * nb))[i] = v;
      if (NA > 3) ((d2*)(o3 + base + (long long)j0 * nb))[i] = v;
    }
  }
}
template <typename F> float timeit(F f, int rep = 5) {
  static hipEvent_t a = nullptr, b = nullptr;
  if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
template <int NA>
double rate(char* p0, char* p1, char* p2, char* p3, int ncol, int mode, int S) {
  int grid = ncol;
  if (mode == 1) grid = ((ncol + S - 1) / S) * S;
  float t = timeit([&] { hipLaunchKernelGGL(flat<NA>, dim3(grid), dim3(512), 0, 0, (double*)p0, (double*)p1, (double*)p2, (double*)p3, nb, nz, 8, ncol, mode, S); });
  return (double)NA * ncol * COLB / t / 1e6;
}
int coprime_near(int n, double frac) {
  int s = (int)(n * frac);
  while (std::gcd(s, n) != 1) ++s;
  return s;
}
int main() {
  const int NCH = 64;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  char* p = (char*)va;
  const int nc = 10000;
  const size_t per = (size_t)nc * COLB, pitch = ((per + (2 << 20) - 1) >> 21) << 21;
  const int sg = coprime_near(nc, 0.6180339887), s3 = coprime_near(nc, 0.3819660113);
  printf("4 arrays of 1e4 columns back to back from chunk k (pitch %zu): plain | S=64 S=512 S=2048 | stride %d, %d\n", pitch, sg, s3);
  for (int k = 0; k + 6 <= NCH; k += 6) {
    char* q = p + (size_t)k * GB;
    printf("  k=%2d  %5.0f | %5.0f %5.0f %5.0f | %5.0f %5.0f\n", k, rate<4>(q, q + pitch, q + 2 * pitch, q + 3 * pitch, nc, 0, 0),
           rate<4>(q, q + pitch, q + 2 * pitch, q + 3 * pitch, nc, 1, 64), rate<4>(q, q + pitch, q + 2 * pitch, q + 3 * pitch, nc, 1, 512),
           rate<4>(q, q + pitch, q + 2 * pitch, q + 3 * pitch, nc, 1, 2048), rate<4>(q, q + pitch, q + 2 * pitch, q + 3 * pitch, nc, 2, sg),
           rate<4>(q, q + pitch, q + 2 * pitch, q + 3 * pitch, nc, 2, s3));
    fflush(stdout);
  }
  printf("single array of 1e4 columns at chunk k: plain | S=64 S=512 S=2048 | stride\n");
  for (int k = 0; k + 2 <= NCH; k += 9) {
    char* q = p + (size_t)k * GB;
    printf("  k=%2d  %5.0f | %5.0f %5.0f %5.0f | %5.0f\n", k, rate<1>(q, q, q, q, nc, 0, 0), rate<1>(q, q, q, q, nc, 1, 64), rate<1>(q, q, q, q, nc, 1, 512),
           rate<1>(q, q, q, q, nc, 1, 2048), rate<1>(q, q, q, q, nc, 2, sg));
    fflush(stdout);
  }
  printf("single array of 4e4 columns (5.76 GB) at chunk k: plain | S=512 | stride\n");
  for (int k = 0; k + 6 <= NCH; k += 12) {
    char* q = p + (size_t)k * GB;
    const int n4 = 40000, s4 = coprime_near(n4, 0.6180339887);
    printf("  k=%2d  %5.0f | %5.0f | %5.0f\n", k, rate<1>(q, q, q, q, n4, 0, 0), rate<1>(q, q, q, q, n4, 1, 512), rate<1>(q, q, q, q, n4, 2, s4));
    fflush(stdout);
  }
  printf("pairs of 1e4 columns, pitch 2 GB from chunk k: plain | S=512 | stride\n");
  for (int k = 0; k + 4 <= NCH; k += 10) {
    char* q = p + (size_t)k * GB;
    printf("  k=%2d  %5.0f | %5.0f | %5.0f\n", k, rate<2>(q, q + 2 * GB, q, q, nc, 0, 0), rate<2>(q, q + 2 * GB, q, q, nc, 1, 512), rate<2>(q, q + 2 * GB, q, q, nc, 2, sg));
    fflush(stdout);
  }
  printf("4 arrays of 1e4 columns at pitch 2 GB from chunk k: plain | S=512\n");
  for (int k = 0; k + 8 <= NCH; k += 8) {
    char* q = p + (size_t)k * GB;
    printf("  k=%2d  %5.0f | %5.0f\n", k, rate<4>(q, q + 2 * GB, q + 4 * GB, q + 6 * GB, nc, 0, 0), rate<4>(q, q + 2 * GB, q + 4 * GB, q + 6 * GB, nc, 1, 512));
    fflush(stdout);
  }
  CK(hipMemUnmap(va, (size_t)NCH * GB));
  for (int i = 0; i < NCH; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)NCH * GB));
  return 0;
}
