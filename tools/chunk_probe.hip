// Where does the fast/slow mode of the column-pattern stores come from?
// Holds NCH physical chunks of 1 GB (hipMemCreate) at once and measures, per chunk and alone: a linear fill, the column
// pattern into one array inside the chunk, and the column pattern into four arrays inside the chunk.  Then builds four output
// arrays [ncol][60][300] from (a) the chunks that were fastest alone, (b) the slowest, and runs the 4-array column pattern
// on them.  Second part: hipMalloc'ed sets back to back (held), each set timed as a whole and each of its arrays alone.
//   hipcc -O3 --offload-arch=gfx950 tools/chunk_probe.hip -o tools/chunk_probe.bin && tools/chunk_probe.bin [NCH]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static const int nz = 60, nb = 300, T = 8;
static const size_t COLB = (size_t)nz * nb * 8;  // bytes per column and array
static const size_t GB = 1ull << 30;

// NA arrays, one workgroup per column, flat flush of T levels per round (the store pattern of k_pipe)
template <int NA>
__global__ __launch_bounds__(512) void flat(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T) {
  const long long base = (long long)blockIdx.x * nz * nb;
  const int chunk2 = T * nb / 2;
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
  (void)chunk2;
}
__global__ __launch_bounds__(256) void fill(d2* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { d2 v; v.x = 1; v.y = 2; __builtin_nontemporal_store(v, p + i); }
}
template <typename F> float timeit(F f, int rep = 5) {
  static hipEvent_t a = nullptr, b = nullptr;
  if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
double col1(double* p, int ncol) {
  float t = timeit([&] { hipLaunchKernelGGL(flat<1>, dim3(ncol), dim3(512), 0, 0, p, p, p, p, nb, nz, T); });
  return (double)ncol * COLB / t / 1e6;
}
double col4(double* o0, double* o1, double* o2, double* o3, int ncol) {
  float t = timeit([&] { hipLaunchKernelGGL(flat<4>, dim3(ncol), dim3(512), 0, 0, o0, o1, o2, o3, nb, nz, T); });
  return 4.0 * ncol * COLB / t / 1e6;
}
double fill1(void* p, size_t bytes) {
  float t = timeit([&] { hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (d2*)p, bytes / 16); });
  return bytes / t / 1e6;
}

int main(int argc, char** argv) {
  const int NCH = argc > 1 ? atoi(argv[1]) : 160;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
  printf("free %.1f GB of %.1f GB; holding %d chunks of 1 GB\n", fr / 1e9, tot / 1e9, NCH);
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  const int nc1 = (int)(GB / COLB), nc4 = (int)(GB / 4 / COLB);
  std::vector<double> s1(NCH), s4(NCH), sf(NCH);
  printf("chunk  fill  col1  col4   (GB/s; col = column-pattern stores, %d / 4 x %d columns)\n", nc1, nc4);
  for (int i = 0; i < NCH; ++i) {
    char* p = (char*)va + (size_t)i * GB;
    sf[i] = fill1(p, GB);
    s1[i] = col1((double*)p, nc1);
    s4[i] = col4((double*)p, (double*)(p + GB / 4), (double*)(p + GB / 2), (double*)(p + 3 * GB / 4), nc4);
    printf("%3d  %5.0f %5.0f %5.0f\n", i, sf[i], s1[i], s4[i]); fflush(stdout);
  }
  // arrays across adjacent chunks in the held VA: the full 2s output set of 1e4 columns (4 x 1.44 GB) at successive offsets
  const int ncol = 10000; const size_t per = (size_t)ncol * COLB, pitch = ((per + (2 << 20) - 1) >> 21) << 21;
  printf("4 arrays of %d columns laid out back to back (pitch %zu) starting at chunk k:\n", ncol, pitch);
  for (int k = 0; k + 6 <= NCH; k += 6) {
    char* p = (char*)va + (size_t)k * GB;
    printf("  k=%3d  %5.0f GB/s\n", k, col4((double*)p, (double*)(p + pitch), (double*)(p + 2 * pitch), (double*)(p + 3 * pitch), ncol)); fflush(stdout);
  }
  // sets built from the chunks that were fastest / slowest alone (2 chunks per array)
  std::vector<int> ord(NCH); for (int i = 0; i < NCH; ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](int x, int y) { return s1[x] > s1[y]; });
  CK(hipMemUnmap(va, (size_t)NCH * GB));
  for (int which = 0; which < 3; ++which) {
    void* a4[4];
    int used[8];
    for (int k = 0; k < 4; ++k) {
      CK(hipMemAddressReserve(&a4[k], 2 * GB, 0, nullptr, 0));
      for (int q = 0; q < 2; ++q) {
        const int r = k * 2 + q;
        const int idx = which == 0 ? ord[r] : which == 1 ? ord[NCH - 1 - r] : ord[(r % 2) ? NCH - 1 - r : r];
        used[r] = idx;
        CK(hipMemMap((char*)a4[k] + q * GB, GB, 0, h[idx], 0));
      }
      CK(hipMemSetAccess(a4[k], 2 * GB, &acc, 1));
    }
    double r = col4((double*)a4[0], (double*)a4[1], (double*)a4[2], (double*)a4[3], ncol);
    printf("set from the %s chunks [", which == 0 ? "FASTEST" : which == 1 ? "SLOWEST" : "fast+slow alternating");
    for (int q = 0; q < 8; ++q) printf("%d(%.0f) ", used[q], s1[used[q]]);
    printf("]: %5.0f GB/s", r);
    for (int k = 0; k < 4; ++k) printf("  a%d alone %5.0f", k, col1((double*)a4[k], ncol));
    printf("\n"); fflush(stdout);
    for (int k = 0; k < 4; ++k) { CK(hipMemUnmap(a4[k], 2 * GB)); CK(hipMemAddressFree(a4[k], 2 * GB)); }
  }
  for (int i = 0; i < NCH; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)NCH * GB));

  // hipMalloc'ed sets, back to back and held: whole set, then every array alone, then pairs (0,1) and (2,3)
  printf("hipMalloc sets (held): set | each array alone | pairs\n");
  std::vector<void*> keep;
  for (int trial = 0; trial < 12; ++trial) {
    double* o[4];
    for (int k = 0; k < 4; ++k) { CK(hipMalloc((void**)&o[k], per)); keep.push_back(o[k]); }
    printf("  set %2d %5.0f |", trial, col4(o[0], o[1], o[2], o[3], ncol));
    for (int k = 0; k < 4; ++k) printf(" %5.0f", col1(o[k], ncol));
    float t = timeit([&] { hipLaunchKernelGGL(flat<2>, dim3(ncol), dim3(512), 0, 0, o[0], o[1], o[0], o[1], nb, nz, T); });
    printf(" | %5.0f", 2.0 * per / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(flat<2>, dim3(ncol), dim3(512), 0, 0, o[2], o[3], o[2], o[3], nb, nz, T); });
    printf(" %5.0f | fill %5.0f\n", 2.0 * per / t / 1e6, fill1(o[0], per / 16 * 16)); fflush(stdout);
  }
  for (void* p : keep) CK(hipFree(p));
  return 0;
}
