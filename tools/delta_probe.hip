// Which RELATIVE placement of the output arrays gives the fast mode of the column-pattern stores?
// tools/chunk_probe.hip showed: every 1 GB physical chunk alone runs the pattern at the same rate (5.3-5.7 TB/s), any four
// arrays that each start at the beginning of a 1 GB physical chunk run at 7.0 TB/s, arrays laid out back to back mostly at 5.6.
// Here: physical chunks of 1 GB mapped in order into one VA range; array 0 at chunk 0, array i at i * (BASE + delta) for a
// sweep of delta -> rate of the 2-array and 4-array pattern as a function of the relative offset.
//   hipcc -O3 --offload-arch=gfx950 tools/delta_probe.hip -o tools/delta_probe.bin && tools/delta_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static const int nz = 60, nb = 300;
static const size_t COLB = (size_t)nz * nb * 8;
static const size_t GB = 1ull << 30;

template <int NA>
__global__ __launch_bounds__(512) void flat(double* o0, double* o1, double* o2, double* o3, int nb, int nz, int T) {
  const long long base = (long long)blockIdx.x * nz * nb;
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
double rate(char* p0, char* p1, char* p2, char* p3, int ncol, int T) {
  float t = timeit([&] { hipLaunchKernelGGL(flat<NA>, dim3(ncol), dim3(512), 0, 0, (double*)p0, (double*)p1, (double*)p2, (double*)p3, nb, nz, T); });
  return (double)NA * ncol * COLB / t / 1e6;
}

int main(int argc, char** argv) {
  const int NCH = 16;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, GB, nullptr, 0));
  printf("VA base %p (mod 1 GB = %zu)\n", va, (size_t)va % GB);
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  char* p = (char*)va;
  const int ncol = 6000;  // 0.864 GB per array
  const int T = 8;
  printf("single array at chunk 0: %.0f GB/s; at chunk 5 + 300 MB: %.0f\n", rate<1>(p, p, p, p, ncol, T), rate<1>(p + 5 * GB + (300 << 20), p, p, p, ncol, T));
  // delta sweep: array i at i * (2 GB + delta)   (4 arrays: needs 3 * (2 GB + delta) + 0.864 GB <= 16 GB)
  std::vector<size_t> deltas = {0};
  for (size_t d = 128; d <= (1ull << 30); d <<= 1) deltas.push_back(d);
  for (size_t d : {(size_t)384, (size_t)3 << 10, (size_t)3 << 12, (size_t)5 << 12, (size_t)3 << 16, (size_t)3 << 20, (size_t)5 << 21, (size_t)3 << 24, (size_t)350 << 20,
                   (size_t)3 << 28, ((size_t)1 << 30) + ((size_t)1 << 29)})
    deltas.push_back(d);
  printf("delta(B)        pair   quad    [array i at i * (2 GB + delta)]\n");
  for (size_t d : deltas) {
    const size_t pitch = 2 * GB + d;
    if (3 * pitch + (size_t)ncol * COLB > (size_t)NCH * GB) continue;
    printf("%12zu  %6.0f %6.0f\n", d, rate<2>(p, p + pitch, p, p, ncol, T), rate<4>(p, p + pitch, p + 2 * pitch, p + 3 * pitch, ncol, T));
    fflush(stdout);
  }
  // pitch = exactly 1 GB multiples and the array size itself (back to back, 2 MB rounded)
  const size_t per = (size_t)ncol * COLB, per2m = ((per + (2 << 20) - 1) >> 21) << 21;
  printf("pitch = array size rounded to 2 MB (%zu): pair %.0f quad %.0f\n", per2m, rate<2>(p, p + per2m, p, p, ncol, T),
         rate<4>(p, p + per2m, p + 2 * per2m, p + 3 * per2m, ncol, T));
  for (size_t al : {(size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)512 << 20, GB}) {
    const size_t pitch = ((per + al - 1) / al) * al;
    printf("pitch rounded up to %4zu MB (%zu): pair %.0f quad %.0f\n", al >> 20, pitch, rate<2>(p, p + pitch, p, p, ncol, T),
           rate<4>(p, p + pitch, p + 2 * pitch, p + 3 * pitch, ncol, T));
  }
  // the real case: 1e4 columns (1.44 GB per array), pitch rounded up to various alignments; also T = 4
  {
    const int nc = 10000; const size_t pr = (size_t)nc * COLB;
    for (size_t al : {(size_t)2 << 20, (size_t)64 << 20, (size_t)256 << 20, (size_t)512 << 20, GB, 2 * GB}) {
      const size_t pitch = ((pr + al - 1) / al) * al;
      if (3 * pitch + pr > (size_t)NCH * GB) continue;
      printf("1e4 columns, pitch rounded up to %4zu MB (%zu): quad T=8 %.0f  T=4 %.0f  T=60 %.0f\n", al >> 20, pitch,
             rate<4>(p, p + pitch, p + 2 * pitch, p + 3 * pitch, nc, 8), rate<4>(p, p + pitch, p + 2 * pitch, p + 3 * pitch, nc, 4),
             rate<4>(p, p + pitch, p + 2 * pitch, p + 3 * pitch, nc, 60));
    }
    // same array base for the whole set but the START of the set shifted inside the chunk (absolute phase)
    for (size_t sh : {(size_t)0, (size_t)4096, (size_t)1 << 20, (size_t)100 << 20, (size_t)512 << 20}) {
      const size_t pitch = 2 * GB;
      printf("1e4 columns, pitch 2 GB, set shifted by %zu: quad %.0f\n", sh, rate<4>(p + sh, p + sh + pitch, p + sh + 2 * pitch, p + sh + 3 * pitch, nc, 8));
    }
    // 7 arrays' worth (zq): two launches are not the same thing; emulate with 4 + pair on further chunks
  }
  CK(hipMemUnmap(va, (size_t)NCH * GB));
  for (int i = 0; i < NCH; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)NCH * GB));
  return 0;
}
