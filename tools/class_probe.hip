// Follow-up of tools/matrix_probe.hip (pairs of 1 GB chunks fall into three classes: same class slow, different classes fast).
// Classifies NCH held chunks against three reference chunks, then measures k-array sets (one 6000-column array per chunk)
// drawn from ONE class (consecutive chunks of a run / chunks far apart) and from several classes, k = 2, 3, 4, 6, 7.
//   hipcc -O3 --offload-arch=gfx950 tools/class_probe.hip -o tools/class_probe.bin && tools/class_probe.bin [NCH]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static const int nz = 60, nb = 300;
static const size_t COLB = (size_t)nz * nb * 8;
static const size_t GB = 1ull << 30;
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
template <typename F> float timeit(F f, int rep = 4) {
  static hipEvent_t a = nullptr, b = nullptr;
  if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
char* base_va;
double rate(const std::vector<int>& chunks, int ncol) {
  Ptrs P = {};
  const int na = (int)chunks.size();
  for (int a = 0; a < na; ++a) P.o[a] = (double*)(base_va + (size_t)chunks[a] * GB);
  float t = timeit([&] { hipLaunchKernelGGL(flatn, dim3(ncol), dim3(512), 0, 0, P, na, nb, nz, 8); });
  return (double)na * ncol * COLB / t / 1e6;
}
void show(const char* what, const std::vector<int>& c, const std::vector<int>& cls, int ncol = 6000) {
  printf("%-44s [", what);
  for (int x : c) printf("%d%c ", x, "XYZ?"[cls[x]]);
  printf("]  %5.0f GB/s\n", rate(c, ncol));
  fflush(stdout);
}
int main(int argc, char** argv) {
  const int NCH = argc > 1 ? atoi(argv[1]) : 64;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  base_va = (char*)va;
  const int nc = 6000;
  // classify: reference 0 = chunk 0; reference 1 = first chunk fast against 0; reference 2 = first chunk fast against both
  std::vector<int> cls(NCH, 3), ref = {0};
  cls[0] = 0;
  for (int j = 1; j < NCH; ++j) {
    int c = -1;
    for (size_t r = 0; r < ref.size(); ++r)
      if (rate({ref[r], j}, nc) < 6200) { c = (int)r; break; }
    if (c < 0 && ref.size() < 3) { c = (int)ref.size(); ref.push_back(j); }
    cls[j] = c < 0 ? 3 : c;
  }
  printf("classes: ");
  for (int j = 0; j < NCH; ++j) printf("%c", "XYZ?"[cls[j]]);
  printf("\n");
  std::vector<int> byc[4];
  for (int j = 0; j < NCH; ++j) byc[cls[j]].push_back(j);
  for (int c = 0; c < 3; ++c) {
    auto& v = byc[c];
    printf("--- class %c: %zu chunks\n", "XYZ"[c], v.size());
    if (v.size() < 2) continue;
    const size_t n = v.size();
    show("pair, first two", {v[0], v[1]}, cls);
    show("pair, first and last", {v[0], v[n - 1]}, cls);
    if (n >= 3) show("triple", {v[0], v[1], v[2]}, cls);
    if (n >= 4) { show("quad, first four", {v[0], v[1], v[2], v[3]}, cls); show("quad, spread", {v[0], v[n / 3], v[2 * n / 3], v[n - 1]}, cls); }
    if (n >= 6) show("six", {v[0], v[1], v[2], v[3], v[4], v[5]}, cls);
    if (n >= 7) show("seven", {v[0], v[1], v[2], v[3], v[4], v[5], v[6]}, cls);
    if (n >= 8) show("eight", {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}, cls);
  }
  printf("--- mixed\n");
  if (byc[0].size() >= 2 && byc[1].size() >= 2) {
    show("pair X Y", {byc[0][0], byc[1][0]}, cls);
    show("quad X X Y Y", {byc[0][0], byc[0][1], byc[1][0], byc[1][1]}, cls);
    if (byc[0].size() >= 3) show("quad X X X Y", {byc[0][0], byc[0][1], byc[0][2], byc[1][0]}, cls);
    if (byc[1].size() >= 6) show("seven X Y Y Y Y Y Y", {byc[0][0], byc[1][0], byc[1][1], byc[1][2], byc[1][3], byc[1][4], byc[1][5]}, cls);
  }
  if (byc[0].size() >= 3 && byc[1].size() >= 2 && byc[2].size() >= 2) {
    show("triple X Y Z", {byc[0][0], byc[1][0], byc[2][0]}, cls);
    show("seven X Y Z X Y Z X", {byc[0][0], byc[1][0], byc[2][0], byc[0][1], byc[1][1], byc[2][1], byc[0][2]}, cls);
  }
  // time structure: arrays of 1e4 columns over two chunks each; all arrays change class at the same moment
  {
    auto two = [&](std::vector<int> idx, const char* what) {  // 2 chunks per array, separate VA
      const int na = (int)idx.size() / 2;
      Ptrs P = {};
      std::vector<void*> vas(na);
      for (int a = 0; a < na; ++a) {
        CK(hipMemAddressReserve(&vas[a], 2 * GB, 0, nullptr, 0));
        for (int q = 0; q < 2; ++q) { CK(hipMemUnmap(base_va + (size_t)idx[2 * a + q] * GB, GB)); CK(hipMemMap((char*)vas[a] + q * GB, GB, 0, h[idx[2 * a + q]], 0)); }
        CK(hipMemSetAccess(vas[a], 2 * GB, &acc, 1));
        P.o[a] = (double*)vas[a];
      }
      float t = timeit([&] { hipLaunchKernelGGL(flatn, dim3(10000), dim3(512), 0, 0, P, na, nb, nz, 8); });
      printf("%-44s [", what);
      for (int x : idx) printf("%d%c ", x, "XYZ?"[cls[x]]);
      printf("]  %5.0f GB/s\n", (double)na * 10000 * COLB / t / 1e6);
      for (int a = 0; a < na; ++a) {
        CK(hipMemUnmap(vas[a], 2 * GB)); CK(hipMemAddressFree(vas[a], 2 * GB));
        for (int q = 0; q < 2; ++q) { CK(hipMemMap(base_va + (size_t)idx[2 * a + q] * GB, GB, 0, h[idx[2 * a + q]], 0)); CK(hipMemSetAccess(base_va + (size_t)idx[2 * a + q] * GB, GB, &acc, 1)); }
      }
    };
    for (int c = 0; c < 3; ++c)
      if (byc[c].size() >= 8) {
        auto& v = byc[c];
        two({v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}, "1e4-col quad, one class throughout");
      }
    if (byc[0].size() >= 4 && byc[1].size() >= 4) {
      auto &x = byc[0], &y = byc[1];
      two({x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3]}, "1e4-col quad, every array X then Y");
      two({x[0], y[0], y[1], x[1], x[2], y[2], y[3], x[3]}, "1e4-col quad, arrays XY / YX alternating");
    }
  }
  CK(hipMemUnmap(va, (size_t)NCH * GB));
  for (int i = 0; i < NCH; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)NCH * GB));
  return 0;
}
