// Is the pair class of tools/matrix_probe.hip a property of the PHYSICAL chunk or of the virtual mapping?
//  1. NCH chunks of 1 GB mapped in order into one reservation (one hipMemSetAccess): classify every chunk against references.
//  2. swap two handles between their VA slots and re-measure their pairs.
//  3. unmap everything; every handle into its own 1 GB reservation (own hipMemSetAccess): classify again.
//  4. unmap; one new reservation, handles mapped in REVERSE order: classify again.
//   hipcc -O3 --offload-arch=gfx950 tools/remap_probe.hip -o tools/remap_probe.bin && tools/remap_probe.bin [NCH]
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
double rate(const std::vector<char*>& p, int ncol = 6000) {
  Ptrs P = {};
  const int na = (int)p.size();
  for (int a = 0; a < na; ++a) P.o[a] = (double*)p[a];
  float t = timeit([&] { hipLaunchKernelGGL(flatn, dim3(ncol), dim3(512), 0, 0, P, na, nb, nz, 8); });
  return (double)na * ncol * COLB / t / 1e6;
}
// where[i] = current address of handle i
std::vector<int> classify(const std::vector<char*>& where, const char* title) {
  const int n = (int)where.size();
  std::vector<int> cls(n, 3), ref = {0};
  cls[0] = 0;
  for (int j = 1; j < n; ++j) {
    int c = -1;
    for (size_t r = 0; r < ref.size(); ++r)
      if (rate({where[ref[r]], where[j]}) < 6200) { c = (int)r; break; }
    if (c < 0 && ref.size() < 3) { c = (int)ref.size(); ref.push_back(j); }
    cls[j] = c < 0 ? 3 : c;
  }
  printf("%-52s ", title);
  for (int j = 0; j < n; ++j) printf("%c", "XYZ?"[cls[j]]);
  printf("\n");
  fflush(stdout);
  return cls;
}
int main(int argc, char** argv) {
  const int NCH = argc > 1 ? atoi(argv[1]) : 40;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  std::vector<char*> where(NCH);
  for (int i = 0; i < NCH; ++i) where[i] = (char*)va + (size_t)i * GB;
  printf("base VA %p\n", va);
  auto c1 = classify(where, "1. in order, one reservation:");
  auto c1b = classify(where, "1b. same again (repeatability):");
  // 2. swap: a = first chunk of class X (0), b = first of another class, c = second chunk of class X
  int a = 0, b = -1, c = -1;
  for (int j = 1; j < NCH; ++j) { if (b < 0 && c1[j] != 0 && c1[j] != 3) b = j; if (c < 0 && c1[j] == 0) c = j; }
  if (b > 0 && c > 0) {
    printf("2. a=%d (X)  b=%d (%c)  c=%d (X):  pair(a,b) %.0f  pair(a,c) %.0f  pair(b,c) %.0f\n", a, b, "XYZ?"[c1[b]], c, rate({where[a], where[b]}),
           rate({where[a], where[c]}), rate({where[b], where[c]}));
    CK(hipMemUnmap(where[b], GB)); CK(hipMemUnmap(where[c], GB));
    CK(hipMemMap(where[b], GB, 0, h[c], 0)); CK(hipMemMap(where[c], GB, 0, h[b], 0));
    CK(hipMemSetAccess(where[b], GB, &acc, 1)); CK(hipMemSetAccess(where[c], GB, &acc, 1));
    printf("   after swapping the handles of b and c between their VA slots: pair(a, slot b [= handle c]) %.0f   pair(a, slot c [= handle b]) %.0f\n",
           rate({where[a], where[b]}), rate({where[a], where[c]}));
    std::swap(where[b], where[c]);
    classify(where, "   classes by HANDLE after the swap:");
  }
  // 3. every handle in its own reservation
  for (int i = 0; i < NCH; ++i) CK(hipMemUnmap(where[i], GB));
  std::vector<void*> own(NCH);
  for (int i = 0; i < NCH; ++i) {
    CK(hipMemAddressReserve(&own[i], GB, 0, nullptr, 0));
    CK(hipMemMap(own[i], GB, 0, h[i], 0));
    CK(hipMemSetAccess(own[i], GB, &acc, 1));
    where[i] = (char*)own[i];
  }
  printf("3. own reservations: first VAs %p %p %p %p (mod 1 GB: %zu %zu)\n", own[0], own[1], own[2], own[3], (size_t)own[0] % GB, (size_t)own[1] % GB);
  classify(where, "3. every handle in its own reservation:");
  {  // the set experiment of chunk_probe: 4 arrays of 6000 columns, all class X / mixed
    std::vector<int> X, Y;
    for (int j = 0; j < NCH; ++j) (c1[j] == 0 ? X : Y).push_back(j);
    if (X.size() >= 4) printf("   quad of four class-X handles (by step 1): %.0f\n", rate({where[X[0]], where[X[1]], where[X[2]], where[X[3]]}));
    if (X.size() >= 2 && Y.size() >= 2) printf("   quad X X Y Y: %.0f\n", rate({where[X[0]], where[X[1]], where[Y[0]], where[Y[1]]}));
  }
  for (int i = 0; i < NCH; ++i) { CK(hipMemUnmap(own[i], GB)); CK(hipMemAddressFree(own[i], GB)); }
  // 4. reverse order in a new reservation
  void* va2; CK(hipMemAddressReserve(&va2, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemMap((char*)va2 + (size_t)(NCH - 1 - i) * GB, GB, 0, h[i], 0)); where[i] = (char*)va2 + (size_t)(NCH - 1 - i) * GB; }
  CK(hipMemSetAccess(va2, (size_t)NCH * GB, &acc, 1));
  classify(where, "4. reverse order, new reservation (by handle):");
  CK(hipMemUnmap(va2, (size_t)NCH * GB));
  for (int i = 0; i < NCH; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)NCH * GB));
  CK(hipMemAddressFree(va2, (size_t)NCH * GB));
  return 0;
}
