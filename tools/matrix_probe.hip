// Pairwise test of 1 GB physical chunks: is the fast/slow mode of the column-pattern stores decided by WHICH chunks are written
// together?  (tools/spread_probe.hip: the workgroup -> column order does not matter; a contiguous 5.76 GB behaves the same as one
// array or as four; so the set of chunks touched decides.)  For every pair (i, j) of NCH held chunks: rate of the 2-array pattern
// with one array in chunk i and one in chunk j.  Prints the matrix ('#' fast, '.' slow), a two-class colouring derived from row 0,
// its consistency, and then the rate of 4-array sets built (a) from one class only, (b) alternating classes.
//   hipcc -O3 --offload-arch=gfx950 tools/matrix_probe.hip -o tools/matrix_probe.bin && tools/matrix_probe.bin [NCH]
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
template <typename F> float timeit(F f, int rep = 4) {
  static hipEvent_t a = nullptr, b = nullptr;
  if (!a) { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < rep; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / rep;
}
template <int NA>
double rate(char* p0, char* p1, char* p2, char* p3, int ncol) {
  float t = timeit([&] { hipLaunchKernelGGL(flat<NA>, dim3(ncol), dim3(512), 0, 0, (double*)p0, (double*)p1, (double*)p2, (double*)p3, nb, nz, 8); });
  return (double)NA * ncol * COLB / t / 1e6;
}
int main(int argc, char** argv) {
  const int NCH = argc > 1 ? atoi(argv[1]) : 48;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  void* va; CK(hipMemAddressReserve(&va, (size_t)NCH * GB, 0, nullptr, 0));
  for (int i = 0; i < NCH; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemMap((char*)va + (size_t)i * GB, GB, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)NCH * GB, &acc, 1));
  char* p = (char*)va;
  const int nc = 7400;  // 1.0656e9 B < 1 GiB
  std::vector<std::vector<float>> M(NCH, std::vector<float>(NCH, 0.f));
  for (int i = 0; i < NCH; ++i)
    for (int j = i + 1; j < NCH; ++j) M[i][j] = M[j][i] = (float)rate<2>(p + i * GB, p + j * GB, p, p, nc);
  for (int i = 0; i < NCH; ++i) M[i][i] = (float)rate<1>(p + i * GB, p, p, p, nc);
  printf("pair matrix (# > 6500, + 6000..6500, . < 6000 GB/s), diagonal = chunk alone:\n    ");
  for (int j = 0; j < NCH; ++j) printf("%d", j % 10);
  printf("\n");
  for (int i = 0; i < NCH; ++i) {
    printf("%3d ", i);
    for (int j = 0; j < NCH; ++j) printf("%c", M[i][j] > 6500 ? '#' : M[i][j] > 6000 ? '+' : '.');
    printf("\n");
  }
  printf("row 0 rates:");
  for (int j = 0; j < NCH; ++j) printf(" %.0f", M[0][j]);
  printf("\n");
  // colour from row 0, check consistency
  std::vector<int> cls(NCH, 0);
  for (int j = 1; j < NCH; ++j) cls[j] = M[0][j] > 6250 ? 1 : 0;
  int ok = 0, bad = 0;
  for (int i = 0; i < NCH; ++i)
    for (int j = i + 1; j < NCH; ++j) ((M[i][j] > 6250) == (cls[i] != cls[j]) ? ok : bad)++;
  printf("two-class colouring from row 0: consistent pairs %d, inconsistent %d\nclass:", ok, bad);
  for (int j = 0; j < NCH; ++j) printf("%d", cls[j]);
  printf("\n");
  // quarters of a chunk: does the class change inside a chunk?  pair (quarter q of chunk x, quarter 0 of a chunk of the other class)
  {
    int other = -1;
    for (int j = 1; j < NCH; ++j) if (cls[j] != cls[0]) { other = j; break; }
    if (other > 0) {
      const int ncq = 1850;
      printf("quarters of chunks 0..7 against quarter 0 of chunk %d (other class than chunk 0) and of chunk 0's class mate:\n", other);
      for (int x = 0; x < 8 && x < NCH; ++x) {
        printf("  chunk %d (class %d):", x, cls[x]);
        for (int q = 0; q < 4; ++q) printf(" %5.0f", rate<2>(p + x * GB + q * (GB / 4), p + other * GB + (x == other ? GB / 2 : 0), p, p, ncq));
        printf("\n");
      }
    }
  }
  // sets of four 1e4-column arrays, each in its own 2 GB VA range backed by two chunks of a chosen class
  std::vector<int> A, B;
  for (int j = 0; j < NCH; ++j) (cls[j] ? B : A).push_back(j);
  printf("class sizes: %zu / %zu\n", A.size(), B.size());
  CK(hipMemUnmap(va, (size_t)NCH * GB));
  auto run_set = [&](const char* name, std::vector<int> idx) {  // idx: 8 chunk ids, two per array
    void* a4[4];
    for (int k = 0; k < 4; ++k) {
      CK(hipMemAddressReserve(&a4[k], 2 * GB, 0, nullptr, 0));
      for (int q = 0; q < 2; ++q) CK(hipMemMap((char*)a4[k] + q * GB, GB, 0, h[idx[2 * k + q]], 0));
      CK(hipMemSetAccess(a4[k], 2 * GB, &acc, 1));
    }
    printf("%s [", name);
    for (int x : idx) printf("%d ", x);
    printf("]: %5.0f GB/s\n", rate<4>((char*)a4[0], (char*)a4[1], (char*)a4[2], (char*)a4[3], 10000));
    for (int k = 0; k < 4; ++k) { CK(hipMemUnmap(a4[k], 2 * GB)); CK(hipMemAddressFree(a4[k], 2 * GB)); }
  };
  if (A.size() >= 8) run_set("all class 0", {A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7]});
  if (B.size() >= 8) run_set("all class 1", {B[0], B[1], B[2], B[3], B[4], B[5], B[6], B[7]});
  if (A.size() >= 4 && B.size() >= 4) {
    run_set("arrays alternate classes (0,0|1,1|0,0|1,1)", {A[0], A[1], B[0], B[1], A[2], A[3], B[2], B[3]});
    run_set("each array = class-0 chunk then class-1 chunk", {A[0], B[0], A[1], B[1], A[2], B[2], A[3], B[3]});
    run_set("three arrays class 0, one class 1", {A[0], A[1], A[2], A[3], A[4 % A.size()], A[5 % A.size()], B[0], B[1]});
  }
  for (int i = 0; i < NCH; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)NCH * GB));
  return 0;
}
