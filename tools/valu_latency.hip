// Micro-benchmark (round 3): how fast does ONE wave advance through a dependent fp64 chain on gfx950, and how does the SIMD's throughput
// scale with the number of such waves?  The tridiagonal / four-stream kernels that are neither HBM- nor VALU-bound (n79 at 107 bands,
// zq_pa, ragged 4s) are chains of this kind.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_latency.hip -o tools/valu_latency.bin && tools/valu_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc) {
  __shared__ double lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = 1e-12 * threadIdx.x;
  __syncthreads();
  double x = 1.0 + 1e-3 * threadIdx.x, y = 1.5 + 1e-3 * threadIdx.x, z = 0.7, w = 0.9;
  const double a = 1.0000001, b = 1e-9;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {  // 8 dependent fp64 FMAs
#pragma unroll
      for (int u = 0; u < 8; ++u) x = __builtin_fma(x, a, b);
    } else if constexpr (MODE == 1) {  // dependent mul, add pairs (8 instructions)
#pragma unroll
      for (int u = 0; u < 4; ++u) { x = x * a; x = x + b; }
    } else if constexpr (MODE == 2) {  // 2 independent chains of 8 (16 instructions)
#pragma unroll
      for (int u = 0; u < 8; ++u) { x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); }
    } else if constexpr (MODE == 3) {  // 4 independent chains of 8 (32 instructions)
#pragma unroll
      for (int u = 0; u < 8; ++u) { x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); z = __builtin_fma(z, a, b); w = __builtin_fma(w, a, b); }
    } else if constexpr (MODE == 4) {  // v_rcp_f64 chain: 4 dependent (rcp, fma) pairs
#pragma unroll
      for (int u = 0; u < 4; ++u) { x = __builtin_amdgcn_rcp(x); x = __builtin_fma(x, a, 1.0); }
    } else if constexpr (MODE == 5) {  // 7 dependent FMAs + one LDS broadcast read that the chain depends on
      const double v = lds[i & 63];
#pragma unroll
      for (int u = 0; u < 7; ++u) x = __builtin_fma(x, a, b);
      x += v;
    } else if constexpr (MODE == 6) {  // fp32 chain of 8 dependent FMAs
      float xf = (float)x;
#pragma unroll
      for (int u = 0; u < 8; ++u) xf = __builtin_fmaf(xf, 1.0000001f, 1e-9f);
      x = xf;
    } else if constexpr (MODE == 7) {  // 8 dependent FMAs + 8 scalar ALU instructions (loop bookkeeping of the sweeps)
      int s = i;
#pragma unroll
      for (int u = 0; u < 8; ++u) { x = __builtin_fma(x, a, b); s = __builtin_amdgcn_readfirstlane(s) * 3 + 1; }
      if (s == 0x7fffffff) x += 1.0;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
int run(const char* name, int instr_per_iter) {
  const int iters = 4096;
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, 256ull * 16 * 256 * 8));
  CHECK(hipMalloc(&cyc, 256ull * 16 * 4 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-44s", name);
  for (int wps : {1, 2, 3, 4, 6, 8}) {  // waves per SIMD: 256 CUs x wps workgroups of 4 waves
    const int grid = 256 * wps;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 64, cyc);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    const double ticks = s / h.size();  // s_memtime ticks per wave: shader clock cycles (8.0 per dependent v_fma_f64 for a wave alone on its SIMD)
    const double ns_per_instr_wave = ticks / ((double)iters * instr_per_iter);
    const double ginstr = (double)grid * 4 * iters * instr_per_iter / (ms * 1e-3) / 1e9;  // wave-instructions per second, whole chip
    printf(" | w/SIMD %d: %.1f cycles/instr/wave, %.0f Ginstr/s", wps, ns_per_instr_wave, ginstr);
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
  return 0;
}

int main() {
  printf("peak: 1024 SIMDs x 1 fp64 wave-instruction per 4 cycles at ~2.3 GHz = ~590 G wave-instr/s\n");
  run<0>("8 dependent v_fma_f64", 8);
  run<1>("dependent v_mul_f64 / v_add_f64", 8);
  run<2>("2 independent chains of v_fma_f64", 16);
  run<3>("4 independent chains of v_fma_f64", 32);
  run<4>("dependent (v_rcp_f64, v_fma_f64) pairs", 8);
  run<5>("7 dependent fma + 1 LDS broadcast read", 8);
  run<6>("8 dependent v_fma_f32", 8);
  run<7>("8 dependent fma + 8 SALU", 8);
  return 0;
}
