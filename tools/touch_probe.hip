// Measurement aid for tools/subrange_probe.py: read a byte range with plain 16-byte loads (nothing kept), as a separate kernel.
//   hipcc -O3 -shared -fPIC --offload-arch=gfx950 tools/touch_probe.hip -o variants/libtouch_probe.so
#include <hip/hip_runtime.h>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_touch(const d2* p, long long n, double* sink) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const d2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e-300) *sink = acc;  // (never true for the data in question; keeps the loads)
}
extern "C" int touch_range(const void* p, long long bytes, double* sink, void* stream) {
  hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const d2*>(p), bytes / 16, sink);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
