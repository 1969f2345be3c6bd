// Dispatch of the tridiagonal column-tile / pipeline kernels.  The kernels themselves are templates in tri_tile_impl.hpp,
// instantiated per (scheme, storage type) from tri_inst.hip and in tri_zqpa.hip (separate translation units: parallel compilation).
#include "crt_internal.hpp"

namespace crt {

#include <stdarg.h>
#include <stdio.h>

namespace {
thread_local char g_last_kernel[192] = "";
}

void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernel, sizeof g_last_kernel, fmt, ap);
  va_end(ap);
}
const char* last_kernel() { return g_last_kernel; }

int launch_tridiag_int(int scheme, const SolveArgs& a, const IntArgs& ia, hipStream_t s) {
  if (scheme == CRT_SCHEME_N79) return a.f32 ? launch_tri_int_n79_f32(a, ia, s) : launch_tri_int_n79_f64(a, ia, s);
  if (scheme == CRT_SCHEME_ZQ) return a.f32 ? launch_tri_int_zq_f32(a, ia, s) : launch_tri_int_zq_f64(a, ia, s);
  return CRT_ERR_BAD_ARG;
}

// returns CRT_OK with done = false when the column-tile kernel does not apply (caller falls back)
int launch_tridiag_tile(int scheme, const SolveArgs& a, hipStream_t s, bool& done) {
  if (scheme == CRT_SCHEME_N79) return a.f32 ? launch_tri_tile_n79_f32(a, s, done) : launch_tri_tile_n79_f64(a, s, done);
  if (scheme == CRT_SCHEME_ZQ) return a.f32 ? launch_tri_tile_zq_f32(a, s, done) : launch_tri_tile_zq_f64(a, s, done);
  done = false;
  return CRT_ERR_BAD_ARG;
}

}  // namespace crt
