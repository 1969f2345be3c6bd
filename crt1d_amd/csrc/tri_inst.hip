// One (scheme, storage type) instantiation of the column-tile / pipeline kernels (tri_tile_impl.hpp).  Compiled four times by
// the Makefile (-DTRI_SCHEME=TriN79|TriZq -DTRI_TAG=n79|zq -DTRI_TIO=double|float -DTRI_TIOTAG=f64|f32) so that the
// instantiations build in parallel.
#include "tri_tile_impl.hpp"

#define TRI_CAT_(a, b, c, d) a##b##_##c##d
#define TRI_CAT(a, b, c, d) TRI_CAT_(a, b, c, d)

namespace crt {

int TRI_CAT(launch_tri_tile_, TRI_TAG, TRI_TIOTAG, )(const SolveArgs& a, hipStream_t s, bool& done) {
  return launch_scheme<TRI_SCHEME, TRI_TIO>(a, s, done);
}

int TRI_CAT(launch_tri_int_, TRI_TAG, TRI_TIOTAG, )(const SolveArgs& a, const IntArgs& ia, hipStream_t s) {
  return launch_int_scheme<TRI_SCHEME, TRI_TIO>(a, ia, s);
}

}  // namespace crt
