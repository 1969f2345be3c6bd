#include "tri_tile_impl.hpp"

// ------------------------------------------------------------------------------------------
// zq_pa (crt1d/solvers/_solve_zq_pa.py:24-418): the zq system on a computational grid of M = min(100, nz) equal
// layers, then linear interpolation of the interface fluxes back to the caller's levels.
//   step 1: the zq column-tile kernel above runs with nz := M on the K0 record of scheme zq_pa (whose first vector is
//           the beam fraction per computational layer and whose tau_i / tau_psi are those of LAI/M) and writes only
//           I_df_d[z] = SWd[z+1], I_df_u[z] = SWu[z], z = 0..M-1, into workspace scratch;
//   step 2: k_zqpa_interp forms SWd[0] := SWd[1], SWu[M] := SWu[M-1] (:310,335), interpolates and writes the outputs.
namespace crt {
namespace {

struct InterpArgs {
  int ncol, nb, nz, M, reclen;
  long long col_stride;
  const double* ws;
  const double* dnz;  // [ncol][M][nb]  SWd[z+1]
  const double* upz;  // [ncol][M][nb]  SWu[z]
  const void* I_dr0;
  void* o[4];
};

template <typename TIO>
__global__ __launch_bounds__(256) void k_zqpa_interp(InterpArgs a) {
  extern __shared__ double lds[];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int nb = a.nb, nz = a.nz, M = a.M;
  const double* src = a.ws + (long long)c * a.reclen;
  for (int i = tid; i < a.reclen; i += 256) lds[i] = src[i];
  __syncthreads();
  const double* rec = lds;
  const double invmu = rec[S_INVMU];
  const double* ekl = rec + REC_HDR + nz;
  const double* kidx = ekl + nz;
  const double* wgt = kidx + nz;
  const long long sb = (long long)c * M * nb;
  // flat sweep over the column's nz * nb outputs: consecutive threads -> consecutive addresses
  for (int i = tid; i < nz * nb; i += 256) {
    const int j = i / nb, b = i - j * nb;
    const int ka = (int)kidx[j], kb = ka - 1;  // interfaces above / below lai[j] (0 = ground)
    const double w = wgt[j];
    // SWd[k] = dnz[max(k,1)-1],  SWu[k] = upz[min(k, M-1)]
    const double da = a.dnz[sb + (long long)(max(ka, 1) - 1) * nb + b], db = a.dnz[sb + (long long)(max(kb, 1) - 1) * nb + b];
    const double ua = a.upz[sb + (long long)min(ka, M - 1) * nb + b], ub = a.upz[sb + (long long)min(kb, M - 1) * nb + b];
    const double dn = da + (db - da) * w;  // :360
    const double up = ua + (ub - ua) * w;  // :361
    const double idr = ldio<TIO>(a.I_dr0, (long long)c * a.col_stride + b) * ekl[j];  // :354-355
    const long long o = (long long)c * nz * nb + i;
    __builtin_nontemporal_store((TIO)idr, outp<TIO>(a.o[0]) + o);
    __builtin_nontemporal_store((TIO)dn, outp<TIO>(a.o[1]) + o);
    __builtin_nontemporal_store((TIO)up, outp<TIO>(a.o[2]) + o);
    __builtin_nontemporal_store((TIO)(idr * invmu + 2 * up + 2 * dn), outp<TIO>(a.o[3]) + o);  // :412
  }
}

// ------------------------------------------------------------------------------------------
// One kernel, no scratch: the pipeline's compute waves solve the computational grid (tile by tile, from the top), its store
// waves interpolate to the caller's levels and write the four profiles.  Both grids are ordered in LAI, so each round can
// emit a contiguous range of output levels: level j needs the computational rows max(ka-2, 0) .. min(ka, Mg-1), ka = kidx[j]
// (k_zqpa_interp's index arithmetic), i.e. it is complete once the tile holding row max(ka-2, 0) has arrived; the one or two
// rows above the tile that it may also need are the two lowest rows of the previous tile, kept in a small halo buffer.
// Same expressions as k_zqpa_interp -> bitwise the same profiles, 5.8 GB less HBM traffic of 11.5 GB at 1e4 x 300 x 60.
template <typename TIO, int M, int T>
__device__ __forceinline__ void zqpa_pipe_store(const SolveArgs& a, const PipeCfg& cfg, double* lds) {
  typedef TIO vt __attribute__((ext_vector_type(2)));
  static_assert(T >= 2, "the halo holds the two lowest rows of a tile");
  const int nb2 = a.nb >> 1, Mg = a.nz, nzo = cfg.nz_out;
  const int c = blockIdx.x;
  const int sid = threadIdx.x - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
  const double* rec = lds;
  const double invmu = rec[S_INVMU];
  const double* ekl = rec + REC_HDR + nzo;
  const double* kidx = ekl + nzo;
  const double* wgt = kidx + nzo;
  const d2* bandc2 = reinterpret_cast<const d2*>(lds + cfg.off_bc);  // I_dr0 of the band pair
  const d2* tile2 = reinterpret_cast<const d2*>(lds + cfg.off_tile);  // [2 buffers][2 arrays: dn, up][T][nb2]
  d2* halo2 = reinterpret_cast<d2*>(lds + cfg.off_halo);              // [2 parities][dn(ktop), up(ktop), up(ktop+1)][nb2]
  const int K = Mg + 1;
  int buf = 0, g = 0, jhi = nzo;
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    for (int i = M - T; i >= 0; i -= T) {
      const int k = k0 + i;
      if (k > kend) continue;
      lds_barrier();  // tile `buf` is complete: computational levels k .. ktop-1
      const int ktop = min(k + T, kend + 1);
      int jlo = 0;  // first output level whose lowest row, max(kidx - 2, 0), lies in or above this tile
      if (k > 0) {
        int lo = 0, hi = jhi;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((int)kidx[mid] >= k + 2)
            hi = mid;
          else
            lo = mid + 1;
        }
        jlo = lo;
      }
      const d2* cur = tile2 + (size_t)buf * (2 * T * nb2);
      // written by the previous round: dn of level ktop, up of levels ktop and ktop+1 (dn is never needed above ktop: its
      // rows are ka-1 and ka-2, and ka <= ktop+1 for every level emitted here)
      const d2* hal = halo2 + (size_t)((g + 1) & 1) * (3 * nb2);
      auto row = [&](int q, int r, int p) -> d2 {  // array q (0 = dn, 1 = up), computational level r, band pair p
        return r < ktop ? cur[(q * T + (r - k)) * nb2 + p] : hal[(q + (r - ktop)) * nb2 + p];
      };
      const int n = (jhi - jlo) * nb2;  // band pairs to emit: output levels jlo .. jhi-1 are one contiguous run per array
      vt* o0 = reinterpret_cast<vt*>(cfg.out[0]) + ((long long)c * nzo + jlo) * nb2;
      vt* o1 = reinterpret_cast<vt*>(cfg.out[1]) + ((long long)c * nzo + jlo) * nb2;
      vt* o2 = reinterpret_cast<vt*>(cfg.out[2]) + ((long long)c * nzo + jlo) * nb2;
      vt* o3 = reinterpret_cast<vt*>(cfg.out[3]) + ((long long)c * nzo + jlo) * nb2;
      auto cvt = [](d2 x) -> vt { vt r; r.x = (TIO)x.x; r.y = (TIO)x.y; return r; };
      {
        const int dt = nst / nb2, dp = nst - dt * nb2;
        int t = sid / nb2, p = sid - (sid / nb2) * nb2;
        for (int idx = sid; idx < n; idx += nst) {
          const int j = jlo + t;
          const int ka = (int)kidx[j];
          const double w = wgt[j];
          const d2 da = row(0, ka - 1, p), db = row(0, max(ka - 2, 0), p);            // SWd[ka], SWd[ka-1]  (:310 clamp)
          const d2 ua = row(1, min(ka, Mg - 1), p), ub = row(1, ka - 1, p);           // SWu[ka], SWu[ka-1]  (:335 clamp)
          const d2 dn = da + (db - da) * w;  // :360
          const d2 up = ua + (ub - ua) * w;  // :361
          const d2 idr = bandc2[p] * ekl[j];  // :354-355
          __builtin_nontemporal_store(cvt(idr), o0 + idx);
          __builtin_nontemporal_store(cvt(dn), o1 + idx);
          __builtin_nontemporal_store(cvt(up), o2 + idx);
          __builtin_nontemporal_store(cvt(idr * invmu + 2 * up + 2 * dn), o3 + idx);  // :412
          p += dp;
          t += dt;
          if (p >= nb2) {
            p -= nb2;
            ++t;
          }
        }
      }
      // halo for the next round: the two lowest levels of this tile
      if (ktop - k >= 2) {
        d2* hw = halo2 + (size_t)(g & 1) * (3 * nb2);
        for (int idx = sid; idx < 3 * nb2; idx += nst)  // dn row 0 | up rows 0, 1 (adjacent in the tile)
          hw[idx] = idx < nb2 ? cur[idx] : cur[(T - 1) * nb2 + idx];
      }
      jhi = jlo;
      buf ^= 1;
      ++g;
    }
  }
}

// Odd nb: the same store role as a flat walk over the run's (jhi - jlo) * nb elements (rows are not pair-aligned), element pairs
// on the pair grid of the arrays (as flush_flat_class does for the other schemes), the unpaired ends by one thread.  Scalar
// forms of the same expressions -> bitwise the same profiles.
template <typename TIO, int M, int T>
__device__ __forceinline__ void zqpa_pipe_store_flat(const SolveArgs& a, const PipeCfg& cfg, double* lds) {
  typedef TIO vt __attribute__((ext_vector_type(2)));
  static_assert(T >= 2, "the halo holds the two lowest rows of a tile");
  const int nb = a.nb, Mg = a.nz, nzo = cfg.nz_out;
  const int c = blockIdx.x;
  const int sid = threadIdx.x - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
  const double* rec = lds;
  const double invmu = rec[S_INVMU];
  const double* ekl = rec + REC_HDR + nzo;
  const double* kidx = ekl + nzo;
  const double* wgt = kidx + nzo;
  const double* bandc = lds + cfg.off_bc;
  const double* tile = lds + cfg.off_tile;  // [2 buffers][dn, up][T][nb]
  double* halo = lds + cfg.off_halo;        // [2 parities][dn(ktop), up(ktop), up(ktop+1)][nb]
  TIO* const o0 = outp<TIO>(cfg.out[0]);
  TIO* const o1 = outp<TIO>(cfg.out[1]);
  TIO* const o2 = outp<TIO>(cfg.out[2]);
  TIO* const o3 = outp<TIO>(cfg.out[3]);
  const int K = Mg + 1;
  const int step = 2 * nst, dt = step / nb, db = step - dt * nb;
  int buf = 0, g = 0, jhi = nzo;
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    for (int i = M - T; i >= 0; i -= T) {
      const int k = k0 + i;
      if (k > kend) continue;
      lds_barrier();
      const int ktop = min(k + T, kend + 1);
      int jlo = 0;
      if (k > 0) {
        int lo = 0, hi = jhi;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((int)kidx[mid] >= k + 2)
            hi = mid;
          else
            lo = mid + 1;
        }
        jlo = lo;
      }
      const double* cur = tile + (size_t)buf * (2 * T * nb);
      const double* hal = halo + (size_t)((g + 1) & 1) * (3 * nb);
      auto row = [&](int q, int r, int b) -> double {
        return r < ktop ? cur[(q * T + (r - k)) * nb + b] : hal[(q + (r - ktop)) * nb + b];
      };
      auto elem = [&](int j, int b, double& idr, double& dn, double& up) {
        const int ka = (int)kidx[j];
        const double w = wgt[j];
        const double da = row(0, ka - 1, b), db_ = row(0, max(ka - 2, 0), b);
        const double ua = row(1, min(ka, Mg - 1), b), ub = row(1, ka - 1, b);
        dn = da + (db_ - da) * w;  // :360
        up = ua + (ub - ua) * w;   // :361
        idr = bandc[b] * ekl[j];   // :354-355
      };
      const int n = (jhi - jlo) * nb;
      const long long g0 = ((long long)c * nzo + jlo) * nb;
      if (n > 0) {
        const int mis = (int)(g0 & 1);
        const int npair = (n - mis) >> 1;
        int e = mis + 2 * sid;
        int t = (int)(((float)e + 0.5f) * (1.0f / (float)nb));
        int b = e - t * nb;
        if (b < 0) { --t; b += nb; }
        if (b >= nb) { ++t; b -= nb; }
        for (int idx = sid; idx < npair; idx += nst) {
          const bool wrap = b + 1 >= nb;
          const int t2 = wrap ? t + 1 : t, b2 = wrap ? 0 : b + 1;
          double ix, dx, ux, iy, dy, uy;
          elem(jlo + t, b, ix, dx, ux);
          elem(jlo + t2, b2, iy, dy, uy);
          vt v;
          v.x = (TIO)ix, v.y = (TIO)iy;
          *reinterpret_cast<vt*>(o0 + g0 + e) = v;
          v.x = (TIO)dx, v.y = (TIO)dy;
          *reinterpret_cast<vt*>(o1 + g0 + e) = v;
          v.x = (TIO)ux, v.y = (TIO)uy;
          *reinterpret_cast<vt*>(o2 + g0 + e) = v;
          v.x = (TIO)(ix * invmu + 2 * ux + 2 * dx), v.y = (TIO)(iy * invmu + 2 * uy + 2 * dy);  // :412
          *reinterpret_cast<vt*>(o3 + g0 + e) = v;
          e += step;
          b += db;
          t += dt;
          if (b >= nb) {
            b -= nb;
            ++t;
          }
        }
        if (sid == 0) {
          auto single = [&](int el, int j, int b) {
            double ix, dx, ux;
            elem(j, b, ix, dx, ux);
            o0[g0 + el] = (TIO)ix;
            o1[g0 + el] = (TIO)dx;
            o2[g0 + el] = (TIO)ux;
            o3[g0 + el] = (TIO)(ix * invmu + 2 * ux + 2 * dx);
          };
          if (mis) single(0, jlo, 0);
          if ((n - mis) & 1) single(n - 1, jhi - 1, nb - 1);
        }
      }
      if (ktop - k >= 2) {
        double* hw = halo + (size_t)(g & 1) * (3 * nb);
        for (int idx = sid; idx < 3 * nb; idx += nst) hw[idx] = idx < nb ? cur[idx] : cur[(T - 1) * nb + idx];
      }
      jhi = jlo;
      buf ^= 1;
      ++g;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Round 3: the interpolation moves INTO the compute lanes.  Timeline of the kernel above (tools/stamp_timeline_tri.py, 1e4 x 300 x 60):
// of a column's 71 us the compute waves spend 11 in the forward sweep, 7 recomputing pairs, 15 substituting back -- and 35 waiting at the
// hand-over barriers for the store waves, whose per-element work (index search, halo selection, four row reads, two interpolations) is
// what bounds the kernel.  But the interpolation runs along the LEVEL axis of one band, i.e. inside a lane of the compute role: the lane
// keeps the two fluxes of the last three computational rows in registers (output level j needs rows max(ka - 2, 0) .. ka, ka = kidx[j])
// and, walking down, emits every output level as soon as its lowest row exists.  The LDS tile then holds FINISHED output levels (I_df_d,
// I_df_u), T of them, and the store role is the plain fused flush of the other tridiagonal schemes (I_dr and F formed while flushing):
// ZqPaOut below is that role's view of the outputs.  Same expressions as k_zqpa_interp -> the same bits as the two-kernel path.
struct ZqPaOut {
  static constexpr int NST = 2;   // staged: I_df_d, I_df_u
  static constexpr int NOUT = 4;  // I_dr, I_df_d, I_df_u, F
  __host__ __device__ static inline int rows(int nz) { return nz; }
  __host__ __device__ static constexpr int out_rows(int, int nz) { return nz; }
  // (record of scheme zq_pa: vector 1 = e^{-K_b lai_j} on the caller's levels)
  __device__ static inline void emit(const double* rec, int nz, int j, d2 bc, double invmu_, const d2 (&st)[NST], d2 (&o)[NOUT]) {
    const d2 idr = bc * rec[REC_HDR + nz + j];  // :354-355
    o[0] = idr;
    o[1] = st[0];
    o[2] = st[1];
    o[3] = idr * invmu_ + 2 * st[1] + 2 * st[0];  // :412
  }
  // the same outputs element by element, for the flat flush of odd band counts (same expressions as emit -> same bits)
  template <int ARR>
  __device__ static inline double value(const double* rec, int nz, int j, double bc, double invmu_, const double* tile, int stride, int idx) {
    if constexpr (ARR == 0) return bc * rec[REC_HDR + nz + j];
    if constexpr (ARR == 1) return tile[idx];
    if constexpr (ARR == 2) return tile[stride + idx];
    return bc * rec[REC_HDR + nz + j] * invmu_ + 2 * tile[stride + idx] + 2 * tile[idx];
  }
  static constexpr bool derived(int arr) { return arr == 0 || arr == 3; }
  static constexpr int staged_slot(int arr) { return arr == 1 ? 0 : 1; }
  static constexpr int park_slots(int, int hi) { return hi ? NST : 0; }
};

// compute role: the zq system on the computational grid (g.nz = Mg rows + 1), checkpointed like tri_pipe_compute, with the emission of
// interpolated output levels behind every back-substitution step.  RS > 0: single tile buffer (two barriers per hand-over).
template <typename TIO, int M, int T, int RS>
__device__ __forceinline__ void zqpa_compute(const SolveArgs& g, const PipeCfg& cfg, double* lds, const TriBand& band) {
  typedef TriZq S;
  const int nb = g.nb, Mg = g.nz, nzo = cfg.nz_out;
  const int tid = threadIdx.x, nthr = cfg.ncomp;
  const double* rec = lds;
  double* bandc = lds + cfg.off_bc;
  double* ck = lds + cfg.off_ck + tid;
  double* tile = lds + cfg.off_tile;  // [RS > 0 ? 1 : 2][2][T][nb]
  const int tstride = T * nb, bstride = 2 * tstride;
  const double* kidx = rec + REC_HDR + 2 * nzo;
  const double* wgt = kidx + nzo;
  const bool active = tid < nb;
  const int b = active ? tid : 0;
#ifdef CRT_STAMP
  double* stamp = const_cast<double*>(g.ws) + (long long)blockIdx.x * g.reclen;
  int nstamp = 0;
#define ZSTAMP() do { if (tid == 0 && nstamp < g.reclen) stamp[nstamp++] = (double)wall_clock64(); } while (0)
#else
#define ZSTAMP() do {} while (0)
#endif
  ZSTAMP();
  S st;
  st.init_band(rec, g, band);
  if (active) bandc[b] = st.band_const();
  const int K = Mg + 1;
  const int seg_top = (K - 1) / M, k_top = seg_top * M;
  double be[M], bf[M];
  typename S::St fs;
  st.first(rec, Mg, fs);
  ZSTAMP();
  tri_forward<S, M>(st, rec, Mg, fs, k_top, [&](int level, const typename S::St& cs) {
    if (level < k_top) {
      const int sidx = level / M - 1;
      double e, f;
      st.pair(cs, e, f);
      ck[(2 * sidx) * nthr] = e;
      ck[(2 * sidx + 1) * nthr] = f;
    }
  });
  st.pair(fs, be[0], bf[0]);
  if constexpr (M % S::RENORM != 0) st.seed(fs, be[0], bf[0]);
#pragma unroll
  for (int i = 1; i < M; ++i) {
    be[i] = be[i - 1];
    bf[i] = bf[i - 1];
    if (k_top + i <= K - 1) tri_step(st, k_top + i - 1, rec, Mg, fs, be[i], bf[i]);
  }
  ZSTAMP();
  // Window of the last three computational rows (k, k + 1, k + 2): d0 = SWd[k + 1], d1 = SWd[k + 2]; u0 = SWu[k], u1 = SWu[k + 1], u2 = SWu[k + 2].
  // Output level j with interface index ka (K0: the node above lai[j], 1 <= ka <= Mg) interpolates (:357-362, with the clamps of :310 / :335)
  //   dn = SWd[ka] + (SWd[max(ka - 1, 1)] - SWd[ka]) w,     up = SWu[min(ka, Mg - 1)] + (SWu[ka - 1] - SWu[min(ka, Mg - 1)]) w
  // and leaves when row k = ka - 2 is done: then SWd[ka] = d1, SWd[ka - 1] = d0, SWu[ka] = u2, SWu[ka - 1] = u1.  The two clamps are built into
  // the window instead of being tested per level (round 3: the general per-level selection was 60 % of this role's instructions, most of
  // them scalar): the first row pushed (k = Mg - 1) also fills u1, so that two steps later u2 reads SWu[Mg - 1] where SWu[Mg] is asked for;
  // and after row 0 one more shift repeats d0 (SWd[1] where SWd[0] is asked for), at which "row -1" the levels with ka = 1 leave.
  double d0 = 0.0, d1 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0;
  int jn = nzo - 1;        // next output level to emit (they leave from the top)
  int slot = jn % T;       // its row of the output tile
  // interface index and weight of level jn, requested as soon as jn is known: the LDS round trip + conversion + readfirstlane overlap with
  // the next back-substitution step instead of heading every step's "does a level leave here?" test
  int ka = __builtin_amdgcn_readfirstlane((int)kidx[jn]);
  double w = wgt[jn];
  int buf = 0;
  // levels that leave behind row k (wave-uniform: the level grid belongs to the column)
  auto emit_levels = [&](int k) {
    while (jn >= 0 && ka - 2 >= k) {  // (== k for any valid level grid; ">" cannot wait for a row that is already gone)
      const double dn = d1 + (d0 - d1) * w;  // :360  (same operations as the store-role interpolation of k_zqpa_pipe: same bits)
      const double up = u2 + (u1 - u2) * w;  // :361
      if (active) {
        double* tl = tile + buf * bstride + slot * nb + b;
        tl[0] = dn;
        tl[tstride] = up;
      }
      if (slot == 0) {  // output tile complete: hand it to the store waves
        ZSTAMP();
        lds_barrier();
        if constexpr (RS > 0)
          lds_barrier();
        else
          buf ^= 1;
        ZSTAMP();
        slot = T;
      }
      --slot;
      --jn;
      if (jn >= 0) {
        ka = __builtin_amdgcn_readfirstlane((int)kidx[jn]);
        w = wgt[jn];
      }
    }
  };
  for (int seg = seg_top; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    ZSTAMP();
    if (seg != seg_top) {
      typename S::St rs;
      if (seg == 0) {
        st.first(rec, Mg, rs);
        st.pair(rs, be[0], bf[0]);
      } else {
        be[0] = ck[(2 * (seg - 1)) * nthr];
        bf[0] = ck[(2 * (seg - 1) + 1) * nthr];
        st.seed(rs, be[0], bf[0]);
      }
#pragma unroll
      for (int i = 1; i < M; ++i) {
        be[i] = be[i - 1];
        bf[i] = bf[i - 1];
        if (k0 + i <= kend) tri_step(st, k0 + i - 1, rec, Mg, rs, be[i], bf[i]);
      }
    }
    ZSTAMP();
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const int k = k0 + i;
      if (k <= kend) {
        double o[S::NST];
        if (k == K - 1) {
          st.top(rec, Mg, be[i], bf[i], o);  // row Mg: boundary only, no output row
        } else {
          st.back(k, rec, Mg, be[i], bf[i], o);
          d1 = d0;
          d0 = o[0];
          u2 = u1;
          u1 = u0;
          u0 = o[1];
          if (k == K - 2) u1 = u0;  // (the clamp at the top, see above)
          emit_levels(k);
        }
      }
    }
  }
  d1 = d0;  // "row -1": SWd[1] once more (the clamp at the ground); u0 is not read by the interpolation
  u2 = u1;
  u1 = u0;
  emit_levels(-1);
  // every hand-over the store waves count on has happened once jn < 0; a level grid that left levels behind (NaN in lai) still gets
  // its barriers, so that no wave of the workgroup is left waiting
  while (jn >= 0) {
    if (jn % T == 0) {
      lds_barrier();
      if constexpr (RS > 0) lds_barrier();
    }
    --jn;
  }
}

template <typename TIO, int M, int T, int MAXT, int RS>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(M <= 8 ? 5 : 4))) void k_zqpa_pipe2(SolveArgs a, PipeCfg cfg) {
  extern __shared__ double lds[];
  TriBand band = {};
  if ((int)threadIdx.x < cfg.ncomp) band = load_tri_band<TIO>(a, blockIdx.x, (int)threadIdx.x < a.nb ? threadIdx.x : 0);
  {
    const double* src = a.ws + (long long)blockIdx.x * a.reclen;
    for (int i = threadIdx.x; i < a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  if ((int)threadIdx.x >= cfg.ncomp) {
    // the store role of the other tridiagonal schemes on the caller's levels and output arrays
    SolveArgs ao = a;
    ao.nz = cfg.nz_out;
    for (int i = 0; i < 4; ++i) ao.o[i] = cfg.out[i];
    if constexpr (RS > 0)
      tri_pipe_store_rs<ZqPaOut, TIO, M, T, RS>(ao, cfg, lds);
    else if constexpr (RS < 0)
      tri_pipe_store_generic<ZqPaOut, TIO, M, T>(ao, cfg, lds);  // odd nb: the flat flush of the other tridiagonal schemes
    else
      tri_pipe_store<ZqPaOut, TIO, M, T>(ao, cfg, lds);
    return;
  }
  zqpa_compute<TIO, M, T, RS>(a, cfg, lds, band);
}

// odd band counts (the reference's 107): the same compute role, double-buffered tile, flat flush in the store waves (whole lines when the
// output arrays are line-aligned)
template <typename TIO, int M, int T>
int launch_zqpa_generic2(const SolveArgs& a, hipStream_t s, int nsw) {
  if (a.nb < 65 || a.nb > 256) return CRT_ERR_UNSUPPORTED;
  SolveArgs ao = a;  // (flat_flush_ok looks at the first ZqPaOut::NOUT output arrays)
  const int flat = a.tune[13] != 1 ? flat_flush_ok<ZqPaOut, TIO>(ao) : 0;
  if (!flat) return CRT_ERR_UNSUPPORTED;
  const int Mg = zqpa_M(a.nz);
  const int ncomp = ((a.nb + 63) / 64) * 64;
  if (nsw <= 0) nsw = ncomp <= 128 ? 1 : 2;  // (3e4 x 107 x 60: 1 / 2 / 3 store waves 1.27 / 1.33 / 2.29 ms; the older kernel 1.48)
  const int nthr = ncomp + 64 * nsw;
  if (nthr > 512) return CRT_ERR_UNSUPPORTED;
  SolveArgs g = a;
  g.nz = Mg;
  for (int i = 0; i < 7; ++i) g.o[i] = nullptr;
  PipeCfg cfg{};
  cfg.ncomp = ncomp;
  cfg.nck = std::max(Mg / M - 1, 0);
  cfg.off_bc = (a.reclen + 1) & ~1;
  cfg.off_ck = cfg.off_bc + ((a.nb + 1) & ~1);
  cfg.off_tile = cfg.off_ck + 2 * cfg.nck * ncomp;
  cfg.off_park = cfg.off_tile + 2 * ZqPaOut::NST * T * a.nb;
  cfg.flat = (flat == 2 && a.nb >= 128 / (int)sizeof(TIO) && a.tune[13] != 2) ? 3 : flat;
  cfg.nz_out = a.nz;
  for (int i = 0; i < 4; ++i) cfg.out[i] = a.o[i];
  const size_t sh = ((size_t)cfg.off_park + (cfg.flat == 3 ? park_doubles<ZqPaOut, TIO>() : 0)) * sizeof(double);
  if (sh > MAX_WG_LDS / 2) return CRT_ERR_UNSUPPORTED;
  auto kern = k_zqpa_pipe2<TIO, M, T, 512, -1>;
  if (sh > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
    return (int)CRT_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, g, cfg);
  if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
  note_kernel("k_zqpa_pipe2<%s> %s M=%d T=%d store_waves=%d lds=%zu", sizeof(TIO) == 8 ? "f64" : "f32", cfg.flat == 3 ? "whole-line flat-flush" : "flat-flush", M, T,
              nsw, sh);
  return (int)CRT_OK;
}

// returns CRT_ERR_UNSUPPORTED when the shape does not fit
template <typename TIO, int M, int T, bool REGSTAGE>
int launch_zqpa_fused2(const SolveArgs& a, hipStream_t s, int nsw, size_t lds_cap = MAX_WG_LDS) {
  if (a.nb < (a.tune[12] > 0 ? a.tune[12] : 16) || a.nb % 2) return CRT_ERR_UNSUPPORTED;  // even nb: the fused (row, band pair) flush
  for (int i = 0; i < 4; ++i)
    if (reinterpret_cast<uintptr_t>(a.o[i]) & (2 * sizeof(TIO) - 1)) return CRT_ERR_UNSUPPORTED;
  const int Mg = zqpa_M(a.nz);
  const int ncomp = ((a.nb + 63) / 64) * 64;
  if (nsw <= 0) nsw = ncomp <= 128 ? 1 : 3;  // (two compute waves: one store wave keeps up, 3e4 x 106 x 60 1.19 -> 1.13 ms)
  if (ncomp + 64 * nsw > 1024) nsw = (1024 - ncomp) / 64;
  if (nsw < 1) return CRT_ERR_UNSUPPORTED;
  if (REGSTAGE && T * (a.nb / 2) > PIPE_RS * 64 * nsw) return CRT_ERR_UNSUPPORTED;
  const int nthr = ncomp + 64 * nsw;
  SolveArgs g = a;  // computational-grid solve: nz := Mg; the outputs go through PipeCfg
  g.nz = Mg;
  for (int i = 0; i < 7; ++i) g.o[i] = nullptr;
  PipeCfg cfg{};
  cfg.ncomp = ncomp;
  cfg.nck = std::max(Mg / M - 1, 0);
  cfg.off_bc = (a.reclen + 1) & ~1;
  cfg.off_ck = cfg.off_bc + ((a.nb + 1) & ~1);
  cfg.off_tile = cfg.off_ck + 2 * cfg.nck * ncomp;
  cfg.nz_out = a.nz;
  for (int i = 0; i < 4; ++i) cfg.out[i] = a.o[i];
  const size_t sh = ((size_t)cfg.off_tile + (size_t)(REGSTAGE ? 1 : 2) * 2 * T * a.nb) * sizeof(double);
  if (sh > lds_cap) return CRT_ERR_UNSUPPORTED;
  auto go = [&](auto kern) {
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return (int)CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, g, cfg);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_zqpa_pipe2<%s> %s M=%d T=%d store_waves=%d lds=%zu", sizeof(TIO) == 8 ? "f64" : "f32", REGSTAGE ? "register-staged" : "double-buffered",
                M, T, nsw, sh);  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  constexpr int RSV = REGSTAGE ? PIPE_RS : 0;
  return nthr <= 512 ? go(k_zqpa_pipe2<TIO, M, T, 512, RSV>) : go(k_zqpa_pipe2<TIO, M, T, 1024, RSV>);
}

template <typename TIO, int M, int T, int MAXT, bool FLAT>
__global__ __launch_bounds__(MAXT) void k_zqpa_pipe(SolveArgs a, PipeCfg cfg) {
  extern __shared__ double lds[];
  TriBand band = {};  // requested together with the record (see TriBand): 104 registers leave room for it here, not in k_tri_pipe (126)
  if ((int)threadIdx.x < cfg.ncomp) band = load_tri_band<TIO>(a, blockIdx.x, (int)threadIdx.x < a.nb ? threadIdx.x : 0);
  {
    const double* src = a.ws + (long long)blockIdx.x * a.reclen;
    for (int i = threadIdx.x; i < a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  if ((int)threadIdx.x >= cfg.ncomp) {
    if constexpr (FLAT)
      zqpa_pipe_store_flat<TIO, M, T>(a, cfg, lds);
    else
      zqpa_pipe_store<TIO, M, T>(a, cfg, lds);
    return;
  }
  tri_pipe_compute<TriZqPa, TIO, M, T, 0, 2>(a, cfg, lds, &band);
}

// returns CRT_ERR_UNSUPPORTED when the shape does not fit (caller falls back to the two-kernel path)
template <typename TIO, int M, int T>
int launch_zqpa_fused(const SolveArgs& a, hipStream_t s, int nsw, size_t lds_cap = MAX_WG_LDS) {
  if (a.nb < (a.tune[12] > 0 ? a.tune[12] : 16)) return CRT_ERR_UNSUPPORTED;  // (tune 12: smallest nb, as for the other pipelines)
  const bool flat = a.nb % 2;  // odd nb: rows are not pair-aligned -> flat store role
  if (flat && a.tune[13] == 1) return CRT_ERR_UNSUPPORTED;
  for (int i = 0; i < 4; ++i)
    if (reinterpret_cast<uintptr_t>(a.o[i]) & (2 * sizeof(TIO) - 1)) return CRT_ERR_UNSUPPORTED;
  const int Mg = zqpa_M(a.nz);
  const int ncomp = ((a.nb + 63) / 64) * 64;
  // store waves (nsw <= 0: automatic).  Measured (tools/ab_zqpa.py), two-kernel path -> fused with 2 / 3 / 4 / 5 store waves:
  //   1e4 x 300 x 60: 3.16 ms -> 1.51 / 1.34 / 1.67 / 1.60;  6e3 x 300 x 100 (one workgroup per CU): 3.16 -> 1.88 / 1.70 / 1.61 / 1.56;
  //   3e4 x 128 x 60: 3.47 -> 1.56 / 2.26 / 2.26 / 2.15
  const size_t lds_doubles = (size_t)((a.reclen + 1) & ~1) + ((a.nb + 1) & ~1) + 2 * (size_t)std::max(Mg / M - 1, 0) * ncomp + 4 * (size_t)T * a.nb + 6 * (size_t)a.nb;
  if (nsw <= 0) nsw = ncomp <= 64 ? 1 : ncomp <= 128 ? 2 : (lds_doubles * sizeof(double) > MAX_WG_LDS / 2 ? 5 : 3);
  if (ncomp + 64 * nsw > 1024) nsw = (1024 - ncomp) / 64;
  if (nsw < 1) return CRT_ERR_UNSUPPORTED;
  const int nthr = ncomp + 64 * nsw;
  SolveArgs g = a;  // computational-grid solve: nz := Mg; the outputs go through PipeCfg
  g.nz = Mg;
  for (int i = 0; i < 7; ++i) g.o[i] = nullptr;
  PipeCfg cfg{};
  cfg.ncomp = ncomp;
  cfg.nck = std::max(Mg / M - 1, 0);  // K = Mg + 1 rows; checkpoints for the segments 1 .. top-1 (see tri_pipe_compute)
  cfg.off_bc = (a.reclen + 1) & ~1;
  cfg.off_ck = cfg.off_bc + ((a.nb + 1) & ~1);
  cfg.off_tile = cfg.off_ck + 2 * cfg.nck * ncomp;
  cfg.off_halo = cfg.off_tile + 2 * 2 * T * a.nb;
  cfg.nz_out = a.nz;
  for (int i = 0; i < 4; ++i) cfg.out[i] = a.o[i];
  const size_t sh = ((size_t)cfg.off_halo + 2 * 3 * a.nb) * sizeof(double);
  if (sh > lds_cap) return CRT_ERR_UNSUPPORTED;
  auto go = [&](auto kern) {
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return (int)CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, g, cfg);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_zqpa_pipe<%s%s> M=%d T=%d store_waves=%d lds=%zu", sizeof(TIO) == 8 ? "f64" : "f32", flat ? ",flat" : "", M, T, nsw, sh);  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  if (flat) return nthr <= 512 ? go(k_zqpa_pipe<TIO, M, T, 512, true>) : go(k_zqpa_pipe<TIO, M, T, 1024, true>);
  return nthr <= 512 ? go(k_zqpa_pipe<TIO, M, T, 512, false>) : go(k_zqpa_pipe<TIO, M, T, 1024, false>);
}

}  // namespace

int launch_zqpa(const SolveArgs& a, double* scratch, hipStream_t s) {
  const int* g_tri_tune = a.tune + 8;  // this call's overrides
  if (g_tri_tune[2] != 1 || a.f32) {  // fused interpolation first (tune key 10 = 1: the two-kernel path with workspace scratch)
    const int nsw = g_tri_tune[3];
    int st;
    // two workgroups per CU first (half of the LDS each): M = 16, T = 4, or -- above ~85 levels at 300 bands -- tiles of 3 levels
    // (M = 15), which is what brings 100 levels from 84 KB to 74 KB; then whatever fits at all
    constexpr size_t HALF = MAX_WG_LDS / 2;
    if (a.f32) {
      st = launch_zqpa_fused<float, 16, 4>(a, s, nsw, HALF);
      if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<float, 15, 3>(a, s, nsw, HALF);
      if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<float, 16, 4>(a, s, nsw);
      if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<float, 12, 4>(a, s, nsw);
      return st;  // f32 storage exists in the fused kernel only (the two-kernel path keeps its computational-grid scratch in fp64)
    }
    st = CRT_ERR_UNSUPPORTED;
    // round 3: interpolation in the compute lanes + the plain fused store role (even nb; tune key 10 = 5 keeps the kernel below; key 10 = 6 /
    // 7 force the double-buffered / register-staged form)
    // Measured (tools/ragged_sweep.py, round 3; both kernels on the division-free sweep): 1e4 x 300 x 60 1.10 (below) vs 1.11-1.13 ms,
    // 6000 x 300 x 100 1.23 vs 1.12-1.15, 3e4 x 106 x 60 1.23 vs 2.67, 1e5 x 38 x 100 3.00 vs 3.58 -> the new form above 128 bands only.
    // Three workgroups per CU (M = 8 capped at 80 registers, 40 B of scratch) gave 1.19 ms: occupancy is not what binds it.
    // (round 3, after the level emission was rewritten: also 65 .. 128 even bands with ONE store wave -- 3e4 x 106 x 60 1.21 -> 1.13 ms; below
    //  65 bands the older kernel stays ahead, 1e5 x 38 x 100 3.01 vs 3.15 ms: profiles/r03/zqpa_pipe2_narrow_tune.txt)
    if (g_tri_tune[2] != 5 && (a.nb > 64 || g_tri_tune[2] >= 6 || g_tri_tune[0] == 8)) {
      constexpr size_t HALF2 = MAX_WG_LDS / 2;
      const int mode = g_tri_tune[2];
      if (g_tri_tune[0] == 8) {  // A/B: short segments (fewer registers: five waves per SIMD, three workgroups per CU)
        if (mode != 6) st = launch_zqpa_fused2<double, 8, 4, true>(a, s, nsw, MAX_WG_LDS / 3);
        if (st == CRT_ERR_UNSUPPORTED && mode != 7) st = launch_zqpa_fused2<double, 8, 4, false>(a, s, nsw, MAX_WG_LDS / 3);
        if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused2<double, 8, 4, true>(a, s, nsw, HALF2);
      } else if (a.nz <= 64) {  // few checkpoints: M = 16 keeps the LDS small; above, M = 16 as well (registers cap M)
        if (mode != 6) st = launch_zqpa_fused2<double, 16, 4, true>(a, s, nsw, HALF2);
        if (st == CRT_ERR_UNSUPPORTED && mode != 7) st = launch_zqpa_fused2<double, 16, 4, false>(a, s, nsw, HALF2);
      } else {
        if (mode != 6) st = launch_zqpa_fused2<double, 16, 4, true>(a, s, nsw, HALF2);
        if (st == CRT_ERR_UNSUPPORTED && mode != 7) st = launch_zqpa_fused2<double, 16, 4, false>(a, s, nsw, HALF2);
      }
      if (st == CRT_ERR_UNSUPPORTED && mode != 6) st = launch_zqpa_fused2<double, 16, 4, true>(a, s, nsw);
      if (st == CRT_ERR_UNSUPPORTED && mode != 7) st = launch_zqpa_fused2<double, 16, 4, false>(a, s, nsw);
      if (st != CRT_ERR_UNSUPPORTED) return st;
    }
    if (a.nb % 2 == 1 && g_tri_tune[2] != 5) {  // odd band counts: the new compute role with the flat flush (tune key 10 = 5: the kernel below)
      st = launch_zqpa_generic2<double, 16, 4>(a, s, nsw);
      if (st != CRT_ERR_UNSUPPORTED) return st;
    }
    // narrow spectra (one compute wave per column): M = 12 needs 92 registers, five waves per SIMD instead of four (1e5 x 38 x 100:
    // 3.74 -> 3.64 ms, 2e5 x 16 x 60 3.23 -> 3.16; at 62 bands the other way, 2.61 -> 2.64).  (tune key 8 = 16 keeps M = 16.)
    if (a.nb <= 48 && g_tri_tune[0] != 16) st = launch_zqpa_fused<double, 12, 4>(a, s, nsw, HALF);
    if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<double, 16, 4>(a, s, nsw, HALF);
    if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<double, 15, 3>(a, s, nsw, HALF);
    if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<double, 16, 4>(a, s, nsw);
    if (st == CRT_ERR_UNSUPPORTED) st = launch_zqpa_fused<double, 12, 4>(a, s, nsw);
    if (st != CRT_ERR_UNSUPPORTED) return st;
  }
  const int M = zqpa_M(a.nz);
  SolveArgs g = a;  // computational-grid solve: nz := M, outputs := scratch
  g.nz = M;
  g.o[0] = scratch;
  g.o[1] = scratch + (size_t)a.ncol * M * a.nb;
  for (int i = 2; i < 7; ++i) g.o[i] = nullptr;
  bool done = false;
  int st = launch_scheme<TriZqPa, double>(g, s, done, 1);
  if (st != CRT_OK) return st;
  if (!done && (st = launch_zqpa_wave(g, s)) != CRT_OK) return st;  // nb > 1024: per-wave kernel (solve_tridiag.hip)
  InterpArgs ia;
  ia.ncol = a.ncol;
  ia.nb = a.nb;
  ia.nz = a.nz;
  ia.M = M;
  ia.reclen = a.reclen;
  ia.col_stride = a.col_stride;
  ia.ws = a.ws;
  ia.dnz = scratch;
  ia.upz = scratch + (size_t)a.ncol * M * a.nb;
  ia.I_dr0 = a.I_dr0;
  for (int i = 0; i < 4; ++i) ia.o[i] = a.o[i];
  const size_t sh = a.reclen * sizeof(double);
  if (sh > MAX_WG_LDS) return CRT_ERR_UNSUPPORTED;
  if (sh > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_zqpa_interp<double>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
    return CRT_ERR_LAUNCH;
  hipLaunchKernelGGL((k_zqpa_interp<double>), dim3(a.ncol), dim3(256), sh, s, ia);
  if (hipGetLastError() != hipSuccess) return CRT_ERR_LAUNCH;
  note_kernel("zq_pa two-kernel path: grid solve + k_zqpa_interp");  // (only a launch that succeeded is reported)
  return CRT_OK;
}

}  // namespace crt
