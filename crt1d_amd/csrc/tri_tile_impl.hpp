// Templates of the column-tile / pipeline kernels for the tridiagonal schemes.  Included by one translation unit per scheme
// (tri_inst.hip x 4, tri_zqpa.hip) so that the instantiations compile in parallel; not part of the ABI.
#pragma once
// Tridiagonal schemes (n79, zq), column-tile kernel for gfx950.
//
// Why a second kernel: the per-wave kernels of solve_tridiag.hip keep 16 nz bytes of Thomas state per lane in
// LDS (2 waves per CU at nz = 60) and let every lane store its own band (512-B runs that are only 8-B aligned when
// nb*8 is not a multiple of 128 B).  Both hurt: 0.33-0.37 of HBM peak.  Here
//
//  * a workgroup owns ONE whole column (all nb bands), so -- as in k_tile of solve_closed.hip -- T consecutive
//    levels of an output array are one contiguous run that is staged in LDS and flushed with 16-B-per-lane stores
//    covering whole 128-B lines;
//  * the Thomas state is CHECKPOINTED: the forward sweep keeps the even-row pair (e, f) only every M-th level
//    (nz/M pairs per lane in LDS).  The back substitution walks the segments from the top; for each segment it
//    re-runs the forward recurrence from the segment's checkpoint into M-1 register-resident pairs and then
//    back-substitutes through them.  Every pair is produced by exactly the arithmetic of the plain sweep, so the
//    result is bitwise what the per-wave kernel computes -- no unstable "shooting" recurrences -- at the price
//    of running the (cheap) forward recurrence twice.  LDS per lane drops from 16 nz to 16 nz/M bytes;
//  * only the 4 profiles that need the solve (dn, up and the two scheme extras) go through the LDS tile; I_dr and
//    the F arrays are linear combinations with per-band / per-level constants and are formed while flushing.
//
// LDS at nb = 300, nz = 60, M = 12, T = 4: checkpoints 25.6 KB + tile 38.4 KB + record/band constants 5.4 KB
// = 69 KB -> two workgroups (10 waves) per CU.
#include <type_traits>

#include "crt_internal.hpp"
#include "tri_schemes.hpp"

namespace crt {
namespace {

constexpr size_t MAX_WG_LDS = 160 * 1024;

struct TriCfg {
  int T;        // levels per flush tile (divides M)
  int nck;      // checkpoints per lane
  int off_bc;   // LDS offsets in doubles: band constants
  int off_ck;   // checkpoints [nck][2][nthr]
  int off_tile; // tile [NST][T][nb]
  int flat;     // odd nb: fused flat flush (flush_flat) instead of the per-array generic flush
};

// ------------------------------------------------------------------------------------------
// flush one output array: rows [j0, j0 + nrows) of column c, one contiguous run
template <class S, typename TIO, int ARR>
__device__ inline void flush_array(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int tstride,
                                   int c, int j0, int nrows, double invmu, float inv_nb, int tid, int nthr) {
  if (nrows <= 0) return;
  typedef TIO vt __attribute__((ext_vector_type(2)));
  const int nb = a.nb, nz = a.nz;
  const int n = nrows * nb;
  TIO* g = outp<TIO>(a.o[ARR]) + ((long long)c * S::out_rows(ARR, nz) + j0) * nb;
  const int mis = (int)((reinterpret_cast<uintptr_t>(g) / sizeof(TIO)) & 1);  // run starts on an odd element?
  const int npair = (n - mis) >> 1;
  auto elem = [&](int i) -> double {
    if constexpr (S::derived(ARR)) {
      int t = (int)(((float)i + 0.5f) * inv_nb);
      int b = i - t * nb;
      if (b < 0) { --t; b += nb; }
      if (b >= nb) { ++t; b -= nb; }
      return S::template value<ARR>(rec, nz, j0 + t, bandc[b], invmu, tile, tstride, i);
    } else {
      return tile[S::staged_slot(ARR) * tstride + i];
    }
  };
  vt* g2 = reinterpret_cast<vt*>(g + mis);
  for (int i = tid; i < npair; i += nthr) {
    vt v;
    v.x = (TIO)elem(mis + 2 * i);
    v.y = (TIO)elem(mis + 2 * i + 1);
    g2[i] = v;  // (plain stores: the runs of this flush begin and end inside 128-B lines, which L2 has to merge)
  }
  if (tid == 0) {
    if (mis) g[0] = (TIO)elem(0);
    if ((n - mis) & 1) g[n - 1] = (TIO)elem(n - 1);
  }
}

// tid / nthr: the threads that take part in the flush (all of the workgroup in k_tri_tile, the store waves in k_tri_pipe)
template <class S, typename TIO, int ARR>
__device__ inline void flush_arrays(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int tstride, int c,
                                    int j0, int T, double invmu, float inv_nb, int tid, int nthr) {
  const int nr = min(j0 + T, S::out_rows(ARR, a.nz)) - j0;
  flush_array<S, TIO, ARR>(a, rec, bandc, tile, tstride, c, j0, nr, invmu, inv_nb, tid, nthr);
  if constexpr (ARR + 1 < S::NOUT) flush_arrays<S, TIO, ARR + 1>(a, rec, bandc, tile, tstride, c, j0, T, invmu, inv_nb, tid, nthr);
}

// ------------------------------------------------------------------------------------------
// Fused FLAT flush for odd nb (round 2).  flush_arrays above walks every output array separately and derives (level, band) of
// every element with a float division; at the reference's 107 bands that index arithmetic, not HBM, bounds the tridiagonal
// kernels (0.59-0.65 of the peak, SQ counters in profiles/r02/rocprof/{n79,zq}_nb107).  Arrays with the same number of rows are
// flat runs of identical length and alignment inside their buffers, so ONE walk over the element pairs of the tile serves all of
// them: (level, band) advance incrementally, the staged values are read once (the compiler merges the LDS reads of the value<>()
// expressions), and every array gets its 16-byte store.  Same expressions as flush_array -> same bits.
// CLS = 0: arrays with nz rows, CLS = 1: arrays with nz - 1 rows (n79's per-leaf-area absorption).
// A pair of bands of one output row.  Arrays with nz rows per column: a column is 8 nz nb bytes and a tile of T rows a whole number of
// 128-B lines at the band counts of the fused flush that matter (300 bands: 75 lines per 4 rows) -> streaming stores, nothing is read
// back (2s 0.92 -> 0.85 ms, zq 1.57 -> 1.50 at 1e4 x 300 x 60).  Arrays with nz - 1 rows (n79's absorbed fluxes) start every column
// and end every tile inside a line: those part-lines are left to L2 to merge (plain stores; streaming them cost n79 3 %).
// The choice is per SCHEME (all arrays of zq have nz rows; n79 keeps plain stores for all six: choosing per array inside the unrolled
// loop over the arrays cost zq 7 %).
template <class S, typename TIO>
__device__ __forceinline__ void store_rows(TIO __attribute__((ext_vector_type(2))) * p, TIO __attribute__((ext_vector_type(2))) v) {
  if constexpr (S::out_rows(S::NOUT - 1, 8) == 8)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

template <class S, typename TIO, int CLS, int ARR, class F>
__device__ __forceinline__ void for_class(int nz, F&& f) {
  if constexpr (ARR < S::NOUT) {
    if constexpr ((S::out_rows(ARR, 8) == 8 ? 0 : 1) == CLS) f(std::integral_constant<int, ARR>{});
    for_class<S, TIO, CLS, ARR + 1>(nz, f);
  }
}

template <class S, typename TIO, int CLS>
__device__ inline void flush_flat_class(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int tstride,
                                        int c, int j0, int T, double invmu, int tid, int nthr, bool stream) {
  typedef TIO vt __attribute__((ext_vector_type(2)));
  constexpr int LINE = 128 / (int)sizeof(TIO);
  const int nb = a.nb, nz = a.nz, rows = nz - CLS;
  const int nr = min(j0 + T, rows) - j0;
  if (nr <= 0) return;
  const int n = nr * nb;
  const long long g0 = ((long long)c * rows + j0) * nb;  // start of the run inside every array of the class (bases are pair-aligned: checked by the launcher)
  const int mis = (int)(g0 & 1);
  const int npair = (n - mis) >> 1;
  const int step = 2 * nthr;
  int e = mis + 2 * tid;
  int t = (int)(((float)e + 0.5f) * (1.0f / (float)nb));
  int b = e - t * nb;
  if (b < 0) { --t; b += nb; }
  if (b >= nb) { ++t; b -= nb; }
  const int dt = step / nb, db = step - dt * nb;
  for (int i = tid; i < npair; i += nthr) {
    const bool wrap = b + 1 >= nb;
    const int t2 = wrap ? t + 1 : t, b2 = wrap ? 0 : b + 1;
    const double bcx = bandc[b], bcy = bandc[b2];
    // the 128-B line this pair lies in is written completely by this run?  (array bases are line-aligned when `stream` is set.)
    // Whole lines are streamed; the part-lines at the two ends of the run wait in L2 for their other half (plain stores --
    // streaming those too: n79 at 107 bands 1.59 -> 1.76 ms)
    const long long l0 = (g0 + e) & ~(long long)(LINE - 1);
    const bool whole = stream && l0 >= g0 && l0 + LINE <= g0 + n;
    for_class<S, TIO, CLS, 0>(nz, [&](auto arr) {
      constexpr int ARRI = decltype(arr)::value;
      vt v;
      v.x = (TIO)S::template value<ARRI>(rec, nz, j0 + t, bcx, invmu, tile, tstride, e);
      v.y = (TIO)S::template value<ARRI>(rec, nz, j0 + t2, bcy, invmu, tile, tstride, e + 1);
      vt* dst = reinterpret_cast<vt*>(outp<TIO>(a.o[ARRI]) + g0 + e);
      if (whole)
        __builtin_nontemporal_store(v, dst);
      else
        *dst = v;
    });
    e += step;
    b += db;
    t += dt;
    if (b >= nb) {
      b -= nb;
      ++t;
    }
  }
  if (tid == 0) {  // the unpaired elements at the two ends of the run
    if (mis)
      for_class<S, TIO, CLS, 0>(nz, [&](auto arr) {
        constexpr int ARRI = decltype(arr)::value;
        outp<TIO>(a.o[ARRI])[g0] = (TIO)S::template value<ARRI>(rec, nz, j0, bandc[0], invmu, tile, tstride, 0);
      });
    if ((n - mis) & 1) {
      const int el = n - 1, tl = nr - 1, bl = nb - 1;
      for_class<S, TIO, CLS, 0>(nz, [&](auto arr) {
        constexpr int ARRI = decltype(arr)::value;
        outp<TIO>(a.o[ARRI])[g0 + el] = (TIO)S::template value<ARRI>(rec, nz, j0 + tl, bandc[bl], invmu, tile, tstride, el);
      });
    }
  }
}

// ------------------------------------------------------------------------------------------
// WHOLE-LINE flat flush (round 3; pipelines only: the tiles of a column are flushed top-down by the same store waves).  A tile of T rows
// of an odd spectrum begins and ends inside 128-B lines; flush_flat_class writes those two part-lines per array and tile with plain stores
// and leaves the merging to L2.  Here the run of a tile is shifted instead: the elements in front of the first line boundary (the head,
// <= LINE - 1 elements of the tile's lowest row) are not written but PARKED -- their staged values copied to a small LDS area -- and the
// tile below, whose run ends where this one began, appends them to its own elements.  Every tile then writes whole lines only, all of
// them streamed; only the first line of a column's bottom tile and the last line of its top tile can be partial (shared with the
// neighbouring columns, which other workgroups write).  Same value<>() expressions as flush_flat_class -> same bits.
// Requires line-aligned array bases (flat_flush_ok == 2) and nb >= LINE (a tile is at least one line long).
// park_in: [NST][LINE] staged values parked by the tile above (row j0 + nr, bands 0 ..); park_out: where this tile parks its own head.
template <class S, typename TIO, int CLS>
__device__ inline void flush_flat_class_wl(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int tstride,
                                           int c, int j0, int T, double invmu, int tid, int nthr, const double* park_in, double* park_out) {
  typedef TIO vt __attribute__((ext_vector_type(2)));
  constexpr int LINE = 128 / (int)sizeof(TIO);
  const int nb = a.nb, nz = a.nz, rows = nz - CLS;
  const int nr = min(j0 + T, rows) - j0;
  if (nr <= 0) return;
  const int n = nr * nb;
  const long long g0 = ((long long)c * rows + j0) * nb;
  const bool first = j0 + nr >= rows;  // holds the top row of the class: nothing was parked above it
  const bool last = j0 == 0;           // bottom tile: its head is written (the line in front belongs to the previous column)
  const int e_lo = last ? 0 : (int)((LINE - (g0 & (LINE - 1))) & (LINE - 1));                  // own elements [0, e_lo) are parked
  const int npin = first ? 0 : (int)((LINE - ((g0 + n) & (LINE - 1))) & (LINE - 1));           // parked elements appended behind element n - 1
  const int mis = (int)((g0 + e_lo) & 1);  // (non-zero in the bottom tile only)
  const int e0 = e_lo + mis;
  const int npair = (n - e0) >> 1;
  const int odd = (n - e0) & 1;
  const int goff = (int)(g0 & (LINE - 1)), run_hi = n + npin;  // (lines in element offsets from g0: 32-bit arithmetic)
  const int step = 2 * nthr;
  int e = e0 + 2 * tid;
  int t = (int)(((float)e + 0.5f) * (1.0f / (float)nb));
  int b = e - t * nb;
  if (b < 0) { --t; b += nb; }
  if (b >= nb) { ++t; b -= nb; }
  const int dt = step / nb, db = step - dt * nb;
  for (int i = tid; i < npair; i += nthr) {
    const bool wrap = b + 1 >= nb;
    const int t2 = wrap ? t + 1 : t, b2 = wrap ? 0 : b + 1;
    const double bcx = bandc[b], bcy = bandc[b2];
    const int l0 = ((goff + e) & ~(LINE - 1)) - goff;
    const bool whole = l0 >= e_lo && l0 + LINE <= run_hi;  // false in the first line of a column / the last line of its top tile only
    for_class<S, TIO, CLS, 0>(nz, [&](auto arr) {
      constexpr int ARRI = decltype(arr)::value;
      vt v;
      v.x = (TIO)S::template value<ARRI>(rec, nz, j0 + t, bcx, invmu, tile, tstride, e);
      v.y = (TIO)S::template value<ARRI>(rec, nz, j0 + t2, bcy, invmu, tile, tstride, e + 1);
      vt* dst = reinterpret_cast<vt*>(outp<TIO>(a.o[ARRI]) + g0 + e);
      if (whole)
        __builtin_nontemporal_store(v, dst);
      else
        *dst = v;
    });
    e += step;
    b += db;
    t += dt;
    if (b >= nb) {
      b -= nb;
      ++t;
    }
  }
  // behind the pairs: the tile's last element when their count is odd, then the parked elements of the row above -- an even number in
  // all (the run ends on a line boundary) unless this is the top tile, whose single last element closes the column
  const int ntail = odd + npin;
  if (2 * tid < ntail) {
    const int k = 2 * tid;  // first tail element of this thread
    auto val = [&](auto arr, int kk) -> double {
      constexpr int ARRI = decltype(arr)::value;
      const int ei = n - odd + kk;
      if (ei < n) return S::template value<ARRI>(rec, nz, j0 + nr - 1, bandc[nb - 1], invmu, tile, tstride, ei);
      return S::template value<ARRI>(rec, nz, j0 + nr, bandc[ei - n], invmu, park_in, LINE, ei - n);
    };
    const long long gi = g0 + n - odd + k;
    for_class<S, TIO, CLS, 0>(nz, [&](auto arr) {
      constexpr int ARRI = decltype(arr)::value;
      if (k + 1 < ntail) {
        vt v;
        v.x = (TIO)val(arr, k);
        v.y = (TIO)val(arr, k + 1);
        vt* dst = reinterpret_cast<vt*>(outp<TIO>(a.o[ARRI]) + gi);
        if (!first)
          __builtin_nontemporal_store(v, dst);  // (completes the run's last line)
        else
          *dst = v;
      } else {
        outp<TIO>(a.o[ARRI])[gi] = (TIO)val(arr, k);
      }
    });
  }
  if (tid == nthr - 1 && mis)  // bottom tile starting on an odd element
    for_class<S, TIO, CLS, 0>(nz, [&](auto arr) {
      constexpr int ARRI = decltype(arr)::value;
      outp<TIO>(a.o[ARRI])[g0] = (TIO)S::template value<ARRI>(rec, nz, j0, bandc[0], invmu, tile, tstride, 0);
    });
  // park the head for the tile below: the staged arrays this class's outputs are formed from (the classes share one [NST][LINE] block:
  // n79's level arrays read slots 0, 1 and its layer arrays slots 2, 3 -- S::park_slots)
  constexpr int QLO = S::park_slots(CLS, 0), QN = S::park_slots(CLS, 1) - QLO;
  if (tid < e_lo) {  // (e_lo < LINE <= the threads of one store wave)
#pragma unroll
    for (int q = QLO; q < QLO + QN; ++q) park_out[q * LINE + tid] = tile[q * tstride + tid];
  }
}

template <class S>
constexpr int flush_classes() { return S::out_rows(S::NOUT - 1, 8) != 8 ? 2 : 1; }
// doubles of LDS the whole-line flush needs: [2 tiles in flight][NST][LINE]
template <class S, typename TIO>
constexpr int park_doubles() { return 2 * S::NST * (128 / (int)sizeof(TIO)); }

template <class S, typename TIO>
__device__ inline void flush_flat_wl(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int tstride, int c,
                                     int j0, int T, double invmu, int tid, int nthr, double* park, int pb) {
  constexpr int LINE = 128 / (int)sizeof(TIO);
  constexpr int PC = S::NST * LINE;  // one class of one tile
  double* pin = park + pb * PC;
  double* pout = park + (pb ^ 1) * PC;
  flush_flat_class_wl<S, TIO, 0>(a, rec, bandc, tile, tstride, c, j0, T, invmu, tid, nthr, pin, pout);
  if constexpr (flush_classes<S>() == 2) flush_flat_class_wl<S, TIO, 1>(a, rec, bandc, tile, tstride, c, j0, T, invmu, tid, nthr, pin, pout);
}

template <class S, typename TIO>
__device__ inline void flush_flat(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int tstride, int c,
                                  int j0, int T, double invmu, int tid, int nthr, bool stream) {
  flush_flat_class<S, TIO, 0>(a, rec, bandc, tile, tstride, c, j0, T, invmu, tid, nthr, stream);
  if constexpr (S::out_rows(S::NOUT - 1, 8) != 8) flush_flat_class<S, TIO, 1>(a, rec, bandc, tile, tstride, c, j0, T, invmu, tid, nthr, stream);
}

// host side: may the flat fused flush be used?  Every output array must start on a pair boundary (2 * sizeof(TIO)).
template <class S, typename TIO>
inline int flat_flush_ok(const SolveArgs& a) {  // 0: no; 1: yes; 2: yes, and every array starts on a 128-B line (whole lines of a run are streamed)
  int mode = 2;
  for (int i = 0; i < S::NOUT; ++i) {
    if (reinterpret_cast<uintptr_t>(a.o[i]) & (2 * sizeof(TIO) - 1)) return 0;
    if (reinterpret_cast<uintptr_t>(a.o[i]) & 127) mode = 1;
  }
  return mode;
}

// ------------------------------------------------------------------------------------------
// Fused flush for even nb (a band pair never straddles a row and every row is 16-B aligned): thread -> (row offset,
// band pair) is fixed for the whole kernel, so there is no index arithmetic per element; the 4 staged pairs are read
// once and all NOUT arrays are stored in the same pass.  Rows of a tile are adjacent in memory, so consecutive threads
// still write consecutive 16-B words: every wave store is a contiguous, line-aligned 1 KiB.
struct FlushMap {
  int t_off;  // row of the tile this thread starts at
  int p;      // band pair
  int rpi;    // rows covered per iteration by the workgroup
  bool on;
};

template <class S, typename TIO, int T>
__device__ inline void flush_fused(const SolveArgs& a, const double* rec, const double* bandc, const double* tile, int c, int j0,
                                   const FlushMap& fm, double invmu) {
  const int nb2 = a.nb >> 1, nz = a.nz;
  const d2* tile2 = reinterpret_cast<const d2*>(tile);
  const d2 bc = reinterpret_cast<const d2*>(bandc)[fm.p];
  for (int t = fm.t_off; t < T; t += fm.rpi) {
    const int j = j0 + t;
    if (!fm.on || j >= nz) continue;
    d2 st[S::NST], o[S::NOUT];
#pragma unroll
    for (int q = 0; q < S::NST; ++q) st[q] = tile2[(q * T + t) * nb2 + fm.p];
    S::emit(rec, nz, j, bc, invmu, st, o);
#pragma unroll
    for (int k = 0; k < S::NOUT; ++k) {
      const int rows = S::out_rows(k, nz);
      if (j < rows) {
        typedef TIO vt __attribute__((ext_vector_type(2)));
        vt v;
        v.x = (TIO)o[k].x;
        v.y = (TIO)o[k].y;
        store_rows<S, TIO>(reinterpret_cast<vt*>(a.o[k]) + ((long long)c * rows + j) * nb2 + fm.p, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The segment loops are fully unrolled: the register-resident pairs be[M], bf[M] need static indices (a runtime-indexed
// array goes to scratch; VGPR-index mode on vector types works but costs ~40 instructions per level, measured).
template <class S, typename TIO, int M, int T, bool FUSED>
__device__ __forceinline__ void tri_tile_body(const SolveArgs& a, const TriCfg& cfg, double* lds) {
  const int nb = a.nb, nz = a.nz;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int c = blockIdx.x;
  const double* rec = lds;
  double* bandc = lds + cfg.off_bc;
  double* ck = lds + cfg.off_ck + tid;       // [nck][2][nthr]
  double* tile = lds + cfg.off_tile;         // [NST][T][nb]
  const int tstride = T * nb;
  const bool active = tid < nb;
  const int b = active ? tid : 0;
  const float inv_nb = 1.0f / (float)nb;
  const double invmu = rec[S_INVMU];
  FlushMap fm;
  {
    const int nb2 = nb >> 1;
    fm.rpi = nthr / nb2;  // >= 2 because nthr >= nb
    fm.t_off = tid / nb2;
    fm.p = tid - fm.t_off * nb2;
    fm.on = fm.t_off < fm.rpi;
  }

  S st;
  st.template init<TIO>(rec, a, c, b);
  if (active) bandc[b] = st.band_const();
  const int K = S::rows(nz);

  // ---- pass 1: forward recurrence (state St, tri_schemes.hpp), keep the even-row pair of every M-th level ----
  static_assert(S::RENORM == 0 || M % S::RENORM == 0, "checkpoints must fall on re-seeding levels");
  typename S::St fs;
  st.first(rec, nz, fs);
  {
    double e, f;
    st.pair(fs, e, f);
    ck[0] = e;
    ck[nthr] = f;
  }
  tri_forward<S, M>(st, rec, nz, fs, K - 1, [&](int level, const typename S::St& cs) {
    const int s = level / M;
    double e, f;
    st.pair(cs, e, f);
    ck[(2 * s) * nthr] = e;
    ck[(2 * s + 1) * nthr] = f;
  });

  // ---- pass 2: segments from the top; recompute the segment's pairs into registers, back-substitute, flush ----
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    double be[M], bf[M];  // pairs of levels k0 .. k0+M-1 (statically indexed: the loops below are fully unrolled)
    be[0] = ck[(2 * seg) * nthr];
    bf[0] = ck[(2 * seg + 1) * nthr];
    typename S::St rs;
    st.seed(rs, be[0], bf[0]);
#pragma unroll
    for (int i = 1; i < M; ++i) {
      be[i] = be[i - 1];
      bf[i] = bf[i - 1];
      if (k0 + i <= kend) tri_step(st, k0 + i - 1, rec, nz, rs, be[i], bf[i]);
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const int k = k0 + i;
      if (k <= kend) {
        double o[S::NST];
        if (k == K - 1)
          st.top(rec, nz, be[i], bf[i], o);
        else
          st.back(k, rec, nz, be[i], bf[i], o);
        if (active) {
#pragma unroll
          for (int q = 0; q < S::NST; ++q) tile[q * tstride + (i % T) * nb + b] = o[q];
        }
      }
      if (i % T == 0) {  // bottom row of a tile (T divides M, segments start at multiples of M)
        if (k <= kend) {
          // LDS-only barriers: __syncthreads() would also drain this wave's global stores (vmcnt(0)) and serialise the
          // recurrence behind the HBM write round trip of the previous tile
          lds_barrier();
          if constexpr (FUSED)
            flush_fused<S, TIO, T>(a, rec, bandc, tile, c, k, fm, invmu);
          else if (cfg.flat)
            flush_flat<S, TIO>(a, rec, bandc, tile, tstride, c, k, T, invmu, tid, nthr, cfg.flat == 2);
          else
            flush_arrays<S, TIO, 0>(a, rec, bandc, tile, tstride, c, k, T, invmu, inv_nb, tid, nthr);
          lds_barrier();
        }
      }
    }
  }
}

// S_UNIF columns run the scheme's uniform-dLAI object (UniformOf, tri_schemes.hpp); the flag is per column = per workgroup
template <class S, typename TIO, int M, int T, int MAXT, bool FUSED>
__global__ __launch_bounds__(MAXT) void k_tri_tile(SolveArgs a, TriCfg cfg) {
  static_assert(M % T == 0, "tile height must divide the checkpoint spacing");
  extern __shared__ double lds[];
  {
    const double* src = a.ws + (long long)blockIdx.x * a.reclen;
    for (int i = threadIdx.x; i < a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  typedef typename UniformOf<S>::type SU;
  if constexpr (!std::is_same<S, SU>::value) {
    if (lds[S_UNIF] != 0.0) {
      tri_tile_body<SU, TIO, M, T, FUSED>(a, cfg, lds);
      return;
    }
  }
  tri_tile_body<S, TIO, M, T, FUSED>(a, cfg, lds);
}

template <class S, typename TIO, int M, int T, bool FUSED>
int launch_mt(const SolveArgs& a, hipStream_t s, int nthr) {
  const int K = S::rows(a.nz);
  TriCfg cfg{};
  cfg.T = T;
  cfg.nck = (K - 1) / M + 1;
  cfg.off_bc = (a.reclen + 1) & ~1;
  cfg.off_ck = cfg.off_bc + ((a.nb + 1) & ~1);
  cfg.off_tile = cfg.off_ck + 2 * cfg.nck * nthr;
  cfg.flat = (!FUSED && a.tune[13] != 1) ? flat_flush_ok<S, TIO>(a) : 0;
  const size_t sh = ((size_t)cfg.off_tile + (size_t)S::NST * T * a.nb) * sizeof(double);
  if (sh > MAX_WG_LDS) return CRT_ERR_UNSUPPORTED;
  const void* fn = nthr <= 256 ? (const void*)k_tri_tile<S, TIO, M, T, 256, FUSED> : nthr <= 512 ? (const void*)k_tri_tile<S, TIO, M, T, 512, FUSED>
                                                                                      : (const void*)k_tri_tile<S, TIO, M, T, 1024, FUSED>;
  if (sh > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
    return CRT_ERR_LAUNCH;
  dim3 grid(a.ncol), block(nthr);
  if (nthr <= 256)
    hipLaunchKernelGGL((k_tri_tile<S, TIO, M, T, 256, FUSED>), grid, block, sh, s, a, cfg);
  else if (nthr <= 512)
    hipLaunchKernelGGL((k_tri_tile<S, TIO, M, T, 512, FUSED>), grid, block, sh, s, a, cfg);
  else
    hipLaunchKernelGGL((k_tri_tile<S, TIO, M, T, 1024, FUSED>), grid, block, sh, s, a, cfg);
  if (hipGetLastError() != hipSuccess) return CRT_ERR_LAUNCH;
  note_kernel("k_tri_tile<%s,%s>%s M=%d T=%d lds=%zu", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", FUSED ? "" : cfg.flat ? " flat-flush" : " generic-flush", M, T, sh);  // (only a launch that succeeded is reported)
  return CRT_OK;
}

// ------------------------------------------------------------------------------------------
// Integrated outputs only (IntArgs, crt_internal.hpp): the same checkpointed sweep, but instead of staging and flushing
// profiles every level's net flux is reduced across the bands with ngroup wave shuffles.
template <class S, typename TIO, int M, bool PROF>
__device__ __forceinline__ void tri_int_body(const SolveArgs& a, const IntArgs& ia, int off_ck, int off_int, double* lds) {
  const int nb = a.nb, nz = a.nz, ng = ia.ngroup;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const int c = blockIdx.x;
  const double* rec = lds;
  double* ck = lds + off_ck + tid;  // [nck][2][nthr]
  const IntLds L = int_lds_carve(lds + off_int, nz, nwave, PROF);
  const bool active = tid < nb;
  const int b = active ? tid : 0;
  S st;
  st.template init<TIO>(rec, a, c, b);
  const double bc = st.band_const();
  const long long ib = (long long)c * a.col_stride + b;
  const double leaf_a = 1 - (ldio<TIO>(a.leaf_r, ib) + ldio<TIO>(a.leaf_t, ib));
  double w[INT_MAXG];
#pragma unroll
  for (int g = 0; g < INT_MAXG; ++g) w[g] = (g < ng && active) ? ia.band_w[(long long)g * nb + b] : 0.0;
#pragma unroll
  for (int g = 0; g < INT_MAXG; ++g)
    if (g < ng) {
      const double t = wave_sum_all(w[g] * leaf_a * bc);
      if (lane == 0) L.pdr[wave * INT_MAXG + g] = t;
      if constexpr (PROF) {
        const double t0 = wave_sum_all(w[g] * bc);
        if (lane == 0) L.pi0[wave * INT_MAXG + g] = t0;
      }
    }
  const int K = S::rows(nz);
  static_assert(S::RENORM == 0 || M % S::RENORM == 0, "checkpoints must fall on re-seeding levels");
  typename S::St fs;
  st.first(rec, nz, fs);
  {
    double e, f;
    st.pair(fs, e, f);
    ck[0] = e;
    ck[nthr] = f;
  }
  tri_forward<S, M>(st, rec, nz, fs, K - 1, [&](int level, const typename S::St& cs) {
    const int sidx = level / M;
    double e, f;
    st.pair(cs, e, f);
    ck[(2 * sidx) * nthr] = e;
    ck[(2 * sidx + 1) * nthr] = f;
  });
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    double be[M], bf[M];
    be[0] = ck[(2 * seg) * nthr];
    bf[0] = ck[(2 * seg + 1) * nthr];
    typename S::St rs;
    st.seed(rs, be[0], bf[0]);
#pragma unroll
    for (int i = 1; i < M; ++i) {
      be[i] = be[i - 1];
      bf[i] = bf[i - 1];
      if (k0 + i <= kend) tri_step(st, k0 + i - 1, rec, nz, rs, be[i], bf[i]);
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const int k = k0 + i;
      if (k <= kend) {
        double o[S::NST];
        if (k == K - 1)
          st.top(rec, nz, be[i], bf[i], o);
        else
          st.back(k, rec, nz, be[i], bf[i], o);
        if (k < nz) int_accumulate<false, PROF>(L, nwave, wave, lane, nz, k, ng, w, active, bc * rec[REC_HDR + k], o[0], o[1]);
      }
    }
  }
  int_finish<false, PROF>(L, ia, nwave, nz, c, rec[S_KB], rec[S_INVMU]);
}

template <class S, typename TIO, int M, int MAXT, bool PROF>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(M <= 8 ? 5 : 3))) void k_tri_int(SolveArgs a, IntArgs ia, int off_ck, int off_int) {
  extern __shared__ double lds[];
  {
    const double* src = a.ws + (long long)blockIdx.x * a.reclen;
    for (int i = threadIdx.x; i < a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  typedef typename UniformOf<S>::type SU;
  if constexpr (!std::is_same<S, SU>::value) {
    if (lds[S_UNIF] != 0.0) {
      tri_int_body<SU, TIO, M, PROF>(a, ia, off_ck, off_int, lds);
      return;
    }
  }
  tri_int_body<S, TIO, M, PROF>(a, ia, off_ck, off_int, lds);
}

template <class S, typename TIO, int M>
int launch_int_m(const SolveArgs& a, const IntArgs& ia, hipStream_t s, int nthr) {
  const int K = S::rows(a.nz);
  const int nck = (K - 1) / M + 1;
  const int off_ck = (a.reclen + 1) & ~1;
  const int off_int = off_ck + 2 * nck * nthr;
  const bool prof = ia.L_dr != nullptr;
  const size_t sh = ((size_t)off_int + int_lds_doubles(a.nz, nthr / 64, false, prof)) * sizeof(double);
  if (sh > MAX_WG_LDS) return CRT_ERR_UNSUPPORTED;
  auto go = [&](auto kern) {
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return (int)CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, a, ia, off_ck, off_int);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_tri_int<%s> M=%d%s", S::NAME, M, prof ? " + level profiles" : "");  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  if (prof) {
    if (nthr <= 256) return go(k_tri_int<S, TIO, M, 256, true>);
    if (nthr <= 512) return go(k_tri_int<S, TIO, M, 512, true>);
    return go(k_tri_int<S, TIO, M, 1024, true>);
  }
  if (nthr <= 256) return go(k_tri_int<S, TIO, M, 256, false>);
  if (nthr <= 512) return go(k_tri_int<S, TIO, M, 512, false>);
  return go(k_tri_int<S, TIO, M, 1024, false>);
}

template <class S, typename TIO>
int launch_int_scheme(const SolveArgs& a, const IntArgs& ia, hipStream_t s) {
  if (a.nb > 1024) return CRT_ERR_UNSUPPORTED;
  const int nthr = ((a.nb + 63) / 64) * 64;
  int st = launch_int_m<S, TIO, 8>(a, ia, s, nthr);  // small M: fewer registers, LDS is not the constraint here
  if (st == CRT_ERR_UNSUPPORTED) st = launch_int_m<S, TIO, 16>(a, ia, s, nthr);
  return st;
}

// crt_options.tune[8..11]: [8] force M (8/12/16), [9] force T (4/8), [10] kernel family, [11] store waves

// instantiated (M, T) pairs
template <class S, typename TIO, bool FUSED>
int launch_cfg(const SolveArgs& a, hipStream_t s, int M, int T, int nthr) {
  if (M == 8 && T == 4) return launch_mt<S, TIO, 8, 4, FUSED>(a, s, nthr);
  if (M == 8 && T == 8) return launch_mt<S, TIO, 8, 8, FUSED>(a, s, nthr);
  if (M == 12 && T == 4) return launch_mt<S, TIO, 12, 4, FUSED>(a, s, nthr);
  if (M == 12 && T == 12) return launch_mt<S, TIO, 12, 12, FUSED>(a, s, nthr);
  if (M == 16 && T == 4) return launch_mt<S, TIO, 16, 4, FUSED>(a, s, nthr);
  if (M == 16 && T == 8) return launch_mt<S, TIO, 16, 8, FUSED>(a, s, nthr);
  return CRT_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// Wave-specialised pipeline.  In k_tri_tile every wave of a workgroup alternates between the recurrence and the flush, and
// because all workgroups run the same schedule the whole chip alternates with them: HBM idles while the recurrences run
// and the ALUs idle while the store queues drain (measured: time = store time + compute time, for 1 or 2 workgroups
// per CU alike).  Here the workgroup carries extra "store waves" that do nothing but flush: the compute waves write
// tile g into one LDS buffer while the store waves stream tile g-1 out of the other, one LDS-only barrier per tile.
//   barrier #n : compute waves arrive after completing tile n-1, store waves before reading it; a compute wave writes
//   tile n+1 (same buffer as n-1) only after barrier #n+1, which the store waves reach after their reads of tile n-1.
struct PipeCfg {
  int ncomp;     // compute threads (multiple of 64); threads beyond are store threads
  int nck;
  int off_bc, off_ck, off_tile;  // LDS offsets in doubles; tile = [2][NST][T][nb]
  int flat;  // generic store role: fused flat flush (flush_flat) instead of the per-array generic flush; 3 = its whole-line form (flush_flat_wl)
  int off_park;  // flat == 3: the parked heads, park_doubles() behind the tiles
  int cpw;       // packed form (RS == 3): columns per workgroup, thread -> (column, band) = (tid / nb, tid % nb); tile = [NST][cpw][T][nb]
  // zq_pa with the interpolation fused into the store waves (tri_zqpa.hip): the caller's level count and output arrays
  int nz_out, off_halo;
  void* out[4];
};

template <class S, typename TIO, int M, int T, int RS, int NSTG = S::NST, bool PACK = false>
__device__ __forceinline__ void tri_pipe_compute(const SolveArgs& a, const PipeCfg& cfg, double* lds, const TriBand* band = nullptr) {
  const int nb = a.nb, nz = a.nz;
  const int tid = threadIdx.x, nthr = cfg.ncomp;
  // PACK (narrow spectra): the compute threads own cfg.cpw consecutive columns, thread -> (column of the pack, band); every lane reads its
  // own column's record, the tile keeps the columns' T x nb blocks one after the other inside each staged array
  const int cl = PACK ? tid / nb : 0;
  const bool active = PACK ? (cl < cfg.cpw && (int)blockIdx.x * cfg.cpw + cl < a.ncol) : tid < nb;
  const int b = active ? tid - cl * nb : 0;
  const int c = PACK ? (int)blockIdx.x * cfg.cpw + (active ? cl : 0) : (int)blockIdx.x;
  const double* rec = PACK ? lds + (active ? cl : 0) * a.reclen : lds;
  double* bandc = lds + cfg.off_bc + (PACK && active ? cl * nb : 0);
  double* ck = lds + cfg.off_ck + tid;  // [nck][2][ncomp]
  double* tile = lds + cfg.off_tile + (PACK && active ? cl * (T * nb) : 0);
  const int tstride = (PACK ? cfg.cpw : 1) * T * nb, bstride = NSTG * tstride;  // NSTG: staged arrays kept in the tile (the first NSTG of NST)
#ifdef CRT_STAMP
  // diagnostic build only (tools/stamp_timeline.py --tri): wall_clock64 stamps of compute wave 0, written over the column's own K0 record
  // in the workspace (already staged in LDS; K0 rewrites it) -- never into an output
  double* stamp = const_cast<double*>(a.ws) + (long long)c * a.reclen;
  int nstamp = 0;
#define TSTAMP() do { if (tid == 0 && nstamp < a.reclen) stamp[nstamp++] = (double)wall_clock64(); } while (0)
#else
#define TSTAMP() do {} while (0)
#endif
  TSTAMP();  // [0] record staged
  S st;
  if (band)  // requested by the kernel together with the record (zq_pa)
    st.init_band(rec, a, *band);
  else
    st.template init<TIO>(rec, a, c, b);
  if (active) bandc[b] = st.band_const();
  const int K = S::rows(nz);
  TSTAMP();  // [1] band set-up done
  // Forward sweep.  Checkpoints (every M levels) are kept for the segments 1 .. top-1 only: segment 0 restarts from first() and the TOP
  // segment -- the first one to be substituted back -- is left in registers by the sweep itself, so it is neither stored nor recomputed
  // (two checkpoint pairs per lane less = 10 KB of LDS at 300 bands; up to M - 1 level steps less before the first tile is handed over).
  const int seg_top = (K - 1) / M, k_top = seg_top * M;
  double be[M], bf[M];  // (e, f) of the levels of the segment being substituted back
  // (M a multiple of the scheme's re-seeding interval: every kernel family then repeats the same operations, tri_schemes.hpp.  zq_pa's
  //  M = 15 form is the exception: its segments restart from (e, 1, f) off the schedule -- equal to the other forms to rounding, not bitwise)
  typename S::St fs;
  st.first(rec, nz, fs);
  tri_forward<S, M>(st, rec, nz, fs, k_top, [&](int level, const typename S::St& cs) {
    if (level < k_top) {
      const int sidx = level / M - 1;  // checkpoint of segment sidx + 1
      double e, f;
      st.pair(cs, e, f);
      ck[(2 * sidx) * nthr] = e;
      ck[(2 * sidx + 1) * nthr] = f;
    }
  });
  st.pair(fs, be[0], bf[0]);
  if constexpr (S::RENORM > 0 && M % (S::RENORM > 0 ? S::RENORM : 1) != 0) st.seed(fs, be[0], bf[0]);  // off-schedule segment start
#pragma unroll
  for (int i = 1; i < M; ++i) {
    be[i] = be[i - 1];
    bf[i] = bf[i - 1];
    if (k_top + i <= K - 1) tri_step(st, k_top + i - 1, rec, nz, fs, be[i], bf[i]);
  }
  TSTAMP();  // [2] forward sweep done (top segment in registers)
  int buf = 0;
  for (int seg = seg_top; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    TSTAMP();  // per segment: start
    if (seg != seg_top) {
      typename S::St rs;
      if (seg == 0) {
        st.first(rec, nz, rs);
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
        if (k0 + i <= kend) tri_step(st, k0 + i - 1, rec, nz, rs, be[i], bf[i]);
      }
    }
    TSTAMP();  // per segment: pairs recomputed
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const int k = k0 + i;
      if (k <= kend) {
        double o[S::NST];
        if (k == K - 1)
          st.top(rec, nz, be[i], bf[i], o);
        else
          st.back(k, rec, nz, be[i], bf[i], o);
        if (active) {
#pragma unroll
          for (int q = 0; q < NSTG; ++q) tile[buf * bstride + q * tstride + (i % T) * nb + b] = o[q];
        }
        if (i % T == 0) {  // tile complete: hand it to the store waves
          TSTAMP();  // per tile: substituted back
          lds_barrier();
          if constexpr (RS > 0)
            lds_barrier();  // single buffer: wait until the store waves hold the tile in registers
          else
            buf ^= 1;
          TSTAMP();  // per tile: handed over
        }
      }
    }
  }
}

// RS > 0: ONE LDS tile buffer; the store waves pull the finished tile into registers (RS band pairs x NST staged values per
// thread), release the buffer with a second barrier and only then issue the stores, so the registers of the store waves
// play the part of the second buffer.  Halves the tile's LDS -> two workgroups per CU, whose forward sweeps (no stores yet)
// hide behind each other's flushes.
template <class S, typename TIO, int M, int T, int RS>
__device__ __forceinline__ void tri_pipe_store_rs(const SolveArgs& a, const PipeCfg& cfg, double* lds) {
  const int nb2 = a.nb >> 1, nz = a.nz;
  const int c = blockIdx.x;
  const int sid = threadIdx.x - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
  const double* rec = lds;
  const d2* bandc2 = reinterpret_cast<const d2*>(lds + cfg.off_bc);
  const d2* tb = reinterpret_cast<const d2*>(lds + cfg.off_tile);
  const double invmu = rec[S_INVMU];
  const int K = S::rows(nz);
  const int dt = nst / nb2, dp = nst - dt * nb2;
  const int t0 = sid / nb2, p0 = sid - t0 * nb2;
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    for (int i = M - T; i >= 0; i -= T) {
      const int k = k0 + i;
      if (k > kend) continue;
      const int jmax = min(kend, nz - 1);
      d2 v[RS][S::NST];
      int tt[RS], pp[RS];
      lds_barrier();  // tile complete
      {
        int t = t0, p = p0;
#pragma unroll
        for (int it = 0; it < RS; ++it) {
          tt[it] = t;
          pp[it] = p;
          if (t < T) {
#pragma unroll
            for (int q = 0; q < S::NST; ++q) v[it][q] = tb[(q * T + t) * nb2 + p];
          }
          p += dp;
          t += dt;
          if (p >= nb2) {
            p -= nb2;
            ++t;
          }
        }
      }
      lds_barrier();  // values are in registers (lgkmcnt(0) inside): the compute waves may overwrite the tile
#pragma unroll
      for (int it = 0; it < RS; ++it) {
        __builtin_amdgcn_sched_barrier(0);  // one pair at a time: keeps the emit temporaries of the RS pairs from being live together
        const int j = k + tt[it];
        if (tt[it] < T && j <= jmax) {
          d2 o[S::NOUT];
          S::emit(rec, nz, j, bandc2[pp[it]], invmu, v[it], o);
#pragma unroll
          for (int r = 0; r < S::NOUT; ++r) {
            const int rows = S::out_rows(r, nz);
            if (j < rows) {
              typedef TIO vt __attribute__((ext_vector_type(2)));
              vt w;
              w.x = (TIO)o[r].x;
              w.y = (TIO)o[r].y;
              // wave-uniform column base (scalar registers) + one 32-bit lane offset shared by all arrays
              vt* colbase = reinterpret_cast<vt*>(a.o[r]) + (long long)c * rows * nb2;
              store_rows<S, TIO>(colbase + (unsigned)(j * nb2 + pp[it]), w);
            }
          }
        }
      }
    }
  }
}

// Packed form of the two-pair register-staged store role (RS == 3; tri_pipe_compute<..., PACK>): the tile holds cfg.cpw columns,
// [NST][cpw][T][nb]; a store thread's two pairs are fixed (column, row, band pair) triples for the whole kernel; every column has its
// own record and its own base address in the outputs.  cpw * nb <= 64 and T = 4, so the tile's cpw * T * nb / 2 <= 128 pairs are two
// per thread of ONE store wave.
template <class S, typename TIO, int M, int T>
__device__ __forceinline__ void tri_pack_store(const SolveArgs& a, const PipeCfg& cfg, double* lds) {
  constexpr int RS = 2;
  const int nb2 = a.nb >> 1, nz = a.nz, cpw = cfg.cpw;
  const int c0 = blockIdx.x * cpw;
  const int ncol_here = min(cpw, a.ncol - c0);
  const int sid = threadIdx.x - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
  const d2* bandc2 = reinterpret_cast<const d2*>(lds + cfg.off_bc);
  const d2* tb = reinterpret_cast<const d2*>(lds + cfg.off_tile);
  const int K = S::rows(nz);
  const int per = T * nb2, qstride = cpw * per;  // pairs of one column's tile / between staged arrays
  int ff[RS], cc[RS], tt[RS], pp[RS];
#pragma unroll
  for (int it = 0; it < RS; ++it) {
    const int f = sid + it * nst;
    const int c = f / per, r = f - c * per, t = r / nb2;
    ff[it] = f;
    cc[it] = c < ncol_here ? c : -1;
    tt[it] = t;
    pp[it] = r - t * nb2;
  }
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    for (int i = M - T; i >= 0; i -= T) {
      const int k = k0 + i;
      if (k > kend) continue;
      const int jmax = min(kend, nz - 1);
      d2 v[RS][S::NST];
      lds_barrier();  // tile complete
#pragma unroll
      for (int it = 0; it < RS; ++it)
        if (cc[it] >= 0) {
#pragma unroll
          for (int q = 0; q < S::NST; ++q) v[it][q] = tb[q * qstride + ff[it]];
        }
      lds_barrier();  // values are in registers: the compute wave may overwrite the tile
#pragma unroll
      for (int it = 0; it < RS; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const int j = k + tt[it];
        if (cc[it] >= 0 && j <= jmax) {
          const double* rec = lds + cc[it] * a.reclen;
          d2 o[S::NOUT];
          S::emit(rec, nz, j, bandc2[cc[it] * nb2 + pp[it]], rec[S_INVMU], v[it], o);
#pragma unroll
          for (int r = 0; r < S::NOUT; ++r) {
            const int rows = S::out_rows(r, nz);
            if (j < rows) {
              typedef TIO vt __attribute__((ext_vector_type(2)));
              vt w;
              w.x = (TIO)o[r].x;
              w.y = (TIO)o[r].y;
              store_rows<S, TIO>(reinterpret_cast<vt*>(a.o[r]) + ((long long)(c0 + cc[it]) * rows + j) * nb2 + pp[it], w);
            }
          }
        }
      }
    }
  }
}

// any nb / alignment: the generic flat flush of k_tri_tile, run by the store waves on the double-buffered tile
template <class S, typename TIO, int M, int T>
__device__ __forceinline__ void tri_pipe_store_generic(const SolveArgs& a, const PipeCfg& cfg, double* lds) {
  const int nb = a.nb, nz = a.nz;
  const int c = blockIdx.x;
  const int sid = threadIdx.x - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
  const double* rec = lds;
  const double* bandc = lds + cfg.off_bc;
  const double* tile = lds + cfg.off_tile;
  const int tstride = T * nb, bstride = S::NST * tstride;
  const double invmu = rec[S_INVMU];
  const float inv_nb = 1.0f / (float)nb;
  const int K = S::rows(nz);
  int buf = 0;
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    for (int i = M - T; i >= 0; i -= T) {
      const int k = k0 + i;
      if (k > kend) continue;
      lds_barrier();  // tile `buf` is complete
      if (cfg.flat == 3)
        flush_flat_wl<S, TIO>(a, rec, bandc, tile + buf * bstride, tstride, c, k, min(T, kend - k + 1), invmu, sid, nst, lds + cfg.off_park, buf);
      else if (cfg.flat)
        flush_flat<S, TIO>(a, rec, bandc, tile + buf * bstride, tstride, c, k, min(T, kend - k + 1), invmu, sid, nst, cfg.flat == 2);
      else
        flush_arrays<S, TIO, 0>(a, rec, bandc, tile + buf * bstride, tstride, c, k, min(T, kend - k + 1), invmu, inv_nb, sid, nst);
      buf ^= 1;
    }
  }
}

template <class S, typename TIO, int M, int T>
__device__ __forceinline__ void tri_pipe_store(const SolveArgs& a, const PipeCfg& cfg, double* lds) {
  const int nb2 = a.nb >> 1, nz = a.nz;
  const int c = blockIdx.x;
  const int sid = threadIdx.x - cfg.ncomp, nst = blockDim.x - cfg.ncomp;
  const double* rec = lds;
  const d2* bandc2 = reinterpret_cast<const d2*>(lds + cfg.off_bc);
  const d2* tile2 = reinterpret_cast<const d2*>(lds + cfg.off_tile);
  const double invmu = rec[S_INVMU];
  const int K = S::rows(nz);
  // thread -> (row, band pair) walk over the T x nb2 pairs of a tile: consecutive threads, consecutive 16-B words
  const int dt = nst / nb2, dp = nst - dt * nb2;
  const int t0 = sid / nb2, p0 = sid - t0 * nb2;
  int buf = 0;
  for (int seg = (K - 1) / M; seg >= 0; --seg) {
    const int k0 = seg * M;
    const int kend = min(k0 + M - 1, K - 1);
    for (int i = M - T; i >= 0; i -= T) {
      const int k = k0 + i;
      if (k > kend) continue;
      lds_barrier();  // tile `buf` is complete
      const d2* tb = tile2 + (size_t)buf * (S::NST * T * nb2);
      const int jmax = min(kend, nz - 1);
      int t = t0, p = p0;
      while (t < T) {
        const int j = k + t;
        if (j <= jmax) {
          d2 st[S::NST], o[S::NOUT];
#pragma unroll
          for (int q = 0; q < S::NST; ++q) st[q] = tb[(q * T + t) * nb2 + p];
          S::emit(rec, nz, j, bandc2[p], invmu, st, o);
#pragma unroll
          for (int r = 0; r < S::NOUT; ++r) {
            const int rows = S::out_rows(r, nz);
            if (j < rows) {
              typedef TIO vt __attribute__((ext_vector_type(2)));
              vt v;
              v.x = (TIO)o[r].x;
              v.y = (TIO)o[r].y;
              vt* colbase = reinterpret_cast<vt*>(a.o[r]) + (long long)c * rows * nb2;
              store_rows<S, TIO>(colbase + (unsigned)(j * nb2 + p), v);
            }
          }
        }
        p += dp;
        t += dt;
        if (p >= nb2) {
          p -= nb2;
          ++t;
        }
      }
      buf ^= 1;
    }
  }
}

// amdgpu_waves_per_eu(4): two 8-wave workgroups per CU need four waves per SIMD, i.e. <= 128 registers.  Left to itself the
// register-staged M = 16 form (levels above ~80) takes 130 and only ONE workgroup stays resident: zq at 6000 x 300 x 100
// 1.74 ms (0.73 of the peak) -> 1.55 ms (0.82) with the cap (12 bytes of scratch per lane).  profiles/r02/kernel_resources.txt lists
// every kernel's registers / occupancy (hipcc -Rpass-analysis=kernel-resource-usage).
// RS = 2 is the narrow-spectrum form (one compute + one store wave per column, two band pairs staged per store thread): five waves per
// SIMD, i.e. ten of these two-wave workgroups per CU instead of eight.
template <class S, typename TIO, int M, int T, int MAXT, int RS>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu((RS == 2 || RS == 3) ? 5 : 4))) void k_tri_pipe(SolveArgs a, PipeCfg cfg) {
  static_assert(M % T == 0, "tile height must divide the checkpoint spacing");
  extern __shared__ double lds[];
  if constexpr (RS == 3) {  // packed: the records of the workgroup's consecutive columns are consecutive in the workspace
    const int c0 = blockIdx.x * cfg.cpw;
    const int nhere = min(cfg.cpw, a.ncol - c0);
    const double* src = a.ws + (long long)c0 * a.reclen;
    for (int i = threadIdx.x; i < nhere * a.reclen; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    if ((int)threadIdx.x >= cfg.ncomp) {
      tri_pack_store<S, TIO, M, T>(a, cfg, lds);
      return;
    }
    typedef typename UniformOf<S>::type SU;
    if constexpr (!std::is_same<S, SU>::value) {
      // the uniform-column object only when EVERY column of the pack is uniform (the barriers of the compute role must not diverge); a
      // mixed pack runs the general object on all of its columns -- K0 fills the per-level vectors of uniform columns too (colpre.hip)
      bool all_unif = true;
      for (int i = 0; i < nhere; ++i) all_unif = all_unif && lds[i * a.reclen + S_UNIF] != 0.0;
      if (all_unif) {
        tri_pipe_compute<SU, TIO, M, T, 2, S::NST, true>(a, cfg, lds);
        return;
      }
    }
    tri_pipe_compute<S, TIO, M, T, 2, S::NST, true>(a, cfg, lds);
    return;
  }
  {
    const double* src = a.ws + (long long)blockIdx.x * a.reclen;
    for (int i = threadIdx.x; i < a.reclen; i += blockDim.x) lds[i] = src[i];
  }
  __syncthreads();
  if ((int)threadIdx.x >= cfg.ncomp) {
    if constexpr (RS > 0)
      tri_pipe_store_rs<S, TIO, M, T, RS>(a, cfg, lds);
    else if constexpr (RS < 0)
      tri_pipe_store_generic<S, TIO, M, T>(a, cfg, lds);
    else
      tri_pipe_store<S, TIO, M, T>(a, cfg, lds);
    return;
  }
  typedef typename UniformOf<S>::type SU;
  if constexpr (!std::is_same<S, SU>::value) {
    if (lds[S_UNIF] != 0.0) {
      tri_pipe_compute<SU, TIO, M, T, RS>(a, cfg, lds);
      return;
    }
  }
  tri_pipe_compute<S, TIO, M, T, RS>(a, cfg, lds);
}

// returns CRT_ERR_UNSUPPORTED when the shape does not fit (caller falls back to k_tri_tile)
constexpr int PIPE_RS = 4;  // band pairs a store thread holds in the register-staged variant

template <class S, typename TIO, int M, int T>
int launch_pipe_generic(const SolveArgs& a, hipStream_t s, int nstore_waves) {
  const int ncomp = ((a.nb + 63) / 64) * 64;
  const int nthr = ncomp + 64 * nstore_waves;
  if (nthr > 512) return CRT_ERR_UNSUPPORTED;  // instantiated for narrow spectra only (the odd-nb case that matters: nb = 107)
  const int K = S::rows(a.nz);
  PipeCfg cfg{};
  cfg.ncomp = ncomp;
  cfg.nck = std::max((K - 1) / M - 1, 0);  // checkpoints kept: segments 1 .. top-1 (see tri_pipe_compute)
  cfg.off_bc = (a.reclen + 1) & ~1;
  cfg.off_ck = cfg.off_bc + ((a.nb + 1) & ~1);
  cfg.off_tile = cfg.off_ck + 2 * cfg.nck * ncomp;
  cfg.flat = a.tune[13] != 1 ? flat_flush_ok<S, TIO>(a) : 0;
  cfg.off_park = cfg.off_tile + 2 * S::NST * T * a.nb;
  // Whole lines only -- for schemes whose arrays all have nz rows (zq).  Measured at 3e4 x 107 x 60 (tools/ragged_sweep.py --tune=13:2 /
  // 13:3, same process): zq 1.797 -> 1.709 ms (0.759 -> 0.798 of the peak), uniform and ragged alike; n79, whose two layer arrays need a
  // second pass per tile, 1.530 -> 1.507 ms on equal-dLAI columns but 1.60 -> 1.72 ms on ragged ones (stamps: the store role then takes
  // 3 us per tile against 2 us of arithmetic, profiles/r03/nb107/): n79 keeps the part-line form.  tune key 13: 2 / 3 force either form.
  const bool wl_default = flush_classes<S>() == 1;
  if (cfg.flat == 2 && a.nb >= 128 / (int)sizeof(TIO) && (a.tune[13] == 3 || (a.tune[13] == 0 && wl_default))) cfg.flat = 3;
  const size_t sh = ((size_t)cfg.off_park + (cfg.flat == 3 ? park_doubles<S, TIO>() : 0)) * sizeof(double);
  if (sh > MAX_WG_LDS) return CRT_ERR_UNSUPPORTED;
  auto kern = k_tri_pipe<S, TIO, M, T, 512, -1>;
  if (sh > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
    return CRT_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, a, cfg);
  if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
  note_kernel("k_tri_pipe<%s,%s> %s M=%d T=%d store_waves=%d lds=%zu", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32", cfg.flat == 3 ? "whole-line flat-flush" : cfg.flat ? "flat-flush" : "generic-flush", M, T,
              nstore_waves, sh);  // (only a launch that succeeded is reported)
  return (int)CRT_OK;
}

// narrow spectra, several columns per workgroup (one compute wave + one store wave)
template <class S, typename TIO>
int launch_tri_pack(const SolveArgs& a, hipStream_t s) {
  constexpr int M = 8, T = 4;
  {
    // up to 32 bands: 64 / nb columns on one compute wave; 33 .. 42 bands (the 36-38-band shards of an 8-rank band partition): three
    // columns on two compute waves (a column may straddle the waves) -- beyond that a pack fills no more lanes than a column by itself
    const int ncw = a.tune[6] > 0 ? a.tune[6] : (a.nb <= 32 ? 1 : 2);
    if (a.nb % 2 || a.nb < 2 || 64 * ncw / a.nb < 2 || (a.nb > 42 && a.tune[6] == 0)) return CRT_ERR_UNSUPPORTED;
    for (int i = 0; i < S::NOUT; ++i)
      if (reinterpret_cast<uintptr_t>(a.o[i]) & (2 * sizeof(TIO) - 1)) return CRT_ERR_UNSUPPORTED;
    const int K = S::rows(a.nz);
    PipeCfg cfg{};
    cfg.ncomp = 64 * ncw;
    cfg.cpw = cfg.ncomp / a.nb;
    const int nsw = (cfg.cpw * T * (a.nb / 2) + 127) / 128;  // two pairs per store thread
    if (cfg.ncomp + 64 * nsw > 512) return CRT_ERR_UNSUPPORTED;
    cfg.nck = std::max((K - 1) / M - 1, 0);
    cfg.off_bc = (cfg.cpw * a.reclen + 1) & ~1;
    cfg.off_ck = cfg.off_bc + cfg.ncomp;
    cfg.off_tile = cfg.off_ck + 2 * cfg.nck * cfg.ncomp;
    const size_t sh = ((size_t)cfg.off_tile + (size_t)S::NST * cfg.cpw * T * a.nb) * sizeof(double);
    if (sh > MAX_WG_LDS / 2) return CRT_ERR_UNSUPPORTED;
    // 33 .. 42 bands, measured (tools/ragged_sweep.py --tune=5:1 / 5:0, profiles/r03/narrow/tri_pack_38_bands_ab.txt): n79 1.5e5 x 38 x 60
    // 0.695 -> 0.775 (ragged 0.63 -> 0.72), but 1e5 x 38 x 100 0.71 -> 0.71 / 0.60 -> 0.57 (55 KB of LDS per pack); zq 1e5 x 38 x 100 0.80 -> 0.83,
    // 1.5e5 x 38 x 60 0.78 -> 0.78, 36 bands 0.83 -> 0.81: taken for n79 while the pack stays below 40 KB, not for zq
    if (a.nb > 32 && a.tune[6] == 0 && (std::is_same<S, typename UniformOf<S>::type>::value || sh > 40 * 1024)) return CRT_ERR_UNSUPPORTED;
    auto kern = k_tri_pipe<S, TIO, M, T, 512, 3>;
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.ncol + cfg.cpw - 1) / cfg.cpw), dim3(cfg.ncomp + 64 * nsw), sh, s, a, cfg);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_tri_pipe<%s,%s> packed columns=%d compute_waves=%d register-staged(2 pairs) M=%d T=%d store_waves=%d lds=%zu", S::NAME,
                sizeof(TIO) == 8 ? "f64" : "f32", cfg.cpw, ncw, M, T, nsw, sh);
    return (int)CRT_OK;
  }
}

template <class S, typename TIO, int M, int T>
int launch_pipe_mt(const SolveArgs& a, hipStream_t s, int nstore_waves, bool regstage) {
  const int ncomp = ((a.nb + 63) / 64) * 64;
  const int nthr = ncomp + 64 * nstore_waves;
  if (nthr > 1024) return CRT_ERR_UNSUPPORTED;
  if (regstage && T * (a.nb / 2) > PIPE_RS * 64 * nstore_waves) return CRT_ERR_UNSUPPORTED;
  const int K = S::rows(a.nz);
  PipeCfg cfg{};
  cfg.ncomp = ncomp;
  cfg.nck = std::max((K - 1) / M - 1, 0);  // checkpoints kept: segments 1 .. top-1 (see tri_pipe_compute)
  cfg.off_bc = (a.reclen + 1) & ~1;
  cfg.off_ck = cfg.off_bc + ((a.nb + 1) & ~1);
  cfg.off_tile = cfg.off_ck + 2 * cfg.nck * ncomp;
  const size_t sh = ((size_t)cfg.off_tile + (size_t)(regstage ? 1 : 2) * S::NST * T * a.nb) * sizeof(double);
  if (sh > (regstage ? MAX_WG_LDS / 2 : MAX_WG_LDS)) return CRT_ERR_UNSUPPORTED;  // register staging only pays with 2 WG/CU
  // (M = 8 only: at M = 12 the n79 compute role does not fit the 96 registers of five waves per SIMD and spills -- 3.45 -> 4.51 ms)
  const bool narrow_rs = T == 4 && M == 8 && regstage && nthr <= 512 && T * (a.nb / 2) <= 2 * 64 * nstore_waves && a.tune[2] != 16;
  auto go = [&](auto kern) {
    if (sh > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
      return (int)CRT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.ncol), dim3(nthr), sh, s, a, cfg);
    if (hipGetLastError() != hipSuccess) return (int)CRT_ERR_LAUNCH;
    note_kernel("k_tri_pipe<%s,%s> %s M=%d T=%d store_waves=%d lds=%zu", S::NAME, sizeof(TIO) == 8 ? "f64" : "f32",
                regstage ? (narrow_rs ? "register-staged(2 pairs)" : "register-staged") : "double-buffered", M, T, nstore_waves, sh);  // (only a launch that succeeded is reported)
    return (int)CRT_OK;
  };
  if constexpr (T == 4 && M == 8) {  // narrow spectra: two staged pairs per store thread cover the tile
    if (narrow_rs) return go(k_tri_pipe<S, TIO, M, T, 512, 2>);  // (crt_options.tune[2] = 16 keeps the four-pair form: A/B)
  }
  if (regstage) return nthr <= 512 ? go(k_tri_pipe<S, TIO, M, T, 512, PIPE_RS>) : go(k_tri_pipe<S, TIO, M, T, 1024, PIPE_RS>);
  return nthr <= 512 ? go(k_tri_pipe<S, TIO, M, T, 512, 0>) : go(k_tri_pipe<S, TIO, M, T, 1024, 0>);
}

template <class S, typename TIO>
int launch_pipe(const SolveArgs& a, hipStream_t s, int M, int T, int nsw, bool regstage) {
  if (M == 8 && T == 4) return launch_pipe_mt<S, TIO, 8, 4>(a, s, nsw, regstage);
  if (M == 12 && T == 4) return launch_pipe_mt<S, TIO, 12, 4>(a, s, nsw, regstage);
  if (M == 16 && T == 4) return launch_pipe_mt<S, TIO, 16, 4>(a, s, nsw, regstage);
  if (M == 16 && T == 8) return launch_pipe_mt<S, TIO, 16, 8>(a, s, nsw, regstage);
  return CRT_ERR_UNSUPPORTED;
}

// min_nb: narrow spectra (the 36-38-band shards of an 8-rank band partition) run one compute wave + ONE store wave per column:
// tools/ab_narrow.py, zq 1e5 x 38 x 100: k_tri_wave 7.54 ms (2.8 TB/s), pipeline with 3 / 2 / 1 store waves 4.5 / 4.2 / 3.74 ms
// (5.7 TB/s); n79 1e5 x 38 x 60: 4.38 -> 2.13 ms.  At 10-15 bands the all-waves tile kernel (one wave per column, no store wave) is the
// best of the three (n79 3e5 x 14 x 60: k_tri_wave 6.73, pipeline 4.29, k_tri_tile 3.88 ms; zq 4e5 x 12 x 60: 6.75 / 6.18 / 5.41);
// below 10 the per-wave kernel, which packs several columns into a wave, wins again (zq 4e5 x 8 x 60: 3.20 vs 4.81 ms).
template <class S, typename TIO>
int launch_scheme(const SolveArgs& a, hipStream_t s, bool& done, int min_nb = 10) {
  done = false;
  const int* g_tri_tune = a.tune + 8;  // this call's overrides (crt_options.tune[8..11])
  if (a.tune[12] > 0) min_nb = a.tune[12];
  if (a.nb <= 64 && a.tune[5] != 1 && g_tri_tune[0] == 0 && g_tri_tune[2] == 0) {  // several columns per compute wave (tune key 5 = 1: off)
    const int st = launch_tri_pack<S, TIO>(a, s);
    if (st != CRT_ERR_UNSUPPORTED) {
      done = st == CRT_OK;
      return st;
    }
  }
  if (a.nb < min_nb || a.nb > 1024) return CRT_OK;  // narrow spectra: the per-wave kernels fill their lanes better
  const int nthr = ((a.nb + 63) / 64) * 64;
  const int K = S::rows(a.nz);
  auto lds_bytes = [&](int M, int T) {
    const int nck = (K - 1) / M + 1;
    return ((size_t)a.reclen + a.nb + 4 + 2 * (size_t)nck * nthr + (size_t)S::NST * T * a.nb) * sizeof(double);
  };
  // Measured on MI355X at 1e4 x 300 (tools/ab_tri.py, interleaved rounds, fused flush):
  //   n79 nz=60 : M12/T4 (2 WG/CU) 1.84 ms | M8/T8 2.25 | M16/T4 2.37 | M8/T4 2.28 | per-wave kernel 2.79
  //   zq  nz=60 : M8/T8 (1 WG/CU) 1.91 ms | M16/T8 1.92 | M12/T4 2.02 | per-wave 2.47
  //   zq  nz=100: M8/T8 3.11 ms | M16/T8 3.16 | M8/T4 3.18 | M12/T4 3.20 | per-wave 5.15
  int M = 0, T = 0;
  const int pref_n79[6][2] = {{12, 4}, {8, 8}, {16, 8}, {8, 4}, {16, 4}, {12, 12}};
  const int pref_zq[6][2] = {{8, 8}, {16, 8}, {12, 4}, {8, 4}, {16, 4}, {12, 12}};
  const int (*pref)[2] = S::NOUT == 6 ? pref_n79 : pref_zq;
  for (int i = 0; i < 6 && !M; ++i) {
    const size_t need = lds_bytes(pref[i][0], pref[i][1]);
    // a (M, 4) choice is only worth it when it leaves two workgroups per CU
    const size_t budget = pref[i][1] == 4 && i == 0 ? 78 * 1024 : MAX_WG_LDS;
    if (need <= budget) { M = pref[i][0]; T = pref[i][1]; }
  }
  if (g_tri_tune[0] > 0) {
    M = g_tri_tune[0];
    T = g_tri_tune[1] > 0 ? g_tri_tune[1] : 4;
    if (lds_bytes(M, T) > MAX_WG_LDS) return CRT_OK;
  }
  if (!M) return CRT_OK;
  // fused flush needs even nb and 16-B aligned output arrays
  bool fused = (a.nb % 2 == 0);
  for (int i = 0; i < S::NOUT && fused; ++i)
    if (reinterpret_cast<uintptr_t>(a.o[i]) & (2 * sizeof(TIO) - 1)) fused = false;
  // odd nb / unaligned outputs: pipeline with the generic flush.  The per-element flush is too much work for a few store
  // waves once the spectrum is wide (tools/ab_tri_odd.py, k_tri_tile -> pipeline with 2 store waves: nb=107 n79 1.81 -> 1.90 ms,
  // zq 2.35 -> 2.29, zq nz=100 3.00 -> 2.22; nb=255 n79 1.60 -> 2.39, zq 2.25 -> 2.86), so only zq on narrow spectra takes it
  // (tune key 10 = 4 forces it for any scheme and nb).
  // With the fused flat flush (round 2; tools/ab_flat.py, 3e4 x 107 x 60: zq 2.25 -> 1.80 ms, n79 1.77 -> 1.61 in k_tri_tile, 1.57 in
  // the pipeline) the narrow pipeline pays for n79 as well.
  const bool flat_ok = a.tune[13] != 1 && flat_flush_ok<S, TIO>(a);
  if (!fused && g_tri_tune[2] != 1 && g_tri_tune[0] == 0 && ((nthr <= 128 && a.nb >= 16 && (S::NOUT == 7 || flat_ok)) || g_tri_tune[2] == 4)) {
    const int nsw = g_tri_tune[3] > 0 ? g_tri_tune[3] : (nthr == 64 ? 1 : 2);
    int st = launch_pipe_generic<S, TIO, 12, 4>(a, s, nsw);
    if (st == CRT_ERR_UNSUPPORTED) st = launch_pipe_generic<S, TIO, 16, 4>(a, s, nsw);
    if (st != CRT_ERR_UNSUPPORTED) {
      done = st == CRT_OK;
      return st;
    }
  }
  if (fused && g_tri_tune[2] != 1 && (a.nb >= 16 || g_tri_tune[2] >= 2)) {  // wave-specialised pipeline first (tune key 10 = 1 disables, key 11 = store waves)
    // Measured on MI355X at 1e4 x 300 (tools/ab_tri.py, profiles/r01/ab_tri_pipe_*.txt; fill probe 6.3-6.8 TB/s):
    //                best k_tri_tile -> double-buffer pipeline (1 WG/CU) -> register-staged pipeline (2 WG/CU)
    //   n79 nz=60 :  1.649 ms        -> 1.522 (M12/T4, 4 store waves)    -> 1.384 (M12/T4, 3 store waves) = 0.93 of the fill rate
    //   zq  nz=60 :  1.852           -> 1.771 (M8/T4, 3)                 -> 1.607 (M12/T4, 3)             = 0.95
    //   zq  nz=100:  3.115           -> 2.902 (M16/T4, 4)                -> 2.857 (M16/T4, 3)             = 0.93
    int nsw = g_tri_tune[3] > 0 ? g_tri_tune[3] : 4;
    if (nthr + 64 * nsw > 1024) nsw = (1024 - nthr) / 64;
    int st = CRT_ERR_UNSUPPORTED;
    if (nsw >= 1) {
      // tune key 10: 0 = automatic, 1 = no pipeline, 2 = double-buffer pipeline only, 3 = register-staged only
      const bool try_rs = g_tri_tune[2] != 2, try_db = g_tri_tune[2] != 3;
      // register-staged: 5 compute + 3 store waves = 8 waves per workgroup, two workgroups fill the 16 wave slots of a CU at <= 128 VGPRs
      const int nsw_rs = g_tri_tune[3] > 0 ? nsw : (nthr == 64 ? 1 : min(nsw, 3));
      if (g_tri_tune[0] > 0) {
        if (try_rs) st = launch_pipe<S, TIO>(a, s, M, T, nsw_rs, true);
        if (st == CRT_ERR_UNSUPPORTED && try_db) st = launch_pipe<S, TIO>(a, s, M, T, nsw, false);
      } else {
        const int pp_n79[3][2] = {{12, 4}, {16, 4}, {8, 4}};
        const int pp_zq[3][2] = {{8, 4}, {12, 4}, {16, 4}};
        // narrow spectra (one compute wave): M = 8 has the two-pair store role and five or six waves per SIMD (tools/ab_narrow_rs.py:
        // n79 1.5e5 x 38 x 60 3.14 -> 3.04 ms, 2e5 x 16 x 60 2.82 -> 2.61; zq 3.37 -> 3.11, 3.22 -> 2.92); above ~80 levels its
        // checkpoints cost n79 the occupancy again (1e5 x 38 x 100: 3.43 with M = 16 against 3.50)
        const int (*pp)[2] = (S::NOUT == 6 && !(nthr == 64 && a.nz <= 80)) ? pp_n79 : pp_zq;
        for (int i = 0; i < 3 && st == CRT_ERR_UNSUPPORTED && try_rs; ++i) st = launch_pipe<S, TIO>(a, s, pp[i][0], pp[i][1], nsw_rs, true);
        for (int i = 0; i < 3 && st == CRT_ERR_UNSUPPORTED && try_db; ++i) st = launch_pipe<S, TIO>(a, s, pp[i][0], pp[i][1], nsw, false);
      }
    }
    if (st != CRT_ERR_UNSUPPORTED) {
      done = st == CRT_OK;
      return st;
    }
  }
  const int st = fused ? launch_cfg<S, TIO, true>(a, s, M, T, nthr) : launch_cfg<S, TIO, false>(a, s, M, T, nthr);
  if (st == CRT_ERR_UNSUPPORTED) return CRT_OK;
  done = st == CRT_OK;
  return st;
}

}  // namespace
}  // namespace crt
