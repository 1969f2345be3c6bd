// Scheme objects of the tridiagonal solvers (n79, zq), shared by the column-tile and pipeline kernels (tri_tile_impl.hpp) and the
// per-wave fallback kernels (solve_tridiag.hip) so that both produce identical bits.  Not part of the ABI.
#pragma once
#include "crt_internal.hpp"

namespace crt {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

// the five spectra of one (column, band); the pipeline kernels request them together with the column record, before the barrier
// that publishes the record (one round trip instead of two ahead of the forward sweep)
struct TriBand {
  double I_dr0, I_df0, leaf_r, leaf_t, soil_r;
};
template <typename TIO>
__device__ inline TriBand load_tri_band(const SolveArgs& a, int c, int b) {
  const long long i = (long long)c * a.col_stride + b;
  TriBand in;
  in.I_dr0 = ldio<TIO>(a.I_dr0, i);
  in.I_df0 = ldio<TIO>(a.I_df0, i);
  in.leaf_r = ldio<TIO>(a.leaf_r, i);
  in.leaf_t = ldio<TIO>(a.leaf_t, i);
  in.soil_r = ldio<TIO>(a.soil_r, i);
  return in;
}

// ------------------------------------------------------------------------------------------
// The forward recurrence of both schemes maps the even-row pair (e, f) of level k to that of level k + 1 by a LINEAR FRACTIONAL
// transformation: e' = (a e + b) / (c e + d), f' = (alpha f + beta(e)) / (c e + d).  Written for e = p / q, f = g / q it is a linear
// recurrence in (p, q, g) WITHOUT a division: the serial dependency chain of a step is 3-5 multiply-adds instead of ~12 operations
// with a reciprocal in the middle (the sweeps are latency-bound: zq_pa spent a third of a workgroup's life in its forward sweep,
// DESIGN 3.7).  A scheme therefore carries a state St through its sweep:
//   first(rec, nz, St&)            state of level 0
//   advance(k, rec, nz, St&)       level k -> level k + 1
//   pair(St, e, f)                 the normalised pair (e, f) = (p / q, g / q) of the state's level -- one reciprocal, OFF the chain
//   seed(St&, e, f)                state (e, 1, f)
// and the chain is re-seeded from its own normalised pair every RENORM-th level (tri_advance / tri_step below): (p, q, g) shrink or grow
// by a bounded factor per level, so nothing over- or underflows in between, and because the schedule depends on the level index only --
// and every checkpoint spacing M in use is a multiple of RENORM -- a recomputation that starts from a checkpoint's (e, 1, f) repeats the
// forward sweep's operations exactly: every kernel family produces the same bits.  (RENORM = 0 would mean "the state is the pair itself"; both schemes are projective now.)
template <class S>
__device__ __forceinline__ void tri_advance(const S& st, int k, const double* rec, int nz, typename S::St& s) {
  st.advance(k, rec, nz, s);
  if constexpr (S::RENORM > 0) {
    if ((k + 1) % S::RENORM == 0) {
      double e, f;
      st.pair(s, e, f);
      st.seed(s, e, f);
    }
  }
}
// The forward sweep from level 0 (state of first()) to level k1, in groups of RENORM levels with the re-seeding at the end of each group
// -- as straight-line code: written as "if ((k + 1) % RENORM == 0)" inside a per-level loop the compiler if-converts the branch and
// evaluates the reciprocal of pair() at EVERY level, back on the chain (ISA of k_zqpa_pipe, round 3).  on_checkpoint(level, state) is
// called at every level that is a multiple of M (state freshly re-seeded there when M is a multiple of RENORM).
template <class S, int M, class F>
__device__ __forceinline__ void tri_forward(const S& st, const double* rec, int nz, typename S::St& s, int k1, F&& on_checkpoint) {
  if constexpr (S::RENORM > 0 && M % (S::RENORM > 0 ? S::RENORM : 1) == 0) {
    constexpr int R = S::RENORM;
    int k = 0;
    for (; k + R <= k1; k += R) {
#pragma unroll
      for (int i = 0; i < R; ++i) st.advance(k + i, rec, nz, s);
      double e, f;
      st.pair(s, e, f);
      st.seed(s, e, f);
      if ((k + R) % M == 0) on_checkpoint(k + R, s);
    }
    for (; k < k1; ++k) st.advance(k, rec, nz, s);  // fewer than RENORM levels left: no re-seeding level, no checkpoint among them
  } else {
    for (int k = 0; k < k1; ++k) {
      tri_advance(st, k, rec, nz, s);
      if ((k + 1) % M == 0) on_checkpoint(k + 1, s);
    }
  }
}
// ... and the same step returning the normalised pair of level k + 1
template <class S>
__device__ __forceinline__ void tri_step(const S& st, int k, const double* rec, int nz, typename S::St& s, double& e, double& f) {
  st.advance(k, rec, nz, s);
  st.pair(s, e, f);
  if constexpr (S::RENORM > 0) {
    if ((k + 1) % S::RENORM == 0) st.seed(s, e, f);
  }
}

// ------------------------------------------------------------------------------------------
// n79 (crt1d/solvers/_solve_n79.py:70-155).  Even row k <-> upward flux at level k, k = 0 .. nz-1.
struct TriN79 {
  // state of the forward sweep: e = p / q, f = g / q (see the note above tri_advance)
  struct St {
    double p, q, g;
  };
  static constexpr int RENORM = 4;
  __device__ static inline void pair(const St& s, double& e, double& f) {
    const double iq = fast_rcp(s.q);
    e = s.p * iq;
    f = s.g * iq;
  }
  __device__ static inline void seed(St& s, double e, double f) {
    s.p = e;
    s.q = 1.0;
    s.g = f;
  }
  static constexpr const char* NAME = "n79";
  static constexpr int NST = 4;   // staged: dn, up, aI_lsl, aI_lsh
  static constexpr int NOUT = 6;  // I_dr, I_df_d, I_df_u, F, aI_lsl, aI_lsh
  double swb, swd, rho, tau, alb, oma, invmu, irho;
  double dn, up;  // back-substitution state (level k+1)

  __host__ __device__ static inline int rows(int nz) { return nz; }               // even rows
  __host__ __device__ static constexpr int out_rows(int arr, int nz) { return arr >= 4 ? nz - 1 : nz; }
  __device__ inline double band_const() const { return swb; }

  template <typename TIO>
  __device__ inline void init(const double* rec, const SolveArgs& a, int c, int b) {
    init_band(rec, a, load_tri_band<TIO>(a, c, b));
  }
  __device__ inline void init_band(const double* rec, const SolveArgs&, const TriBand& in) {
    swb = in.I_dr0;
    swd = in.I_df0;
    rho = in.leaf_r;
    tau = in.leaf_t;
    alb = in.soil_r;
    oma = 1 - (rho + tau);  // :56,145
    irho = fast_rcp(rho);
    invmu = rec[S_INVMU];
  }
  // K0 vectors of the record (colpre.hip): [0] tbcum = e^{-K_b lai}, [1] 1 - tb, [2] 1 - td (by expm1), [3] fs / (fs dlai), [4] 1 / (fs dlai),
  // [5] (1 - fs) / ((1 - fs) dlai), [6] 1 / (1 - td).  Explicit FMAs throughout (the build runs with -ffp-contract=off): the sweeps are
  // dependency chains, every fused pair is one step less on them.
  // layer scattering coefficients (:85-88 / :102-105): r = trand/refld, s = refld - trand^2/refld.
  // 1/refld = (1/rho) * 1/(1 - td_j): a per-band register times a per-level K0 vector, no division here.
  __device__ inline void layer(const double* rec, int nz, int j, double& r, double& s) const {
    const double omt = rec[REC_HDR + 2 * nz + j];
    const double t = 1 - omt;
    const double refld = omt * rho;
    const double trand = __builtin_fma(omt, tau, t);
    r = trand * (irho * rec[REC_HDR + 6 * nz + j]);
    s = __builtin_fma(-trand, r, refld);
  }
  __device__ inline void first(const double* rec, int nz, St& st) const {
    st.p = -alb;  // row 0: soil, upward (:79-82)
    st.q = 1.0;
    st.g = swb * rec[REC_HDR] * alb;
  }
  // even pair of level k -> even pair of level k+1: the odd row of level k (layer m) and the even row of level k+1
  // (layer k) of the Thomas sweep (:180-192 applied to rows :85-129), merged into one rational update
  //   A = 1 + s_m e,  D = A - r_k r_m,   e' = -s_k A / D,   f' = (d_even A + r_k (d_odd + s_m f)) / D     (|r r| >> |A|: no cancellation in D)
  // and multiplied through by q (e = p / q, f = g / q), which takes the division off the chain:   t = q + s_m p  (= A q),
  //   p' = -s_k t,   q' = t - r_k r_m q,   g' = d_even t + r_k (d_odd q + s_m g).
  // |q| grows by about r r per level (r = trand / refld, 1e2..1e3 for thin layers): re-seeded every RENORM = 4 levels, far inside the range.
  __device__ inline void advance(int k, const double* rec, int nz, St& st) const {
    const double* tbcum = rec + REC_HDR;
    const double* omtb = tbcum + nz;
    const int mk = k == 0 ? 1 : k;  // the first downward row uses layer index 1 (:85-92), as the reference
    double rm, sm, r, s;
    layer(rec, nz, mk, rm, sm);
    if (k == 0) {
      layer(rec, nz, 0, r, s);
    } else {
      r = rm;
      s = sm;
    }
    const double src = swb * tbcum[k + 1];
    const double d_odd = (src * omtb[mk]) * __builtin_fma(-rho, rm, tau);   // (:92, :119)
    const double d_even = (src * omtb[k]) * __builtin_fma(-tau, r, rho);    // (:109, :129)
    const double t = __builtin_fma(sm, st.p, st.q);
    st.g = __builtin_fma(d_even, t, r * __builtin_fma(sm, st.g, d_odd * st.q));
    st.p = -s * t;
    st.q = __builtin_fma(-(r * rm), st.q, t);
  }
  // top even row (k = nz-1): dn = sky diffuse (:132-135); emits output level nz-1 (no layer above it)
  __device__ inline void top(const double* rec, int nz, double e, double f, double (&o)[NST]) {
    dn = swd;
    up = __builtin_fma(-e, dn, f);
    o[0] = dn;
    o[1] = up;
    o[2] = 0.0;
    o[3] = 0.0;
  }
  // level k from level k+1; emits output level k and layer k
  __device__ inline void back(int k, const double* rec, int nz, double e, double f, double (&o)[NST]) {
    const double* tbcum = rec + REC_HDR;
    const double omt = rec[REC_HDR + 2 * nz + k];
    const double t = 1 - omt;
    const double refld = omt * rho;
    const double trand = __builtin_fma(omt, tau, t);
    const double src = swb * tbcum[k + 1] * rec[REC_HDR + nz + k];
    const double dn1 = dn;
    // dn_k from the upward equation of level k+1 (layer k):  -r dn_k + up_{k+1} - s dn_{k+1} = d
    const double k_dn = __builtin_fma(trand, trand, -(refld * refld));
    const double k_src = __builtin_fma(rho, refld, -(tau * trand));
    dn = __builtin_fma(refld, up, __builtin_fma(k_dn, dn1, -(src * k_src))) * fast_rcp(trand);
    up = __builtin_fma(-e, dn, f);
    const double direct = src * oma;                       // :145
    const double diffuse = (dn1 + up) * (omt * oma);       // :146
    o[0] = dn;
    o[1] = up;
    o[2] = __builtin_fma(diffuse, rec[REC_HDR + 3 * nz + k], direct * rec[REC_HDR + 4 * nz + k]);  // :154  (diffuse fs + direct) / (fs dlai)
    o[3] = diffuse * rec[REC_HDR + 5 * nz + k];                                                    // :155  diffuse (1 - fs) / ((1 - fs) dlai)
  }
  // value of output array `arr` at tile row t (level j), band b; st[] = staged arrays at that element
  template <int ARR>
  __device__ static inline double value(const double* rec, int nz, int j, double bc, double invmu_, const double* tile, int stride,
                                        int idx) {
    if constexpr (ARR == 0) return bc * rec[REC_HDR + j];                                                          // :151
    if constexpr (ARR == 1) return tile[idx];
    if constexpr (ARR == 2) return tile[stride + idx];
    if constexpr (ARR == 3) return bc * rec[REC_HDR + j] * invmu_ + 2 * tile[idx] + 2 * tile[stride + idx];        // :161
    if constexpr (ARR == 4) return tile[2 * stride + idx];
    return tile[3 * stride + idx];
  }
  static constexpr bool derived(int arr) { return arr == 0 || arr == 3; }
  static constexpr int staged_slot(int arr) { return arr == 1 ? 0 : arr == 2 ? 1 : arr == 4 ? 2 : 3; }
  // staged slots [lo, hi) that the output arrays of a flush class read (class 0: the nz-row arrays, class 1: the nz-1-row arrays)
  static constexpr int park_slots(int cls, int hi) { return cls == 0 ? (hi ? 2 : 0) : (hi ? 4 : 2); }
  // all outputs of one (level j, band pair) from the staged pairs st[]; same expressions as value<>()
  __device__ static inline void emit(const double* rec, int nz, int j, d2 bc, double invmu_, const d2 (&st)[NST], d2 (&o)[NOUT]) {
    const d2 idr = bc * rec[REC_HDR + j];
    o[0] = idr;
    o[1] = st[0];
    o[2] = st[1];
    o[3] = idr * invmu_ + 2 * st[0] + 2 * st[1];
    o[4] = st[2];
    o[5] = st[3];
  }
};

// n79 on an equal-dLAI column (record flag S_UNIF; every LAI generator of the reference produces such columns): K0 then
// writes ONE (tb, td, 1/(1-td)) for all layers, the layer coefficients r, s and every product of per-band and per-layer
// constants leave the level loop, and the back substitution needs no reciprocal.  ~14 + rcp flops per advance instead
// of ~40, ~20 per back step instead of ~35 (the tridiagonal kernels are ALU-limited next to their stores).
struct TriN79U : TriN79 {
  double r, s, rr, k_odd, k_even, omtb;        // advance
  double refld, k_dn, k_src, itrand, omt_oma;  // back
  template <typename TIO>
  __device__ inline void init(const double* rec, const SolveArgs& a, int c, int b) {
    init_band(rec, a, load_tri_band<TIO>(a, c, b));
  }
  __device__ inline void init_band(const double* rec, const SolveArgs& a, const TriBand& in) {
    TriN79::init_band(rec, a, in);
    const int nz = a.nz;
    layer(rec, nz, 1, r, s);
    rr = r * r;
    const double omt = rec[REC_HDR + 2 * nz + 1], t = 1 - omt;
    omtb = rec[REC_HDR + nz + 1];
    k_odd = omtb * __builtin_fma(-rho, r, tau);   // (:92, :119)
    k_even = omtb * __builtin_fma(-tau, r, rho);  // (:109, :129)
    refld = omt * rho;
    const double trand = __builtin_fma(omt, tau, t);
    itrand = fast_rcp(trand);
    k_dn = __builtin_fma(trand, trand, -(refld * refld));
    k_src = omtb * __builtin_fma(rho, refld, -(tau * trand));
    omt_oma = omt * oma;
  }
  __device__ inline void advance(int k, const double* rec, int nz, St& st) const {  // (the projective update of TriN79::advance)
    const double src = swb * rec[REC_HDR + k + 1];
    const double t = __builtin_fma(s, st.p, st.q);
    st.g = __builtin_fma(src * k_even, t, r * __builtin_fma(s, st.g, (src * k_odd) * st.q));
    st.p = -s * t;
    st.q = __builtin_fma(-rr, st.q, t);
  }
  __device__ inline void back(int k, const double* rec, int nz, double e, double f, double (&o)[NST]) {
    const double src = swb * rec[REC_HDR + k + 1];
    const double dn1 = dn;
    dn = __builtin_fma(refld, up, __builtin_fma(k_dn, dn1, -(src * k_src))) * itrand;
    up = __builtin_fma(-e, dn, f);
    const double direct = src * omtb * oma;         // :145
    const double diffuse = (dn1 + up) * omt_oma;    // :146
    o[0] = dn;
    o[1] = up;
    o[2] = __builtin_fma(diffuse, rec[REC_HDR + 3 * nz + k], direct * rec[REC_HDR + 4 * nz + k]);  // :154
    o[3] = diffuse * rec[REC_HDR + 5 * nz + k];                                                    // :155
  }
};

// scheme object to use for a column whose record has S_UNIF set (same type = no specialisation)
template <class S>
struct UniformOf {
  typedef S type;
};
template <>
struct UniformOf<TriN79> {
  typedef TriN79U type;
};

// ------------------------------------------------------------------------------------------
// zq (crt1d/solvers/_solve_zq.py:74-219).  Even row k <-> SWu0[k], k = 0 .. m (m = nz); output level z = k, k < m.
struct TriZq {
  static constexpr const char* NAME = "zq";
  static constexpr int NST = 4;   // staged: I_df_d, I_df_u, I_df_d_ss, I_df_u_ss
  static constexpr int NOUT = 7;  // I_dr, I_df_d, I_df_u, F, I_df_d_ss, I_df_u_ss, F_ss
  double I_dr0, I_df0, rho, fwd, q, q0, cu, cd, invmu;
  double idq;     // 1 / (1 - q^2): dlo = dhi = 1 - q^2 on every interior level -- their two reciprocals per back step become none
  double ilo1;    // 1 / (1 - q0 q): dlo of the lowest layer.  (dhi of the top layer is 1.)  Written as "interior ? idq : fast_rcp(...)" the
                  // compiler evaluated the reciprocal at EVERY level and selected afterwards (ISA of k_zqpa_pipe2, round 3)
  double xd, xu;  // SWd0[li], SWu0[li] of the level above

  __host__ __device__ static inline int rows(int nz) { return nz + 1; }
  __host__ __device__ static constexpr int out_rows(int, int nz) { return nz; }
  __device__ inline double band_const() const { return I_dr0; }

  template <typename TIO>
  __device__ inline void init(const double* rec, const SolveArgs& a, int c, int b) {
    init_band(rec, a, load_tri_band<TIO>(a, c, b));
  }
  __device__ inline void init_band(const double* rec, const SolveArgs&, const TriBand& in) {
    I_dr0 = in.I_dr0;
    I_df0 = in.I_df0;
    const double bL = in.leaf_r, tL = in.leaf_t;
    rho = in.soil_r;
    const double mu = rec[S_MU], t = rec[S_TAUI], t_psi = rec[S_TPSI];
    invmu = rec[S_INVMU];
    const double aL = 1 - (bL + tL);                                             // :87
    const double r_i = 2.0 / 3 * (bL / (bL + tL)) + 1.0 / 3 * (tL / (bL + tL));  // eq. 23 :40-43
    const double r_psi = 0.5 + 0.3334 * ((bL - tL) / (bL + tL)) * mu;            // eq. 22 :35-38
    fwd = t + (1 - t) * (1 - aL) * (1 - r_i);                                    // :116
    q = r_i * (1 - aL) * (1 - t);
    q0 = 1.0 * (1 - (1 - rho)) * (1 - 0.0);  // ground "layer": r=1, t=0, a=1-rho (:106-108)
    cu = r_psi * (1 - t_psi) * (1 - aL);        // :139
    cd = (1 - t_psi) * (1 - aL) * (1 - r_psi);  // :142
    idq = fast_rcp(__builtin_fma(-q, q, 1.0));
    ilo1 = fast_rcp(__builtin_fma(-q0, q, 1.0));
  }
  // state of the forward sweep: e = p / q, f = g / q (see the note above tri_advance)
  struct St {
    double p, q, g;
  };
  static constexpr int RENORM = 4;
  __device__ static inline void pair(const St& s, double& e, double& f) {
    const double iq = fast_rcp(s.q);
    e = s.p * iq;
    f = s.g * iq;
  }
  __device__ static inline void seed(St& s, double e, double f) {
    s.p = e;
    s.q = 1.0;
    s.g = f;
  }
  __device__ inline void first(const double* rec, int nz, St& s) const {
    s.p = 0.0;  // row 0: x0 = rho S_0 (:115,136)
    s.q = 1.0;
    s.g = rho * (I_dr0 * rec[REC_HDR]);
  }
  // even pair of li-1 -> even pair of li: rows 2li-1 (sub -fwd, dia -qlo fwd, sup dlo; :116-118, rhs :137-139) and
  // 2li (sub dhi, dia -qhi fwd, sup -fwd; :119-121, rhs :140-142) of the sweep, merged into one rational update
  //   B = fwd (e - qlo),  D = qhi fwd B + dhi dlo,   e' = fwd B / D,   f' = (dhi (C1 + fwd f) - C2 B) / D,   C1 = dlo cu S, C2 = dhi cd S
  // and multiplied through by q (e = p / q, f = g / q):   u = p - qlo q,
  //   p' = fwd^2 u,   q' = qhi p' + dhi dlo q,   g' = dhi (fwd g + S (dlo cu q - cd fwd u))
  __device__ inline void advance(int k, const double* rec, int m, St& s) const {
    const int li = k + 1;
    const double S = I_dr0 * rec[REC_HDR + li - 1];  // :130
    const double qlo = (li == 1) ? q0 : q;
    const double qhi = (li == m) ? 0.0 : q;
    const double dlo = __builtin_fma(-qlo, q, 1.0);  // :118
    const double dhi = __builtin_fma(-q, qhi, 1.0);  // :119
    const double u = __builtin_fma(-qlo, s.q, s.p);
    const double pn = (fwd * fwd) * u;
    const double t = __builtin_fma(dlo * cu, s.q, -((cd * fwd) * u));
    s.g = dhi * __builtin_fma(fwd, s.g, S * t);
    s.q = __builtin_fma(qhi, pn, (dhi * dlo) * s.q);
    s.p = pn;
  }
  // k = m: x[2m+1] = I_df0 (:122,143); no output row at k = m
  __device__ inline void top(const double* rec, int m, double e, double f, double (&o)[NST]) {
    xd = I_df0;
    xu = __builtin_fma(-e, xd, f);
    o[0] = o[1] = o[2] = o[3] = 0.0;
  }
  __device__ inline void back(int k, const double* rec, int m, double e, double f, double (&o)[NST]) {
    const int li = k + 1;
    const double S = I_dr0 * rec[REC_HDR + k];
    const double qlo = (li == 1) ? q0 : q;
    const double qhi = (li == m) ? 0.0 : q;
    const double dhi = __builtin_fma(-q, qhi, 1.0);
    // SWd0[li-1] from the original row 2li:  dhi x_{2li-1} - qhi fwd x_{2li} - fwd x_{2li+1} = C
    // (1/dhi and 1/dlo take three values per band -- 1 / (1 - q^2), 1 / (1 - q0 q), 1 -- kept in two registers since round 3)
    const double rdhi = (li == m) ? 1.0 : idq;  // 1 / dhi: the same bits fast_rcp(dhi) gives (dhi is 1 at the top, 1 - q^2 elsewhere)
    const double xdl = __builtin_fma(dhi * cd, S, fwd * __builtin_fma(qhi, xu, xd)) * rdhi;
    const double xul = __builtin_fma(-e, xdl, f);  // SWu0[li-1]
    const double iden = (li == 1) ? ilo1 : idq;   // 1 / dlo; multiple-scattering correction, eqs. 24/25 (:180-187)
    o[0] = __builtin_fma(q, xul, xd) * iden;
    o[1] = __builtin_fma(qlo, xd, xul) * iden;
    o[2] = xd;   // I_df_d_ss :197
    o[3] = xul;  // I_df_u_ss :199
    xd = xdl;
    xu = xul;
  }
  template <int ARR>
  __device__ static inline double value(const double* rec, int nz, int j, double bc, double invmu_, const double* tile, int stride,
                                        int idx) {
    if constexpr (ARR == 0) return bc * rec[REC_HDR + j];                                                              // :219
    if constexpr (ARR == 1) return tile[idx];
    if constexpr (ARR == 2) return tile[stride + idx];
    if constexpr (ARR == 3) return bc * rec[REC_HDR + j] * invmu_ + 2 * tile[stride + idx] + 2 * tile[idx];            // :202
    if constexpr (ARR == 4) return tile[2 * stride + idx];
    if constexpr (ARR == 5) return tile[3 * stride + idx];
    return bc * rec[REC_HDR + j] * invmu_ + 2 * tile[3 * stride + idx] + 2 * tile[2 * stride + idx];                   // :201
  }
  static constexpr bool derived(int arr) { return arr == 0 || arr == 3 || arr == 6; }
  static constexpr int staged_slot(int arr) { return arr == 1 ? 0 : arr == 2 ? 1 : arr == 4 ? 2 : 3; }
  static constexpr int park_slots(int, int hi) { return hi ? NST : 0; }  // one flush class, all staged arrays
  __device__ static inline void emit(const double* rec, int nz, int j, d2 bc, double invmu_, const d2 (&st)[NST], d2 (&o)[NOUT]) {
    const d2 S = bc * rec[REC_HDR + j];
    o[0] = S;
    o[1] = st[0];
    o[2] = st[1];
    o[3] = S * invmu_ + 2 * st[1] + 2 * st[0];
    o[4] = st[2];
    o[5] = st[3];
    o[6] = S * invmu_ + 2 * st[3] + 2 * st[2];
  }
};

// ------------------------------------------------------------------------------------------
// zq_pa computational-grid solve (crt1d/solvers/_solve_zq_pa.py:24-418): the zq system with nz := M, only the two
// single-scattering interface fluxes are kept (see tri_zqpa.hip, launch_zqpa).
struct TriZqPa : TriZq {
  static constexpr const char* NAME = "zq_pa grid";
  static constexpr int NOUT = 2;
  __host__ __device__ static constexpr int out_rows(int, int nz) { return nz; }
  template <int ARR>
  __device__ static inline double value(const double*, int, int, double, double, const double* tile, int stride, int idx) {
    return tile[ARR * stride + idx];
  }
  static constexpr bool derived(int) { return false; }
  static constexpr int staged_slot(int arr) { return arr; }
  __device__ static inline void emit(const double*, int, int, d2, double, const d2 (&st)[NST], d2 (&o)[NOUT]) {
    o[0] = st[0];
    o[1] = st[1];
  }
};

}  // namespace
}  // namespace crt
