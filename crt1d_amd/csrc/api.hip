// C-ABI entry points of libcrt1d_hip.so (see include/crt1d_hip.h), argument validation,
// the absorption + band-integral epilogue kernel and the HBM bandwidth probes.
#include <math.h>

#include <algorithm>
#include <type_traits>

#include "crt_internal.hpp"

namespace crt {
namespace {

// ------------------------------------------------------------------------------------------
// Epilogue: model.py:573-647 (_calc_absorption) fused with diagnostics.py:39-108 (band sums).
//
// k_absorb_bandsum: ONE streaming pass over the three profiles.  Workgroup = column, thread = band (NBT bands per thread when
// nb > 1024); the previous level's values stay in registers, so every profile byte is read exactly once (round 1 gave a
// (column, layer) to each wave and read rows k and k+1: every row twice, 0.27 of the HBM roofline).  Per level only TWO
// band sums per group are needed, because the sunlit/shaded split uses band-independent factors (as in the integrated
// kernels, crt_internal.hpp):
//     A_g(k) = sum_b w_g[b] aI(k, b),   D_g(k) = sum_b w_g[b] (1 - r - t)[b] I_dr(k+1, b)
//     aI_dr = (1 - e^{-K_b dlai_k}) D,  aI_df = A - aI_dr,  aI_sl = aI_df f_sl(k) + aI_dr,  aI_sh = aI_df (1 - f_sl(k))
// Wave sums by DPP (wave_sum_lane63), one LDS slot per wave and level; the levels are processed in chunks of BS_CH whose loads
// are all issued before the arithmetic (24 rows in flight per thread), a double-buffered partial-sum area needs one LDS-only
// barrier per chunk, and the first BS_CH * ngroup threads turn the partials of the previous chunk into outputs.
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int MAXG = 4;
constexpr int BS_CH = 8;

struct EpiArgs {
  int ncol, nb, nz, ngroup;
  long long col_stride;
  const double* psi;
  const double* lai;
  const int32_t* g_kind;
  const double* g_param;
  const double* g_at_psi;
  const double* leaf_r;
  const double* leaf_t;
  const double* I_dr;
  const double* I_df_d;
  const double* I_df_u;
  const double* band_w;
  double* aI;
  double* aI_sl;
  double* aI_sh;
  double* totals;
  // optional (crt_bandsum_out): the direct-beam part of the absorption [ncol][nz-1][ngroup], and the band-integrated LEVEL profiles
  // [ncol][nz][ngroup] of every irradiance variable diagnostics.band sums (diagnostics.py:84-91): I_dr, I_df_d, I_df_u, F, I_d
  double* aI_dr;
  double* L_dr;
  double* L_dn;
  double* L_up;
  double* L_F;
  double* L_Id;
};

// level outputs from the three level sums: F = I_dr / mu + 2 I_df_u + 2 I_df_d is linear in the profiles (every scheme forms its F this
// way, e.g. _solve_2s.py:156), so its band sum is the same combination of the band sums; I_d = I_dr + I_df_d (model.py:425)
__device__ inline void store_level_profiles(const EpiArgs& a, long long o, double R, double Dn, double Up, double invmu, bool accumulate) {
  if (accumulate) {
    a.L_dr[o] += R;
    a.L_dn[o] += Dn;
    a.L_up[o] += Up;
    a.L_F[o] += R * invmu + 2 * (Up + Dn);
    a.L_Id[o] += R + Dn;
  } else {
    __builtin_nontemporal_store(R, a.L_dr + o);
    __builtin_nontemporal_store(Dn, a.L_dn + o);
    __builtin_nontemporal_store(Up, a.L_up + o);
    __builtin_nontemporal_store(R * invmu + 2 * (Up + Dn), a.L_F + o);
    __builtin_nontemporal_store(R + Dn, a.L_Id + o);
  }
}

template <int MAXT, bool PROF>
__global__ __launch_bounds__(MAXT) void k_absorb_bandsum(EpiArgs a, int b0, int nbs, int accumulate) {
  // bands [b0, b0 + nbs) of every row (nbs <= blockDim.x <= 1024; spectra wider than 1024 bands take several launches, the
  // later ones adding to the outputs of the first)
  extern __shared__ double lds[];
  const int c = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const int nz = a.nz, nb = a.nb, ng = a.ngroup;
  const double psi = a.psi[c];
  const int kind = a.g_kind[c];
  const double G = (kind == CRT_G_TABLE) ? a.g_at_psi[c] : G_closed(kind, a.g_param ? a.g_param[c] : 0.0, cos(psi), sin(psi));
  const double Kb = G / cos(psi);
  const double* __restrict__ lai = a.lai + (long long)c * nz;
  const long long cb = (long long)c * nz * nb + b0;
  const double* __restrict__ R = a.I_dr + cb;
  const double* __restrict__ D = a.I_df_d + cb;
  const double* __restrict__ U = a.I_df_u + cb;
  // LDS: part[2][BS_CH][nwave][NV][MAXG] (NV = 2 sums per level and group: A, D; PROF: + the level sums of I_dr, I_df_d, I_df_u),
  // ends[nwave][4][MAXG], PROF: lev0[nwave][3][MAXG] (the level sums of row 0)
  constexpr int NV = PROF ? 5 : 2;
  double* part = lds;
  double* ends = lds + 2 * BS_CH * nwave * NV * MAXG;
  double* lev0 = ends + nwave * 4 * MAXG;
  const int pstride = nwave * NV * MAXG;  // doubles per level slot
  const double invmu = 1.0 / cos(psi);

  const bool act = tid < nbs;
  const int bi = act ? tid : 0;
  double w[MAXG], wa[MAXG];
  {
    const long long ib = (long long)c * a.col_stride + b0 + bi;
    const double la = act ? 1 - (a.leaf_r[ib] + a.leaf_t[ib]) : 0.0;  // :584
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
      w[g] = (g < ng && act) ? a.band_w[(long long)g * nb + b0 + bi] : 0.0;
      wa[g] = w[g] * la;
    }
  }
  double r0 = R[bi], d0 = D[bi], u0 = U[bi];
  if constexpr (PROF) {  // level sums of row 0
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < ng) {
        const double t0 = wave_sum_lane63(w[g] * r0), t1 = wave_sum_lane63(w[g] * d0), t2 = wave_sum_lane63(w[g] * u0);
        if (lane == 63) {
          lev0[(wave * 3 + 0) * MAXG + g] = t0;
          lev0[(wave * 3 + 1) * MAXG + g] = t1;
          lev0[(wave * 3 + 2) * MAXG + g] = t2;
        }
      }
  }
  // energy-balance terms at the ground (diagnostics.py:476-530): transmitted I_d[0], soil-reflected I_df_u[0]
  if (a.totals) {
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < ng) {
        const double t2 = wave_sum_lane63(w[g] * (r0 + d0)), t3 = wave_sum_lane63(w[g] * u0);
        if (lane == 63) {
          ends[(wave * 4 + 2) * MAXG + g] = t2;
          ends[(wave * 4 + 3) * MAXG + g] = t3;
        }
      }
  }

  auto finish_chunk = [&](int k0, int nlev, int buf) {  // first BS_CH * ng threads: partials of levels k0 .. k0+nlev-1 -> outputs
    if (tid < BS_CH * ng) {
      const int t = tid / ng, g = tid - t * ng;
      if (t < nlev) {
        const int k = k0 + t;
        const double* p = part + (buf * BS_CH + t) * pstride;
        double A = 0.0, Dg = 0.0;
        for (int wv = 0; wv < nwave; ++wv) {
          A += p[(wv * NV + 0) * MAXG + g];
          Dg += p[(wv * NV + 1) * MAXG + g];
        }
        const double dl = lai[k] - lai[k + 1];                      // model.py:248
        const double fsl = exp(-Kb * ((lai[k] + lai[k + 1]) / 2));  // :601-602
        const double adr = (1 - exp(-Kb * dl)) * Dg;                // :617-621
        const double adf = A - adr;                                 // :628
        const long long o = ((long long)c * (nz - 1) + k) * ng + g;
        if (accumulate) {
          a.aI[o] += A;
          a.aI_sl[o] += adf * fsl + adr;
          a.aI_sh[o] += adf * (1 - fsl);
        } else {
          a.aI[o] = A;
          a.aI_sl[o] = adf * fsl + adr;                             // :631-633
          a.aI_sh[o] = adf * (1 - fsl);
        }
        if constexpr (PROF) {
          if (accumulate) a.aI_dr[o] += adr; else a.aI_dr[o] = adr;
          double sR = 0.0, sD = 0.0, sU = 0.0;
          for (int wv = 0; wv < nwave; ++wv) {
            sR += p[(wv * NV + 2) * MAXG + g];
            sD += p[(wv * NV + 3) * MAXG + g];
            sU += p[(wv * NV + 4) * MAXG + g];
          }
          store_level_profiles(a, ((long long)c * nz + k + 1) * ng + g, sR, sD, sU, invmu, accumulate != 0);  // level k + 1
        }
      }
    }
  };

  int buf = 0, prev_k0 = -1, prev_n = 0;
  for (int k0 = 0; k0 < nz - 1; k0 += BS_CH) {
    const int nlev = min(BS_CH, nz - 1 - k0);
    // all loads of the chunk first: rows k0+1 .. k0+nlev of the three profiles
    double r1[BS_CH], d1[BS_CH], u1[BS_CH];
#pragma unroll
    for (int t = 0; t < BS_CH; ++t)
      if (t < nlev) {
        const long long row = (long long)(k0 + t + 1) * nb + bi;
        r1[t] = __builtin_nontemporal_load(R + row);
        d1[t] = __builtin_nontemporal_load(D + row);
        u1[t] = __builtin_nontemporal_load(U + row);
      }
    if (prev_k0 >= 0) finish_chunk(prev_k0, prev_n, buf ^ 1);  // the previous chunk's outputs while this chunk's loads are in flight
#pragma unroll
    for (int t = 0; t < BS_CH; ++t)
      if (t < nlev) {
        const double av = r1[t] - r0 + d1[t] - d0 + u0 - u1[t];  // :609
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
          if (g < ng) {
            const double ta = wave_sum_lane63(w[g] * av), td = wave_sum_lane63(wa[g] * r1[t]);
            double* p = part + (buf * BS_CH + t) * pstride;
            if (lane == 63) {
              p[(wave * NV + 0) * MAXG + g] = ta;
              p[(wave * NV + 1) * MAXG + g] = td;
            }
            if constexpr (PROF) {
              const double s0 = wave_sum_lane63(w[g] * r1[t]), s1 = wave_sum_lane63(w[g] * d1[t]), s2 = wave_sum_lane63(w[g] * u1[t]);
              if (lane == 63) {
                p[(wave * NV + 2) * MAXG + g] = s0;
                p[(wave * NV + 3) * MAXG + g] = s1;
                p[(wave * NV + 4) * MAXG + g] = s2;
              }
            }
          }
        r0 = r1[t];
        d0 = d1[t];
        u0 = u1[t];
      }
    lds_barrier();  // chunk complete (and every thread is past its reads of the other buffer)
    prev_k0 = k0;
    prev_n = nlev;
    buf ^= 1;
  }
  // canopy top: incoming I_d[top], reflected I_df_u[top] (r0, d0, u0 now hold level nz-1)
  if (a.totals) {
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
      if (g < ng) {
        const double t0 = wave_sum_lane63(w[g] * (r0 + d0)), t1 = wave_sum_lane63(w[g] * u0);
        if (lane == 63) {
          ends[(wave * 4 + 0) * MAXG + g] = t0;
          ends[(wave * 4 + 1) * MAXG + g] = t1;
        }
      }
  }
  if (prev_k0 >= 0) finish_chunk(prev_k0, prev_n, buf ^ 1);
  if (a.totals || PROF) {
    lds_barrier();
    if (a.totals && tid < ng * 4) {
      const int g = tid >> 2, q = tid & 3;
      double t = 0.0;
      for (int wv = 0; wv < nwave; ++wv) t += ends[(wv * 4 + q) * MAXG + g];
      double* o = a.totals + ((long long)c * ng + g) * 4 + q;
      *o = accumulate ? *o + t : t;
    }
    if constexpr (PROF) {
      if (tid < ng) {  // level 0
        double sR = 0.0, sD = 0.0, sU = 0.0;
        for (int wv = 0; wv < nwave; ++wv) {
          sR += lev0[(wv * 3 + 0) * MAXG + tid];
          sD += lev0[(wv * 3 + 1) * MAXG + tid];
          sU += lev0[(wv * 3 + 2) * MAXG + tid];
        }
        store_level_profiles(a, (long long)c * nz * ng + tid, sR, sD, sU, invmu, accumulate != 0);
      }
    }
  }
}

// The per-column tail of the band-sum kernels below: raw band sums A_g(k), D_g(k) (LDS) -> the level outputs, lanes over (level, group).
// Streaming stores: the outputs are not read again by this kernel (1e5 x 38 x 100: 2.05 -> 1.91 ms).
__device__ inline double column_kb(const EpiArgs& a, int c) {
  const double psi = a.psi[c];
  const int kind = a.g_kind[c];
  const double G = (kind == CRT_G_TABLE) ? a.g_at_psi[c] : G_closed(kind, a.g_param ? a.g_param[c] : 0.0, cos(psi), sin(psi));
  return G / cos(psi);
}
template <int NGT, bool PROF = false>
__device__ inline void bandsum_finish(const EpiArgs& a, int c, const double* raw, const double* ends, const double* __restrict__ lai, double Kb, int l,
                                      int nlanes, const double* lev = nullptr) {
  const int ng = a.ngroup, nl = a.nz - 1;
  const long long ob = (long long)c * nl * ng;
  for (int i = l; i < nl * ng; i += nlanes) {
    const int k = i / ng, g = i - k * ng;
    const double A = raw[(k * NGT + g) * 2], Dg = raw[(k * NGT + g) * 2 + 1];
    const double fsl = exp(-Kb * ((lai[k] + lai[k + 1]) / 2));       // model.py:601-602
    const double adr = (1 - exp(-Kb * (lai[k] - lai[k + 1]))) * Dg;  // :617-621
    const double adf = A - adr;                                          // :628
    __builtin_nontemporal_store(A, a.aI + ob + i);
    __builtin_nontemporal_store(adf * fsl + adr, a.aI_sl + ob + i);      // :631-633
    __builtin_nontemporal_store(adf * (1 - fsl), a.aI_sh + ob + i);
    if constexpr (PROF) __builtin_nontemporal_store(adr, a.aI_dr + ob + i);
  }
  if constexpr (PROF) {  // band-integrated level profiles (diagnostics.py:81): lev = [nz][NGT][3] level sums of I_dr, I_df_d, I_df_u
    const double invmu = 1.0 / cos(a.psi[c]);
    const long long lb = (long long)c * a.nz * ng;
    for (int i = l; i < a.nz * ng; i += nlanes) {
      const int j = i / ng, g = i - j * ng;
      const double* q = lev + (j * NGT + g) * 3;
      store_level_profiles(a, lb + i, q[0], q[1], q[2], invmu, false);
    }
  }
  if (a.totals && l < 4 * ng) {  // incoming, reflected, transmitted, soil-reflected
    const int g = l >> 2, q = l & 3;
    a.totals[((long long)c * ng + g) * 4 + q] = ends[(q < 2 ? 2 * NGT : 0) + 2 * g + (q & 1)];
  }
}

// k_absorb_bandsum_w: the same pass with ONE WAVE PER COLUMN (nb <= 512): lane l owns bands l, l + 64, ... (NBT of them), so
// the cross-band reduction of a level costs one set of lane exchanges per COLUMN instead of one per wave of a multi-wave
// workgroup -- the kernel above spends most of its instructions there (2 ngroup reductions x 18 VALU instructions x 5 waves per
// level at nb = 300: VALU-bound at 2.1 TB/s, measured) -- and the exchanges themselves reduce FOUR values at a time (wave_sum4,
// crt_internal.hpp: 5 instructions per value instead of 18).  No barriers, no cross-wave traffic: the waves of a workgroup are
// independent columns.  The raw band sums A_g(k), D_g(k) go to LDS; at the end the lanes turn them into the level outputs
// (level factors f_sl(k), 1 - e^{-K_b dlai_k} evaluated there, lanes over levels) and write them coalesced.
template <int NBT, int CH, int NGT, bool PF = false, bool PROF = false>
__global__ __launch_bounds__(256) void k_absorb_bandsum_w(EpiArgs a, int wpb, int per_wave) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave-uniform column index in a scalar register: the profile pointers below then are scalar bases, and every load is
  // "scalar base + one 32-bit lane offset" shared by the three arrays (15 offset registers instead of 90 address registers)
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + wave);
  if (c >= a.ncol) return;  // (no workgroup barrier anywhere in this kernel)
  const int nz = a.nz, nb = a.nb, ng = a.ngroup, nl = nz - 1;
  double* raw = lds + (size_t)wave * per_wave;  // [nl][NGT][2]: A_g(k), D_g(k)
  double* ends = raw + 2 * NGT * nl;            // [2][NGT][2]: ground (I_d, I_df_u), top (I_d, I_df_u)
  double* lev = ends + 4 * NGT;                 // PROF: [nz][NGT][3] level sums of I_dr, I_df_d, I_df_u
  const long long cb = (long long)c * nz * nb;
  const double* __restrict__ R = a.I_dr + cb;
  const double* __restrict__ D = a.I_df_d + cb;
  const double* __restrict__ U = a.I_df_u + cb;
  int bi[NBT];
  double w[NBT][NGT], la[NBT], r0[NBT], d0[NBT], u0[NBT];  // NGT >= ngroup: weight registers for the groups in use only
#pragma unroll
  for (int i = 0; i < NBT; ++i) {
    const int b = lane + 64 * i;
    const bool act = b < nb;
    bi[i] = act ? b : 0;
    const long long ib = (long long)c * a.col_stride + bi[i];
    la[i] = act ? 1 - (a.leaf_r[ib] + a.leaf_t[ib]) : 0.0;  // :584
#pragma unroll
    for (int g = 0; g < NGT; ++g) w[i][g] = (g < ng && act) ? a.band_w[(long long)g * nb + bi[i]] : 0.0;
    r0[i] = R[bi[i]];
    d0[i] = D[bi[i]];
    u0[i] = U[bi[i]];
  }
  auto end_terms = [&](double* dst) {  // sum_b w (I_dr + I_df_d), sum_b w I_df_u of the level held in r0, d0, u0
    double v[2 * NGT];
#pragma unroll
    for (int g = 0; g < NGT; ++g) {
      v[2 * g] = v[2 * g + 1] = 0.0;
#pragma unroll
      for (int i = 0; i < NBT; ++i) {
        v[2 * g] += w[i][g] * (r0[i] + d0[i]);
        v[2 * g + 1] += w[i][g] * u0[i];
      }
    }
    wave_sum_store(v, dst, 2 * NGT, lane);
  };
  const double Kb = column_kb(a, c);  // now, not in the tail: three dependent round trips with nothing else of the wave in flight there
  if (a.totals) end_terms(ends);  // ground: transmitted I_d[0], soil-reflected I_df_u[0]  (diagnostics.py:476-530)
  if constexpr (PROF) {  // level sums of row 0
    double v[3 * NGT];
#pragma unroll
    for (int g = 0; g < NGT; ++g) {
      v[3 * g] = v[3 * g + 1] = v[3 * g + 2] = 0.0;
#pragma unroll
      for (int i = 0; i < NBT; ++i) {
        v[3 * g] += w[i][g] * r0[i];
        v[3 * g + 1] += w[i][g] * d0[i];
        v[3 * g + 2] += w[i][g] * u0[i];
      }
    }
    wave_sum_store(v, lev, 3 * NGT, lane);
  }
  // a chunk = CH levels: `fetch` issues its loads, `reduce` forms the level terms and the band sums
  auto fetch = [&](int k0, double (&r1)[CH][NBT], double (&d1)[CH][NBT], double (&u1)[CH][NBT]) {
    const int nlev = min(CH, nl - k0);
#pragma unroll
    for (int t = 0; t < CH; ++t)
      if (t < nlev) {
        const unsigned row = (unsigned)(k0 + t + 1) * (unsigned)nb;  // nz * nb < 2^31 (checked by the launcher)
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
          const unsigned off = row + (unsigned)bi[i];
          r1[t][i] = __builtin_nontemporal_load(R + off);
          d1[t][i] = __builtin_nontemporal_load(D + off);
          u1[t][i] = __builtin_nontemporal_load(U + off);
        }
      }
  };
  auto reduce = [&](int k0, const double (&r1)[CH][NBT], const double (&d1)[CH][NBT], const double (&u1)[CH][NBT]) {
    const int nlev = min(CH, nl - k0);
    double v[CH * NGT * 2];  // [t][g][A, D]
#pragma unroll
    for (int t = 0; t < CH; ++t) {
#pragma unroll
      for (int g = 0; g < NGT; ++g) v[(t * NGT + g) * 2] = v[(t * NGT + g) * 2 + 1] = 0.0;
      if (t < nlev) {
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
          const double av = r1[t][i] - r0[i] + d1[t][i] - d0[i] + u0[i] - u1[t][i];  // :609
          const double ar = la[i] * r1[t][i];                                         // :617-621 without the level factor
#pragma unroll
          for (int g = 0; g < NGT; ++g) {
            v[(t * NGT + g) * 2] += w[i][g] * av;
            v[(t * NGT + g) * 2 + 1] += w[i][g] * ar;
          }
          r0[i] = r1[t][i];
          d0[i] = d1[t][i];
          u0[i] = u1[t][i];
        }
      }
    }
    wave_sum_store(v, raw + (size_t)k0 * NGT * 2, nlev * NGT * 2, lane);
    if constexpr (PROF) {  // the level sums of rows k0 + 1 .. k0 + nlev
      double v3[CH * NGT * 3];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
#pragma unroll
        for (int g = 0; g < NGT; ++g) {
          double s0 = 0.0, s1 = 0.0, s2 = 0.0;
          if (t < nlev) {
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
              s0 += w[i][g] * r1[t][i];
              s1 += w[i][g] * d1[t][i];
              s2 += w[i][g] * u1[t][i];
            }
          }
          v3[(t * NGT + g) * 3] = s0;
          v3[(t * NGT + g) * 3 + 1] = s1;
          v3[(t * NGT + g) * 3 + 2] = s2;
        }
      }
      wave_sum_store(v3, lev + (size_t)(k0 + 1) * NGT * 3, nlev * NGT * 3, lane);
    }
  };
  if constexpr (PF) {
    // narrow spectra (one band per lane): a chunk is only CH rows of nb * 8 bytes per array, too little in flight to cover the
    // HBM latency at the occupancy the registers allow -> the next chunk's loads are issued before this chunk is reduced
    double rA[CH][NBT], dA[CH][NBT], uA[CH][NBT], rB[CH][NBT], dB[CH][NBT], uB[CH][NBT];
    fetch(0, rA, dA, uA);
    for (int k0 = 0; k0 < nl; k0 += 2 * CH) {
      if (k0 + CH < nl) fetch(k0 + CH, rB, dB, uB);
      reduce(k0, rA, dA, uA);
      if (k0 + CH < nl) {
        if (k0 + 2 * CH < nl) fetch(k0 + 2 * CH, rA, dA, uA);
        reduce(k0 + CH, rB, dB, uB);
      }
    }
  } else {
    for (int k0 = 0; k0 < nl; k0 += CH) {
      double r1[CH][NBT], d1[CH][NBT], u1[CH][NBT];
      fetch(k0, r1, d1, u1);
      reduce(k0, r1, d1, u1);
    }
  }
  if (a.totals) end_terms(ends + 2 * NGT);  // canopy top: incoming I_d[top], reflected I_df_u[top]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave: LDS operations complete in order; make the sums visible to all lanes
  bandsum_finish<NGT, PROF>(a, c, raw, ends, a.lai + (long long)c * nz, Kb, lane, 64, lev);
}

// k_absorb_bandsum_l: narrow spectra (32 < nb <= 48, even; nz <= 257): one wave per column, the lanes over LAYERS.
//   * lanes over layers: in the kernel above a narrow spectrum leaves lanes idle and still pays the full cross-lane reduction per
//     level.  Here a slab of LS + 1 = 17 consecutive rows of the three profiles is staged in LDS as the flat copy of its global run
//     (16-byte loads into registers -- issued one slab ahead, in flight while the previous slab is reduced -- then 16-byte LDS
//     writes) and lane (t = lane & 15, p = lane >> 4) sums layer k0 + t over the bands p, p + 4, ... serially; what is left of the
//     reduction is the sum over the four p, four values per set of six lane swaps.
//     Rows are nbp = nb or nb + 2 doubles apart in LDS, whichever is 2 (mod 4): the 16 layers of a read fall into 16 bank groups.
//   * the head and the tail of a column have nothing in flight to hide behind, so they are kept short: everything the column needs
//     (first slab, LAI levels, leaf optics, geometry) is requested in one go, the level factors (two exponentials per layer) are
//     formed while the second slab is in flight, and the tail is LDS reads, three multiply-adds and streaming stores.
//     Measured per column at 1e5 x 38 x 100 before that (wall_clock64 stamps): first slab after 7.7 us (three dependent round
//     trips), 3.3 us per further slab, then 2.2 + 2.6 + 0.9 us (last reduction, tail arithmetic, store acknowledgement).
//     (A persistent form -- waves walking columns w, w + G, ... with the next column's first slab prefetched -- was tried: the
//     loop-carried state costs 316 registers, one wave per SIMD, 3.3 ms instead of 2.0; capped at 256 it spills, 2.3 ms.)
template <int NGT>
__global__ __launch_bounds__(64) void k_absorb_bandsum_l(EpiArgs a, int nbp) {
  constexpr int LS = 16, NP = 4, NLD = 9;  // NLD 16-byte loads per lane cover 17 rows of <= 64 bands: 17 * 32 <= 9 * 64
  constexpr int NLAI = 5;                  // LAI levels per lane: nz <= 64 * 4 + 1 (the launcher checks)
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int nz = a.nz, nb = a.nb, ng = a.ngroup, nl = nz - 1, nb2 = nb >> 1;
  const int SS = (LS + 1) * nbp;          // one array's slab
  double* slab = lds;                     // [3][LS + 1][nbp]
  double* wts = slab + 3 * SS;            // [NGT + 1][nbp]: the groups' band weights, then 1 - (leaf_r + leaf_t) of the column
  double* raw = wts + (NGT + 1) * nbp;    // [nl][NGT][2]: A_g(k), D_g(k)
  double* ends = raw + 2 * NGT * nl;      // [2][NGT][2]
  double* lai_s = ends + 4 * NGT;         // [nz]
  double* lvl = lai_s + nz;               // [nl][2]: f_sl(k), 1 - e^{-K_b dlai_k}
  // where this lane's i-th 16-byte piece of a slab goes in LDS (the same for every slab): piece lane + 64 i of the flat run
  int dst[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = lane + 64 * i;
    const int row = idx / nb2;
    dst[i] = row * nbp + 2 * (idx - row * nb2);
  }
  const int t = lane & 15, p = lane >> 4;
  const int slot = wave_sum4_slot(p);
  const bool band_lane = lane < nb;
  // registers of the slab in flight, and of the column it opens (only loaded with a column's first slab)
  d2 sr[NLD], sd[NLD], su[NLD];
  double c_lr = 0.0, c_lt = 0.0, c_lai[NLAI], c_psi = 0.0, c_gp = 0.0, c_ga = 0.0;
  int c_kind = 0;
  const int c = blockIdx.x;
  auto fetch = [&](int k0) {
    const long long cb2 = (long long)c * nz * nb2;
    const d2* __restrict__ R2 = reinterpret_cast<const d2*>(a.I_dr) + cb2;
    const d2* __restrict__ D2 = reinterpret_cast<const d2*>(a.I_df_d) + cb2;
    const d2* __restrict__ U2 = reinterpret_cast<const d2*>(a.I_df_u) + cb2;
    const int n2 = (min(LS, nl - k0) + 1) * nb2;
    const unsigned base = (unsigned)k0 * (unsigned)nb2;  // nz * nb < 2^31 (checked by the launcher)
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (lane + 64 * i < n2) {
        sr[i] = __builtin_nontemporal_load(R2 + base + lane + 64 * i);
        sd[i] = __builtin_nontemporal_load(D2 + base + lane + 64 * i);
        su[i] = __builtin_nontemporal_load(U2 + base + lane + 64 * i);
      }
    if (k0 == 0) {
      const long long ib = (long long)c * a.col_stride + (band_lane ? lane : 0);
      c_lr = a.leaf_r[ib];
      c_lt = a.leaf_t[ib];
#pragma unroll
      for (int j = 0; j < NLAI; ++j) c_lai[j] = a.lai[(long long)c * nz + min(lane + 64 * j, nz - 1)];
      c_psi = a.psi[c];
      c_kind = a.g_kind[c];
      c_gp = a.g_param ? a.g_param[c] : 0.0;
      c_ga = a.g_at_psi ? a.g_at_psi[c] : 0.0;
    }
  };
  auto end_terms = [&](int row, double* out) {  // sum_b w (I_dr + I_df_d), sum_b w I_df_u of slab row `row`
    double v[2 * NGT];
    const int o = row * nbp + (band_lane ? lane : 0);
    const double rd = band_lane ? slab[o] + slab[SS + o] : 0.0, uu = band_lane ? slab[2 * SS + o] : 0.0;
#pragma unroll
    for (int g = 0; g < NGT; ++g) {
      const double wg = wts[g * nbp + (band_lane ? lane : 0)];
      v[2 * g] = wg * rd;
      v[2 * g + 1] = wg * uu;
    }
    wave_sum_store(v, out, 2 * NGT, lane);
  };
  fetch(0);  // first: everything below queues up behind it
  {  // band weights: all requested at once (a load inside `g < ng ? ... : 0` is waited for before the next one is issued)
    double wv[NGT];
#pragma unroll
    for (int g = 0; g < NGT; ++g) wv[g] = a.band_w[(long long)min(g, ng - 1) * nb + (band_lane ? lane : 0)];
    if (band_lane) {
#pragma unroll
      for (int g = 0; g < NGT; ++g) wts[g * nbp + lane] = g < ng ? wv[g] : 0.0;
    }
  }
  for (int k0 = 0; k0 < nl; k0 += LS) {
    const int nr = min(LS, nl - k0);
    {  // the slab in flight -> LDS
      const int n2 = (nr + 1) * nb2;
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (lane + 64 * i < n2) {
          *reinterpret_cast<d2*>(slab + dst[i]) = sr[i];
          *reinterpret_cast<d2*>(slab + SS + dst[i]) = sd[i];
          *reinterpret_cast<d2*>(slab + 2 * SS + dst[i]) = su[i];
        }
    }
    if (k0 == 0) {  // ... and the column it opens
      if (band_lane) wts[NGT * nbp + lane] = 1 - (c_lr + c_lt);  // :584
#pragma unroll
      for (int j = 0; j < NLAI; ++j)
        if (lane + 64 * j < nz) lai_s[lane + 64 * j] = c_lai[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: LDS operations complete in order; the slab is visible to all lanes
    if (k0 + LS < nl) fetch(k0 + LS);  // the next slab is in flight while this one is reduced
    if (k0 == 0) {  // level factors of the column (model.py:601-602, 617-621), lanes over layers
      const double G = (c_kind == CRT_G_TABLE) ? c_ga : G_closed(c_kind, c_gp, cos(c_psi), sin(c_psi));
      const double Kb = G / cos(c_psi);
      for (int k = lane; k < nl; k += 64) {
        const double l0 = lai_s[k], l1 = lai_s[k + 1];
        lvl[2 * k] = exp(-Kb * ((l0 + l1) / 2));
        lvl[2 * k + 1] = 1 - exp(-Kb * (l0 - l1));
      }
      if (a.totals) end_terms(0, ends);  // ground: transmitted I_d[0], soil-reflected I_df_u[0]  (diagnostics.py:476-530)
    }
    double v[2 * NGT];
#pragma unroll
    for (int j = 0; j < 2 * NGT; ++j) v[j] = 0.0;
    if (t < nr) {
      const double* r0p = slab + t * nbp;
      for (int b = p; b < nb; b += NP) {
        const double r0 = r0p[b], r1 = r0p[nbp + b];
        const double d0 = r0p[SS + b], d1 = r0p[SS + nbp + b];
        const double u0 = r0p[2 * SS + b], u1 = r0p[2 * SS + nbp + b];
        const double av = r1 - r0 + d1 - d0 + u0 - u1;  // :609
        const double ar = wts[NGT * nbp + b] * r1;       // :617-621 without the level factor
#pragma unroll
        for (int g = 0; g < NGT; ++g) {
          const double wg = wts[g * nbp + b];
          v[2 * g] += wg * av;
          v[2 * g + 1] += wg * ar;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < (2 * NGT + 3) / 4; ++j) {  // sum over the four p: afterwards DPP row p holds value 4 j + slot of its layer
      const double z = swap_add16(swap_add32(v[4 * j], 4 * j + 1 < 2 * NGT ? v[4 * j + 1] : 0.0),
                                  swap_add32(4 * j + 2 < 2 * NGT ? v[4 * j + 2] : 0.0, 4 * j + 3 < 2 * NGT ? v[4 * j + 3] : 0.0));
      if (t < nr && 4 * j + slot < 2 * NGT) raw[(k0 + t) * NGT * 2 + 4 * j + slot] = z;
    }
    if (k0 + LS >= nl) {  // the column is complete
      if (a.totals) end_terms(nr, ends + 2 * NGT);  // canopy top: incoming I_d[top], reflected I_df_u[top]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long ob = (long long)c * nl * ng;
      for (int i = lane; i < nl * ng; i += 64) {
        const int k = i / ng, g = i - k * ng;
        const double A = raw[(k * NGT + g) * 2], Dg = raw[(k * NGT + g) * 2 + 1];
        const double fsl = lvl[2 * k];
        const double adr = lvl[2 * k + 1] * Dg;  // :617-621
        const double adf = A - adr;              // :628
        __builtin_nontemporal_store(A, a.aI + ob + i);
        __builtin_nontemporal_store(adf * fsl + adr, a.aI_sl + ob + i);  // :631-633
        __builtin_nontemporal_store(adf * (1 - fsl), a.aI_sh + ob + i);
      }
      if (a.totals && lane < 4 * ng) {  // incoming, reflected, transmitted, soil-reflected
        const int g = lane >> 2, q = lane & 3;
        a.totals[((long long)c * ng + g) * 4 + q] = ends[(q < 2 ? 2 * NGT : 0) + 2 * g + (q & 1)];
      }
    }
  }
}

// k_absorb_bandsum_h: very narrow spectra (nb <= 32): a column per HALF wave (lane l of a
// half owns band l), so a wave instruction serves two columns instead of leaving half of the lanes idle;
// the reductions stay inside the halves (half_sum2).  Otherwise the same pass.
template <int NBT, int CH, int NGT>
__global__ __launch_bounds__(256) void k_absorb_bandsum_h(EpiArgs a, int wpb, int per_col) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
  const int c0 = 2 * (blockIdx.x * wpb + wave);
  if (c0 >= a.ncol) return;  // wave-uniform; no workgroup barrier in this kernel
  const bool colok = c0 + half < a.ncol;
  const int c = colok ? c0 + half : c0;  // a missing second column repeats the first and writes nothing
  const int nz = a.nz, nb = a.nb, ng = a.ngroup, nl = nz - 1;
  double* raw = lds + (size_t)(2 * wave + half) * per_col;  // [nl][NGT][2]
  double* ends = raw + 2 * NGT * nl;                        // [2][NGT][2]
  const long long cb = (long long)c * nz * nb;
  const double* __restrict__ R = a.I_dr + cb;
  const double* __restrict__ D = a.I_df_d + cb;
  const double* __restrict__ U = a.I_df_u + cb;
  int bi[NBT];
  double w[NBT][NGT], la[NBT], r0[NBT], d0[NBT], u0[NBT];
#pragma unroll
  for (int i = 0; i < NBT; ++i) {
    const int b = l + 32 * i;
    const bool act = b < nb;
    bi[i] = act ? b : 0;
    const long long ib = (long long)c * a.col_stride + bi[i];
    la[i] = act ? 1 - (a.leaf_r[ib] + a.leaf_t[ib]) : 0.0;  // :584
#pragma unroll
    for (int g = 0; g < NGT; ++g) w[i][g] = (g < ng && act) ? a.band_w[(long long)g * nb + bi[i]] : 0.0;
    r0[i] = R[bi[i]];
    d0[i] = D[bi[i]];
    u0[i] = U[bi[i]];
  }
  auto end_terms = [&](double* dst) {
    double v[2 * NGT];
#pragma unroll
    for (int g = 0; g < NGT; ++g) {
      v[2 * g] = v[2 * g + 1] = 0.0;
#pragma unroll
      for (int i = 0; i < NBT; ++i) {
        v[2 * g] += w[i][g] * (r0[i] + d0[i]);
        v[2 * g + 1] += w[i][g] * u0[i];
      }
    }
    half_sum_store(v, dst, 2 * NGT, lane);
  };
  if (a.totals) end_terms(ends);
  for (int k0 = 0; k0 < nl; k0 += CH) {
    const int nlev = min(CH, nl - k0);
    double r1[CH][NBT], d1[CH][NBT], u1[CH][NBT];
#pragma unroll
    for (int t = 0; t < CH; ++t)
      if (t < nlev) {
        const unsigned row = (unsigned)(k0 + t + 1) * (unsigned)nb;
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
          const unsigned off = row + (unsigned)bi[i];
          r1[t][i] = __builtin_nontemporal_load(R + off);
          d1[t][i] = __builtin_nontemporal_load(D + off);
          u1[t][i] = __builtin_nontemporal_load(U + off);
        }
      }
    double v[CH * NGT * 2];  // [t][g][A, D]
#pragma unroll
    for (int t = 0; t < CH; ++t) {
#pragma unroll
      for (int g = 0; g < NGT; ++g) v[(t * NGT + g) * 2] = v[(t * NGT + g) * 2 + 1] = 0.0;
      if (t < nlev) {
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
          const double av = r1[t][i] - r0[i] + d1[t][i] - d0[i] + u0[i] - u1[t][i];  // :609
          const double ar = la[i] * r1[t][i];
#pragma unroll
          for (int g = 0; g < NGT; ++g) {
            v[(t * NGT + g) * 2] += w[i][g] * av;
            v[(t * NGT + g) * 2 + 1] += w[i][g] * ar;
          }
          r0[i] = r1[t][i];
          d0[i] = d1[t][i];
          u0[i] = u1[t][i];
        }
      }
    }
    half_sum_store(v, raw + (size_t)k0 * NGT * 2, nlev * NGT * 2, lane);
  }
  if (a.totals) end_terms(ends + 2 * NGT);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!colok) return;
  bandsum_finish<NGT>(a, c, raw, ends, a.lai + (long long)c * nz, column_kb(a, c), l, 32);
}

// ------------------------------------------------------------------------------------------
// Per-band layer absorption, model.py:573-647: the seven (nz-1, nb) arrays + laim, f_slm of the reference's
// `Model.absorption` dict.  One workgroup per column, lanes over bands, previous level kept in registers.
struct AbsArgs {
  int ncol, nb, nz;
  long long col_stride;
  const double* psi;
  const double* lai;
  const int32_t* g_kind;
  const double* g_param;
  const double* g_at_psi;
  const double* leaf_r;
  const double* leaf_t;
  const double* I_dr;
  const double* I_df_d;
  const double* I_df_u;
  double* o[7];  // aI, aI_df, aI_dr, aI_sh, aI_sl, aI_df_sl, aI_df_sh
  double* laim;
  double* f_slm;
};

__global__ __launch_bounds__(256) void k_absorb(AbsArgs a) {
  const int c = blockIdx.x;
  const int nz = a.nz, nb = a.nb;
  const double psi = a.psi[c];
  const int kind = a.g_kind[c];
  const double G = (kind == CRT_G_TABLE) ? a.g_at_psi[c] : G_closed(kind, a.g_param ? a.g_param[c] : 0.0, cos(psi), sin(psi));
  const double Kb = G / cos(psi);
  const double* lai = a.lai + (long long)c * nz;
  const long long cb = (long long)c * nz * nb;
  const long long cm = (long long)c * (nz - 1) * nb;
  for (int b = threadIdx.x; b < nb; b += 256) {
    const double leaf_a = 1 - (a.leaf_r[(long long)c * a.col_stride + b] + a.leaf_t[(long long)c * a.col_stride + b]);  // :584
    double r0 = a.I_dr[cb + b], d0 = a.I_df_d[cb + b], u0 = a.I_df_u[cb + b];
    for (int k = 0; k < nz - 1; ++k) {
      const long long i1 = cb + (long long)(k + 1) * nb + b;
      const double r1 = a.I_dr[i1], d1 = a.I_df_d[i1], u1 = a.I_df_u[i1];
      const double dl = lai[k] - lai[k + 1];
      const double fsl = exp(-Kb * ((lai[k] + lai[k + 1]) / 2));   // :601-602
      const double av = r1 - r0 + d1 - d0 + u0 - u1;               // :609
      const double adr = r1 * (1 - exp(-Kb * dl)) * leaf_a;        // :617-621
      const double adf = av - adr;                                 // :628
      const double adfsl = adf * fsl, adfsh = adf * (1 - fsl);     // :631-632
      const long long o = cm + (long long)k * nb + b;
      a.o[0][o] = av;
      a.o[1][o] = adf;
      a.o[2][o] = adr;
      a.o[3][o] = adfsh;
      a.o[4][o] = adfsl + adr;
      a.o[5][o] = adfsl;
      a.o[6][o] = adfsh;
      if (b == 0) {
        a.laim[(long long)c * (nz - 1) + k] = (lai[k] + lai[k + 1]) / 2;
        a.f_slm[(long long)c * (nz - 1) + k] = fsl;
      }
      r0 = r1;
      d0 = d1;
      u0 = u1;
    }
  }
}

// k_absorb_tile (even nb, 16-byte aligned arrays): a column's seven outputs are each ONE contiguous run of (nz-1) nb doubles, so
// the kernel walks the flat element index with 16 bytes per lane -- every wave store is a contiguous, line-aligned 1 KiB whatever nb
// is (with lanes on bands a 300-band row starts and ends inside a 128-B line: 0.50 of the HBM peak).  The three input profiles go
// through an LDS ring of T + 1 rows: each round loads T new rows (one contiguous run per array, 16 bytes per lane), the row on top
// of the previous round stays where it is, so every input byte is read exactly once.  Level factors and (1 - r - t) come from LDS.
__global__ __launch_bounds__(256) void k_absorb_tile(AbsArgs a, int T) {
  extern __shared__ double lds[];
  const int c = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int nz = a.nz, nb = a.nb, nl = nz - 1, nb2 = nb >> 1, NS = T + 1;
  double* fsl = lds;           // [nl]
  double* absd = lds + nl;     // [nl]
  double* la = lds + 2 * nl;   // [nb]
  d2* ring = reinterpret_cast<d2*>(lds + 2 * nl + nb);  // [3][NS][nb2]
  const double psi = a.psi[c];
  const int kind = a.g_kind[c];
  const double G = (kind == CRT_G_TABLE) ? a.g_at_psi[c] : G_closed(kind, a.g_param ? a.g_param[c] : 0.0, cos(psi), sin(psi));
  const double Kb = G / cos(psi);
  const double* lai = a.lai + (long long)c * nz;
  for (int k = tid; k < nl; k += nthr) {
    const double lm = (lai[k] + lai[k + 1]) / 2;       // model.py:601
    const double f = exp(-Kb * lm);                    // :602
    fsl[k] = f;
    absd[k] = 1 - exp(-Kb * (lai[k] - lai[k + 1]));    // :619
    a.laim[(long long)c * nl + k] = lm;
    a.f_slm[(long long)c * nl + k] = f;
  }
  for (int b = tid; b < nb; b += nthr)
    la[b] = 1 - (a.leaf_r[(long long)c * a.col_stride + b] + a.leaf_t[(long long)c * a.col_stride + b]);  // :584
  const long long cb = (long long)c * nz * nb, cm = (long long)c * nl * nb;
  const d2* R2 = reinterpret_cast<const d2*>(a.I_dr + cb);
  const d2* D2 = reinterpret_cast<const d2*>(a.I_df_d + cb);
  const d2* U2 = reinterpret_cast<const d2*>(a.I_df_u + cb);
  const int dt = nthr / nb2, dp = nthr - dt * nb2;       // one round of the workgroup advances (dt rows, dp pairs)
  const int t00 = tid / nb2, p00 = tid - t00 * nb2;
  // row 0 of the column into slot 0
  for (int p = tid; p < nb2; p += nthr) {
    ring[p] = R2[p];
    ring[NS * nb2 + p] = D2[p];
    ring[2 * NS * nb2 + p] = U2[p];
  }
  int base = 0;  // ring slot of the row below the current chunk (level k0)
  for (int k0 = 0; k0 < nl; k0 += T) {
    const int nlev = min(T, nl - k0), n2 = nlev * nb2;
    // rows k0+1 .. k0+nlev -> slots base+1 .. base+nlev (mod NS): one contiguous run of nlev * nb doubles per array
    {
      const long long g0 = (long long)(k0 + 1) * nb2;
      int t = t00, p = p00;
      for (int i2 = tid; i2 < n2; i2 += nthr) {
        int slot = base + 1 + t;
        if (slot >= NS) slot -= NS;
        const int li = slot * nb2 + p;
        ring[li] = __builtin_nontemporal_load(R2 + g0 + i2);
        ring[NS * nb2 + li] = __builtin_nontemporal_load(D2 + g0 + i2);
        ring[2 * NS * nb2 + li] = __builtin_nontemporal_load(U2 + g0 + i2);
        p += dp;
        t += dt;
        if (p >= nb2) {
          p -= nb2;
          ++t;
        }
      }
    }
    __syncthreads();
    {
      int t = t00, p = p00;
      const long long o0 = (cm >> 1) + (long long)k0 * nb2;
      for (int i2 = tid; i2 < n2; i2 += nthr) {
        int s0 = base + t;
        if (s0 >= NS) s0 -= NS;
        int s1 = s0 + 1;
        if (s1 >= NS) s1 -= NS;
        const int l0 = s0 * nb2 + p, l1 = s1 * nb2 + p;
        const d2 r0 = ring[l0], r1 = ring[l1], d0 = ring[NS * nb2 + l0], d1 = ring[NS * nb2 + l1], u0 = ring[2 * NS * nb2 + l0],
                 u1 = ring[2 * NS * nb2 + l1];
        const double f = fsl[k0 + t], ab = absd[k0 + t];
        const d2 l = reinterpret_cast<const d2*>(la)[p];
        d2 av, adr, adf, sl, dsl, dsh;
        av.x = r1.x - r0.x + d1.x - d0.x + u0.x - u1.x;   // :609
        av.y = r1.y - r0.y + d1.y - d0.y + u0.y - u1.y;
        adr.x = r1.x * ab * l.x;                          // :617-621
        adr.y = r1.y * ab * l.y;
        adf.x = av.x - adr.x;                             // :628
        adf.y = av.y - adr.y;
        dsl.x = adf.x * f;                                // :631-632
        dsl.y = adf.y * f;
        dsh.x = adf.x * (1 - f);
        dsh.y = adf.y * (1 - f);
        sl.x = dsl.x + adr.x;
        sl.y = dsl.y + adr.y;
        const long long o = o0 + i2;
        __builtin_nontemporal_store(av, reinterpret_cast<d2*>(a.o[0]) + o);
        __builtin_nontemporal_store(adf, reinterpret_cast<d2*>(a.o[1]) + o);
        __builtin_nontemporal_store(adr, reinterpret_cast<d2*>(a.o[2]) + o);
        __builtin_nontemporal_store(dsh, reinterpret_cast<d2*>(a.o[3]) + o);
        __builtin_nontemporal_store(sl, reinterpret_cast<d2*>(a.o[4]) + o);
        __builtin_nontemporal_store(dsl, reinterpret_cast<d2*>(a.o[5]) + o);
        __builtin_nontemporal_store(dsh, reinterpret_cast<d2*>(a.o[6]) + o);
        p += dp;
        t += dt;
        if (p >= nb2) {
          p -= nb2;
          ++t;
        }
      }
    }
    base += nlev;
    if (base >= NS) base -= NS;
    lds_barrier();  // all reads of this chunk's slots are done before the next round overwrites them (LDS only: the stores keep flowing)
  }
}

// ------------------------------------------------------------------------------------------
// bandwidth probes: plain 16-B-per-lane streaming fill / copy, grid-stride
// out[row][g] = sum_b w[g][b] X[row][b]: diagnostics.band's reduction (diagnostics.py:81) for ANY variable with a trailing wavelength
// axis -- one wave per row, lanes over bands, four group totals per set of lane exchanges (wave_sum4)
__global__ __launch_bounds__(256) void k_band_reduce(const double* __restrict__ X, long long nrow, int nb, const double* __restrict__ w, int ng,
                                                     double* __restrict__ out) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= nrow) return;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double* x = X + row * nb;
  for (int b = lane; b < nb; b += 64) {
    const double v = x[b];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (g < ng) acc[g] += w[(long long)g * nb + b] * v;
  }
  const double z = wave_sum4(acc[0], acc[1], acc[2], acc[3]);
  const int g = wave_sum4_slot(lane >> 4);
  if ((lane & 15) == 0 && g < ng) out[row * ng + g] = z;
}

__global__ __launch_bounds__(256) void k_probe_math(const double* x, size_t n, double* e, double* sn, double* cs) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  e[i] = fexp(x[i]);
  double s_, c_;
  fast_sincos(x[i], s_, c_);
  sn[i] = s_;
  cs[i] = c_;
}

__global__ __launch_bounds__(256) void k_fill(d2* dst, size_t n2, double v) {
  d2 t;
  t.x = v;
  t.y = v;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(t, dst + i);
}

__global__ __launch_bounds__(256) void k_copy(d2* dst, const d2* src, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// store-pattern probe: what the solve kernels' flush does and nothing else -- workgroup c writes, for column c of EVERY array of the
// set, run after run of `run` doubles (T levels x nb bands) with 16-byte streaming stores, all arrays in step.  Run on a Plan's own
// output set it measures the store rate that placement allows (the linear fill above writes one array, i.e. one memory class, at a time).
struct StoreSetArgs {
  double* p[8];
  int na;
  long long ncol, col;  // columns, doubles per column
  int run;              // doubles per run (a multiple of 2)
};
__global__ __launch_bounds__(192) void k_store_set(StoreSetArgs a, double v) {
  d2 t;
  t.x = v;
  t.y = v;
  const long long c = blockIdx.x;
  const long long base = c * a.col;
  for (long long r0 = 0; r0 < a.col; r0 += a.run) {
    const int n2 = (int)(min((long long)a.run, a.col - r0) >> 1);
    for (int k = 0; k < a.na; ++k) {
      d2* dst = reinterpret_cast<d2*>(a.p[k] + base + r0);
      for (int i = threadIdx.x; i < n2; i += 192) __builtin_nontemporal_store(t, dst + i);
    }
  }
}

bool scheme_ok(int s) { return s >= 0 && s < CRT_NUM_SCHEMES; }

}  // namespace
}  // namespace crt

using namespace crt;

extern "C" {

int crt_hip_abi_version(void) { return CRT_ABI_VERSION; }

const char* crt_hip_strerror(int st) {
  switch (st) {
    case CRT_OK: return "ok";
    case CRT_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or invalid option)";
    case CRT_ERR_WORKSPACE: return "workspace too small (see crt_hip_workspace_bytes)";
    case CRT_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels (nz too large for LDS, or grid overflow)";
    case CRT_ERR_LAUNCH: return "HIP runtime / launch error";
    case CRT_ERR_SHAPE: return "shape violates a reference assertion";
    default: return "unknown status";
  }
}

size_t crt_hip_workspace_bytes(int scheme, int32_t ncol, int32_t nz) {
  if (!scheme_ok(scheme) || ncol <= 0 || nz <= 0) return 0;
  return (size_t)ncol * (size_t)rec_len(scheme, nz) * sizeof(double);
}

size_t crt_hip_workspace_bytes_nb(int scheme, int32_t ncol, int32_t nz, int32_t nb) {
  size_t n = crt_hip_workspace_bytes(scheme, ncol, nz);
  if (n == 0 || nb <= 0) return 0;
  if (scheme == CRT_SCHEME_ZQ_PA) n += 2 * (size_t)ncol * (size_t)zqpa_M(nz) * (size_t)nb * sizeof(double);
  return n;
}

int crt_hip_quad_nodes(double mu_s, double* psi_nodes) {
  if (!psi_nodes || !(mu_s > 0.0 && mu_s < 1.0)) return CRT_ERR_BAD_ARG;
  host_quad_nodes(mu_s, psi_nodes);
  return CRT_OK;
}

static int solve_impl(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                      const crt_outputs* out, void* workspace, size_t workspace_bytes, crt_stream_t stream, int f32,
                      const IntArgs* integ = nullptr) {
  if (!scheme_ok(scheme) || !cols || !bands || !out) return CRT_ERR_BAD_ARG;
  const int ncol = cols->ncol, nz = cols->nz, nb = bands->nb;
  if (ncol <= 0 || nz <= 0 || nb <= 0) return CRT_ERR_BAD_ARG;
  if (!cols->psi || !cols->lai || !cols->g_kind) return CRT_ERR_BAD_ARG;
  if (scheme == CRT_SCHEME_2S && !cols->mla) return CRT_ERR_BAD_ARG;
  if (!bands->I_dr0 || !bands->I_df0 || !bands->leaf_r || !bands->leaf_t) return CRT_ERR_BAD_ARG;
  if (scheme != CRT_SCHEME_BL && !bands->soil_r) return CRT_ERR_BAD_ARG;
  if (bands->col_stride != 0 && bands->col_stride < nb) return CRT_ERR_BAD_ARG;
  if (!integ && (!out->I_dr || !out->I_df_d || !out->I_df_u || !out->F)) return CRT_ERR_BAD_ARG;
  const bool tri = scheme == CRT_SCHEME_N79 || scheme == CRT_SCHEME_ZQ;
  const int nextra = scheme == CRT_SCHEME_N79 ? 2 : (scheme == CRT_SCHEME_ZQ || scheme == CRT_SCHEME_G77 || scheme == CRT_SCHEME_BF) ? 3 : 0;
  if (!integ) {
    if (nextra >= 1 && !out->x0) return CRT_ERR_BAD_ARG;
    if (nextra >= 2 && !out->x1) return CRT_ERR_BAD_ARG;
    if (nextra >= 3 && !out->x2) return CRT_ERR_BAD_ARG;
  }
  if (nz < 2) return CRT_ERR_SHAPE;
  if (scheme == CRT_SCHEME_N79 && nz < 3) return CRT_ERR_SHAPE;  // td[1]/tb[1] of _solve_n79.py:85-92
  double mu_s = 0.501;
  int method = CRT_TAU_D_QUAD, flags = 0;
  int tune[CRT_NTUNE] = {};
  if (opts) {
    mu_s = opts->mu_s;
    method = opts->tau_d_method;
    flags = opts->flags;
    for (int i = 0; i < CRT_NTUNE; ++i) tune[i] = opts->tune[i];
  }
  {  // crt_options.tune is a measurement aid, but it is part of the ABI: out-of-range values are rejected here, before any launch,
     // instead of reaching the kernel configurations (keys: crt_internal.hpp, SolveArgs::tune)
    static const struct { int key, lo, hi; } range[] = {
        {0, 0, 160 * 1024}, {1, 0, 64}, {2, 0, 255}, {3, 0, 12}, {4, 0, 32}, {8, 0, 16}, {9, 0, 12}, {10, 0, 7}, {11, 0, 12}, {12, 0, 1024}, {13, 0, 3}, {5, 0, 2}, {6, 0, 4}};
    bool known[CRT_NTUNE] = {};
    for (const auto& r : range) {
      known[r.key] = true;
      if (tune[r.key] < r.lo || tune[r.key] > r.hi) return CRT_ERR_BAD_ARG;
    }
    if (tune[8] != 0 && tune[8] != 8 && tune[8] != 12 && tune[8] != 16) return CRT_ERR_BAD_ARG;
    if (tune[9] != 0 && tune[9] != 4 && tune[9] != 8 && tune[9] != 12) return CRT_ERR_BAD_ARG;
    for (int i = 0; i < CRT_NTUNE; ++i)
      if (!known[i] && tune[i] != 0) return CRT_ERR_BAD_ARG;  // reserved keys stay zero
  }
  if (scheme == CRT_SCHEME_4S && !(mu_s > 0.0 && mu_s < 1.0)) return CRT_ERR_BAD_ARG;
  if (method != CRT_TAU_D_QUAD && method != CRT_TAU_D_9SKY) return CRT_ERR_BAD_ARG;  // ValueError, common.py:78
  const size_t need = crt_hip_workspace_bytes_nb(scheme, ncol, nz, nb);
  if (!workspace || workspace_bytes < need) return CRT_ERR_WORKSPACE;

  hipStream_t s = static_cast<hipStream_t>(stream);
  ColArgs ca;
  ca.ncol = ncol;
  ca.nz = nz;
  ca.scheme = scheme;
  ca.tau_d_method = method;
  ca.mu_s = mu_s;
  ca.psi = cols->psi;
  ca.lai = cols->lai;
  ca.mla = cols->mla;
  ca.g_kind = cols->g_kind;
  ca.g_param = cols->g_param;
  ca.g_at_psi = cols->g_at_psi;
  ca.g_table = cols->g_table;
  ca.ws = static_cast<double*>(workspace);
  if (!(flags & CRT_FLAG_SKIP_PRECOMPUTE)) {
    int st = launch_colpre(ca, s);
    if (st != CRT_OK) return st;
  }
  if (flags & CRT_FLAG_PRECOMPUTE_ONLY) return CRT_OK;

  SolveArgs sa;
  sa.ncol = ncol;
  sa.nb = nb;
  sa.nz = nz;
  sa.reclen = rec_len(scheme, nz);
  sa.col_stride = bands->col_stride;
  sa.ws = static_cast<const double*>(workspace);
  sa.I_dr0 = bands->I_dr0;
  sa.I_df0 = bands->I_df0;
  sa.leaf_r = bands->leaf_r;
  sa.leaf_t = bands->leaf_t;
  sa.soil_r = bands->soil_r;
  sa.o[0] = out->I_dr;
  sa.o[1] = out->I_df_d;
  sa.o[2] = out->I_df_u;
  sa.o[3] = out->F;
  sa.o[4] = out->x0;
  sa.o[5] = out->x1;
  sa.o[6] = out->x2;
  sa.mu_s = mu_s;
  sa.f32 = f32;
  for (int i = 0; i < CRT_NTUNE; ++i) sa.tune[i] = tune[i];
  if (integ) {
    if (scheme == CRT_SCHEME_ZQ_PA) return CRT_ERR_UNSUPPORTED;
    return tri ? launch_tridiag_int(scheme, sa, *integ, s) : launch_closed_int(scheme, sa, *integ, s);
  }
  if (scheme == CRT_SCHEME_ZQ_PA)
    return launch_zqpa(sa, static_cast<double*>(workspace) + (size_t)ncol * sa.reclen, s);
  const int force = (flags & CRT_FLAG_DIRECT_STORES) ? 1 : 0;
  return tri ? launch_tridiag(scheme, sa, s, force) : launch_closed(scheme, sa, s, force);
}


int crt_hip_solve_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                      const crt_outputs* out, void* workspace, size_t workspace_bytes, crt_stream_t stream) {
  return solve_impl(scheme, cols, bands, opts, out, workspace, workspace_bytes, stream, 0);
}

// the f32 structs have the layout of the f64 ones (pointers + sizes); only the element type behind the pointers differs
int crt_hip_solve_f32(int scheme, const crt_columns* cols, const crt_bands_f32* bands, const crt_options* opts,
                      const crt_outputs_f32* out, void* workspace, size_t workspace_bytes, crt_stream_t stream) {
  static_assert(sizeof(crt_bands_f32) == sizeof(crt_bands) && sizeof(crt_outputs_f32) == sizeof(crt_outputs), "layout");
  return solve_impl(scheme, cols, reinterpret_cast<const crt_bands*>(bands), opts, reinterpret_cast<const crt_outputs*>(out), workspace,
                    workspace_bytes, stream, 1);
}

#define CRT_ENTRY(name, id)                                                                                        \
  int name(const crt_columns* c, const crt_bands* b, const crt_options* o, const crt_outputs* out, void* ws,       \
           size_t wsb, crt_stream_t s) {                                                                           \
    return crt_hip_solve_f64(id, c, b, o, out, ws, wsb, s);                                                        \
  }
CRT_ENTRY(crt_hip_2s_f64, CRT_SCHEME_2S)
CRT_ENTRY(crt_hip_4s_f64, CRT_SCHEME_4S)
CRT_ENTRY(crt_hip_n79_f64, CRT_SCHEME_N79)
CRT_ENTRY(crt_hip_zq_f64, CRT_SCHEME_ZQ)
CRT_ENTRY(crt_hip_bl_f64, CRT_SCHEME_BL)
CRT_ENTRY(crt_hip_g77_f64, CRT_SCHEME_G77)
CRT_ENTRY(crt_hip_bf_f64, CRT_SCHEME_BF)
CRT_ENTRY(crt_hip_zq_pa_f64, CRT_SCHEME_ZQ_PA)
#undef CRT_ENTRY

#define CRT_ENTRY32(name, id)                                                                                      \
  int name(const crt_columns* c, const crt_bands_f32* b, const crt_options* o, const crt_outputs_f32* out, void* ws, \
           size_t wsb, crt_stream_t s) {                                                                           \
    return crt_hip_solve_f32(id, c, b, o, out, ws, wsb, s);                                                        \
  }
CRT_ENTRY32(crt_hip_2s_f32, CRT_SCHEME_2S)
CRT_ENTRY32(crt_hip_4s_f32, CRT_SCHEME_4S)
CRT_ENTRY32(crt_hip_n79_f32, CRT_SCHEME_N79)
CRT_ENTRY32(crt_hip_zq_f32, CRT_SCHEME_ZQ)
CRT_ENTRY32(crt_hip_bl_f32, CRT_SCHEME_BL)
CRT_ENTRY32(crt_hip_g77_f32, CRT_SCHEME_G77)
CRT_ENTRY32(crt_hip_bf_f32, CRT_SCHEME_BF)
CRT_ENTRY32(crt_hip_zq_pa_f32, CRT_SCHEME_ZQ_PA)
#undef CRT_ENTRY32

int crt_hip_absorb_bandsum2_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                                const double* I_df_u, const double* band_w, int32_t ngroup, const crt_bandsum_out* out, crt_stream_t stream) {
  if (!cols || !bands || !I_dr || !I_df_d || !I_df_u || !band_w || !out || !out->aI || !out->aI_sl || !out->aI_sh) return CRT_ERR_BAD_ARG;
  double *aI = out->aI, *aI_sl = out->aI_sl, *aI_sh = out->aI_sh, *totals = out->totals;
  // the optional outputs come all together or not at all: the direct-beam part of the absorption + the five level profiles
  const int nopt = (out->aI_dr != nullptr) + (out->I_dr != nullptr) + (out->I_df_d != nullptr) + (out->I_df_u != nullptr) + (out->F != nullptr) +
                   (out->I_d != nullptr);
  if (nopt != 0 && nopt != 6) return CRT_ERR_BAD_ARG;
  const bool prof = nopt == 6;
  if (cols->ncol <= 0 || cols->nz < 2 || bands->nb <= 0 || ngroup <= 0 || ngroup > MAXG) return CRT_ERR_BAD_ARG;
  if (!cols->psi || !cols->lai || !cols->g_kind || !bands->leaf_r || !bands->leaf_t) return CRT_ERR_BAD_ARG;
  EpiArgs a;
  a.ncol = cols->ncol;
  a.nb = bands->nb;
  a.nz = cols->nz;
  a.ngroup = ngroup;
  a.col_stride = bands->col_stride;
  a.psi = cols->psi;
  a.lai = cols->lai;
  a.g_kind = cols->g_kind;
  a.g_param = cols->g_param;
  a.g_at_psi = cols->g_at_psi;
  a.leaf_r = bands->leaf_r;
  a.leaf_t = bands->leaf_t;
  a.I_dr = I_dr;
  a.I_df_d = I_df_d;
  a.I_df_u = I_df_u;
  a.band_w = band_w;
  a.aI = aI;
  a.aI_sl = aI_sl;
  a.aI_sh = aI_sh;
  a.totals = totals;
  a.aI_dr = out->aI_dr;
  a.L_dr = out->I_dr;
  a.L_dn = out->I_df_d;
  a.L_up = out->I_df_u;
  a.L_F = out->F;
  a.L_Id = out->I_d;
  // measured (tools/epilogue_bench.py): nb = 20: 1.15 ms per wave-column vs 0.85 ms per half-wave-column; nb = 38: 1.11 vs 1.26 (the second band
  // slot of a half is nearly empty and doubles the per-band work) -> halves only up to 32 bands
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(I_dr) | reinterpret_cast<uintptr_t>(I_df_d) | reinterpret_cast<uintptr_t>(I_df_u)) & 15) == 0;
  // measured (tools/bandsum_shapes.py, 9.1 GB of profiles): lanes over layers vs lanes over bands (one band per lane, next chunk prefetched):
  //   1e5 x 34 x 100: 1.69 vs 1.83 ms;  1e5 x 38 x 100: 1.89 vs 2.09;  1e5 x 48 x 80: 1.85 vs 1.88;  1e5 x 64 x 60: 2.02 vs 1.66 -> up to 48 bands
  // (with the level profiles requested the band-lane kernels below serve every width: they hold each level's values in registers anyway)
  if (!prof && a.nb > 32 && a.nb <= 48 && a.nb % 2 == 0 && aligned16 && a.nz <= 257 && (long long)a.nz * a.nb < (1ll << 31)) {  // lanes over layers
    const int nl = a.nz - 1;
    const int ngt = a.ngroup == 1 ? 1 : a.ngroup <= 3 ? 3 : 4;
    const int nbp = (a.nb % 4 == 2) ? a.nb : a.nb + 2;
    const size_t shl = (size_t)(3 * 17 * nbp + (ngt + 1) * nbp + 2 * ngt * nl + 4 * ngt + a.nz + 2 * nl) * sizeof(double);
    if (shl <= 64 * 1024) {
      hipStream_t sl = static_cast<hipStream_t>(stream);
      const dim3 gl(a.ncol);
      if (ngt == 1) hipLaunchKernelGGL((k_absorb_bandsum_l<1>), gl, dim3(64), shl, sl, a, nbp);
      else if (ngt == 3) hipLaunchKernelGGL((k_absorb_bandsum_l<3>), gl, dim3(64), shl, sl, a, nbp);
      else hipLaunchKernelGGL((k_absorb_bandsum_l<4>), gl, dim3(64), shl, sl, a, nbp);
      return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
    }
  }
  if (!prof && a.nb <= 32 && (long long)a.nz * a.nb < (1ll << 31)) {  // a column per half wave
    const int nl = a.nz - 1;
    const int ngt = a.ngroup == 1 ? 1 : a.ngroup <= 3 ? 3 : 4;
    const int per_col = 2 * ngt * nl + 4 * ngt;
    int wpb = 4;
    while (wpb > 1 && (size_t)2 * wpb * per_col * sizeof(double) > 60 * 1024) wpb >>= 1;
    const size_t shw = (size_t)2 * wpb * per_col * sizeof(double);
    if (shw <= 64 * 1024) {
      const dim3 grid((a.ncol + 2 * wpb - 1) / (2 * wpb)), block(64 * wpb);
      hipStream_t sw = static_cast<hipStream_t>(stream);
      auto launch = [&](auto ngt_c) {
        constexpr int NGT = decltype(ngt_c)::value;
        if (a.nb <= 32) hipLaunchKernelGGL((k_absorb_bandsum_h<1, 4, NGT>), grid, block, shw, sw, a, wpb, per_col);
        else hipLaunchKernelGGL((k_absorb_bandsum_h<2, 4, NGT>), grid, block, shw, sw, a, wpb, per_col);
      };
      if (a.ngroup == 1) launch(std::integral_constant<int, 1>{});
      else if (a.ngroup <= 3) launch(std::integral_constant<int, 3>{});
      else launch(std::integral_constant<int, 4>{});
      return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
    }
  }
  if (a.nb <= 512 && (long long)a.nz * a.nb < (1ll << 31)) {  // one wave per column
    const int nl = a.nz - 1;
    const int ngt = a.ngroup == 1 ? 1 : a.ngroup <= 3 ? 3 : 4;
    const int per_wave = 2 * ngt * nl + 4 * ngt + (prof ? 3 * ngt * a.nz : 0);
    int wpb = 4;
    while (wpb > 1 && (size_t)wpb * per_wave * sizeof(double) > 60 * 1024) wpb >>= 1;
    const size_t shw = (size_t)wpb * per_wave * sizeof(double);
    if (shw <= 64 * 1024) {
      const int nbt = (a.nb + 63) / 64;
      const dim3 grid((a.ncol + wpb - 1) / wpb), block(64 * wpb);
      hipStream_t sw = static_cast<hipStream_t>(stream);
      auto launch_p = [&](auto ngt) {  // with the level profiles (chunks of two levels: the extra sums live in registers too)
        constexpr int NGT = decltype(ngt)::value;
        switch (nbt) {
          case 1: hipLaunchKernelGGL((k_absorb_bandsum_w<1, 2, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 2: hipLaunchKernelGGL((k_absorb_bandsum_w<2, 2, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 3: hipLaunchKernelGGL((k_absorb_bandsum_w<3, 2, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 4: hipLaunchKernelGGL((k_absorb_bandsum_w<4, 2, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 5: hipLaunchKernelGGL((k_absorb_bandsum_w<5, 2, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 6: hipLaunchKernelGGL((k_absorb_bandsum_w<6, 1, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          default: hipLaunchKernelGGL((k_absorb_bandsum_w<8, 1, NGT, false, true>), grid, block, shw, sw, a, wpb, per_wave); break;
        }
      };
      if (prof) {
        if (a.ngroup == 1) launch_p(std::integral_constant<int, 1>{});
        else if (a.ngroup <= 3) launch_p(std::integral_constant<int, 3>{});
        else launch_p(std::integral_constant<int, 4>{});
        return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
      }
      auto launch = [&](auto ngt) {
        constexpr int NGT = decltype(ngt)::value;
        switch (nbt) {
          case 1: hipLaunchKernelGGL((k_absorb_bandsum_w<1, 4, NGT, true>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 2: hipLaunchKernelGGL((k_absorb_bandsum_w<2, 4, NGT>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 3: hipLaunchKernelGGL((k_absorb_bandsum_w<3, 4, NGT>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 4: hipLaunchKernelGGL((k_absorb_bandsum_w<4, 2, NGT>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 5: hipLaunchKernelGGL((k_absorb_bandsum_w<5, 2, NGT>), grid, block, shw, sw, a, wpb, per_wave); break;
          case 6: hipLaunchKernelGGL((k_absorb_bandsum_w<6, 2, NGT>), grid, block, shw, sw, a, wpb, per_wave); break;
          default: hipLaunchKernelGGL((k_absorb_bandsum_w<8, 2, NGT>), grid, block, shw, sw, a, wpb, per_wave); break;
        }
      };
      if (a.ngroup == 1) launch(std::integral_constant<int, 1>{});
      else if (a.ngroup <= 3) launch(std::integral_constant<int, 3>{});
      else launch(std::integral_constant<int, 4>{});
      return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
    }
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int b0 = 0; b0 < a.nb; b0 += 1024) {  // one launch per 1024 bands (the usual case: one)
    const int nbs = std::min(1024, a.nb - b0);
    const int nthr = ((nbs + 63) / 64) * 64;
    const size_t sh = ((size_t)2 * BS_CH * (nthr / 64) * (prof ? 5 : 2) * MAXG + (size_t)(nthr / 64) * (4 + (prof ? 3 : 0)) * MAXG) * sizeof(double);
    if (prof) {
      if (nthr <= 256) hipLaunchKernelGGL((k_absorb_bandsum<256, true>), dim3(a.ncol), dim3(nthr), sh, s, a, b0, nbs, b0 > 0);
      else if (nthr <= 512) hipLaunchKernelGGL((k_absorb_bandsum<512, true>), dim3(a.ncol), dim3(nthr), sh, s, a, b0, nbs, b0 > 0);
      else hipLaunchKernelGGL((k_absorb_bandsum<1024, true>), dim3(a.ncol), dim3(nthr), sh, s, a, b0, nbs, b0 > 0);
    } else {
      if (nthr <= 256) hipLaunchKernelGGL((k_absorb_bandsum<256, false>), dim3(a.ncol), dim3(nthr), sh, s, a, b0, nbs, b0 > 0);
      else if (nthr <= 512) hipLaunchKernelGGL((k_absorb_bandsum<512, false>), dim3(a.ncol), dim3(nthr), sh, s, a, b0, nbs, b0 > 0);
      else hipLaunchKernelGGL((k_absorb_bandsum<1024, false>), dim3(a.ncol), dim3(nthr), sh, s, a, b0, nbs, b0 > 0);
    }
  }
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_absorb_bandsum_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                               const double* I_df_u, const double* band_w, int32_t ngroup, double* aI, double* aI_sl,
                               double* aI_sh, double* totals, crt_stream_t stream) {
  crt_bandsum_out o = {};
  o.aI = aI;
  o.aI_sl = aI_sl;
  o.aI_sh = aI_sh;
  o.totals = totals;
  return crt_hip_absorb_bandsum2_f64(cols, bands, I_dr, I_df_d, I_df_u, band_w, ngroup, &o, stream);
}

int crt_hip_integrated2_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                            const double* band_w, int32_t ngroup, const crt_bandsum_out* out, void* workspace, size_t workspace_bytes,
                            crt_stream_t stream) {
  if (!band_w || !out || !out->aI || !out->aI_sl || !out->aI_sh || ngroup <= 0 || ngroup > INT_MAXG || !cols) return CRT_ERR_BAD_ARG;
  const int nopt = (out->aI_dr != nullptr) + (out->I_dr != nullptr) + (out->I_df_d != nullptr) + (out->I_df_u != nullptr) + (out->F != nullptr) +
                   (out->I_d != nullptr);
  if (nopt != 0 && nopt != 6) return CRT_ERR_BAD_ARG;
  IntArgs ia;
  ia.lai = cols->lai;
  ia.band_w = band_w;
  ia.ngroup = ngroup;
  ia.aI = out->aI;
  ia.aI_sl = out->aI_sl;
  ia.aI_sh = out->aI_sh;
  ia.totals = out->totals;
  ia.aI_dr = out->aI_dr;
  ia.L_dr = out->I_dr;
  ia.L_dn = out->I_df_d;
  ia.L_up = out->I_df_u;
  ia.L_F = out->F;
  ia.L_Id = out->I_d;
  crt_outputs none = {};
  return solve_impl(scheme, cols, bands, opts, &none, workspace, workspace_bytes, stream, 0, &ia);
}

int crt_hip_integrated_f64(int scheme, const crt_columns* cols, const crt_bands* bands, const crt_options* opts,
                           const double* band_w, int32_t ngroup, double* aI, double* aI_sl, double* aI_sh, double* totals,
                           void* workspace, size_t workspace_bytes, crt_stream_t stream) {
  crt_bandsum_out o = {};
  o.aI = aI;
  o.aI_sl = aI_sl;
  o.aI_sh = aI_sh;
  o.totals = totals;
  return crt_hip_integrated2_f64(scheme, cols, bands, opts, band_w, ngroup, &o, workspace, workspace_bytes, stream);
}

int crt_hip_band_reduce_f64(const double* X, int64_t nrow, int32_t nb, const double* band_w, int32_t ngroup, double* out, crt_stream_t stream) {
  if (!X || !band_w || !out || nrow < 0 || nb <= 0 || ngroup <= 0 || ngroup > 4) return CRT_ERR_BAD_ARG;
  if (nrow == 0) return CRT_OK;
  const long long nblk = (nrow + 3) / 4;
  if (nblk > 0x7fffffffLL) return CRT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_band_reduce, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream), X, (long long)nrow, nb, band_w, ngroup, out);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_absorb_f64(const crt_columns* cols, const crt_bands* bands, const double* I_dr, const double* I_df_d,
                       const double* I_df_u, double* const* out7, double* laim, double* f_slm, crt_stream_t stream) {
  if (!cols || !bands || !I_dr || !I_df_d || !I_df_u || !out7 || !laim || !f_slm) return CRT_ERR_BAD_ARG;
  if (cols->ncol <= 0 || cols->nz < 2 || bands->nb <= 0) return CRT_ERR_BAD_ARG;
  if (!cols->psi || !cols->lai || !cols->g_kind || !bands->leaf_r || !bands->leaf_t) return CRT_ERR_BAD_ARG;
  AbsArgs a;
  a.ncol = cols->ncol;
  a.nb = bands->nb;
  a.nz = cols->nz;
  a.col_stride = bands->col_stride;
  a.psi = cols->psi;
  a.lai = cols->lai;
  a.g_kind = cols->g_kind;
  a.g_param = cols->g_param;
  a.g_at_psi = cols->g_at_psi;
  a.leaf_r = bands->leaf_r;
  a.leaf_t = bands->leaf_t;
  a.I_dr = I_dr;
  a.I_df_d = I_df_d;
  a.I_df_u = I_df_u;
  for (int i = 0; i < 7; ++i) {
    if (!out7[i]) return CRT_ERR_BAD_ARG;
    a.o[i] = out7[i];
  }
  a.laim = laim;
  a.f_slm = f_slm;
  bool flat = a.nb % 2 == 0 && a.col_stride % 2 == 0 && (long long)a.nz * a.nb < (1ll << 31);
  const void* ptrs[] = {I_dr, I_df_d, I_df_u, a.o[0], a.o[1], a.o[2], a.o[3], a.o[4], a.o[5], a.o[6]};
  for (const void* q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) & 15) flat = false;
  // ring of T + 1 rows of the three inputs: T rows per round, as many as keep the workgroup at ~40 KB of LDS (4 per CU)
  int T = (int)((40 * 1024 / sizeof(double) - 2 * (a.nz - 1) - a.nb) / (3 * (size_t)a.nb)) - 1;
  T = std::max(1, std::min(T, std::min(16, a.nz - 1)));
  const size_t shf = (size_t)(2 * (a.nz - 1) + a.nb + 3 * (T + 1) * a.nb) * sizeof(double);
  if (flat && a.nb >= 2 && shf <= 64 * 1024)
    hipLaunchKernelGGL(k_absorb_tile, dim3(a.ncol), dim3(256), shf, static_cast<hipStream_t>(stream), a, T);
  else
    hipLaunchKernelGGL(k_absorb, dim3(a.ncol), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_tau_d_f64(const double* kb_nodes, const double* L, int64_t n, int32_t method, double* out, crt_stream_t stream) {
  if (!kb_nodes || !L || !out || n < 0) return CRT_ERR_BAD_ARG;
  if (method != CRT_TAU_D_QUAD && method != CRT_TAU_D_9SKY) return CRT_ERR_BAD_ARG;  // ValueError, common.py:78
  if (n == 0) return CRT_OK;
  return crt::launch_tau_d(kb_nodes, L, (long long)n, method, out, static_cast<hipStream_t>(stream));
}

const char* crt_hip_last_kernel(void) { return last_kernel(); }

// The fill probe is a YARDSTICK, not a ceiling: a linear fill writes one narrow window of memory at a time, i.e. into one class
// of physical memory (csrc/buffers.hip), and is subject to the same single-class limit as the column pattern -- 6.2-6.6 TB/s with
// this 256-workgroup grid, 4.2-4.5 TB/s with 2048 workgroups (tools/chunk_probe.hip: the wider window of the larger grid is worse,
// so the grid was NOT enlarged) -- while the solve kernels' pattern reaches 7.0 TB/s into a class-balanced set of arrays.
int crt_hip_probe_fill_f64(double* dst, size_t n, double value, crt_stream_t stream) {
  if (!dst || n == 0 || (n & 1) || (reinterpret_cast<uintptr_t>(dst) & 15)) return CRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<d2*>(dst), n / 2, value);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_probe_store_set_f64(double* const* arrays, int32_t narrays, int64_t ncol, int64_t col_doubles, int32_t run_doubles, double value,
                                crt_stream_t stream) {
  if (!arrays || narrays < 1 || narrays > 8 || ncol < 1 || ncol > 0x7fffffff || col_doubles < 2 || (col_doubles & 1) || run_doubles < 2 ||
      (run_doubles & 1))
    return CRT_ERR_BAD_ARG;
  StoreSetArgs a;
  for (int i = 0; i < narrays; ++i) {
    if (!arrays[i] || (reinterpret_cast<uintptr_t>(arrays[i]) & 15)) return CRT_ERR_BAD_ARG;
    a.p[i] = arrays[i];
  }
  a.na = narrays;
  a.ncol = ncol;
  a.col = col_doubles;
  a.run = run_doubles;
  hipLaunchKernelGGL(k_store_set, dim3((unsigned)ncol), dim3(192), 0, static_cast<hipStream_t>(stream), a, value);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

// the device math the kernels use instead of ocml's exp / sincos (crt_internal.hpp: fexp, fast_sincos), exposed so that a test can
// bound their error in ulps over the arguments the schemes produce
int crt_hip_probe_math_f64(const double* x, size_t n, double* e, double* sn, double* cs, crt_stream_t stream) {
  if (!x || !e || !sn || !cs || n == 0 || n > 0x7fffffffull * 256) return CRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, e, sn, cs);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

int crt_hip_probe_copy_f64(double* dst, const double* src, size_t n, crt_stream_t stream) {
  if (!dst || !src || n == 0 || (n & 1) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15))
    return CRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<d2*>(dst),
                     reinterpret_cast<const d2*>(src), n / 2);
  return hipGetLastError() == hipSuccess ? CRT_OK : CRT_ERR_LAUNCH;
}

}  // extern "C"
