// Stress equilibration of RT_2 in ONE tiled launch: the two row-wise semi-explicit equilibrations
// (se/solve_patch_semiexplt.hpp:212-1163) and the weak-symmetry step of the same patch
// (se/solve_patch_weaksym.hpp:59-233, se/PatchData.hpp:598-663) are done by the same lanes while the
// patch-local stress rows are still in registers; the corrected (cell, vertex) rows of both stress rows
// go to the LDS slots of the tile and are flushed like the flux rows of k_se_patch_tiled.  No slot
// buffer, no separate weak-symmetry pass, no reduction pass.
//
// Applies to the case without flux boundary conditions on the stress rows (pure primal Dirichlet
// data: the benchmark and the reference's convergence tests) and to the FULL patches of a mesh -
// interior, 4 or 8 cells = as many cells as lanes of their group: on them the patch shape is a
// compile-time constant and the body needs 244 registers, no scratch.  Every other patch (boundary
// patches, interior patches of 3, 5, 6, 7 or more than 8 cells) and meshes with stress flux BCs run on
// the generic kernels (row sweeps into the slot buffer, eqlb_se_weaksym.hip, compact reduction over the
// cells they touch) in the same call.  Round 2 also ran a generic instance of this body inside this
// kernel: it needs 40 registers more than exist, and the scratch memory a kernel reserves for its worst
// instance slows EVERY wave of it (measured on the EV kernels in round 3: 332 B of scratch per lane,
// never touched on the benchmark's path, cost 27 %).  The tile lists are padded to whole wave-blocks
// with copies of a full patch that own no cell (they compute and store nothing).
//
// Weak symmetry in lane-parallel form (tests/proto_stress_lanes.py is the numpy statement): lane i of
// a patch group <-> cell T_{i+1} <-> facet row E_i <-> ring point i (the outer node of E_i).
//   * A = [Z C^T; C A_c]: border (d, x_0) + tridiagonal chain, as in the semi-explicit solver; all
//     solves with A share ONE parallel cyclic reduction over the lanes: its right-hand side columns
//     are the two loads of the stress rows, the two coupling columns of the border and the columns of
//     B_0, B_1 (row E_i of B_k meets the ring points i-1, i, i+1 and the patch node only);
//   * the border part of column c, z^(c) = Zs^-1 q^(c), is computed by lane c from its neighbours;
//   * S = sum_k B_k^T A^-1 B_k = T + sum_k Q_k^T Zs^-1 Q_k, three products per entry of T;
//   * S gamma = -R without pivoting (S is SPD on boundary patches; on interior patches S 1 = 0, the
//     mean-value multiplier is eliminated analytically: lambda = sum R / sum M, gamma_node := 0);
//   * u_k = -A^-1 (B_k gamma) re-uses the multipliers of the cyclic reduction.
#include "eqlb_device_common.h"
#ifndef EQLB_STRESS_RING_BPERM
#define EQLB_STRESS_RING_BPERM 1 // cyclic neighbours of full patches by ds_bpermute (2 per double) instead of two DPP moves + select per dword: 0.538 -> 0.531 ms
#endif
#ifndef EQLB_STRESS_REPCR
#define EQLB_STRESS_REPCR 0
#endif
#ifndef EQLB_STRESS_REBUILD_B
#define EQLB_STRESS_REBUILD_B 1
#endif

namespace eqlb
{

struct StressRows
{
  const double* g[2]; // projected stress rows, DG_1^2 [ncells][3][2]
  const double* f[2]; // projected right-hand sides [ncells][3]
  double* x[2];       // equilibrated rows [ncells][8]
};

namespace
{
constexpr int SK = 2, SND = 3, SNQ = 3, SNH = 3, SNRT = 8, SNPK = 6;

// ---- lane exchange inside a group of P lanes -------------------------------------------------------
// FULL (every patch of the wave-block has exactly P cells and is interior): cyclic neighbours by DPP
// row shifts; otherwise ds_bpermute with the lane computed at run time
template <int P, bool FULL>
__device__ __forceinline__ double from_prev(double v, int gbase, int sub, int prevl)
{
#if EQLB_STRESS_RING_BPERM
  if constexpr (FULL)
    return shfl_d(v, gbase + ((sub + P - 1) & (P - 1)));
#else
  if constexpr (FULL)
  {
    const double a = dpp_d<0x111>(v), b = dpp_d<0x100 + (P - 1)>(v);
    return (sub == 0) ? b : a;
  }
#endif
  else
    return shfl_d(v, gbase + prevl);
}
template <int P, bool FULL>
__device__ __forceinline__ double from_next(double v, int gbase, int sub, int nextl)
{
#if EQLB_STRESS_RING_BPERM
  if constexpr (FULL)
    return shfl_d(v, gbase + ((sub + 1) & (P - 1)));
#else
  if constexpr (FULL)
  {
    const double a = dpp_d<0x101>(v), b = dpp_d<0x110 + (P - 1)>(v);
    return (sub == P - 1) ? b : a;
  }
#endif
  else
    return shfl_d(v, gbase + nextl);
}
__device__ __forceinline__ double from_lane(double v, int lane) { return shfl_d(v, lane); }
// two consecutive doubles of a 16-byte aligned LDS row
__device__ __forceinline__ void ldrow_pair(const double* p, double (&r)[2])
{
  const double2 v = *reinterpret_cast<const double2*>(p);
  r[0] = v.x;
  r[1] = v.y;
}

// ---- parallel cyclic reduction of the chain with NC right-hand side columns -------------------------
// b: diagonal, am: coupling to row i - 1 (0 in lane 0), rows outside the chain are identity rows with
// zero couplings; what a shift drags in from a neighbouring patch group is multiplied by an exact zero.
// The multipliers of the levels are returned for later columns (pcr_apply).
#ifndef EQLB_PCR_SELECTS
#define EQLB_PCR_SELECTS 0 // see eqlb_se_kernels.hip
#endif
template <int P>
struct PcrMult
{
  static constexpr int NL = (P > 4) ? 3 : ((P > 2) ? 2 : 1);
  double al[NL], ga[NL], ibf;
};

template <int P, int S, int L, int NC>
__device__ __forceinline__ void pcr_level(double& b, double& am, double (&r)[NC], const int sub, bool& posdef,
                                          PcrMult<P>& m)
{
  if constexpr (P > S)
  {
    posdef = posdef && (b > 0.0);
    const double ib = rcp_d(b);
    const double ib_lo = dpp_d<0x110 + S>(ib), a_lo = dpp_d<0x110 + S>(am);
    const double ib_hi = dpp_d<0x100 + S>(ib), a_hi = dpp_d<0x100 + S>(am);
    const double cp = EQLB_PCR_SELECTS ? ((sub + S < P) ? a_hi : 0.0) : a_hi; // coupling to row i + S (eqlb_se_kernels.hip)
    const double al = am * ib_lo, ga = cp * ib_hi;
    b = __builtin_fma(-ga, cp, __builtin_fma(-al, am, b));
#pragma unroll
    for (int c = 0; c < NC; ++c)
    {
      const double lo = dpp_d<0x110 + S>(r[c]), hi = dpp_d<0x100 + S>(r[c]);
      r[c] = __builtin_fma(-ga, hi, __builtin_fma(-al, lo, r[c]));
    }
    am = EQLB_PCR_SELECTS ? ((sub >= 2 * S) ? -al * a_lo : 0.0) : -al * a_lo;
    m.al[L] = al;
    m.ga[L] = ga;
  }
}

template <int P, int NC>
__device__ __forceinline__ void pcr_chain(double b, double am, double (&r)[NC], const int sub, bool& posdef,
                                          PcrMult<P>& m)
{
  pcr_level<P, 1, 0, NC>(b, am, r, sub, posdef, m);
  pcr_level<P, 2, 1, NC>(b, am, r, sub, posdef, m);
  pcr_level<P, 4, 2, NC>(b, am, r, sub, posdef, m);
  posdef = posdef && (b > 0.0);
  m.ibf = rcp_d(b);
#pragma unroll
  for (int c = 0; c < NC; ++c)
    r[c] *= m.ibf;
}

template <int P, int S, int L, int NC>
__device__ __forceinline__ void pcr_apply_level(double (&r)[NC], const PcrMult<P>& m)
{
  if constexpr (P > S)
  {
#pragma unroll
    for (int c = 0; c < NC; ++c)
    {
      const double lo = dpp_d<0x110 + S>(r[c]), hi = dpp_d<0x100 + S>(r[c]);
      r[c] = __builtin_fma(-m.ga[L], hi, __builtin_fma(-m.al[L], lo, r[c]));
    }
  }
}

template <int P, int NC>
__device__ __forceinline__ void pcr_apply(double (&r)[NC], const PcrMult<P>& m)
{
  pcr_apply_level<P, 1, 0, NC>(r, m);
  pcr_apply_level<P, 2, 1, NC>(r, m);
  pcr_apply_level<P, 4, 2, NC>(r, m);
#pragma unroll
  for (int c = 0; c < NC; ++c)
    r[c] *= m.ibf;
}

// ---- the same for the columns of B_k in ROTATED storage (full interior patches: P chain / border rows, P ring points) --
// B_k couples row E_i to the ring points i - 1, i, i + 1 (cyclic).  Stored by ring point, its P columns have their
// three entries in different registers in every lane (a chain of selects per column to build them, and no way
// for the compiler to skip the zeros).  Stored by OFFSET - entry d of lane i belongs to ring point (i + d) mod P -
// the three entries sit in the registers 0, 1, P - 1 of every lane, a level of the reduction with stride s reads
// offset d + s from the lane s below and d - s from the lane s above (register indices fixed at compile time), and
// the zeros are known: offsets farther than 2^l from 0 are still zero before level l.
#ifndef EQLB_STRESS_ROTATED
#define EQLB_STRESS_ROTATED 1
#endif
#ifndef EQLB_STRESS_NFIX
#define EQLB_STRESS_NFIX 1 // MIXED kernel: instances for interior patches with P - 1, P - 2, P - 3 cells (0: generic instance)
#endif
__host__ __device__ constexpr bool rot_nz(int P, int L, int d)
{
  const int dd = ((d % P) + P) % P, dist = (dd < P - dd) ? dd : P - dd;
  return dist <= (1 << L);
}
// (N <= P ring points - interior patches with N cells in groups of P lanes: offsets modulo N in the first N registers)
template <int P, int N, int S, int L, int NA>
__device__ __forceinline__ void pcr_apply_level_rot(double (&r)[NA], const PcrMult<P>& m)
{
  if constexpr (P > S)
  {
    double n[N];
#pragma unroll
    for (int d = 0; d < N; ++d)
    {
      double v = rot_nz(N, L, d) ? r[d] : 0.0;
      if (rot_nz(N, L, d + S))
        v = __builtin_fma(-m.al[L], dpp_d<0x110 + S>(r[(d + S) % N]), v);
      if (rot_nz(N, L, d - S))
        v = __builtin_fma(-m.ga[L], dpp_d<0x100 + S>(r[((d - S) % N + N) % N]), v);
      n[d] = v;
    }
#pragma unroll
    for (int d = 0; d < N; ++d)
      r[d] = n[d];
  }
}
template <int P, int N, int NA>
__device__ __forceinline__ void pcr_apply_rot(double (&r)[NA], const PcrMult<P>& m)
{
  pcr_apply_level_rot<P, N, 1, 0, NA>(r, m);
  pcr_apply_level_rot<P, N, 2, 1, NA>(r, m);
  pcr_apply_level_rot<P, N, 4, 2, NA>(r, m);
#pragma unroll
  for (int d = 0; d < N; ++d)
    r[d] *= m.ibf;
}

// ---- the patch body ------------------------------------------------------------------------------------
// lds: F | H | D | TE | WQ | HB (as k_se_patch_tiled) | V | VQ
// NFIX > 0 (with FULL = false): every patch of the wave-block is interior with exactly NFIX < P cells (the tile lists
// are ordered by it) - the generic instance with the patch shape known at compile time: no node column, columns of B_k
// stored by offset modulo NFIX like in the full-patch instance
template <int P, bool FULL, int NFIX = 0>
__device__ __forceinline__ void stress_patch_body(const SeArgs& a, const StressRows& rows, const double* lds,
                                                  const int64_t lane_index, double* tile_slots, const int tc)
{
  using Z = Sizes<2, 1, P>;
  constexpr int K = SK, ND = SND, NQ = SNQ, NH = SNH, NRT = SNRT;
  constexpr int NTES = Z::NTES, NCOLS = Z::NCOLS;
  const double* sF = lds;
  const double* sH = sF + Z::NF;
  const double* sTE = sH + Z::NHT + Z::NDT;
  const double* sWQ = sTE + Z::NTET;
  const double* sV = lds + Z::NTAB; // [3][NRT][2]
  const double* sVQ = sV + Z::NVT;  // [NCOMBO][2][NH][3]

  // opaque lane index: keeps the compiler from hoisting the lane predicates (sub == 1, sub < n, ...) of all
  // four instances of this body out of the wave-block loops of the kernel, where they would occupy
  // registers across all bins
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
  const int sub = lane % P, gbase = lane - sub;
  const int64_t tl = lane_index;
  const int64_t patch_local = tl / P;
  constexpr bool FIXN = !FULL && NFIX > 0;
  constexpr bool INTK = FULL || FIXN;        // interior patches of a known size
  constexpr int NR = FIXN ? NFIX : P;        // ring points (where known)
  const bool pvalid = INTK ? true : (patch_local < a.npatch);
  const int64_t slot = a.slot_offset + tl;
  const int64_t patch = a.patch_offset + patch_local;

  int n = P;
  int32_t cell_raw = -1;
  uint32_t info = 0u;
  uint8_t flag0 = (uint8_t)PFLAG_INTERIOR;
  if (pvalid)
  {
    if constexpr (!INTK)
    {
      n = (int)a.pn[patch];
      flag0 = a.pflag[patch];
    }
    if constexpr (FIXN)
      n = NFIX;
    cell_raw = a.slot_cell[slot];
    info = a.slot_info[slot];
  }
  else
    n = 0;
  const bool active = FULL ? true : (cell_raw >= 0);
  const int32_t cell = active ? cell_raw : 0;
  const int fm = (info >> INFO_FM_SHIFT) & 3, fp = (info >> INFO_FP_SHIFT) & 3, ln = (info >> INFO_LN_SHIFT) & 3;
  const bool rev_m = (info & INFO_REV_M) != 0, rev_p = (info & INFO_REV_P) != 0;
  const int ci = active ? combo_index(fm, fp, rev_m) : 0;

  // ---- data of the cell: J and the two stress rows ----
  double J00 = 1.0, J01 = 0.0, J10 = 0.0, J11 = 1.0;
  double2 gdat[2][ND];
  double fdat[2][ND];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      gdat[r][i] = make_double2(0.0, 0.0);
      fdat[r][i] = 0.0;
    }
  if (active)
  {
    const double2* Jp = reinterpret_cast<const double2*>(a.cellJ + 4 * (int64_t)cell);
    const double2 j0 = Jp[0], j1 = Jp[1];
#pragma unroll
    for (int r = 0; r < 2; ++r)
    {
      const double2* gp_ = reinterpret_cast<const double2*>(rows.g[r] + (int64_t)cell * (ND * 2));
      const double* fp_ = rows.f[r] + (int64_t)cell * ND;
#pragma unroll
      for (int i = 0; i < ND; ++i)
        gdat[r][i] = gp_[i];
#pragma unroll
      for (int i = 0; i < ND; ++i)
        fdat[r][i] = fp_[i];
    }
    J00 = j0.x;
    J01 = j0.y;
    J10 = j1.x;
    J11 = j1.y;
  }
  const double detJ = J00 * J11 - J01 * J10;
  const bool neg_det = !(detJ > 0.0);
  const double sgn = neg_det ? -1.0 : 1.0;
  const double pf_m = (fm == 1) ? sgn : -sgn, pf_p = (fp == 1) ? sgn : -sgn;

  // ---- neighbours ----
  const bool interior = INTK ? true : ((flag0 & PFLAG_INTERIOR) != 0);
  const int nf = interior ? n : n + 1;
  const int nn = (n > 0) ? n : 1;
  const int nextl = (sub + 1 < nn) ? sub + 1 : (interior ? 0 : sub);
  const int prevl = (sub > 0) ? sub - 1 : (interior ? nn - 1 : 0);
  const bool has_next = active && (interior || sub < n - 1);
  const bool has_prev = active && (interior || sub > 0);
  const bool row_valid = pvalid && sub < nf;
  const bool has_prevcell = row_valid && (sub > 0 || interior); // the cell before facet E_sub exists
  const bool in_chain = pvalid && sub >= 1 && sub < nf;
  const bool wraps = interior && sub == n - 1; // coupling of E_{n-1} to E_n == E_0
  // the lane after this one in facet order (its row is the up-unknown of this cell)
  const int upl = interior ? nextl : ((sub + 1 < P) ? sub + 1 : sub);
  // values of the lanes that own the facet rows / ring points before and after this lane's (cyclic on
  // interior patches); 0 where there is none.  _nc: without the wrap (enough where lane 0 holds a zero)
  const bool has_rprev = row_valid && (sub > 0 || interior), has_rnext = row_valid && (sub + 1 < nf || interior);
  const int rprevl = (sub > 0) ? sub - 1 : ((nf > 0) ? nf - 1 : 0), rnextl = (sub + 1 < nf) ? sub + 1 : 0;
  auto ring_prev = [&](double v) {
    const double t_ = from_prev<P, FULL>(v, gbase, sub, rprevl);
    return (FULL || has_rprev) ? t_ : 0.0;
  };
  auto ring_next = [&](double v) {
    const double t_ = from_next<P, FULL>(v, gbase, sub, rnextl);
    return (FULL || has_rnext) ? t_ : 0.0;
  };
  auto ring_next_nc = [&](double v) {
    if constexpr (FULL)
      return dpp_d<0x101>(v) * ((sub == P - 1) ? 0.0 : 1.0);
    else
      return ring_next(v);
  };

  // ---- phase A (both rows): facet moments of hat G, moments of hat (f - div G) ----
  double gm[2][K], gpv[2][K], Rq[2][NQ];
  {
    const double a00 = J11, a01 = -J01, a10 = -J10, a11 = J00;
    const double nmx = (fm == 2) ? 0.0 : -1.0, nmy = (fm == 0) ? -1.0 : ((fm == 1) ? 0.0 : 1.0);
    const double npx = (fp == 2) ? 0.0 : -1.0, npy = (fp == 0) ? -1.0 : ((fp == 1) ? 0.0 : 1.0);
    const double num0 = a00 * nmx + a10 * nmy, num1 = a01 * nmx + a11 * nmy;
    const double nup0 = a00 * npx + a10 * npy, nup1 = a01 * npx + a11 * npy;
    double rm[ND][K], rp[ND][K], rH[Z::HROW];
    {
      const double* tF_m = sF + (fm * 3 + ln) * ND * K;
      const double* tF_p = sF + (fp * 3 + ln) * ND * K;
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        ldrow_pair(tF_m + i * K, rm[i]);
        ldrow_pair(tF_p + i * K, rp[i]);
      }
      const double* tH = sH + ln * Z::HROW;
#pragma unroll
      for (int e2 = 0; e2 < Z::HROW / 2; ++e2)
      {
        const double2 v = reinterpret_cast<const double2*>(tH)[e2];
        rH[2 * e2] = v.x;
        rH[2 * e2 + 1] = v.y;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
    {
#pragma unroll
      for (int j = 0; j < K; ++j)
        gm[r][j] = gpv[r][j] = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        Rq[r][q] = 0.0;
      double fdv[ND], dvg = 0.0;
#pragma unroll
      for (int i = 0; i < ND; ++i)
      {
        const double2 g2 = gdat[r][i];
        const double gnm = g2.x * num0 + g2.y * num1, gnp = g2.x * nup0 + g2.y * nup1;
        const double gh0 = a00 * g2.x + a01 * g2.y, gh1 = a10 * g2.x + a11 * g2.y;
        fdv[i] = detJ * fdat[r][i];
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          gm[r][j] += rm[i][j] * gnm;
          gpv[r][j] += rp[i][j] * gnp;
        }
        // P1 data: div_ref(adj G) is constant on the cell, folded into the nodal values of detJ f
        if (i == 0)
          dvg = -(gh0 + gh1);
        else if (i == 1)
          dvg += gh0;
        else
          dvg += gh1;
      }
#pragma unroll
      for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          Rq[r][q] += (fdv[i] - dvg) * rH[i * NQ + q];
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        gm[r][j] = active ? gm[r][j] * pf_m : 0.0;
        gpv[r][j] = active ? gpv[r][j] * pf_p : 0.0;
      }
    }
  }

  // ---- phase B (both rows, no flux BCs): neighbour exchange -> particular solution ----
  double mu_m[2][K], mu_p[2][K];
#pragma unroll
  for (int r = 0; r < 2; ++r)
  {
    double Jv[K];
    {
      const double g0n = from_next<P, FULL>(gm[r][0], gbase, sub, nextl);
      const double g1n = from_next<P, FULL>(gm[r][1], gbase, sub, nextl);
      Jv[0] = has_next ? gpv[r][0] + g0n : 0.0;
      Jv[1] = has_next ? gpv[r][1] + (rev_p ? g0n - g1n : g1n) : 0.0;
    }
    const double Jprev0 = from_prev<P, FULL>(Jv[0], gbase, sub, prevl);
    double t = active ? (sgn * Rq[r][0] + (has_prev ? Jprev0 : 0.0)) : 0.0;
    {
      double o = lane_down_d<P, 1>(t, gbase, sub);
      if (sub >= 1)
        t += o;
      o = lane_down_d<P, 2>(t, gbase, sub);
      if (sub >= 2)
        t += o;
      if constexpr (P > 4)
      {
        o = lane_down_d<P, 4>(t, gbase, sub);
        if (sub >= 4)
          t += o;
      }
    }
    mu_p[r][0] = t;
    mu_p[r][1] = 0.0;
    const double v0 = from_prev<P, FULL>(t + Jv[0], gbase, sub, prevl);
    const double v1 = from_prev<P, FULL>(Jv[1], gbase, sub, prevl);
    mu_m[r][0] = has_prev ? -v0 : 0.0;
    mu_m[r][1] = has_prev ? (rev_m ? -(v0 - v1) : -v1) : 0.0;
  }

  // ---- phase C: element matrix (shared) and the two loads ----
  double Te[NH][NH], Le[2][NH];
  {
    const double ia = active ? rcp_d(fabs(detJ)) : 0.0;
    const double g0 = (J00 * J00 + J10 * J10) * ia, g1 = (J00 * J01 + J10 * J11) * ia,
                 g2 = (J01 * J01 + J11 * J11) * ia;
    const double* te = sTE + ci * 3 * NTES;
    double tev[NTES];
#pragma unroll
    for (int e2 = 0; e2 < NTES / 2; ++e2)
    {
      const double2 t0 = reinterpret_cast<const double2*>(te)[e2];
      const double2 t1 = reinterpret_cast<const double2*>(te + NTES)[e2];
      const double2 t2 = reinterpret_cast<const double2*>(te + 2 * NTES)[e2];
      tev[2 * e2] = g0 * t0.x + g1 * t1.x + g2 * t2.x;
      tev[2 * e2 + 1] = g0 * t0.y + g1 * t1.y + g2 * t2.y;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int g = 0; g <= h; ++g)
        Te[h][g] = Te[g][h] = tev[h * (h + 1) / 2 + g];
    double full[2][NCOLS];
#pragma unroll
    for (int r = 0; r < 2; ++r)
    {
      full[r][0] = mu_m[r][0];
      full[r][1] = mu_m[r][1];
      full[r][2] = mu_p[r][0];
      full[r][3] = mu_p[r][1];
      full[r][4] = sgn * Rq[r][1];
      full[r][5] = sgn * Rq[r][2];
    }
    const double* wq = sWQ + ci * 3 * NH * NCOLS;
#pragma unroll
    for (int h = 0; h < NH; ++h)
    {
      double s[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
      for (int c2 = 0; c2 < NCOLS / 2; ++c2)
      {
        const double2 w0 = reinterpret_cast<const double2*>(wq + h * NCOLS)[c2];
        const double2 w1 = reinterpret_cast<const double2*>(wq + (NH + h) * NCOLS)[c2];
        const double2 w2 = reinterpret_cast<const double2*>(wq + (2 * NH + h) * NCOLS)[c2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
        {
          const double f0 = full[r][2 * c2], f1 = full[r][2 * c2 + 1];
          s[r][0] = __builtin_fma(w0.y, f1, __builtin_fma(w0.x, f0, s[r][0]));
          s[r][1] = __builtin_fma(w1.y, f1, __builtin_fma(w1.x, f0, s[r][1]));
          s[r][2] = __builtin_fma(w2.y, f1, __builtin_fma(w2.x, f0, s[r][2]));
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
        Le[r][h] = -(g0 * s[r][0] + g1 * s[r][1] + g2 * s[r][2]);
    }
  }

  const double Ce = active ? fabs(detJ) * (1.0 / 6.0) : 0.0;

  // ---- rows of the patch system A: lane i owns facet row E_i = own minus-side + plus-side of the
  // previous cell ----
  // (cross-lane moves are executed by ALL lanes - a lane masked out by a branch could not be read -, the
  // select comes afterwards)
  auto prevv = [&](double v) {
    const double t_ = from_prev<P, FULL>(v, gbase, sub, prevl);
    return has_prevcell ? t_ : 0.0;
  };
  double bt = Te[0][1] + prevv(Te[0][2]);
  double Dg = Te[1][1] + prevv(Te[2][2]);
  double Off = Te[1][2];
  double rr[2];
  rr[0] = Le[0][1] + prevv(Le[0][2]);
  rr[1] = Le[1][1] + prevv(Le[1][2]);
  double alpha = group_sum_d<P>(Te[0][0], gbase, sub);
  double rd[2] = {group_sum_d<P>(Le[0][0], gbase, sub), group_sum_d<P>(Le[1][0], gbase, sub)};
  if (!row_valid)
  {
    bt = rr[0] = rr[1] = 0.0;
    Dg = 1.0;
    Off = 0.0;
  }
  if (!pvalid)
  {
    alpha = 1.0;
    rd[0] = rd[1] = 0.0;
  }

  // ---- border data (lane 0 of the group) and the chain ----
  const double Z00 = alpha, Z10 = from_lane(bt, gbase), Z11 = from_lane(Dg, gbase);
  const double Off0 = from_lane(Off, gbase);
  const double rz1[2] = {from_lane(rr[0], gbase), from_lane(rr[1], gbase)};
  const double Dp = in_chain ? Dg : 1.0;
  const double OffC = (in_chain && !wraps) ? Off : 0.0;
  const double B1 = in_chain ? bt : 0.0;
  const double B2 = in_chain ? (((sub == 1) ? Off0 : 0.0) + (wraps ? Off : 0.0)) : 0.0;
  double am = dpp_d<0x111>(OffC);
  if (sub == 0)
    am = 0.0;
  const double Dp_keep = Dp, am_keep = am;
  (void)Dp_keep;
  (void)am_keep;
  // chain reduction with the columns [load row 0 | load row 1 | column of d | column of x_0]; the
  // multipliers are kept for the columns of the weak-symmetry step
  double col4[4] = {in_chain ? rr[0] : 0.0, in_chain ? rr[1] : 0.0, B1, B2};
  bool posdef = true;
  PcrMult<P> mult;
  pcr_chain<P, 4>(Dp, am, col4, sub, posdef, mult);
  const double s1 = col4[2], s2 = col4[3]; // A_c^-1 [column of d | column of x_0]

  // ---- Schur complement of the chain on the border [d ; x_0]: Zi = (Z - C^T A_c^-1 C)^-1 ----
  double Zi00, Zi01, Zi11;
  {
    const double S00 = group_sum_d<P>(B1 * s1, gbase, sub), S10 = group_sum_d<P>(B2 * s1, gbase, sub),
                 S11 = group_sum_d<P>(B2 * s2, gbase, sub);
    const double l00 = Z00 - S00, l10 = Z10 - S10, l11 = Z11 - S11;
    const double det = __builtin_fma(l00, l11, -l10 * l10);
    posdef = posdef && (l00 > 0.0) && (det > 0.0);
    const double id = rcp_d(det);
    Zi00 = l11 * id;
    Zi11 = l00 * id;
    Zi01 = -l10 * id;
  }
  int status_local = (!posdef && pvalid) ? 1 : 0;

  // ---- the two row-wise solutions: RT coefficients into the LDS slots of the owned cell (halo lanes drop
  // them); packed row of the (cell, vertex): [facet with the smaller local id: 2 | the other: 2 | interior: 2].
  // The weak-symmetry corrections are added to the same slots at the end (only this lane touches them).
  const uint32_t loc = info >> INFO_LOCAL_SHIFT;
  const bool owned = active && loc != 0u;
  const int pm = fm - ((fm > ln) ? 1 : 0), pp = fp - ((fp > ln) ? 1 : 0);
  double* const orow = tile_slots + ((int64_t)(owned ? loc - 1 : 0) * 3 + ln) * SNPK;
  const int64_t row_stride = (int64_t)tc * 3 * SNPK;
  double Ll = 0.0, Lfp = 0.0, Lfm = 0.0; // Lc_e[j] = -int psi_j (s01 - s10): patch node / vertex fp / vertex fm
  {
    double cm[2][K], cpl[2][K];
#pragma unroll
    for (int r = 0; r < 2; ++r)
    {
      const double q0 = rd[r] - group_sum_d<P>(B1 * col4[r], gbase, sub);
      const double q1 = rz1[r] - group_sum_d<P>(B2 * col4[r], gbase, sub);
      const double zd = Zi00 * q0 + Zi01 * q1, zx = Zi01 * q0 + Zi11 * q1;
      double xs = in_chain ? __builtin_fma(-s2, zx, __builtin_fma(-s1, zd, col4[r])) : 0.0;
      if (sub == 0)
        xs = zx;
      const double up = from_next<P, FULL>(xs, gbase, sub, upl);
      // own-frame moments: mu_m -= Bm [d; um], mu_p += [d; up]
      cm[r][0] = pf_m * (mu_m[r][0] - zd);
      cm[r][1] = pf_m * (mu_m[r][1] - (rev_m ? zd - xs : xs));
      cpl[r][0] = pf_p * (mu_p[r][0] + zd);
      cpl[r][1] = pf_p * (mu_p[r][1] + up);
      if (owned)
      {
        double* o = orow + r * row_stride;
        o[pm * K + 0] = cm[r][0];
        o[pm * K + 1] = cm[r][1];
        o[pp * K + 0] = cpl[r][0];
        o[pp * K + 1] = cpl[r][1];
        o[2 * K + 0] = Rq[r][1];
        o[2 * K + 1] = Rq[r][2];
      }
    }
    // w = J-row combination of the two coefficient rows, per non-zero DOF of the row
#pragma unroll
    for (int e = 0; e < 6; ++e)
    {
      const double c0 = (e < 2) ? cm[0][e] : ((e < 4) ? cpl[0][e - 2] : Rq[0][e - 3]);
      const double c1 = (e < 2) ? cm[1][e] : ((e < 4) ? cpl[1][e - 2] : Rq[1][e - 3]);
      const double w0 = c0 * J10 - c1 * J00, w1 = c0 * J11 - c1 * J01;
      const int i = (e < 2) ? fm * K + e : ((e < 4) ? fp * K + (e - 2) : 3 * K + (e - 4));
      const double2 vl = reinterpret_cast<const double2*>(sV)[ln * NRT + i];
      const double2 vp = reinterpret_cast<const double2*>(sV)[fp * NRT + i];
      const double2 vm = reinterpret_cast<const double2*>(sV)[fm * NRT + i];
      Ll -= sgn * (w0 * vl.x + w1 * vl.y);
      Lfp -= sgn * (w0 * vp.x + w1 * vp.y);
      Lfm -= sgn * (w0 * vm.x + w1 * vm.y);
    }
    if (!active)
      Ll = Lfp = Lfm = 0.0;
  }
  const double Rc = group_sum_d<P>(Ll, gbase, sub);
  const double p_lfm = prevv(Lfm), p_ce = prevv(Ce);
  const double Rring = row_valid ? Lfp + p_lfm : 0.0;
  const double Mc = group_sum_d<P>(Ce, gbase, sub);
  const double Mr = row_valid ? Ce + p_ce : 0.0;

  // ---- weak symmetry: S row of ring point `sub` in its lane ([0] = column of the patch node), row of the
  // patch node in every lane; last entry = right-hand side.  One stress row k at a time. ----
  constexpr int NPT = P + 1;
  double rv[NPT + 1], rn[NPT + 1];
#pragma unroll
  for (int j = 0; j <= NPT; ++j)
    rv[j] = rn[j] = 0.0;
  // B_k: row E_i x ring points i-1, i, i+1 and x patch node; d row x ring point i, x patch node
  double Brow[2][3], bcn[2], Bd[2], Bdc[2];
  // ring point of the facets before / after facet E_sub (cyclic on interior patches)
  const int cm1 = (sub > 0) ? sub - 1 : (interior ? nf - 1 : -1);
  const int cp1 = (sub + 1 < nf) ? sub + 1 : (interior ? 0 : -1);
  // rows of B_k owned by this lane (from the tensor VQ and J; also re-derived after the Schur solve, see below)
  auto build_B = [&](const int k, const int ci_, const double ja, const double jb, double (&Br)[3], double& bcn_,
                     double& Bd_, double& Bdc_) {
    // Be[h][j]: k = 0: int (Phi_h)_y psi_j, k = 1: -int (Phi_h)_x psi_j; j = patch node / vertex on the minus
    // facet (local vertex fp) / vertex on the plus facet (local vertex fm)
    const double* vq = sVQ + ci_ * 2 * NH * 3;
    double Bl[NH], Bfp[NH], Bfm[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
    {
      Bl[h] = active ? (ja * vq[h * 3 + ln] + jb * vq[(NH + h) * 3 + ln]) : 0.0;
      Bfp[h] = active ? (ja * vq[h * 3 + fp] + jb * vq[(NH + h) * 3 + fp]) : 0.0;
      Bfm[h] = active ? (ja * vq[h * 3 + fm] + jb * vq[(NH + h) * 3 + fm]) : 0.0;
    }
    const double p_fp2 = prevv(Bfp[2]), p_fm2 = prevv(Bfm[2]), p_l2 = prevv(Bl[2]), p_fm0 = prevv(Bfm[0]);
    Br[0] = row_valid ? p_fp2 : 0.0;
    Br[1] = row_valid ? Bfp[1] + p_fm2 : 0.0;
    Br[2] = row_valid ? Bfm[1] : 0.0;
    bcn_ = row_valid ? Bl[1] + p_l2 : 0.0;
    Bd_ = row_valid ? Bfp[0] + p_fm0 : 0.0;
    Bdc_ = group_sum_d<P>(Bl[0], gbase, sub);
  };
#pragma unroll
  for (int k = 0; k < 2; ++k)
  {
    build_B(k, ci, (k == 0) ? J10 : -J00, (k == 0) ? J11 : -J01, Brow[k], bcn[k], Bd[k], Bdc[k]);
    // columns of B_k through the chain reduction: ring points, and the patch node on boundary patches
    constexpr int NCEN = INTK ? 0 : 1;
    constexpr bool ROT = INTK && EQLB_STRESS_ROTATED;
    double col[P + NCEN];
    if constexpr (ROT)
    {
      // rotated storage (pcr_apply_rot): entry d = ring point (sub + d) mod NR; lane 0 (and lanes >= NR) are no chain rows
      const double mch = in_chain ? 1.0 : 0.0;
#pragma unroll
      for (int d = 0; d < P + NCEN; ++d)
        col[d] = 0.0;
      col[0] = mch * Brow[k][1];
      col[1] = mch * Brow[k][2];
      col[NR - 1] = mch * Brow[k][0];
      pcr_apply_rot<P, NR, P + NCEN>(col, mult);
    }
    else
    {
#pragma unroll
    for (int c = 0; c < P; ++c)
    {
      const double v = (c == sub) ? Brow[k][1] : ((c == cm1) ? Brow[k][0] : ((c == cp1) ? Brow[k][2] : 0.0));
      col[c] = (in_chain && c < nf) ? v : 0.0;
    }
    if constexpr (!INTK)
      col[P] = in_chain ? bcn[k] : 0.0;
    pcr_apply<P, P + NCEN>(col, mult);
    }
    // border part of every column, in the lane of its ring point: q^(c) = b_border - C^T A_c^-1 b_chain with
    // the sparse original column (rows c-1, c, c+1; lane 0 is no chain row: s1 = s2 = 0 there, so the wrap
    // nf - 1 -> ring point 0 is the only cyclic term)
    double q0, q1, z0, z1;
    {
      const double t1 = s1 * Brow[k][1] + ring_prev(s1 * Brow[k][2]) + ring_next_nc(s1 * Brow[k][0]);
      const double t2 = s2 * Brow[k][1] + ring_prev(s2 * Brow[k][2]) + ring_next_nc(s2 * Brow[k][0]);
      // row E_0 of the column: ring point 0 itself, ring point 1, ring point nf - 1 (interior patches)
      const double r0m = from_lane(Brow[k][0], gbase), r0c = from_lane(Brow[k][1], gbase),
                   r0p = from_lane(Brow[k][2], gbase);
      const double bx0 = (sub == 0) ? r0c : ((sub == 1) ? r0p : ((interior && sub == nf - 1) ? r0m : 0.0));
      q0 = row_valid ? Bd[k] - t1 : 0.0;
      q1 = row_valid ? bx0 - t2 : 0.0;
      z0 = Zi00 * q0 + Zi01 * q1;
      z1 = Zi01 * q0 + Zi11 * q1;
    }
    if constexpr (ROT)
    {
      // rv[1 + d] collects S[sub][(sub + d) mod P] here (turned into ring-point order behind the loop over k):
      // the row before this lane holds that column at offset d + 1, the row after it at offset d - 1
#pragma unroll
      for (int d = 0; d < NR; ++d)
      {
        double v = Brow[k][1] * col[d];
        v += ring_prev(Brow[k][2] * col[(d + 1) % NR]);
        v += ring_next_nc(Brow[k][0] * col[(d + NR - 1) % NR]);
        const int src = gbase + (FULL ? ((sub + d) & (P - 1)) : ((sub + d >= NR) ? sub + d - NR : sub + d));
        v = __builtin_fma(q1, from_lane(z1, src), __builtin_fma(q0, from_lane(z0, src), v));
        asm volatile("" : "+v"(v)); // keeps the exchanges of the columns apart (register pressure)
        rv[1 + d] += v;
      }
    }
    else
    {
#pragma unroll
    for (int c = 0; c < P; ++c)
    {
      const double tc_ = col[c];
      // T[r][c] = sum over the chain rows r-1, r, r+1 of B[E_i][r] t_i^(c)
      double v = Brow[k][1] * tc_;
      v += ring_prev(Brow[k][2] * tc_);
      v += ring_next_nc(Brow[k][0] * tc_);
      // border part q^(r)^T Zs^-1 q^(c)
      v = __builtin_fma(q1, from_lane(z1, gbase + c), __builtin_fma(q0, from_lane(z0, gbase + c), v));
      asm volatile("" : "+v"(v)); // keeps the exchanges of the columns apart (register pressure)
      rv[1 + c] += v;
    }
    }
    if constexpr (!INTK)
    {
      // the patch node (needed on boundary patches): dense column, all rows E_i carry bcn_i
      const double bch = in_chain ? bcn[k] : 0.0;
      const double qn0 = Bdc[k] - group_sum_d<P>(s1 * bch, gbase, sub);
      const double qn1 = from_lane(bcn[k], gbase) - group_sum_d<P>(s2 * bch, gbase, sub);
      const double zn0 = Zi00 * qn0 + Zi01 * qn1, zn1 = Zi01 * qn0 + Zi11 * qn1;
      const double tn = col[P];
      rv[0] += Brow[k][1] * tn + ring_prev(Brow[k][2] * tn) + ring_next_nc(Brow[k][0] * tn) + q0 * zn0 + q1 * zn1;
      rn[0] += group_sum_d<P>(bch * tn, gbase, sub) + qn0 * zn0 + qn1 * zn1;
    }
  }
  if constexpr (INTK && EQLB_STRESS_ROTATED)
  {
    // S[sub][c] = rotated[(c - sub) mod NR]: a rotation of the NR registers by `sub`, one conditional step per bit
#pragma unroll
    for (int st = 1; st < NR; st <<= 1)
    {
      const bool bit = (sub & st) != 0;
      double t_[NR];
#pragma unroll
      for (int c = 0; c < NR; ++c)
        t_[c] = bit ? rv[1 + ((c - st) % NR + NR) % NR] : rv[1 + c];
#pragma unroll
      for (int c = 0; c < NR; ++c)
        rv[1 + c] = t_[c];
    }
  }
  if constexpr (!INTK)
  {
#pragma unroll
    for (int c = 0; c < P; ++c)
      rn[1 + c] = from_lane(rv[0], gbase + c); // symmetry: S[node][c] = S[c][node]
  }

  // ---- S gamma = -R: no pivoting ----
  const bool own_pt = row_valid; // the lane owns ring point `sub`
  double gam_own = 0.0, gam0 = 0.0;
  int sing = 0;
  {
    const bool meanvalue = interior;
    double q_own = own_pt ? -Rring : 0.0, q0 = pvalid ? -Rc : 0.0;
    const double sum_m = (pvalid ? Mc : 1.0) + group_sum_d<P>(Mr, gbase, sub);
    if (meanvalue)
    {
      const double sum_q = q0 + group_sum_d<P>(q_own, gbase, sub);
      const double lam = sum_q * rcp_d(sum_m); // = -lambda
      q_own -= lam * Mr;
#pragma unroll
      for (int c = 0; c < NPT; ++c)
        rn[c] = (c == 0) ? 1.0 : 0.0; // gamma_node := 0
      rv[0] = 0.0;
      q0 = 0.0;
    }
    rv[NPT] = q_own;
    rn[NPT] = q0;
    if (!own_pt) // lanes without a ring point: identity row
    {
#pragma unroll
      for (int c = 0; c <= NPT; ++c)
        rv[c] = (c == sub + 1) ? 1.0 : 0.0;
    }
    if (!pvalid)
    {
#pragma unroll
      for (int c = 0; c <= NPT; ++c)
        rn[c] = (c == 0) ? 1.0 : 0.0;
    }
    // elimination: pivot 0 is the replicated node row, pivot p the row of lane p - 1
    {
      if (!(rn[0] > 0.0))
        sing = 1;
      const double f = rv[0] * rcp_d(rn[0]);
#pragma unroll
      for (int c = 1; c <= NPT; ++c)
        rv[c] -= f * rn[c];
    }
#ifndef EQLB_STRESS_GJ
#define EQLB_STRESS_GJ 1
#endif
    if constexpr (EQLB_STRESS_GJ)
    {
      // Gauss-Jordan: the rows ABOVE the pivot are lanes that execute the row update anyway (their factor was a
      // forced zero), so eliminating there as well costs no instruction - and leaves every lane with its own
      // diagonal entry only: no back substitution, i.e. no second chain of eight dependent broadcasts and
      // reciprocals.  (Full interior patches: the node multiplier is fixed to zero above, rn = e_0.)
      double dinv = 1.0;
#pragma unroll
      for (int p = 1; p < NPT; ++p)
      {
        double pr[NPT + 1];
#pragma unroll
        for (int c = p; c <= NPT; ++c)
          pr[c] = from_lane(rv[c], gbase + p - 1);
        if (!(pr[p] > 0.0))
          sing = 1;
        const double ip = rcp_d(pr[p]);
        const bool own = sub + 1 == p;
        const double f = own ? 0.0 : rv[p] * ip;
        dinv = own ? ip : dinv;
#pragma unroll
        for (int c = p + 1; c <= NPT; ++c)
          rv[c] -= f * pr[c];
      }
      gam_own = rv[NPT] * dinv;
      if constexpr (INTK)
        gam0 = rn[NPT] * rcp_d(rn[0]);
      else
      {
        // node row (boundary patches: a multiplier of its own): the ring multipliers are all known at once - eight
        // independent broadcasts, no chain
        double t = rn[NPT];
#pragma unroll
        for (int c = 1; c < NPT; ++c)
          t -= rn[c] * from_lane(gam_own, gbase + c - 1);
        gam0 = t * rcp_d(rn[0]);
      }
    }
    else
    {
    double gam[NPT];
#pragma unroll
    for (int p = 1; p < NPT; ++p)
    {
      double pr[NPT + 1];
#pragma unroll
      for (int c = p; c <= NPT; ++c)
        pr[c] = from_lane(rv[c], gbase + p - 1);
      if (!(pr[p] > 0.0))
        sing = 1;
      const double f = (sub + 1 > p) ? rv[p] * rcp_d(pr[p]) : 0.0;
#pragma unroll
      for (int c = p + 1; c <= NPT; ++c)
        rv[c] -= f * pr[c];
    }
    // back substitution
#pragma unroll
    for (int p = NPT - 1; p >= 1; --p)
    {
      double t = rv[NPT];
#pragma unroll
      for (int c = p + 1; c < NPT; ++c)
        t -= rv[c] * gam[c];
      t *= rcp_d(rv[p]);
      if (sub + 1 == p)
        gam_own = t;
      gam[p] = from_lane(t, gbase + p - 1);
    }
    {
      double t = rn[NPT];
#pragma unroll
      for (int c = 1; c < NPT; ++c)
        t -= rn[c] * gam[c];
      gam0 = t * rcp_d(rn[0]);
    }
    }
    if (meanvalue)
    {
      const double shift = (Mc * gam0 + group_sum_d<P>(Mr * gam_own, gbase, sub)) * rcp_d(sum_m);
      gam0 -= shift;
      gam_own -= shift;
    }
    if (!own_pt)
      gam_own = 0.0;
  }
  if (sing && pvalid)
    status_local = 1;

  // ---- u_k = -A^-1 (B_k gamma): one more solve per stress row with the stored multipliers; the correction
  // (se/solve_patch_weaksym.hpp:189-232) is added to the rows in the LDS slots ----
  {
    const double g_prev = ring_prev(gam_own);
    const double g_next = ring_next(gam_own);
    double vr[2], vd[2];
#if EQLB_STRESS_REBUILD_B
    // B_k is NOT carried across the Schur solve (12 doubles per lane at the point of the highest register
    // pressure): re-derived from the tensor and J; the opaque copies keep the compiler from merging the two
    // computations back into held values
    int ci2 = ci;
    double j00 = J00, j01 = J01, j10 = J10, j11 = J11;
    asm volatile("" : "+v"(ci2), "+v"(j00), "+v"(j01), "+v"(j10), "+v"(j11));
#pragma unroll
    for (int k = 0; k < 2; ++k)
      build_B(k, ci2, (k == 0) ? j10 : -j00, (k == 0) ? j11 : -j01, Brow[k], bcn[k], Bd[k], Bdc[k]);
#endif
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      vr[k] = row_valid ? -(Brow[k][0] * g_prev + Brow[k][1] * gam_own + Brow[k][2] * g_next + bcn[k] * gam0) : 0.0;
      vd[k] = -(group_sum_d<P>(Bd[k] * gam_own, gbase, sub) + Bdc[k] * gam0);
    }
    double cu[2] = {in_chain ? vr[0] : 0.0, in_chain ? vr[1] : 0.0};
#if EQLB_STRESS_REPCR
    {
      bool pd2 = true;
      PcrMult<P> m2;
      pcr_chain<P, 2>(Dp_keep, am_keep, cu, sub, pd2, m2);
    }
#else
    pcr_apply<P, 2>(cu, mult);
#endif
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      const double q0 = vd[k] - group_sum_d<P>(B1 * cu[k], gbase, sub);
      const double q1 = from_lane(vr[k], gbase) - group_sum_d<P>(B2 * cu[k], gbase, sub);
      const double zd = Zi00 * q0 + Zi01 * q1, zx = Zi01 * q0 + Zi11 * q1;
      double xs = in_chain ? __builtin_fma(-s2, zx, __builtin_fma(-s1, zd, cu[k])) : 0.0;
      if (sub == 0)
        xs = zx;
      const double up = from_next<P, FULL>(xs, gbase, sub, upl);
      if (owned)
      {
        // (slot address and orientation signs rebuilt from the descriptor: fewer values live across the
        // Schur solve)
        const double sg2 = neg_det ? -1.0 : 1.0;
        const double pfm2 = (fm == 1) ? sg2 : -sg2, pfp2 = (fp == 1) ? sg2 : -sg2;
        double* o = tile_slots + (((int64_t)k * tc + (int64_t)(loc - 1)) * 3 + ln) * SNPK;
        const int pm2 = fm - ((fm > ln) ? 1 : 0), pp2 = fp - ((fp > ln) ? 1 : 0);
        o[pm2 * K + 0] += pfm2 * (-zd);
        o[pm2 * K + 1] += pfm2 * (rev_m ? -(zd - xs) : -xs);
        o[pp2 * K + 0] += pfp2 * zd;
        o[pp2 * K + 1] += pfp2 * up;
      }
    }
  }
  if (status_local)
    atomicOr(a.status, 2);
}

// DOF i of a cell from its three packed (cell, vertex) rows (k_se_patch_tiled: packed_sum)
__device__ __forceinline__ double packed_sum2(const double* rows, int i)
{
  constexpr int K = SK, NPK = SNPK;
  if (i >= 3 * K)
  {
    const int q = i - K;
    return (rows[q] + rows[NPK + q]) + rows[2 * NPK + q];
  }
  const int f = i / K, j = i - f * K;
  const int la = (f == 0) ? 1 : 0, lb = (f == 2) ? 1 : 2;
  return rows[la * NPK + (f - ((f > la) ? 1 : 0)) * K + j] + rows[lb * NPK + (f - ((f > lb) ? 1 : 0)) * K + j];
}

__device__ __forceinline__ int xcd_remap2(int b, int n)
{
  const int q = n / 8, r = n % 8, x = b % 8;
  return ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

#ifndef EQLB_STRESS_THREADS
#define EQLB_STRESS_THREADS 512
#endif
#ifndef EQLB_STRESS_WAVES
#define EQLB_STRESS_WAVES 2 // waves per SIMD asked from the register allocator
#endif
#ifndef EQLB_STRESS_TILE_CELLS
#define EQLB_STRESS_TILE_CELLS 524
#endif
// LARGEST tile: 2 rows x 524 x 18 doubles + 11.9 KB of tables = 159.3 KB; the body needs the whole register budget
// of two waves per SIMD, so one 8-wave workgroup per CU is resident anyway and may use all of its LDS.  The
// tile builder picks the size below this that fills whole rounds of the 256 workgroup slots (1M triangles:
// 489 cells = 2 045 tiles = 8 rounds, 0.504 ms; 448 cells = 2 233 tiles = 8.7 rounds cost 0.538 ms)
constexpr int STRESS_TCMAX = EQLB_STRESS_TILE_CELLS;
} // namespace

// MIXED = false: the tile lists hold full patches only (the specialised instance of the body: 0 B of scratch) - the
// crossed benchmark meshes, where everything else is 0.4 % of the patches and goes with the rest.  MIXED = true: the
// tile lists hold every patch of up to 8 lanes, the full ones first; whole wave-blocks of full patches run the
// specialised instance, the others the generic one (interior patches with fewer cells than lanes, boundary patches
// without stress flux BCs).  The spills of the generic instance cost every wave of the kernel they are in, which is
// why the two kernels exist: on unstructured meshes (valence 5 - 7 in groups of 8 lanes) most patches are generic.
template <bool MIXED>
__global__ void __launch_bounds__(EQLB_STRESS_THREADS, EQLB_STRESS_WAVES)
k_se_stress_tiled(const SeArgs a0, const TileArgs ta, const StressRows rows)
{
  constexpr int THREADS = EQLB_STRESS_THREADS;
  extern __shared__ __align__(16) double lds[];
  using Z = Sizes<2, 1, 8>;
  constexpr int NVW = Z::NVT + Z::NVQT;
  const int TC = ta.tc;
  const int tile = ta.tile_first + xcd_remap2(blockIdx.x, ta.ntiles);
  double* sSlots = lds + Z::NTAB + NVW;
  for (int i = threadIdx.x; i < Z::NTAB; i += THREADS)
    lds[i] = a0.tables[Z::NS + i];
  for (int i = threadIdx.x; i < NVW; i += THREADS)
    lds[Z::NTAB + i] = a0.tables[Z::OFF_V + i];
  // rows of vertices whose patch is not solved here (node mask, patches of more than 8 facets: they go
  // through the generic kernels) must read as zero in the flush
  if (ta.tiles[tile].zero)
    for (int i = threadIdx.x; i < 2 * TC * 3 * SNPK; i += THREADS)
      sSlots[i] = 0.0;
  __syncthreads();

  const TileDesc& td = ta.tiles[tile];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = THREADS / 64;
  int u = wave;
  SeArgs a = a0;
#define EQLB_STRESS_BIN(B, PP)                                                                      \
  {                                                                                                 \
    /* only FULL patches are listed (interior, PP cells), padded to whole wave-blocks with copies that own \
       no cell: the generic instance of the body (boundary patches, fewer cells than lanes) needs 40 \
       registers more than there are and the scratch it brings slows every wave of the kernel; those \
       patches run on the generic kernels (slot path) in the same call */                           \
    const int np = td.npatch[B];                                                                    \
    const int nwb = MIXED ? ((np * PP + 63) >> 6) : ((np * PP) >> 6);                               \
    a.npatch = np;                                                                                  \
    a.slot_offset = td.slot_start[B];                                                               \
    a.patch_offset = td.patch_start[B];                                                             \
    const int nwb_full = MIXED ? ((td.nfull[B] * PP) >> 6) : nwb;                                   \
    /* separate loops, not one loop with a branch: the register allocation of the full-patch instance (no spills \
       on its own) is then not tied to the others (same wave-block -> wave assignment) */            \
    for (; u < nwb_full; u += NW)                                                                   \
      stress_patch_body<PP, true>(a, rows, lds, (int64_t)u * 64 + lane, sSlots, TC);                \
    if constexpr (MIXED)                                                                            \
    {                                                                                               \
      /* interior patches with PP - 1, PP - 2, PP - 3 cells (the tile lists are ordered by it): the whole wave-blocks \
         inside their ranges run the instance with that patch size at compile time */                \
      constexpr int PER = 64 / PP;                                                                  \
      int c0[3], c1[3];                                                                             \
      _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                 \
      {                                                                                             \
        const int first = (j == 0) ? td.nfull[B] : td.nval[B][j - 1];                               \
        c0[j] = (first + PER - 1) / PER;                                                            \
        c1[j] = (EQLB_STRESS_NFIX && PP - 1 - j >= 3) ? td.nval[B][j] / PER : 0;                    \
        if (c1[j] < c0[j])                                                                          \
          c1[j] = c0[j];                                                                            \
      }                                                                                             \
      if constexpr (EQLB_STRESS_NFIX && PP - 1 >= 3)                                                \
        for (int v = c0[0] + wave; v < c1[0]; v += NW)                                              \
          stress_patch_body<PP, false, PP - 1>(a, rows, lds, (int64_t)v * 64 + lane, sSlots, TC);   \
      if constexpr (EQLB_STRESS_NFIX && PP - 2 >= 3)                                                \
        for (int v = c0[1] + ((wave + 3) & (NW - 1)); v < c1[1]; v += NW)                           \
          stress_patch_body<PP, false, (PP - 2 >= 3 ? PP - 2 : 0)>(a, rows, lds, (int64_t)v * 64 + lane, sSlots, TC); \
      if constexpr (EQLB_STRESS_NFIX && PP - 3 >= 3)                                                \
        for (int v = c0[2] + ((wave + 5) & (NW - 1)); v < c1[2]; v += NW)                           \
          stress_patch_body<PP, false, (PP - 3 >= 3 ? PP - 3 : 0)>(a, rows, lds, (int64_t)v * 64 + lane, sSlots, TC); \
      /* everything else: the generic instance */                                                   \
      for (; u < nwb; u += NW)                                                                      \
      {                                                                                             \
        if ((u >= c0[0] && u < c1[0]) || (u >= c0[1] && u < c1[1]) || (u >= c0[2] && u < c1[2]))    \
          continue;                                                                                 \
        stress_patch_body<PP, false>(a, rows, lds, (int64_t)u * 64 + lane, sSlots, TC);             \
      }                                                                                             \
    }                                                                                               \
    u -= nwb;                                                                                       \
  }
  EQLB_STRESS_BIN(0, 4)
  EQLB_STRESS_BIN(1, 8)
#undef EQLB_STRESS_BIN

  // ---- flush: x[r][cell][i] (+)= row(v0) + row(v1) + row(v2), fixed order; the old values are fetched
  // before the barrier ----
  const int32_t* cells = ta.tile_cells + (int64_t)tile * TC;
  constexpr int NIT = (2 * STRESS_TCMAX * SNRT / 2 + THREADS - 1) / THREADS;
  double2 xv[NIT];
  int64_t xi[NIT];
  const int per_row = TC * SNRT;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
  {
    const int e = (it * THREADS + threadIdx.x) * 2;
    const int r = (e >= per_row) ? 1 : 0;
    const int er = e - r * per_row;
    const int cl = er / SNRT, i = er - cl * SNRT;
    const int32_t cell = (e < 2 * per_row) ? cells[cl] : -1;
    xi[it] = (cell >= 0) ? (int64_t)cell * SNRT + i : -1;
    xv[it] = make_double2(0.0, 0.0);
    if (xi[it] >= 0 && ta.accumulate)
      xv[it] = *reinterpret_cast<const double2*>(rows.x[r] + xi[it]);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it)
  {
    const int e = (it * THREADS + threadIdx.x) * 2;
    const int r = (e >= per_row) ? 1 : 0;
    const int er = e - r * per_row;
    const int cl = er / SNRT, i = er - cl * SNRT;
    if (xi[it] >= 0)
    {
      const double* sl = sSlots + ((int64_t)r * TC + cl) * 3 * SNPK;
      double2 t;
      t.x = xv[it].x + packed_sum2(sl, i);
      t.y = xv[it].y + packed_sum2(sl, i + 1);
      *reinterpret_cast<double2*>(rows.x[r] + xi[it]) = t;
    }
  }
}

int stress_tile_cells() { return STRESS_TCMAX; }

int launch_se_stress_tiled(const SeArgs& a, const TileArgs& t, const double* const* g, const double* const* f,
                           double* const* x, hipStream_t stream, bool mixed)
{
  using Z = Sizes<2, 1, 8>;
  if (t.tc < 1 || t.tc > STRESS_TCMAX)
    return EQLB_ERR_UNSUPPORTED;
  const size_t lds_bytes = sizeof(double) * ((size_t)Z::NTAB + Z::NVT + Z::NVQT + (size_t)2 * t.tc * 3 * SNPK);
  if (lds_bytes > 160 * 1024)
    return EQLB_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set)
  {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_se_stress_tiled<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess
        || hipFuncSetAttribute(reinterpret_cast<const void*>(k_se_stress_tiled<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EQLB_ERR_DEVICE;
    attr_set = true;
  }
  if (t.ntiles == 0)
    return 0;
  StressRows rows{{g[0], g[1]}, {f[0], f[1]}, {x[0], x[1]}};
  if (!mixed)
    hipLaunchKernelGGL(k_se_stress_tiled<false>, dim3((unsigned)t.ntiles), dim3(EQLB_STRESS_THREADS), lds_bytes, stream,
                       a, t, rows);
  else
    hipLaunchKernelGGL(k_se_stress_tiled<true>, dim3((unsigned)t.ntiles), dim3(EQLB_STRESS_THREADS), lds_bytes, stream,
                       a, t, rows);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

} // namespace eqlb
