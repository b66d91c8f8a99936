// Weak symmetry of equilibrated stresses on gfx950: impose_weak_symmetry
// (cpp/dolfinx_eqlb/se/solve_patch_weaksym.hpp:59-233) with assemble_stressminimiser
// (se/assembly.hpp:292-472), the kernel of se/stressmin_kernel.hpp:76-248 and the Schur solve
// PatchData::solve_constrained_minimisation (se/PatchData.hpp:598-663).
//
// Runs after the row-wise flux kernels: the slot buffer then holds exactly the patch-local stress
// rows sigma_a (one slot row per (cell, vertex)), which is what the reference feeds into the
// constrained minimisation (:134-142).  Per patch (P lanes, one per cell, as in k_se_patch):
//   saddle system  [A 0 B0; 0 A B1; B0^T B1^T 0(+mean-value row)] [u0; u1; gamma] = [0; 0; Lc]
//   B0(i,j) = int (Phi_i)_y psi_j,  B1(i,j) = -int (Phi_i)_x psi_j,  Lc(j) = -int psi_j (s01 - s10)
// over the patch-wise H(div=0) functions Phi_i and the patch P1 functions psi_j.  Quadrature-free:
// Phi/psi products are contractions with the constant tensors V, VQ (tools/gen_tables.py).
// Solve: Cholesky A = L L^T in LDS, Y_k = L^-1 B_k (one column per lane, no synchronisation),
// Schur complement C = -sum_k Y_k^T Y_k (+ multiplier row), dense LU with partial pivoting of the
// (npnt+1)^2 system, u_k = -L^-T (Y_k gamma); the corrections are added to the slot rows in place.
// With flux BCs on a stress row the masked A (and B rows) of that row are used (:611-656).
#include "eqlb_device_common.h"
#include "eqlb_tables_gen.h"

namespace eqlb
{

// relative size (against the largest entry of the Schur system) below which a pivot counts as zero
#ifndef EQLB_WS_PIVOT_RTOL
#define EQLB_WS_PIVOT_RTOL 1e-11
#endif

template <int K, int P>
struct WsSizes
{
  using Z = Sizes<K, K - 1, P>;
  static constexpr int KB = Z::KB, NADD = Z::NADD, NH = Z::NH, NRT = Z::NRT, NTE = Z::NTE;
  static constexpr int DIMMAX = Z::DIMMAX;      // H(div=0) unknowns of a patch
  static constexpr int NPMAX = P + 2;           // patch nodes (multiplier DOFs)
  static constexpr int DCMAX = NPMAX + 1;       // + mean-value multiplier
  static constexpr int TRI = DIMMAX * (DIMMAX + 1) / 2;
  // per patch: A0, A1 (packed lower) | Y [DIMMAX][2*NPMAX] | C [DCMAX][DCMAX] | rhs_c [DCMAX] | w [2][DIMMAX]
  static constexpr int OFF_A1 = TRI, OFF_Y = 2 * TRI, OFF_C = OFF_Y + DIMMAX * 2 * NPMAX;
  static constexpr int OFF_R = OFF_C + DCMAX * DCMAX, OFF_W = OFF_R + DCMAX;
  static constexpr int OFF_D = OFF_W + 2 * DIMMAX; // 1 / L_ii of both factors
  static constexpr int GROUP = OFF_D + 2 * DIMMAX;
  // register variants of the serial phases (columns of Y, rows of the Schur system) where they fit
  static constexpr bool REG_Y = DIMMAX <= 32;
  // small patches (k = 2, up to 8 cells): every lane keeps its own copy of the Cholesky factor in
  // registers - the factorisation and all substitutions run without LDS round trips or syncs
  static constexpr bool REG_A = DIMMAX <= 9;
  static constexpr bool REG_LU = P <= 16;
  static constexpr int NTAB = Z::NTET + Z::NVT + Z::NVQT; // TE | V | VQ
  // one wave per block; as many patch groups as fit the 160 KB of LDS (high-valence bins of k = 3
  // run fewer groups per wave)
  static constexpr int GROUPS_FIT = (160 * 1024 / 8 - NTAB) / GROUP;
  static constexpr int GROUPS = (GROUPS_FIT < 1) ? 1 : ((GROUPS_FIT < 64 / P) ? GROUPS_FIT : 64 / P);
  static constexpr int BLOCK = GROUPS * P;
  static constexpr int lds_doubles() { return NTAB + GROUPS * GROUP; }
};

template <int K, int P>
__global__ void __launch_bounds__(64) k_se_weaksym(const SeArgs a)
{
  using W = WsSizes<K, P>;
  using Z = typename W::Z;
  constexpr int KB = W::KB, NADD = W::NADD, NH = W::NH, NRT = W::NRT, NTE = W::NTE;
  constexpr int NPMAX = W::NPMAX, DCMAX = W::DCMAX, LDY = 2 * W::NPMAX;

  extern __shared__ double lds[];
  double* sTE = lds;
  double* sV = sTE + Z::NTET;
  double* sVQ = sV + Z::NVT;
  double* sG = sVQ + Z::NVQT;

  const int tid = threadIdx.x;
  for (int i = tid; i < Z::NTET; i += W::BLOCK)
    sTE[i] = a.tables[Z::OFF_TE + i];
  for (int i = tid; i < Z::NVT + Z::NVQT; i += W::BLOCK)
    sV[i] = a.tables[Z::OFF_V + i];
  __syncthreads();

  const int lane = tid & 63;
  const int sub = lane % P;
  const int64_t patch_local = ((int64_t)blockIdx.x * W::BLOCK + tid) / P;
  const int64_t slot = a.slot_offset + patch_local * P + sub;
  const int64_t patch = a.patch_offset + patch_local;
  // two-cell patches of a group have no weak-symmetry step of their own (se/reconstruction.hpp:
  // 181-229: it is imposed once, on the internal patch of the group)
  const uint8_t flag0 = (patch_local < a.npatch) ? a.pflag[patch] : (uint8_t)PFLAG_INTERIOR;
  // (patches of another level of overlapping groups are left to that level's pass; plain patches: level 0)
  const bool pvalid = patch_local < a.npatch && (flag0 & PFLAG_WS_SKIP) == 0
                      && (int)((flag0 >> PFLAG_WS_LEVEL_SHIFT) & 3) == a.ws_level;
  const bool grouped = (flag0 & PFLAG_WS_GROUP) != 0;
  const int n = pvalid ? (int)a.pn[patch] : 0;
  const bool active = pvalid && sub < n;
  const int32_t cell = active ? a.slot_cell[slot] : 0;
  const uint32_t info = active ? a.slot_info[slot] : 0u;
  const int fm = (info >> INFO_FM_SHIFT) & 3, fp = (info >> INFO_FP_SHIFT) & 3;
  const int ln = (info >> INFO_LN_SHIFT) & 3;
  const bool rev_m = (info & INFO_REV_M) != 0;
  const int ci = active ? combo_index(fm, fp, rev_m) : 0;

  double J[2][2] = {{1.0, 0.0}, {0.0, 1.0}};
  if (active)
  {
    const double2* Jp = reinterpret_cast<const double2*>(a.cellJ + 4 * (int64_t)cell);
    const double2 j0 = Jp[0], j1 = Jp[1];
    J[0][0] = j0.x;
    J[0][1] = j0.y;
    J[1][0] = j1.x;
    J[1][1] = j1.y;
  }
  const double detJ = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  const double sgn = (detJ > 0.0) ? 1.0 : -1.0;
  const double pf_m = (fm == 1) ? sgn : -sgn, pf_p = (fp == 1) ? sgn : -sgn;

  const uint8_t flag1 = pvalid ? a.pflag[a.npatch_total + patch] : (uint8_t)PFLAG_INTERIOR;
  const bool interior = !pvalid || (flag0 & PFLAG_INTERIOR) != 0;
  const int nf = interior ? n : n + 1;
  const int nn = (n > 0) ? n : 1;
  const int fi_p = interior ? ((sub + 1 < nn) ? sub + 1 : 0) : sub + 1;
  const int dim = pvalid ? 1 + KB * nf + NADD * n : 0;
  const int npnt = nf + 1;
  // flux BCs of the two rows (bits as in k_se_patch); PatchData::reinitialisation :175-206
  const bool bc0[2] = {(flag0 & PFLAG_BC0) != 0, (flag1 & PFLAG_BC0) != 0};
  const bool bcn[2] = {(flag0 & PFLAG_BCN) != 0, (flag1 & PFLAG_BCN) != 0};
  const bool requires_bcs = bc0[0] || bcn[0] || bc0[1] || bcn[1];
  // mean-value multiplier unless some row has a primal-Dirichlet end (type essnt_primal or mixed)
  const bool row_dual[2] = {!interior && bc0[0] && bcn[0], !interior && bc0[1] && bcn[1]};
  const bool meanvalue = interior || (row_dual[0] && row_dual[1]);
  const int dim_c = meanvalue ? npnt + 1 : npnt;

  double* Ag = sG + (tid / P) * W::GROUP; // A of row 0 (and of both rows without flux BCs)
  double* Yg = Ag + W::OFF_Y;
  double* Cg = Ag + W::OFF_C;
  double* Rg = Ag + W::OFF_R;
  double* Wg = Ag + W::OFF_W;
  double* Dg = Ag + W::OFF_D;

  // ---- element quantities ----
  double Te[NH][NH];
  {
    const double ia = active ? 1.0 / fabs(detJ) : 0.0;
    const double g0 = (J[0][0] * J[0][0] + J[1][0] * J[1][0]) * ia,
                 g1 = (J[0][0] * J[0][1] + J[1][0] * J[1][1]) * ia,
                 g2 = (J[0][1] * J[0][1] + J[1][1] * J[1][1]) * ia;
    const double* te = sTE + ci * 3 * Z::NTES;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int g = 0; g <= h; ++g)
      {
        const int e = h * (h + 1) / 2 + g;
        const double v = g0 * te[e] + g1 * te[Z::NTES + e] + g2 * te[2 * Z::NTES + e];
        Te[h][g] = v;
        Te[g][h] = v;
      }
  }
  // Be[k][h][j]: k = 0: int (Phi_h)_y psi_j ; k = 1: -int (Phi_h)_x psi_j
  double Be[2][NH][3];
  {
    const double* vq = sVQ + ci * 2 * NH * 3;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
        const double v0 = vq[h * 3 + j], v1 = vq[(NH + h) * 3 + j]; // X = 0, 1
        Be[0][h][j] = active ? (J[1][0] * v0 + J[1][1] * v1) : 0.0;
        Be[1][h][j] = active ? -(J[0][0] * v0 + J[0][1] * v1) : 0.0;
      }
  }
  // patch-local stress rows from the slots; Lc_e[j] = -int psi_j (s01 - s10), Ce = |detJ|/6
  double* srow[2] = {nullptr, nullptr};
  double Lce[3] = {0.0, 0.0, 0.0};
  if (active)
  {
    srow[0] = a.out + (((int64_t)0 * a.ncells + cell) * 3 + ln) * NRT;
    srow[1] = a.out + (((int64_t)1 * a.ncells + cell) * 3 + ln) * NRT;
    // grouped patches (modified_patch, se/solve_patch_weaksym.hpp:100-131): the stress accumulated by
    // the group so far = own rows + the rows of the group's two-cell patches on this cell
    const uint32_t grows = grouped ? ((info >> INFO_GROUPROW_SHIFT) & 7u) : 0u;
#pragma unroll
    for (int i = 0; i < NRT; ++i)
    {
      double c0 = srow[0][i], c1 = srow[1][i];
      if (grows)
      {
#pragma unroll
        for (int v = 0; v < 3; ++v)
          if (grows & (1u << v))
          {
            c0 += a.out[(((int64_t)0 * a.ncells + cell) * 3 + v) * NRT + i];
            c1 += a.out[(((int64_t)1 * a.ncells + cell) * 3 + v) * NRT + i];
          }
      }
      const double w0 = c0 * J[1][0] - c1 * J[0][0], w1 = c0 * J[1][1] - c1 * J[0][1];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Lce[j] -= sgn * (w0 * sV[(j * NRT + i) * 2] + w1 * sV[(j * NRT + i) * 2 + 1]);
    }
  }
  const double Ce = active ? fabs(detJ) / 6.0 : 0.0;

  // ---- numbering ----
  int gi[NH];
  gi[0] = 0;
#pragma unroll
  for (int j = 0; j < KB; ++j)
  {
    gi[1 + j] = 1 + sub * KB + j;
    gi[1 + KB + j] = 1 + fi_p * KB + j;
  }
#pragma unroll
  for (int q = 0; q < NADD; ++q)
    gi[1 + 2 * KB + q] = 1 + nf * KB + sub * NADD + q;
  // multiplier DOF of the cell's local vertex j (se/Patch.hpp:621-708)
  int pj[3];
  {
    const int v_ea = 3 - fp - ln, v_eam1 = 3 - fm - ln;
    const int p_ea = interior ? sub + 1 : ((sub + 1 == n) ? nf : sub + 1);
    const int p_eam1 = interior ? ((sub == 0) ? n : sub) : ((sub == 0) ? nf - 1 : sub);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      pj[j] = (j == ln) ? 0 : ((j == v_ea) ? p_ea : ((j == v_eam1) ? p_eam1 : 0));
  }
  // fixed (flux-BC) unknowns per row k (se/assembly.hpp:46-98): local unknown h of this lane
  auto fixed = [&](int k, int h) {
    const bool dfx = bc0[k] || bcn[k];
    if (h == 0)
      return dfx;
    if (h <= KB)
      return bc0[k] && sub == 0;
    if (h <= 2 * KB)
      return bcn[k] && sub == n - 1;
    return false;
  };

  // ---- zero the tile ----
  for (int e = sub; e < W::GROUP; e += P)
    Ag[e] = 0.0;
  wave_sync();

  // ---- assembly: A (per row if masked), B, C column, Lc  (LDS atomics: entries are shared by
  //      neighbouring cells; the addition order is fixed by the lane order) ----
  if (active)
  {
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      if (k == 1 && !requires_bcs)
        break;
      double* Ak = Ag + k * W::OFF_A1;
#pragma unroll
      for (int h = 0; h < NH; ++h)
      {
        if (fixed(k, h) && requires_bcs)
          continue;
#pragma unroll
        for (int g = 0; g < NH; ++g)
          if (gi[h] >= gi[g] && !(fixed(k, g) && requires_bcs))
            atomicAdd(&Ak[tri(gi[h], gi[g])], Te[h][g]);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int h = 0; h < NH; ++h)
      {
        if (fixed(k, h) && requires_bcs)
          continue; // rows of fixed unknowns are dropped (se/assembly.hpp:430-436)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          atomicAdd(&Yg[gi[h] * LDY + k * NPMAX + pj[j]], Be[k][h][j]);
      }
#pragma unroll
    for (int j = 0; j < 3; ++j)
    {
      atomicAdd(&Rg[pj[j]], Lce[j]);
      if (meanvalue)
      {
        atomicAdd(&Cg[pj[j] * DCMAX + npnt], Ce);
        atomicAdd(&Cg[npnt * DCMAX + pj[j]], Ce);
      }
    }
  }
  wave_sync();
  // identity rows of the fixed unknowns
  if (requires_bcs && sub == 0 && pvalid)
  {
    for (int k = 0; k < 2; ++k)
    {
      double* Ak = Ag + k * W::OFF_A1;
      if (bc0[k] || bcn[k])
        Ak[0] = 1.0;
      if (bc0[k])
        for (int j = 0; j < KB; ++j)
          Ak[tri(1 + j, 1 + j)] = 1.0;
      if (bcn[k])
        for (int j = 0; j < KB; ++j)
          Ak[tri(1 + n * KB + j, 1 + n * KB + j)] = 1.0;
    }
  }
  wave_sync();

  int status_local = 0;
  // ---- Cholesky of A (both rows if masked) ----
  constexpr int NLR = W::REG_A ? W::TRI : 1, NDR = W::REG_A ? W::DIMMAX : 1;
  double Lr[NLR], Dr[NDR];
  const bool use_reg = W::REG_A && !requires_bcs; // uniform within a patch group
#ifdef EQLB_WS_SKIP_CHOL // timing experiment (wrong results)
  for (int i = 0; i < NLR; ++i)
    Lr[i] = 1.0;
  for (int i = 0; i < NDR; ++i)
    Dr[i] = 1.0;
#else
  if constexpr (W::REG_A)
  {
    if (use_reg)
    {
      // rows beyond dim are padded with the identity
#pragma unroll
      for (int i = 0; i < W::DIMMAX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
          Lr[tri(i, j)] = (i < dim) ? Ag[tri(i, j)] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
      for (int j = 0; j < W::DIMMAX; ++j)
      {
        const double ajj = Lr[tri(j, j)];
        if (!(ajj > 0.0))
          status_local = 1;
        const double inv = rsqrt_d(ajj > 0.0 ? ajj : 1.0);
        Lr[tri(j, j)] = (ajj > 0.0 ? ajj : 1.0) * inv;
        Dr[j] = inv;
#pragma unroll
        for (int i = j + 1; i < W::DIMMAX; ++i)
          Lr[tri(i, j)] *= inv;
#pragma unroll
        for (int i = j + 1; i < W::DIMMAX; ++i)
#pragma unroll
          for (int kk = j + 1; kk <= i; ++kk)
            Lr[tri(i, kk)] -= Lr[tri(i, j)] * Lr[tri(kk, j)];
      }
    }
  }
  if (!use_reg)
    for (int k = 0; k < 2; ++k)
    {
      if (k == 1 && !requires_bcs)
        break;
      double* Ak = Ag + k * W::OFF_A1;
      for (int j = 0; j < dim; ++j)
      {
        const double ajj = Ak[tri(j, j)];
        if (!(ajj > 0.0))
          status_local = 1;
        const double inv = rsqrt_d(ajj > 0.0 ? ajj : 1.0);
        const double ljj = (ajj > 0.0 ? ajj : 1.0) * inv;
        wave_sync();
        if (sub == 0)
          Dg[k * W::DIMMAX + j] = inv;
        for (int i = j + sub; i < dim; i += P)
          Ak[tri(i, j)] = (i == j) ? ljj : Ak[tri(i, j)] * inv;
        wave_sync();
        for (int i = j + 1 + sub; i < dim; i += P)
        {
          const double lij = Ak[tri(i, j)];
          for (int kk = j + 1; kk <= i; ++kk)
            Ak[tri(i, kk)] -= lij * Ak[tri(kk, j)];
        }
        wave_sync();
      }
    }
#ifndef EQLB_WS_SKIP_Y
#endif
  // ---- Y_k = L_k^-1 B_k: every lane forward-substitutes whole columns ----
  for (int col = sub; col < 2 * npnt; col += P)
  {
    const int k = col / npnt, c = col - k * npnt;
    const int ko = (requires_bcs && k == 1) ? 1 : 0;
    const double* Ak = Ag + ko * W::OFF_A1;
    const double* Dk = Dg + ko * W::DIMMAX;
    double* y = Yg + k * NPMAX + c;
    if constexpr (W::REG_Y)
    {
      // the column lives in registers: no LDS round trip inside the dependent chain
      double yr[W::DIMMAX];
#pragma unroll
      for (int i = 0; i < W::DIMMAX; ++i)
        yr[i] = (i < dim) ? y[i * LDY] : 0.0;
      bool done = false;
      if constexpr (W::REG_A)
      {
        if (use_reg)
        {
#pragma unroll
          for (int i = 0; i < W::DIMMAX; ++i)
          {
            double t = yr[i];
#pragma unroll
            for (int q = 0; q < i; ++q)
              t -= Lr[tri(i, q)] * yr[q];
            yr[i] = t * Dr[i];
          }
          done = true;
        }
      }
      if (!done)
      {
#pragma unroll
        for (int i = 0; i < W::DIMMAX; ++i)
        {
          if (i < dim)
          {
            double t = yr[i];
#pragma unroll
            for (int q = 0; q < i; ++q)
              t -= Ak[tri(i, q)] * yr[q];
            yr[i] = t * Dk[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < W::DIMMAX; ++i)
        if (i < dim)
          y[i * LDY] = yr[i];
    }
    else
    {
      for (int i = 0; i < dim; ++i)
      {
        double t = y[i * LDY];
        for (int q = 0; q < i; ++q)
          t -= Ak[tri(i, q)] * y[q * LDY];
        y[i * LDY] = t * Dk[i];
      }
    }
  }
  wave_sync();
#ifndef EQLB_WS_SKIP_SCHUR
#endif
  // ---- Schur complement C -= sum_k Y_k^T Y_k ----
  for (int e = sub; e < npnt * npnt; e += P)
  {
    const int r = e / npnt, c = e - r * npnt;
    double t = 0.0;
    for (int k = 0; k < 2; ++k)
      for (int i = 0; i < dim; ++i)
        t += Yg[i * LDY + k * NPMAX + r] * Yg[i * LDY + k * NPMAX + c];
    Cg[r * DCMAX + c] -= t;
  }
  wave_sync();
#endif
#ifndef EQLB_WS_SKIP_LU
  // ---- dense LU with partial pivoting of the (npnt [+1])^2 Schur system ----
  if constexpr (W::REG_LU)
  {
    // rows r = sub and r = sub + P of [C | rhs] in registers; per column: cross-lane arg-max of the
    // pivot candidates (first maximum, like a serial partial-pivot LU), broadcast of the pivot row,
    // elimination in all rows not used yet (implicit row permutation), then back substitution
    const int gb = (tid & 63) - sub;
    double r0[DCMAX + 1], r1[DCMAX + 1];
#pragma unroll
    for (int j = 0; j < DCMAX; ++j)
    {
      r0[j] = (sub < dim_c && j < dim_c) ? Cg[sub * DCMAX + j] : 0.0;
      r1[j] = (sub + P < dim_c && j < dim_c) ? Cg[(sub + P) * DCMAX + j] : 0.0;
    }
    r0[DCMAX] = (sub < dim_c) ? Rg[sub] : 0.0;
    r1[DCMAX] = (sub + P < dim_c) ? Rg[sub + P] : 0.0;
    bool free0 = sub < dim_c, free1 = sub + P < dim_c;
    // Rank-revealing threshold.  With DIFFERENT boundary types on the two stress rows the Schur matrix can
    // lose rank beyond the constant mode (tests/stress_rank.py); a null vector z of it has B_k z = 0 for both
    // rows, so it does not change u_k = -A^-1 B_k gamma, and on a Galerkin stress the system is consistent:
    // the multiplier of a column without pivot is set to zero where the reference's PartialPivLU
    // (se/PatchData.hpp:631-637) divides by a rounding-level pivot - same stress, any member of gamma.
    double cscale = 0.0;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j)
      cscale = fmax(cscale, fmax(fabs(r0[j]), fabs(r1[j])));
#pragma unroll
    for (int off = 1; off < P; off <<= 1)
      cscale = fmax(cscale, __shfl(cscale, gb + (sub ^ off), 64));
    const double ptol = EQLB_WS_PIVOT_RTOL * cscale;
    int prow[DCMAX]; // pivot row of column c (-1: no pivot, multiplier 0)
#pragma unroll
    for (int c = 0; c < DCMAX; ++c)
    {
      prow[c] = 0;
      if (c < dim_c)
      {
        double bv = free0 ? fabs(r0[c]) : -1.0;
        int br = sub;
        const double v1 = free1 ? fabs(r1[c]) : -1.0;
        if (v1 > bv)
        {
          bv = v1;
          br = sub + P;
        }
#pragma unroll
        for (int off = 1; off < P; off <<= 1)
        {
          const double ov = __shfl(bv, gb + (sub ^ off), 64);
          const int orow = __shfl(br, gb + (sub ^ off), 64);
          if (ov > bv || (ov == bv && orow < br))
          {
            bv = ov;
            br = orow;
          }
        }
        if (!(bv > ptol))
        {
          prow[c] = -1; // uniform within the group: bv, br are reduced over all lanes
          continue;
        }
        prow[c] = br;
        const int owner = gb + (br % P);
        const bool second = br >= P;
        double pr[DCMAX + 1];
#pragma unroll
        for (int j = c; j <= DCMAX; ++j)
          pr[j] = __shfl(second ? r1[j] : r0[j], owner, 64);
        const double ip = rcp_d((pr[c] != 0.0) ? pr[c] : 1.0);
        if (br == sub)
          free0 = false;
        if (br == sub + P)
          free1 = false;
        const double f0 = free0 ? r0[c] * ip : 0.0, f1 = free1 ? r1[c] * ip : 0.0;
#pragma unroll
        for (int j = c; j <= DCMAX; ++j)
        {
          r0[j] -= f0 * pr[j];
          r1[j] -= f1 * pr[j];
        }
      }
    }
    // back substitution: gamma[c] from the pivot row of column c, broadcast to the group
    double gam[DCMAX];
#pragma unroll
    for (int c = DCMAX - 1; c >= 0; --c)
    {
      gam[c] = 0.0;
      if (c < dim_c && prow[c] >= 0)
      {
        const int br = prow[c];
        const bool second = br >= P;
        double t = second ? r1[DCMAX] : r0[DCMAX];
#pragma unroll
        for (int j = c + 1; j < DCMAX; ++j)
          t -= (second ? r1[j] : r0[j]) * gam[j];
        const double d = second ? r1[c] : r0[c];
        t *= rcp_d((d != 0.0) ? d : 1.0);
        gam[c] = __shfl(t, gb + (br % P), 64);
      }
    }
    wave_sync();
#pragma unroll
    for (int c = 0; c < DCMAX; ++c)
      if (sub == 0 && pvalid && c < dim_c)
        Rg[c] = gam[c];
  }
  else if (sub == 0 && pvalid)
  {
    // serial variant (bins of more than 16 lanes): Gauss-Jordan with row pivoting and the same
    // rank-revealing threshold; rows are consumed in order, a column without pivot gets multiplier 0
    double cscale = 0.0;
    for (int r = 0; r < dim_c; ++r)
      for (int j = 0; j < dim_c; ++j)
        cscale = fmax(cscale, fabs(Cg[r * DCMAX + j]));
    const double ptol = EQLB_WS_PIVOT_RTOL * cscale;
    int nr = 0; // rows used so far
    int8_t pcol[DCMAX]; // row that eliminated column c, -1: none
    for (int c = 0; c < dim_c; ++c)
    {
      int piv = nr;
      double best = (nr < dim_c) ? fabs(Cg[nr * DCMAX + c]) : 0.0;
      for (int r = nr + 1; r < dim_c; ++r)
      {
        const double v = fabs(Cg[r * DCMAX + c]);
        if (v > best)
        {
          best = v;
          piv = r;
        }
      }
      if (!(best > ptol))
      {
        pcol[c] = -1;
        continue;
      }
      if (piv != nr)
      {
        for (int j = 0; j < dim_c; ++j)
        {
          const double t = Cg[nr * DCMAX + j];
          Cg[nr * DCMAX + j] = Cg[piv * DCMAX + j];
          Cg[piv * DCMAX + j] = t;
        }
        const double t = Rg[nr];
        Rg[nr] = Rg[piv];
        Rg[piv] = t;
      }
      const double ip = 1.0 / Cg[nr * DCMAX + c];
      for (int r = nr + 1; r < dim_c; ++r)
      {
        const double f = Cg[r * DCMAX + c] * ip;
        for (int j = c; j < dim_c; ++j)
          Cg[r * DCMAX + j] -= f * Cg[nr * DCMAX + j];
        Rg[r] -= f * Rg[nr];
      }
      pcol[c] = (int8_t)nr;
      ++nr;
    }
    // back substitution over the pivot columns, right to left; gamma goes to Wg first (Rg holds the rows)
    double* gm = Wg; // free until the u_k step
    for (int c = dim_c - 1; c >= 0; --c)
    {
      if (pcol[c] < 0)
      {
        gm[c] = 0.0;
        continue;
      }
      const int r = pcol[c];
      double t = Rg[r];
      for (int j = c + 1; j < dim_c; ++j)
        t -= Cg[r * DCMAX + j] * gm[j];
      gm[c] = t / Cg[r * DCMAX + c];
    }
    for (int c = 0; c < dim_c; ++c)
      Rg[c] = gm[c];
  }
  wave_sync();
#endif
  // ---- u_k = -L_k^-T (Y_k gamma) ----
  for (int e = sub; e < 2 * dim; e += P)
  {
    const int k = e / dim, i = e - k * dim;
    double t = 0.0;
    for (int c = 0; c < npnt; ++c)
      t -= Yg[i * LDY + k * NPMAX + c] * Rg[c];
    Wg[k * W::DIMMAX + i] = t;
  }
  wave_sync();
  if (W::REG_A && use_reg)
  {
    if constexpr (W::REG_A)
    {
      if (sub < 2 && pvalid)
      {
        double* w = Wg + sub * W::DIMMAX;
        double wr[W::DIMMAX];
#pragma unroll
        for (int i = 0; i < W::DIMMAX; ++i)
          wr[i] = (i < dim) ? w[i] : 0.0;
#pragma unroll
        for (int i = W::DIMMAX - 1; i >= 0; --i)
        {
          double t = wr[i];
#pragma unroll
          for (int q = i + 1; q < W::DIMMAX; ++q)
            t -= Lr[tri(q, i)] * wr[q];
          wr[i] = t * Dr[i];
        }
#pragma unroll
        for (int i = 0; i < W::DIMMAX; ++i)
          if (i < dim)
            w[i] = wr[i];
      }
    }
  }
  else if (sub < 2 && pvalid)
  {
    const int k = sub;
    const int ko = (requires_bcs && k == 1) ? 1 : 0;
    const double* Ak = Ag + ko * W::OFF_A1;
    const double* Dk = Dg + ko * W::DIMMAX;
    double* w = Wg + k * W::DIMMAX;
    for (int i = dim - 1; i >= 0; --i)
    {
      double t = w[i];
      for (int q = i + 1; q < dim; ++q)
        t -= Ak[tri(q, i)] * w[q];
      w[i] = t * Dk[i];
    }
  }
  wave_sync();

  // ---- back-map and add to the slot rows (se/solve_patch_weaksym.hpp:189-232) ----
  if (active)
  {
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      const double* w = Wg + k * W::DIMMAX;
      double ul[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h)
        ul[h] = w[gi[h]];
      double* o = srow[k];
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c)
          s -= (rev_m ? bcoef(j, c) : ((j == c) ? 1.0 : 0.0)) * ul[c];
        const double yp = (j == 0) ? ul[0] : ul[KB + j];
        o[fm * K + j] += pf_m * s;
        o[fp * K + j] += pf_p * yp;
      }
#pragma unroll
      for (int q = 0; q < NADD; ++q)
        o[3 * K + Z::NDIV + q] += sgn * ul[1 + 2 * KB + q];
    }
  }
  if (status_local)
    atomicOr(a.status, 2);
}

// ---- lean variant: k = 2, patches of up to 8 cells, NO flux BCs on the stress rows -----------------
// (the benchmark / pure-Dirichlet case).  Same solution, a fraction of the LDS and no pivoting: the
// Cholesky factor of A (<= 9 x 9) is computed in registers and parked in LDS, Y_k = L^-1 B_k is formed
// per stress row in one 9 x 10 buffer (every lane keeps the node column and the column of its ring
// point), the Schur matrix S = sum_k Y_k^T Y_k is symmetric positive (semi-)definite and is eliminated
// without pivoting - row of ring point sub + 1 in lane sub, row of the patch node in every lane, the
// mean-value multiplier of interior patches analytically (see below).
// 184 doubles of LDS per patch instead of 438.
#ifndef EQLB_WS_LEAN_WAVES
#define EQLB_WS_LEAN_WAVES 2
#endif
template <int P>
struct WsLean
{
  static constexpr int K = 2;
  using Z = Sizes<K, K - 1, P>;
  static constexpr int NH = Z::NH, NRT = Z::NRT, NTE = Z::NTE;
  static constexpr int DM = Z::DIMMAX;     // 1 + P  (<= 9)
  static constexpr int TRI = DM * (DM + 1) / 2;
  static constexpr int NPM = P + 2, DCM = NPM + 1, LDB = NPM;
  // per patch: L (packed lower) | 1/L_ii | Yb [2][DM][LDB] | Mv [NPM] | R [DCM] | W [2][DM]
  static constexpr int OFF_D = TRI, OFF_Y = OFF_D + DM, OFF_M = OFF_Y + 2 * DM * LDB, OFF_R = OFF_M + NPM,
                       OFF_W = OFF_R + DCM, GROUP = OFF_W + 2 * DM;
  static constexpr int NTAB = Z::NTET + Z::NVT + Z::NVQT;
  static constexpr int BLOCK = 256;
  static constexpr int lds_doubles() { return NTAB + (BLOCK / P) * GROUP; }
};

template <int P>
__global__ void __launch_bounds__(256, EQLB_WS_LEAN_WAVES) k_se_weaksym_lean(const SeArgs a)
{
  using W = WsLean<P>;
  using Z = typename W::Z;
  constexpr int K = 2, KB = 1, NH = W::NH, NRT = W::NRT, NTE = W::NTE, DM = W::DM, LDB = W::LDB;
  static_assert(DM <= 9 && NH == 3, "lean weak-symmetry kernel: k = 2, at most 8 cells");

  extern __shared__ double lds[];
  double* sTE = lds;
  double* sV = sTE + Z::NTET;
  double* sVQ = sV + Z::NVT;
  double* sG = sVQ + Z::NVQT;
  const int tid = threadIdx.x;
  for (int i = tid; i < Z::NTET; i += W::BLOCK)
    sTE[i] = a.tables[Z::OFF_TE + i];
  for (int i = tid; i < Z::NVT + Z::NVQT; i += W::BLOCK)
    sV[i] = a.tables[Z::OFF_V + i];
  __syncthreads();

  const int lane = tid & 63;
  const int sub = lane % P;
  const int gb = lane - sub;
  const int64_t patch_local = ((int64_t)blockIdx.x * W::BLOCK + tid) / P;
  const bool pvalid = patch_local < a.npatch;
  const int64_t slot = a.slot_offset + patch_local * P + sub;
  const int64_t patch = a.patch_offset + patch_local;
  // two load batches: (1) the descriptors of the lane (independent loads; the unused lanes of a
  // patch group hold cell = -1), (2) J and the two stress rows of the cell
  int n = 0;
  int32_t cell_raw = -1;
  uint32_t info = 0u;
  uint8_t flag0 = (uint8_t)PFLAG_INTERIOR;
  if (pvalid)
  {
    n = (int)a.pn[patch];
    cell_raw = a.slot_cell[slot];
    info = a.slot_info[slot];
    flag0 = a.pflag[patch];
  }
  const bool active = cell_raw >= 0;
  const int32_t cell = active ? cell_raw : 0;
  const int fm = (info >> INFO_FM_SHIFT) & 3, fp = (info >> INFO_FP_SHIFT) & 3;
  const int ln = (info >> INFO_LN_SHIFT) & 3;
  const bool rev_m = (info & INFO_REV_M) != 0;
  const int ci = active ? combo_index(fm, fp, rev_m) : 0;

  double J[2][2] = {{1.0, 0.0}, {0.0, 1.0}};
  double* srow[2] = {nullptr, nullptr};
  double2 sv[2][W::NRT / 2]; // the patch-local stress rows sigma_a of the cell
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < W::NRT / 2; ++i)
      sv[r][i] = make_double2(0.0, 0.0);
  if (active)
  {
    const double2* Jp = reinterpret_cast<const double2*>(a.cellJ + 4 * (int64_t)cell);
    srow[0] = a.out + (((int64_t)0 * a.ncells + cell) * 3 + ln) * W::NRT;
    srow[1] = a.out + (((int64_t)1 * a.ncells + cell) * 3 + ln) * W::NRT;
    const double2 j0 = Jp[0], j1 = Jp[1];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < W::NRT / 2; ++i)
        sv[r][i] = reinterpret_cast<const double2*>(srow[r])[i];
    J[0][0] = j0.x;
    J[0][1] = j0.y;
    J[1][0] = j1.x;
    J[1][1] = j1.y;
  }
  const double detJ = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  const double sgn = (detJ > 0.0) ? 1.0 : -1.0;
  const double pf_m = (fm == 1) ? sgn : -sgn, pf_p = (fp == 1) ? sgn : -sgn;

  const bool interior = (flag0 & PFLAG_INTERIOR) != 0;
  const int nf = interior ? n : n + 1;
  const int nn = (n > 0) ? n : 1;
  const int fi_p = interior ? ((sub + 1 < nn) ? sub + 1 : 0) : sub + 1;
  const int dim = pvalid ? 1 + nf : 0;
  const int npnt = nf + 1;
  const bool meanvalue = interior; // no flux BCs: boundary patches are of type essnt_primal

  double* Lg = sG + (tid / P) * W::GROUP;
  double* Dg = Lg + W::OFF_D;
  double* Yb = Lg + W::OFF_Y;
  double* Mv = Lg + W::OFF_M;
  double* Rg = Lg + W::OFF_R;
  double* Wg = Lg + W::OFF_W;

  // numbering (as in k_se_weaksym)
  int gi[NH];
  gi[0] = 0;
  gi[1] = 1 + sub;
  gi[2] = 1 + fi_p;
  int pj[3];
  {
    const int v_ea = 3 - fp - ln, v_eam1 = 3 - fm - ln;
    const int p_ea = interior ? sub + 1 : ((sub + 1 == n) ? nf : sub + 1);
    const int p_eam1 = interior ? ((sub == 0) ? n : sub) : ((sub == 0) ? nf - 1 : sub);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      pj[j] = (j == ln) ? 0 : ((j == v_ea) ? p_ea : ((j == v_eam1) ? p_eam1 : 0));
  }

  // ---- zero the tile, assemble A, the mean-value coupling and the right-hand side ----
  for (int e = sub; e < W::GROUP; e += P)
    Lg[e] = 0.0;
  wave_sync();
  if (active)
  {
    const double ia = rcp_d(fabs(detJ));
    const double g0 = (J[0][0] * J[0][0] + J[1][0] * J[1][0]) * ia,
                 g1 = (J[0][0] * J[0][1] + J[1][0] * J[1][1]) * ia,
                 g2 = (J[0][1] * J[0][1] + J[1][1] * J[1][1]) * ia;
    const double* te = sTE + ci * 3 * Z::NTES;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int g = 0; g < NH; ++g)
      {
        const int hh = (h > g) ? h : g, gg = (h > g) ? g : h;
        const int e = hh * (hh + 1) / 2 + gg;
        if (gi[h] >= gi[g])
          atomicAdd(&Lg[tri(gi[h], gi[g])], g0 * te[e] + g1 * te[Z::NTES + e] + g2 * te[2 * Z::NTES + e]);
      }
    double Lce[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NRT; ++i)
    {
      const double c0 = (i % 2 == 0) ? sv[0][i / 2].x : sv[0][i / 2].y;
      const double c1 = (i % 2 == 0) ? sv[1][i / 2].x : sv[1][i / 2].y;
      const double w0 = c0 * J[1][0] - c1 * J[0][0], w1 = c0 * J[1][1] - c1 * J[0][1];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Lce[j] -= sgn * (w0 * sV[(j * NRT + i) * 2] + w1 * sV[(j * NRT + i) * 2 + 1]);
    }
    const double Ce = fabs(detJ) / 6.0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
    {
      atomicAdd(&Rg[pj[j]], Lce[j]);
      atomicAdd(&Mv[pj[j]], Ce);
    }
    // B_k of this lane's cell into Yb[k]: Be[h][j], k = 0: int (Phi_h)_y psi_j, k = 1: -int (Phi_h)_x psi_j
    const double* vq = sVQ + ci * 2 * NH * 3;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
        const double v0 = vq[h * 3 + j], v1 = vq[(NH + h) * 3 + j];
        atomicAdd(&Yb[gi[h] * LDB + pj[j]], J[1][0] * v0 + J[1][1] * v1);
        atomicAdd(&Yb[DM * LDB + gi[h] * LDB + pj[j]], -(J[0][0] * v0 + J[0][1] * v1));
      }
  }
  wave_sync();

  // ---- Cholesky of A in registers (every lane of the group the same), parked in LDS ----
  int status_local = 0;
  {
    double Lr[W::TRI];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
        Lr[tri(i, j)] = (i < dim) ? Lg[tri(i, j)] : ((i == j) ? 1.0 : 0.0);
    wave_sync();
#pragma unroll
    for (int j = 0; j < DM; ++j)
    {
      const double ajj = Lr[tri(j, j)];
      if (!(ajj > 0.0))
        status_local = pvalid ? 1 : status_local;
      const double inv = rsqrt_d(ajj > 0.0 ? ajj : 1.0);
      Lr[tri(j, j)] = (ajj > 0.0 ? ajj : 1.0) * inv;
      if (sub == 0)
        Dg[j] = inv;
#pragma unroll
      for (int i = j + 1; i < DM; ++i)
        Lr[tri(i, j)] *= inv;
#pragma unroll
      for (int i = j + 1; i < DM; ++i)
#pragma unroll
        for (int kk = j + 1; kk <= i; ++kk)
          Lr[tri(i, kk)] -= Lr[tri(i, j)] * Lr[tri(kk, j)];
    }
    // each lane writes a share of the factor back
#pragma unroll
    for (int e = 0; e < W::TRI; ++e)
      if ((e % P) == sub)
        Lg[e] = Lr[e];
  }
  wave_sync();

  // B_k of this lane's cell: Be[h][j], k = 0: int (Phi_h)_y psi_j, k = 1: -int (Phi_h)_x psi_j
  // Columns of this lane: point 0 (the patch node; the same in every lane) and ring point sub + 1,
  // of Y_0 and Y_1 together (one sweep over the factor for the four vectors): y = L^-1 b
  const bool own_pt = sub + 1 < npnt; // the lane owns ring point sub + 1
  double ys0[2][DM], ys1[2][DM]; // kept for the back substitution
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int i = 0; i < DM; ++i)
    {
      ys0[k][i] = Yb[k * DM * LDB + i * LDB];
      ys1[k][i] = own_pt ? Yb[k * DM * LDB + i * LDB + sub + 1] : 0.0;
    }
#pragma unroll
  for (int i = 0; i < DM; ++i)
  {
    double t00 = ys0[0][i], t01 = ys1[0][i], t10 = ys0[1][i], t11 = ys1[1][i];
#pragma unroll
    for (int q = 0; q < i; ++q)
    {
      const double l = Lg[tri(i, q)];
      t00 -= l * ys0[0][q];
      t01 -= l * ys1[0][q];
      t10 -= l * ys0[1][q];
      t11 -= l * ys1[1][q];
    }
    const double d = Dg[i];
    asm volatile("" : "+v"(t00), "+v"(t01), "+v"(t10), "+v"(t11));
    ys0[0][i] = t00 * d; // rows >= dim of the factor are identity rows, the data there is zero
    ys1[0][i] = t01 * d;
    ys0[1][i] = t10 * d;
    ys1[1][i] = t11 * d;
  }

  // ---- Schur system in registers: row of ring point sub + 1 in its lane, row of point 0 in every
  // lane.  S = sum_k Y_k^T Y_k is symmetric positive (semi-)definite, so no pivoting is needed
  // (the reference factorises the bordered indefinite system with partial pivoting,
  // se/PatchData.hpp:598-663; same solution):
  //   boundary patches (essnt_primal):  S gamma = -R
  //   interior patches: S 1 = 0 (the patch functions are divergence free with zero normal trace),
  //   the mean-value row fixes the constant: lambda = sum R / sum M, gamma_0 := 0 in
  //   S gamma = -(R - lambda M), then gamma -= (M . gamma) / sum M.
  constexpr int NPT = P + 1; // points of a patch: the node + at most P ring points
  double rv[NPT + 1], rn[NPT + 1]; // [S row | rhs] of the ring point / of the node
#pragma unroll
  for (int j = 0; j <= NPT; ++j)
    rv[j] = rn[j] = 0.0;
  // publish the substituted columns, then S[r][c] = sum_k y_r . y_c for the lane's two rows
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int i = 0; i < DM; ++i)
    {
      if (sub == 0)
        Yb[k * DM * LDB + i * LDB] = ys0[k][i];
      if (own_pt)
        Yb[k * DM * LDB + i * LDB + sub + 1] = ys1[k][i];
    }
  wave_sync();
#pragma unroll
  for (int c = 0; c < NPT; ++c)
  {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < DM; ++i)
      {
        const double yc = Yb[k * DM * LDB + i * LDB + c]; // columns >= npnt are zero
        t0 += ys0[k][i] * yc;
        t1 += ys1[k][i] * yc;
      }
    // pin the two dot products here: otherwise the compiler sinks the FMAs of all columns below
    // the LDS reads of all columns (live VGPRs, spilled)
    asm volatile("" : "+v"(t0), "+v"(t1));
    rn[c] = t0;
    rv[c] = t1;
  }
  wave_sync();
  double gam[NPT], g_own = 0.0;
  int sing = 0;
  {
    double m_own = own_pt ? Mv[sub + 1] : 0.0;
    const double m0 = pvalid ? Mv[0] : 1.0;
    double q_own = own_pt ? -Rg[sub + 1] : 0.0, q0 = pvalid ? -Rg[0] : 0.0; // right-hand side -R
    const double sum_m = m0 + group_sum_d<P>(m_own, gb, sub);
    double lam = 0.0;
    if (meanvalue)
    {
      const double sum_q = q0 + group_sum_d<P>(q_own, gb, sub);
      lam = sum_q * rcp_d(sum_m); // = -lambda
      q_own -= lam * m_own;
#pragma unroll
      for (int c = 0; c < NPT; ++c)
        rn[c] = (c == 0) ? 1.0 : 0.0; // gamma_0 := 0
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
#ifdef EQLB_WSL_SKIP_LU // timing experiment (wrong results)
#pragma unroll
    for (int c = 0; c < NPT; ++c)
      gam[c] = rv[c] + rn[c];
    g_own = rv[NPT];
#else
    // elimination without pivoting: pivot 0 is the replicated node row, pivot p the row of lane p - 1
    {
      if (!(rn[0] > 0.0))
        sing = 1;
      const double f = rv[0] * rcp_d(rn[0]);
#pragma unroll
      for (int c = 1; c <= NPT; ++c)
        rv[c] -= f * rn[c];
    }
#pragma unroll
    for (int p = 1; p < NPT; ++p)
    {
      double pr[NPT + 1];
#pragma unroll
      for (int c = p; c <= NPT; ++c)
        pr[c] = __shfl(rv[c], gb + p - 1, 64);
      if (!(pr[p] > 0.0))
        sing = 1;
      const double f = (sub + 1 > p) ? rv[p] * rcp_d(pr[p]) : 0.0;
#pragma unroll
      for (int c = p + 1; c <= NPT; ++c)
        rv[c] -= f * pr[c];
    }
    // back substitution: gamma replicated in every lane
#pragma unroll
    for (int p = NPT - 1; p >= 1; --p)
    {
      double t = rv[NPT];
#pragma unroll
      for (int c = p + 1; c < NPT; ++c)
        t -= rv[c] * gam[c];
      t *= rcp_d(rv[p]); // the row of lane p - 1 (identity rows: pivot 1, right-hand side 0)
      if (sub + 1 == p)
        g_own = t;
      gam[p] = __shfl(t, gb + p - 1, 64);
    }
    {
      double t = rn[NPT];
#pragma unroll
      for (int c = 1; c < NPT; ++c)
        t -= rn[c] * gam[c];
      gam[0] = t * rcp_d(rn[0]);
    }
#endif
    if (meanvalue)
    {
      const double shift = (m0 * gam[0] + group_sum_d<P>(m_own * g_own, gb, sub)) * rcp_d(sum_m);
#pragma unroll
      for (int c = 0; c < NPT; ++c)
        gam[c] -= shift;
      g_own -= shift;
    }
    if (!own_pt)
      g_own = 0.0;
  }
  if (sing && pvalid)
    status_local = 1;

  // ---- u_k = -L^-T (Y_k gamma): node column (every lane the same) + group sum over the ring columns ----
#pragma unroll
  for (int k = 0; k < 2; ++k)
  {
    const double(&y0)[DM] = ys0[k];
    const double(&y1)[DM] = ys1[k];
    double w[DM];
#pragma unroll
    for (int i = 0; i < DM; ++i)
      w[i] = -(y0[i] * gam[0] + group_sum_d<P>(y1[i] * g_own, gb, sub));
#pragma unroll
    for (int i = DM - 1; i >= 0; --i)
    {
      double t = w[i];
#pragma unroll
      for (int q = i + 1; q < DM; ++q)
        t -= Lg[tri(q, i)] * w[q];
      w[i] = (i < dim) ? t * Dg[i] : 0.0;
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < DM; ++i)
      if (sub == 0 && pvalid)
        Wg[k * DM + i] = w[i];
    wave_sync();
  }


  // ---- back-map and add to the slot rows (se/solve_patch_weaksym.hpp:189-232) ----
  if (active)
  {
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      const double* w = Wg + k * DM;
      double ul[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h)
        ul[h] = w[gi[h]];
      double* o = srow[k];
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c)
          sacc -= (rev_m ? bcoef(j, c) : ((j == c) ? 1.0 : 0.0)) * ul[c];
        const double yp = (j == 0) ? ul[0] : ul[KB + j];
        o[fm * K + j] += pf_m * sacc;
        o[fp * K + j] += pf_p * yp;
      }
    }
  }
  if (status_local)
    atomicOr(a.status, 2);
}

template <int P>
static int launch_ws_lean(const SeArgs& a, hipStream_t stream)
{
  using W = WsLean<P>;
  const size_t lds_bytes = sizeof(double) * (size_t)W::lds_doubles();
  auto kern = k_se_weaksym_lean<P>;
  if (lds_bytes > 64 * 1024)
  {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
        != hipSuccess)
      return EQLB_ERR_DEVICE;
  }
  const int64_t grid = (a.npatch * P + W::BLOCK - 1) / W::BLOCK;
  if (grid == 0)
    return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W::BLOCK), lds_bytes, stream, a);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

template <int K, int P>
static int launch_ws_t(const SeArgs& a, hipStream_t stream)
{
  using W = WsSizes<K, P>;
  const size_t lds_bytes = sizeof(double) * (size_t)W::lds_doubles();
  if (lds_bytes > 160 * 1024)
    return EQLB_ERR_UNSUPPORTED;
  auto kern = k_se_weaksym<K, P>;
  if (lds_bytes > 64 * 1024)
  {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
        != hipSuccess)
      return EQLB_ERR_DEVICE;
  }
  const int64_t grid = (a.npatch * P + W::BLOCK - 1) / W::BLOCK;
  if (grid == 0)
    return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W::BLOCK), lds_bytes, stream, a);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

template <int K>
static int launch_ws_k(int P, const SeArgs& a, hipStream_t stream)
{
  switch (P)
  {
  case 4:
    return launch_ws_t<K, 4>(a, stream);
  case 8:
    return launch_ws_t<K, 8>(a, stream);
  case 16:
    return launch_ws_t<K, 16>(a, stream);
  case 32:
    return launch_ws_t<K, 32>(a, stream);
  case 64:
    return launch_ws_t<K, 64>(a, stream);
  }
  return EQLB_ERR_UNSUPPORTED;
}

int launch_se_weaksym(int k, int P, bool no_flux_bcs, const SeArgs& a, hipStream_t stream)
{
  if (k == 2 && no_flux_bcs && P == 4)
    return launch_ws_lean<4>(a, stream);
  if (k == 2 && no_flux_bcs && P == 8)
    return launch_ws_lean<8>(a, stream);
  if (k == 2)
    return launch_ws_k<2>(P, a, stream);
  if (k == 3)
    return launch_ws_k<3>(P, a, stream);
  if (k == 4 && P == 4) // RT_4: patches of up to 8 facets, like its row sweeps
    return launch_ws_t<4, 4>(a, stream);
  if (k == 4 && P == 8)
    return launch_ws_t<4, 8>(a, stream);
  return EQLB_ERR_UNSUPPORTED;
}

} // namespace eqlb
