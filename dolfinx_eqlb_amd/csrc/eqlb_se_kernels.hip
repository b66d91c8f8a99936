// Semi-explicit patch equilibration on gfx950: the hot path of
// cpp/dolfinx_eqlb/se/solve_patch_semiexplt.hpp:212-1163 (explicit step, patch assembly via
// se/assembly.hpp:119-274 + se/fluxmin_kernel.hpp:60-190, factorise/solve of
// se/PatchData.hpp:576-595, back-map and RT DOF scatter :1082-1161) as one HIP kernel.
//
// Mapping: one wavefront processes 64/P patches, P = 4..64 lanes per patch, ONE LANE PER PATCH
// CELL (lane i <-> cell T_{i+1} with minus facet E_i and plus facet E_{i+1}).  Nothing of the
// reference's quadrature/tabulation structure is kept: on affine cells every patch integral is
// a contraction with constant reference tensors (tools/gen_tables.py) staged in LDS, so a lane
// needs only J (4 doubles), G and f of its cell.  All facet quantities are OUTWARD flux moments
// mu_j = int_E (w . n_out) s^j in the cell's own facet parameter; RT coefficients are
// c_{f,j} = pf_f mu_j.  Neighbouring cells exchange k moments with wave shuffles, a segmented
// prefix sum fixes the zero-order moments (the recurrence of :581,844,876-901), and the reduced
// SPD system in [d | (k-1) moments per patch facet | interior DOFs per cell] is solved either
//   SOLVER 0: dense Cholesky of the patch tile in LDS, cooperatively by the P lanes, or
//   SOLVER 1: block-tridiagonal (+ border) system held in registers: parallel cyclic reduction over
//             the lanes of a patch with DPP row shifts for P <= 16 (scalar blocks for RT_2, 2 x 2
//             blocks for RT_3), sequential lane-to-lane elimination for the large-valence bins.
// Result scatter: SCATTER 0 writes each (cell, vertex) contribution once into a slot buffer that
// a streaming kernel reduces in fixed order (bitwise reproducible); SCATTER 1 uses fp64 global
// atomics; SCATTER 2 (k_se_patch_tiled, the default for k <= 2) keeps the rows of a tile of cells in
// LDS and adds them to flux_hdiv in the same launch.  DESIGN.md derives the formulation and logs
// the measurements behind every variant; tests/proto_gpu_math.py is its numpy statement.
#include "eqlb_device_common.h"
#include "eqlb_tables_gen.h"

// Parallel cyclic reduction, couplings across the ends of a chain: the coupling a_i of row i to row i - s is an exact
// zero for i < s at every level s (lane 0 is the border row, a_1 couples to it and is dropped; a level maps
// a_i -> -a_i / b_{i-s} * a_{i-s}, which keeps the zeros of i < 2 s), in EVERY patch group of a wave (idle groups:
// identity rows).  So what a row shift drags in from the first rows of the NEXT group as "coupling to row i + s" is
// already zero, and so is the product that forms the new a_i for i < 2 s: the selects that forced these zeros
// (1: as in rounds 1 - 2) cost 4 v_cndmask per level.
#ifndef EQLB_PCR_SELECTS
#define EQLB_PCR_SELECTS 0
#endif

namespace eqlb
{

size_t table_doubles(int k, int deg)
{
  const int nrt = nrt_of(k), nd = nd_of(deg), nq = nq_of(k);
  const int kb = k - 1, nadd = (k - 1) * (k - 2) / 2, ndiv = k * (k + 1) / 2 - 1;
  const int nh = 1 + 2 * kb + nadd, ncol = 2 * k + ndiv;
  const int hrow = nd * nq + ((nd * nq) & 1); // Sizes::HROW
  const int nte = nh * (nh + 1) / 2;         // rows of TE / WQ padded like Sizes::NTES / NCOLS
  return (size_t)3 * nrt * nrt + (size_t)9 * nd * k + (size_t)3 * hrow + (size_t)6 * nd * nq
         + (size_t)NCOMBO * 3 * (nte + (nte & 1)) + (size_t)NCOMBO * 3 * nh * (ncol + (ncol & 1))
         + (size_t)9 * k * k + (size_t)3 * nrt * 2 + (size_t)NCOMBO * 2 * nh * 3
         + (size_t)hrow + (size_t)NCOMBO * nh * nd * 2;
}

template <int K, int DEG>
static void fill_tables_t(std::vector<double>& out)
{
  using R = eqlb_tables::Ref<K, DEG>;
  out.clear();
  out.insert(out.end(), R::S, R::S + R::S_SIZE);
  out.insert(out.end(), R::F, R::F + R::F_SIZE);
  static_assert(R::H_SIZE % 3 == 0, "H holds one row per local vertex");
  for (int n = 0; n < 3; ++n) // rows padded to an even length (Sizes::HROW)
  {
    out.insert(out.end(), R::H + n * (R::H_SIZE / 3), R::H + (n + 1) * (R::H_SIZE / 3));
    if ((R::H_SIZE / 3) & 1)
      out.push_back(0.0);
  }
  out.insert(out.end(), R::D, R::D + R::D_SIZE);
  {
    using Z = Sizes<K, DEG, 8>;
    static_assert(R::TE_SIZE == NCOMBO * 3 * Z::NTE && R::WQ_SIZE == NCOMBO * 3 * Z::NH * Z::NCOL, "tensor sizes");
    for (int r = 0; r < NCOMBO * 3; ++r) // rows padded to Sizes::NTES
    {
      out.insert(out.end(), R::TE + r * Z::NTE, R::TE + (r + 1) * Z::NTE);
      out.insert(out.end(), Z::NTES - Z::NTE, 0.0);
    }
    for (int r = 0; r < NCOMBO * 3 * Z::NH; ++r) // rows padded to Sizes::NCOLS
    {
      out.insert(out.end(), R::WQ + r * Z::NCOL, R::WQ + (r + 1) * Z::NCOL);
      out.insert(out.end(), Z::NCOLS - Z::NCOL, 0.0);
    }
  }
  out.insert(out.end(), R::HB, R::HB + R::HB_SIZE);
  out.insert(out.end(), R::V, R::V + R::V_SIZE);
  out.insert(out.end(), R::VQ, R::VQ + R::VQ_SIZE);
  out.insert(out.end(), R::HG, R::HG + R::HG_SIZE);
  if (R::HG_SIZE & 1) // Sizes::NHG
    out.push_back(0.0);
  out.insert(out.end(), R::WG, R::WG + R::WG_SIZE);
}

int fill_tables_host(int k, int deg, std::vector<double>& out)
{
  if (k == 1 && deg == 0)
    fill_tables_t<1, 0>(out);
  else if (k == 2 && deg == 1)
    fill_tables_t<2, 1>(out);
  else if (k == 3 && deg == 2)
    fill_tables_t<3, 2>(out);
  else if (k == 4 && deg == 3)
    fill_tables_t<4, 3>(out);
  else if (k == 2 && deg == 0)
    fill_tables_t<2, 0>(out);
  else if (k == 3 && deg == 1)
    fill_tables_t<3, 1>(out);
  else if (k == 3 && deg == 0)
    fill_tables_t<3, 0>(out);
  else
    return EQLB_ERR_UNSUPPORTED;
  return (out.size() == table_doubles(k, deg)) ? 0 : EQLB_ERR_UNSUPPORTED;
}

#ifndef EQLB_EV_FMA
#define EQLB_EV_FMA 1
#endif
#ifndef EQLB_LE_FMA
#define EQLB_LE_FMA 1
#endif
#ifndef EQLB_P2_DIV_NODAL
#define EQLB_P2_DIV_NODAL 1 // P2 data: divergence of the projected flux through its nodal values instead of the tensor D
#endif
#ifndef EQLB_CHAIN_PCR
#define EQLB_CHAIN_PCR 1 // RT_2 chain solve by parallel cyclic reduction (0: sequential elimination)
#endif

// ---- the patch kernel ---------------------------------------------------------------------------
// MODE 0: semi-explicit equilibration.  MODE 1: the constrained-minimisation patch problem of
// ev/solve_patch.hpp:58-238 (mixed RT_k x DG_{k-1} saddle point, (ndof+1)^2 LU per patch in the
// reference) solved in the SAME reduced unknowns: the divergence constraint fixes the div-moment
// DOFs and the zero-order facet moments explicitly (conforming particular solution: no jump data),
// what remains is the SPD minimisation of || sigma - hat_a G || over the patch-wise H(div=0) space,
// i.e. the same matrix with the additional load (phi_h, hat_a G) (tensors HG, WG).
template <bool AL>
__device__ __forceinline__ const double* row16(const double* p)
{
  if constexpr (AL)
    return static_cast<const double*>(__builtin_assume_aligned(p, 16));
  else
    return p;
}

// N doubles (N even) from a 16-byte aligned LDS row: 128-bit reads
template <int N>
__device__ __forceinline__ void ldrow16(const double* p, double (&r)[N])
{
  static_assert(N % 2 == 0, "row length");
  const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int i = 0; i < N / 2; ++i)
  {
    const double2 v = q[i];
    r[2 * i] = v.x;
    r[2 * i + 1] = v.y;
  }
}

// Moments of a trace seen from the other side of a facet: y = (reversed ? B : I) x with the binomial matrix B of
// bcoef - written as y_j = x_j + rho * sum_c (B_jc - delta_jc) x_c with rho = 1 / 0: one multiply-add per output
// instead of a select per coefficient or per value (the partial sums of rows 1 and 2 coincide and are shared).
// _t: the same with B^T (loads instead of unknowns).
#ifndef EQLB_REV_FMA
#define EQLB_REV_FMA 1
#endif
template <int K>
__device__ __forceinline__ void reversal_apply(const double (&x)[K], const double rho, double (&y)[K])
{
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    double d = 0.0;
    bool any = false;
#pragma unroll
    for (int c = 0; c <= j; ++c)
    {
      const double coef = bcoef(j, c) - ((j == c) ? 1.0 : 0.0);
      if (coef != 0.0)
      {
        d = any ? __builtin_fma(coef, x[c], d) : coef * x[c];
        any = true;
      }
    }
    y[j] = any ? __builtin_fma(rho, d, x[j]) : x[j];
  }
}
template <int K>
__device__ __forceinline__ void reversal_apply_t(const double (&x)[K], const double rho, double (&y)[K])
{
#pragma unroll
  for (int h = 0; h < K; ++h)
  {
    double d = 0.0;
    bool any = false;
#pragma unroll
    for (int j = h; j < K; ++j)
    {
      const double coef = bcoef(j, h) - ((j == h) ? 1.0 : 0.0);
      if (coef != 0.0)
      {
        d = any ? __builtin_fma(coef, x[j], d) : coef * x[j];
        any = true;
      }
    }
    y[h] = any ? __builtin_fma(rho, d, x[h]) : x[h];
  }
}

template <int K, int DEG, int P, int SOLVER, int SCATTER, int BLOCK, int MODE = 0, bool FULL = false, bool INTR = false>
__device__ __forceinline__ void se_patch_body(const SeArgs& a, const int64_t block_id, double* lds,
                                              const bool tables_staged = false,
                                              const int64_t lane_index = -1, double* tile_slots = nullptr)
{
  using Z = Sizes<K, DEG, P>;
  constexpr int KB = Z::KB, NADD = Z::NADD, NDIV = Z::NDIV, NRT = Z::NRT, ND = Z::ND, NQ = Z::NQ;
  constexpr int NCOL = Z::NCOL, NH = Z::NH;

  // RT_2 with P1 data: every segment and every row used below has an even number of doubles, the
  // rows start on 16-byte boundaries (the dynamic LDS segment does): 128-bit LDS reads
  constexpr bool AL = K == 2 && DEG == 1;
  static_assert(!AL || (Z::NF % 2 == 0 && Z::NHT % 2 == 0 && Z::NDT % 2 == 0 && Z::NTET % 2 == 0
                        && Z::NWQT % 2 == 0 && (ND * K) % 2 == 0 && Z::HROW % 2 == 0
                        && Z::NTES % 2 == 0 && Z::NCOLS % 2 == 0 && Z::NHB % 2 == 0
                        && Z::NHG % 2 == 0),
                "table rows are not 16-byte aligned");
  double* sF = lds;             // [3][3][ND][K]
  double* sH = sF + Z::NF;      // [3][ND][NQ]
  double* sD = sH + Z::NHT;     // [3][ND][2][NQ]
  double* sTE = sD + Z::NDT;    // [NCOMBO][3][NTE]
  constexpr bool HALFWQ = SCATTER == 2 && K == 3; // Sizes::NWQH
  double* sWQ = sTE + Z::NTET;  // [NCOMBO][3][NH][NCOLS]; HALFWQ: [NCOMBO / 2][3][NH][NCOLS]
  double* sHB = sWQ + (HALFWQ ? Z::NWQH : Z::NWQT); // [3][3][K][K]
  double* sHG = sHB + Z::NHB;                     // MODE 1: [ND][NQ]
  // MODE 1: [NCOMBO][NH][ND][2]; RT_4: straight from the table buffer in global memory (Sizes::NEV_LDS)
  const double* sWG = (K >= 4) ? a.tables + Z::OFF_HG + Z::NHG : sHG + Z::NHG;
  (void)sWG;
  double* sA = sHB + Z::NHB + (MODE ? Z::NEV_LDS : 0); // SOLVER 0: per-group tiles
  (void)sA;

  int tid_ = threadIdx.x;
  // tiled EV launch: an opaque lane index keeps the compiler from hoisting the lane predicates
  // (sub == 1, sub < n, ... for every P) out of the wave-block loops of the kernel, where they occupy
  // registers across all bins - that kernel sits at its 128-VGPR budget and would spill (the SE kernel
  // fits and is 1 % faster with the hoisted predicates)
#ifndef EQLB_OPAQUE_K3
#define EQLB_OPAQUE_K3 1 // RT_3 tiled: 219 -> 207 VGPRs, 0.345 -> 0.338 ms at 1M triangles
#endif
  if constexpr (SCATTER == 2 && (MODE == 1 || (EQLB_OPAQUE_K3 && K >= 3)))
    asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  if (!tables_staged)
  {
    for (int i = tid; i < Z::NTAB; i += BLOCK)
      lds[i] = a.tables[Z::NS + i];
    if constexpr (MODE == 1)
      for (int i = tid; i < Z::NEV_LDS; i += BLOCK)
        sHG[i] = a.tables[Z::OFF_HG + i];
    __syncthreads();
  }

  const int lane = tid & 63;
  const int sub = lane % P;          // lane within the patch group == cell index i
  const int gbase = lane - sub;      // first lane of the group within the wave
  // lane index within the lane space of the bin (slots are lane-contiguous)
  const int64_t tl = (lane_index >= 0) ? lane_index : block_id * BLOCK + tid;
  const int64_t patch_local = tl / P;
#ifndef EQLB_FULL_SIMPLE
#define EQLB_FULL_SIMPLE 1 // 0: the lane predicates of the generic instance in the full-patch instance too (A/B: RT_3 0.293 against 0.300 ms, 8M triangles 2.24 against 2.29, EV RT_3 0.3285 against 0.3315)
#endif
#ifndef EQLB_FULL_SIMPLE_EV
#define EQLB_FULL_SIMPLE_EV 0 // RT_2 in EV mode keeps the generic predicates (measured: 0.0844 against 0.0880 ms with the simplified ones)
#endif
  constexpr bool FULLS = FULL && EQLB_FULL_SIMPLE && (MODE == 0 || K >= 3 || EQLB_FULL_SIMPLE_EV);
  // (FULL: whole wave-blocks of full patches, every lane belongs to one)
  const bool pvalid = FULLS ? true : (patch_local < a.npatch);
  const int64_t slot = a.slot_offset + tl;
  const int64_t patch = a.patch_offset + patch_local;

  // Two memory round trips per wave-block: (1) all descriptors of the lane in ONE batch of
  // independent loads (the unused lanes of a patch group hold cell = -1: the SoA is initialised
  // so), (2) J, G, f of the cell, issued together as soon as the cell index is there.
  const int r = a.rhs; // one right-hand side per launch
  int n = 0;
  int32_t cell_raw = -1;
  uint32_t info = 0u;
  uint8_t flag0 = (uint8_t)PFLAG_INTERIOR, flag = (uint8_t)0;
  if (pvalid)
  {
    n = (int)a.pn[patch];
    cell_raw = a.slot_cell[slot];
    info = a.slot_info[slot];
    flag0 = a.pflag[patch];
    flag = a.pflag[(int64_t)r * a.npatch_total + patch];
  }
  // FULL: the caller guarantees that every patch of this wave-block is interior with exactly P
  // cells (tiled launch, leading patches of a bin): the patch shape becomes a compile-time constant
  // and the masks for missing neighbours, boundary conditions and idle lanes fold away
  if constexpr (FULL)
  {
    n = P;
    flag0 = (uint8_t)PFLAG_INTERIOR;
    flag = (uint8_t)0;
  }
  // INTR: every patch of the wave-block is interior (any number of cells): no boundary facet, no flux BC -
  // the boundary branches fold away, the lane masks stay (unstructured meshes: valence 5 - 7 in groups of 8)
  if constexpr (INTR)
  {
    flag0 = (uint8_t)PFLAG_INTERIOR;
    flag = (uint8_t)0;
  }
  const bool active = FULL ? true : (cell_raw >= 0);
  const int32_t cell = active ? cell_raw : 0;
  const int fm = (info >> INFO_FM_SHIFT) & 3, fp = (info >> INFO_FP_SHIFT) & 3;
  const int ln = (info >> INFO_LN_SHIFT) & 3;
  const bool rev_m = (info & INFO_REV_M) != 0, rev_p = (info & INFO_REV_P) != 0;
  const int ci = active ? combo_index(fm, fp, rev_m) : 0; // row of the reduced tensors
  const double rho_m = rev_m ? 1.0 : 0.0, rho_p = rev_p ? 1.0 : 0.0; // reversal_apply
  (void)rho_m;
  (void)rho_p;

  // ---- geometry (cached affine map) and the DG data of the cell ----
  double J00 = 1.0, J01 = 0.0, J10 = 0.0, J11 = 1.0;
  double2 gdat[ND];
  double fdat[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i)
  {
    gdat[i] = make_double2(0.0, 0.0);
    fdat[i] = 0.0;
  }
  if (active)
  {
    const double2* Jp = reinterpret_cast<const double2*>(a.cellJ + 4 * (int64_t)cell);
    const double2* gp_ = reinterpret_cast<const double2*>(a.flux_dg + ((int64_t)a.rhs_in * a.ncells + cell) * (ND * 2));
    const double* fp_ = a.rhs_dg + ((int64_t)a.rhs_in * a.ncells + cell) * ND;
    const double2 j0 = Jp[0], j1 = Jp[1];
#pragma unroll
    for (int i = 0; i < ND; ++i)
      gdat[i] = gp_[i];
#pragma unroll
    for (int i = 0; i < ND; ++i)
      fdat[i] = fp_[i];
    J00 = j0.x;
    J01 = j0.y;
    J10 = j1.x;
    J11 = j1.y;
  }
  const double detJ = J00 * J11 - J01 * J10;
  const double sgn = (detJ > 0.0) ? 1.0 : -1.0;
  const double pf_m = (fm == 1) ? sgn : -sgn; // facet 1 measures the outward flux
  const double pf_p = (fp == 1) ? sgn : -sgn;

  // ---- neighbour lanes ----
  const bool interior_geo = (flag0 & PFLAG_INTERIOR) != 0;
  const int nf = interior_geo ? n : n + 1;
  const int nn = (n > 0) ? n : 1;
  // (FULL: the P cells of the patch form a ring)
  const int next = FULLS ? ((sub + 1) & (P - 1)) : ((sub + 1 < nn) ? sub + 1 : (interior_geo ? 0 : sub));
  const int prev = FULLS ? ((sub + P - 1) & (P - 1)) : ((sub > 0) ? sub - 1 : (interior_geo ? nn - 1 : 0));
  const bool has_next = active && (interior_geo || sub < n - 1);
  const bool has_prev = active && (interior_geo || sub > 0);
  const int fi_p = FULLS ? next : (interior_geo ? ((sub + 1 < nn) ? sub + 1 : 0) : sub + 1);
  const int dim = pvalid ? 1 + KB * nf + NADD * n : 0;
  (void)dim;

  int status_local = 0;

  // one right-hand side per launch (a.rhs): a loop over the RHS here makes the compiler hoist the
  // ~70 loop-invariant table loads of phase C above the loop and hold them in ~140 VGPRs
  {
    const bool bc0 = (flag & PFLAG_BC0) != 0, bcn = (flag & PFLAG_BCN) != 0;
    const bool d_fixed = bc0 || bcn;

    // ---- phase A: cell-local integrals: facet moments of hat*G, moments of hat*(f - div G) ----
    // full[] collects the particular solution in the load-tensor column order
    // [mu_m (K) | mu_p (K) | sgn * c_div (NDIV)]
    double gm[K], gpv[K], Rq[NQ];
    double LeG[MODE ? NH : 1]; // MODE 1: (phi_h, hat_a G)
    (void)LeG;
    {
      // adj[X][d] = detJ * K[X][d];  nu_f = adj^T N_f, N = {(-1,-1), (-1,0), (0,1)}
      const double a00 = J11, a01 = -J01, a10 = -J10, a11 = J00;
      const double nmx = (fm == 2) ? 0.0 : -1.0, nmy = (fm == 0) ? -1.0 : ((fm == 1) ? 0.0 : 1.0);
      const double npx = (fp == 2) ? 0.0 : -1.0, npy = (fp == 0) ? -1.0 : ((fp == 1) ? 0.0 : 1.0);
      const double num0 = a00 * nmx + a10 * nmy, num1 = a01 * nmx + a11 * nmy;
      const double nup0 = a00 * npx + a10 * npy, nup1 = a01 * npx + a11 * npy;
#pragma unroll
      for (int j = 0; j < K; ++j)
        gm[j] = gpv[j] = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        Rq[q] = 0.0;
      if constexpr (MODE == 1)
      {
#pragma unroll
        for (int h = 0; h < NH; ++h)
          LeG[h] = 0.0;
        if (active)
        {
          const double* tH = row16<AL>(sH + ln * Z::HROW);
          const double* wg = row16<AL>(sWG + ci * NH * ND * 2);

          // reference gradient of the hat function of the patch node
          const double dh0 = (ln == 0) ? -1.0 : ((ln == 1) ? 1.0 : 0.0);
          const double dh1 = (ln == 0) ? -1.0 : ((ln == 2) ? 1.0 : 0.0);
#pragma unroll
          for (int i = 0; i < ND; ++i)
          {
            const double2 g2 = gdat[i];
            const double fd = detJ * fdat[i];
            // detJ * grad hat . G_i = grad_ref hat . (adj G_i)
            const double gg = dh0 * (a00 * g2.x + a01 * g2.y) + dh1 * (a10 * g2.x + a11 * g2.y);
            const double jt0 = J00 * g2.x + J10 * g2.y, jt1 = J01 * g2.x + J11 * g2.y; // J^T G_i
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#if EQLB_EV_FMA
              Rq[q] = __builtin_fma(gg, sHG[i * NQ + q], __builtin_fma(fd, tH[i * NQ + q], Rq[q]));
#else
              Rq[q] += fd * tH[i * NQ + q] + gg * sHG[i * NQ + q];
#endif
#pragma unroll
            for (int h = 0; h < NH; ++h)
            {
              if constexpr (AL)
              {
                const double2 w2 = reinterpret_cast<const double2*>(wg)[h * ND + i];
#if EQLB_EV_FMA
                LeG[h] = __builtin_fma(w2.y, jt1, __builtin_fma(w2.x, jt0, LeG[h]));
#else
                LeG[h] += w2.x * jt0 + w2.y * jt1;
#endif
              }
              else
#if EQLB_EV_FMA
                LeG[h] = __builtin_fma(wg[(h * ND + i) * 2 + 1], jt1, __builtin_fma(wg[(h * ND + i) * 2], jt0, LeG[h]));
#else
                LeG[h] += wg[(h * ND + i) * 2] * jt0 + wg[(h * ND + i) * 2 + 1] * jt1;
#endif
            }
          }
        }
      }
      else if (active)
      {
        const double* tF_m = row16<AL>(sF + (fm * 3 + ln) * ND * K);
        const double* tF_p = row16<AL>(sF + (fp * 3 + ln) * ND * K);
        const double* tH = row16<AL>(sH + ln * Z::HROW);
        const double* tD = sD + ln * ND * 2 * NQ;
        double fdv[ND], dvg = 0.0;
        double ax[(DEG == 2) ? ND : 1], by[(DEG == 2) ? ND : 1];
        (void)fdv;
        (void)tD;
        (void)ax;
        (void)by;
#pragma unroll
        for (int i = 0; i < ND; ++i)
        {
          const double2 g2 = gdat[i];
          const double fv = fdat[i];
          const double gnm = g2.x * num0 + g2.y * num1;
          const double gnp = g2.x * nup0 + g2.y * nup1;
          const double gh0 = a00 * g2.x + a01 * g2.y; // (adj G_i)_X
          const double gh1 = a10 * g2.x + a11 * g2.y;
          const double fd = detJ * fv;
          if constexpr (AL)
          {
            double rm[K], rp[K];
            ldrow16<K>(tF_m + i * K, rm);
            ldrow16<K>(tF_p + i * K, rp);
#pragma unroll
            for (int j = 0; j < K; ++j)
            {
              gm[j] += rm[j] * gnm;
              gpv[j] += rp[j] * gnp;
            }
          }
          else
          {
#pragma unroll
            for (int j = 0; j < K; ++j)
            {
              gm[j] += tF_m[i * K + j] * gnm;
              gpv[j] += tF_p[i * K + j] * gnp;
            }
          }
          if constexpr (DEG == 1)
          {
            // P1 data: div_ref(adj G) is constant on the cell, D[ln][i][X][q] = d_X psi_i * sum_i' H[ln][i'][q]
            // (the psi_i sum to one): fold the divergence into the nodal values of detJ f
            fdv[i] = fd;
            if (i == 0)
              dvg = -(gh0 + gh1);
            else if (i == 1)
              dvg += gh0;
            else
              dvg += gh1;
          }
          else if constexpr (DEG == 2 && EQLB_P2_DIV_NODAL)
          {
            // P2 data: div_ref(adj G) is in P1, hence exactly representable by its values at the six P2 nodes -
            // three vertex values from the derivatives of the P2 basis there (constants of the reference element,
            // node order of elmtlib/lagrange.py: vertices, then the edges (v1,v2), (v0,v2), (v0,v1)), the edge
            // values are their means.  Replaces the contraction with the tensor D: 72 -> 24 multiply-adds per lane
            // and no reads of D.
            fdv[i] = fd;
            ax[i] = gh0;
            by[i] = gh1;
          }
          else
          {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              Rq[q] = __builtin_fma(-gh1, tD[(i * 2 + 1) * NQ + q],
                                    __builtin_fma(-gh0, tD[(i * 2 + 0) * NQ + q], __builtin_fma(fd, tH[i * NQ + q], Rq[q])));
          }
        }
        if constexpr (DEG == 2 && EQLB_P2_DIV_NODAL)
        {
          double dvn[ND];
          dvn[0] = 4.0 * (ax[5] + by[4]) - 3.0 * (ax[0] + by[0]) - ax[1] - by[2];
          dvn[1] = ax[0] + by[0] + 3.0 * ax[1] - by[2] + 4.0 * (by[3] - by[5] - ax[5]);
          dvn[2] = ax[0] + by[0] - ax[1] + 3.0 * by[2] + 4.0 * (ax[3] - ax[4] - by[4]);
          dvn[3] = 0.5 * (dvn[1] + dvn[2]);
          dvn[4] = 0.5 * (dvn[0] + dvn[2]);
          dvn[5] = 0.5 * (dvn[0] + dvn[1]);
#pragma unroll
          for (int i = 0; i < ND; ++i)
          {
            const double w = fdv[i] - dvn[i];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              Rq[q] = __builtin_fma(w, tH[i * NQ + q], Rq[q]);
          }
        }
        if constexpr (DEG == 1)
        {
          double rH[Z::HROW];
          if constexpr (AL)
            ldrow16<Z::HROW>(tH, rH);
          else
          {
#pragma unroll
            for (int e = 0; e < ND * NQ; ++e)
              rH[e] = tH[e];
          }
#pragma unroll
          for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              Rq[q] += (fdv[i] - dvg) * rH[i * NQ + q];
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          gm[j] *= pf_m;
          gpv[j] *= pf_p;
        }
      }
    }

    // ---- phase B: neighbour exchange -> particular solution in own-frame outward moments ----
    double mu_m[K], mu_p[K];
    {
      // jump moments on the plus facet (owner frame)
      double Jv[K];
      {
        double gmn[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
          gmn[j] = shfl_d(gm[j], gbase + next);
#if EQLB_REV_FMA
        double gt[K];
        reversal_apply<K>(gmn, rho_p, gt);
#pragma unroll
        for (int j = 0; j < K; ++j)
          Jv[j] = has_next ? gpv[j] + gt[j] : 0.0;
#else
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          double t = 0.0;
          if (rev_p)
          {
#pragma unroll
            for (int i = 0; i < K; ++i)
              t += bcoef(j, i) * gmn[i];
          }
          else
            t = gmn[j];
          Jv[j] = has_next ? gpv[j] + t : 0.0;
        }
#endif
      }
      // zero-order chain: inclusive prefix sum of R0 + J0(previous facet)
      const double Jprev0 = shfl_d(Jv[0], gbase + prev);
      double t = active ? (sgn * Rq[0] + (has_prev ? Jprev0 : 0.0)) : 0.0;
      {
        double o = lane_down_d<P, 1>(t, gbase, sub);
        if (sub >= 1)
          t += o;
        if constexpr (P > 2)
        {
          o = lane_down_d<P, 2>(t, gbase, sub);
          if (sub >= 2)
            t += o;
        }
        if constexpr (P > 4)
        {
          o = lane_down_d<P, 4>(t, gbase, sub);
          if (sub >= 4)
            t += o;
        }
        if constexpr (P > 8)
        {
          o = lane_down_d<P, 8>(t, gbase, sub);
          if (sub >= 8)
            t += o;
        }
#pragma unroll
        for (int off = 16; off < P; off <<= 1)
        {
          o = shfl_d(t, gbase + ((sub >= off) ? sub - off : sub));
          if (sub >= off)
            t += o;
        }
      }
      // prescribed outward moments of sigma_a on flux-BC end facets: pf * (hat_a g DOFs) - (hat_a G)
      // with the per-patch boundary DOFs of BoundaryData::calculate_patch_bc
      // (base/BoundaryData.cpp:687-745) = HB[f][ln] b_facet; all |b| < 1e-7 -> skipped (:714-725)
      double bnd_m[K], bnd_p[K];
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        bnd_m[j] = -gm[j];
        bnd_p[j] = -gpv[j];
      }
      if (a.bvals != nullptr && active && d_fixed)
      {
        const bool at0 = bc0 && sub == 0, atn = bcn && sub == n - 1;
        if (at0 || atn)
        {
          const double* bv = a.bvals + ((int64_t)r * a.ncells + cell) * NRT;
#pragma unroll
          for (int side = 0; side < 2; ++side)
          {
            if (!(side == 0 ? at0 : atn))
              continue;
            const int fb = side == 0 ? fm : fp;
            const double pfb = side == 0 ? pf_m : pf_p;
            double bg[K];
            bool allzero = true;
#pragma unroll
            for (int j = 0; j < K; ++j)
            {
              bg[j] = bv[fb * K + j];
              allzero = allzero && (fabs(bg[j]) < 1e-7);
            }
            if (!allzero)
            {
              const double* hb = sHB + (fb * 3 + ln) * K * K;
#pragma unroll
              for (int i = 0; i < K; ++i)
              {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j)
                  s += hb[i * K + j] * bg[j];
                if (side == 0)
                  bnd_m[i] += pfb * s;
                else
                  bnd_p[i] += pfb * s;
              }
            }
          }
        }
      }
      double delta = 0.0;
      if (d_fixed)
      {
        const double m0_first = shfl_d(bnd_m[0], gbase);
        const double p0_last = shfl_d(bnd_p[0], gbase + nn - 1);
        const double t_last = shfl_d(t, gbase + nn - 1);
        delta = bc0 ? -m0_first : (p0_last - t_last);
      }
      mu_p[0] = t + delta;
#pragma unroll
      for (int j = 1; j < K; ++j)
        mu_p[j] = (bcn && sub == n - 1) ? bnd_p[j] : 0.0;
      double vprev[K];
#pragma unroll
      for (int j = 0; j < K; ++j)
        vprev[j] = shfl_d(mu_p[j] + Jv[j], gbase + prev);
      if (has_prev)
      {
#if EQLB_REV_FMA
        double vt[K];
        reversal_apply<K>(vprev, rho_m, vt);
#pragma unroll
        for (int j = 0; j < K; ++j)
          mu_m[j] = -vt[j];
#else
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < K; ++c)
            s -= (rev_m ? bcoef(j, c) : ((j == c) ? 1.0 : 0.0)) * vprev[c];
          mu_m[j] = s;
        }
#endif
      }
      else
      {
        mu_m[0] = -delta;
#pragma unroll
        for (int j = 1; j < K; ++j)
          mu_m[j] = bc0 ? bnd_m[j] : 0.0;
      }
    }

    // ---- phase C: element matrix and load from the reduced reference tensors ----
    // Te = sum_x g_x TE[ci][x], Le = -sum_x g_x WQ[ci][x] [mu_m; mu_p; sgn c_div], g = J^T J/|detJ|
    double Te[NH][NH], Le[NH];
    {
      const double ia = active ? rcp_d(fabs(detJ)) : 0.0;
      const double g0 = (J00 * J00 + J10 * J10) * ia, g1 = (J00 * J01 + J10 * J11) * ia,
                   g2 = (J01 * J01 + J11 * J11) * ia;
      constexpr int NTES = Z::NTES, NCOLS = Z::NCOLS;
      const double* te = row16<AL>(sTE + ci * 3 * NTES);
      if constexpr (AL)
      {
        double tev[NTES]; // two entries per step: one 128-bit read from each metric row
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
          {
            const double v = tev[h * (h + 1) / 2 + g];
            Te[h][g] = v;
            Te[g][h] = v;
          }
      }
      else
      {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int g = 0; g <= h; ++g)
          {
            const int e = h * (h + 1) / 2 + g;
            const double v = g0 * te[e] + g1 * te[NTES + e] + g2 * te[2 * NTES + e];
            Te[h][g] = v;
            Te[g][h] = v;
          }
      }
      double full[NCOL];
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        full[j] = mu_m[j];
        full[K + j] = mu_p[j];
      }
#pragma unroll
      for (int q = 0; q < NDIV; ++q)
        full[2 * K + q] = sgn * Rq[1 + q];
      const double* wq = row16<AL>(sWQ + (HALFWQ ? (ci >> 1) * Z::NCMBH : ci * 3 * NH * NCOLS));
#pragma unroll
      for (int h = 0; h < NH; ++h)
      {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if constexpr (AL)
        {
#pragma unroll
          for (int c2 = 0; c2 < NCOLS / 2; ++c2)
          {
            const double2 w0 = reinterpret_cast<const double2*>(wq + h * NCOLS)[c2];
            const double2 w1 = reinterpret_cast<const double2*>(wq + (NH + h) * NCOLS)[c2];
            const double2 w2 = reinterpret_cast<const double2*>(wq + (2 * NH + h) * NCOLS)[c2];
            const double f0 = full[2 * c2], f1 = (2 * c2 + 1 < NCOL) ? full[2 * c2 + 1] : 0.0; // pad column
            // (two fused multiply-adds per sum: "s += a * b + c * d" compiles to mul + fma + add)
#if EQLB_LE_FMA
            s0 = __builtin_fma(w0.y, f1, __builtin_fma(w0.x, f0, s0));
            s1 = __builtin_fma(w1.y, f1, __builtin_fma(w1.x, f0, s1));
            s2 = __builtin_fma(w2.y, f1, __builtin_fma(w2.x, f0, s2));
#else
            s0 += w0.x * f0 + w0.y * f1;
            s1 += w1.x * f0 + w1.y * f1;
            s2 += w2.x * f0 + w2.y * f1;
#endif
          }
        }
        else
        {
#pragma unroll
          for (int c = 0; c < NCOL; ++c)
          {
            s0 += wq[h * NCOLS + c] * full[c];
            s1 += wq[(NH + h) * NCOLS + c] * full[c];
            s2 += wq[(2 * NH + h) * NCOLS + c] * full[c];
          }
        }
        Le[h] = -(g0 * s0 + g1 * s1 + g2 * s2);
      }
      if constexpr (HALFWQ)
      {
        // the staged rows are those of the unreversed minus facet: [d | um] rows through B^T where it is reversed
        double lt[K];
#if EQLB_REV_FMA
        double lk[K];
#pragma unroll
        for (int h = 0; h < K; ++h)
          lk[h] = Le[h];
        reversal_apply_t<K>(lk, rho_m, lt);
#pragma unroll
        for (int h = 0; h < K; ++h)
          Le[h] = lt[h];
#else
#pragma unroll
        for (int h = 0; h < K; ++h)
        {
          double t = 0.0;
#pragma unroll
          for (int j = h; j < K; ++j)
            t += bcoef(j, h) * Le[j];
          lt[h] = t;
        }
#pragma unroll
        for (int h = 0; h < K; ++h)
          Le[h] = rev_m ? lt[h] : Le[h];
#endif
      }
      if constexpr (MODE == 1)
      {
#pragma unroll
        for (int h = 0; h < NH; ++h)
          Le[h] += LeG[h];
      }
    }

    // ---- solve the reduced system ----
    double ul[NH];
    if constexpr (K == 1)
    {
      // only d: u = sum Le / sum Te  (se/PatchData.hpp:589)
      const double sa = group_sum_d<P>(Te[0][0], gbase, sub), sl = group_sum_d<P>(Le[0], gbase, sub);
      ul[0] = (d_fixed || !pvalid) ? 0.0 : sl / sa;
    }
#ifdef EQLB_EXP_SOLVER9
    else if constexpr (SOLVER == 9)
    {
      // timing-only build (results are wrong): no solve, to price the solver
#pragma unroll
      for (int h = 0; h < NH; ++h)
        ul[h] = Le[h] + Te[h][0];
    }
#endif
    else if constexpr (SOLVER == 0)
    {
      double* Ag = sA + (tid / P) * Z::LDS_GROUP;
      double* bg = Ag + Z::TRI;
      const int ntri = dim * (dim + 1) / 2;
      for (int e = sub; e < ntri; e += P)
        Ag[e] = 0.0;
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
      wave_sync();

      // Conflict-free assembly of the patch tile (se/assembly.hpp:182-272 scatter-add): lane i owns
      // the rows of facet E_i = own minus-side blocks + plus-side blocks of the previous cell
      // (fetched by shuffle), the E_i/E_{i+1} coupling of its cell and its interior unknowns; every
      // tile entry is stored exactly once, fixed (flux-BC) unknowns become identity rows (:209-251).
      const int prevl = (sub > 0) ? sub - 1 : nn - 1;
      const bool has_prevcell = pvalid && sub < nf && (sub > 0 || interior_geo);
      const bool fx_m = (bc0 && sub == 0) || (bcn && sub == n); // facet E_sub fixed
      const bool fx_p = bcn && sub == n - 1;                     // facet E_{sub+1} fixed
      {
        double sdd = Te[0][0], sld = Le[0];
#pragma unroll
        for (int off = 1; off < P; off <<= 1)
        {
          sdd += shfl_d(sdd, gbase + (sub ^ off));
          sld += shfl_d(sld, gbase + (sub ^ off));
        }
        if (sub == 0 && pvalid)
        {
          Ag[0] = d_fixed ? 1.0 : sdd;
          bg[0] = d_fixed ? 0.0 : sld;
        }
      }
#pragma unroll
      for (int aa = 0; aa < KB; ++aa)
      {
        const double dp_prev = shfl_d(Te[0][1 + KB + aa], gbase + prevl);
        const double lp_prev = shfl_d(Le[1 + KB + aa], gbase + prevl);
        const int row = 1 + sub * KB + aa;
        if (pvalid && sub < nf)
        {
          const double bt = Te[0][1 + aa] + (has_prevcell ? dp_prev : 0.0);
          const double rr = Le[1 + aa] + (has_prevcell ? lp_prev : 0.0);
          Ag[tri(row, 0)] = (fx_m || d_fixed) ? 0.0 : bt;
          bg[row] = fx_m ? 0.0 : rr;
        }
#pragma unroll
        for (int bb = 0; bb <= aa; ++bb)
        {
          const double pp_prev = shfl_d(Te[1 + KB + aa][1 + KB + bb], gbase + prevl);
          if (pvalid && sub < nf)
          {
            const double dg = Te[1 + aa][1 + bb] + (has_prevcell ? pp_prev : 0.0);
            Ag[tri(row, 1 + sub * KB + bb)] = fx_m ? ((aa == bb) ? 1.0 : 0.0) : dg;
          }
        }
        if (active)
        {
          // coupling of E_sub (um) with E_{fi_p} (up) inside the own cell
#pragma unroll
          for (int bb = 0; bb < KB; ++bb)
          {
            const int col = 1 + fi_p * KB + bb;
            const double v = (fx_m || fx_p) ? 0.0 : Te[1 + aa][1 + KB + bb];
            if (row > col)
              Ag[tri(row, col)] = v;
            else
              Ag[tri(col, row)] = v;
          }
        }
      }
      if (active)
      {
#pragma unroll
        for (int q = 0; q < NADD; ++q)
        {
          const int row = 1 + nf * KB + sub * NADD + q;
          Ag[tri(row, 0)] = d_fixed ? 0.0 : Te[0][1 + 2 * KB + q];
          bg[row] = Le[1 + 2 * KB + q];
#pragma unroll
          for (int aa = 0; aa < KB; ++aa)
          {
            Ag[tri(row, 1 + sub * KB + aa)] = fx_m ? 0.0 : Te[1 + aa][1 + 2 * KB + q];
            Ag[tri(row, 1 + fi_p * KB + aa)] = fx_p ? 0.0 : Te[1 + KB + aa][1 + 2 * KB + q];
          }
#pragma unroll
          for (int q2 = 0; q2 <= q; ++q2)
            Ag[tri(row, row - q + q2)] = Te[1 + 2 * KB + q][1 + 2 * KB + q2];
        }
      }
      wave_sync();
#ifdef EQLB_DEBUG
      if (patch == 0 && r == 0 && sub == 0)
        for (int i = 0; i < dim; ++i)
          for (int j = 0; j <= i; ++j)
            printf("[dbg] A[%d][%d] = %g   b %g\n", i, j, Ag[tri(i, j)], bg[i]);
      wave_sync();
#endif
      // Cholesky, column by column
      for (int j = 0; j < dim; ++j)
      {
        const double ajj = Ag[tri(j, j)];
        if (!(ajj > 0.0))
          status_local = 1;
        const double ljj = sqrt(ajj > 0.0 ? ajj : 1.0);
        const double inv = 1.0 / ljj;
        wave_sync();
        for (int i = j + sub; i < dim; i += P)
          Ag[tri(i, j)] = (i == j) ? ljj : Ag[tri(i, j)] * inv;
        wave_sync();
        for (int i = j + 1 + sub; i < dim; i += P)
        {
          const double lij = Ag[tri(i, j)];
          for (int kk = j + 1; kk <= i; ++kk)
            Ag[tri(i, kk)] -= lij * Ag[tri(kk, j)];
        }
        wave_sync();
      }
      // forward / backward substitution
      for (int j = 0; j < dim; ++j)
      {
        const double yj = bg[j] / Ag[tri(j, j)];
        wave_sync();
        for (int i = j + sub; i < dim; i += P)
        {
          if (i == j)
            bg[j] = yj;
          else
            bg[i] -= Ag[tri(i, j)] * yj;
        }
        wave_sync();
      }
      for (int j = dim - 1; j >= 0; --j)
      {
        const double xj = bg[j] / Ag[tri(j, j)];
        wave_sync();
        for (int i = sub; i <= j; i += P)
        {
          if (i == j)
            bg[j] = xj;
          else
            bg[i] -= Ag[tri(j, i)] * xj;
        }
        wave_sync();
      }
#pragma unroll
      for (int h = 0; h < NH; ++h)
        ul[h] = active ? bg[gi[h]] : 0.0;
      wave_sync();
    }
    else
    {
      // ---- SOLVER 1: everything in registers, lane-to-lane hand-off by shuffles ----
      // Unknown layout: border z = [d ; x_0] (x_i = the KB higher moments of facet E_i), chain
      // x_1 .. x_{nf-1} block tridiagonal.  Lane i holds the block row of facet E_i.
      static_assert(NADD <= 3, "interior-DOF condensation is written for at most three DOFs per cell (k <= 4)");
      constexpr int W = K;         // border width
      constexpr int NC = 1 + 2 * KB; // core local unknowns [d | um | up]
      const bool fx_m = (bc0 && sub == 0) || (bcn && sub == n); // facet E_sub fixed (flux BC)
      const bool fx_p = bcn && sub == n - 1;                     // facet E_{sub+1} fixed
      const int prevl = FULLS ? prev : ((sub > 0) ? sub - 1 : nn - 1);
      const bool has_prevcell = pvalid && sub < nf && (sub > 0 || interior_geo);

      // (a) element system with the rows/columns of fixed (flux-BC) unknowns cleared (identity rows
      // of se/assembly.hpp:209-251); patches with flux BCs are rare, so the masking sits behind a
      // branch that whole waves skip
      double(&T)[NH][NH] = Te;
      double(&Lv)[NH] = Le;
      if (d_fixed || fx_m || fx_p)
      {
#pragma unroll
        for (int h = 0; h < NH; ++h)
        {
          const bool fh = (h == 0) ? d_fixed : ((h <= KB) ? fx_m : ((h <= 2 * KB) ? fx_p : false));
          if (fh)
            Lv[h] = 0.0;
#pragma unroll
          for (int g = 0; g < NH; ++g)
          {
            const bool fg = (g == 0) ? d_fixed : ((g <= KB) ? fx_m : ((g <= 2 * KB) ? fx_p : false));
            if (fh || fg)
              T[h][g] = 0.0;
          }
        }
      }
      // (b) static condensation of the cell-interior unknowns (k = 3: one, k = 4: three per cell)
      constexpr int NA1 = (NADD > 0) ? NADD : 1;
      double ca[NC][NA1], la[NA1];
#pragma unroll
      for (int h = 0; h < NC; ++h)
#pragma unroll
        for (int q = 0; q < NA1; ++q)
          ca[h][q] = 0.0;
#pragma unroll
      for (int q = 0; q < NA1; ++q)
        la[q] = 0.0;
      if constexpr (NADD == 1)
      {
        const double piv = active ? T[NC][NC] : 1.0;
        const double ip = rcp_d(piv);
        la[0] = Lv[NC] * ip;
#pragma unroll
        for (int h = 0; h < NC; ++h)
          ca[h][0] = T[h][NC] * ip;
#pragma unroll
        for (int h = 0; h < NC; ++h)
        {
          Lv[h] -= T[h][NC] * la[0];
#pragma unroll
          for (int g = 0; g <= h; ++g) // symmetric update: one triangle, mirrored
          {
            const double v = T[h][g] - ca[h][0] * T[NC][g];
            T[h][g] = v;
            T[g][h] = v;
          }
        }
      }
      else if constexpr (NADD > 1)
      {
        // inverse of the SPD interior block by Cholesky (L L^T = T_ii, inactive lanes: identity), then
        // ca = T_ci T_ii^-1, la = T_ii^-1 L_i, Schur update of the core block and load
        double Li[NADD][NADD], iLd[NADD], Inv[NADD][NADD];
#pragma unroll
        for (int i = 0; i < NADD; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j)
            Li[i][j] = active ? T[NC + i][NC + j] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
        for (int j = 0; j < NADD; ++j)
        {
          double dj = Li[j][j];
#pragma unroll
          for (int q = 0; q < j; ++q)
            dj -= Li[j][q] * Li[j][q];
          if (!(dj > 0.0))
          {
            status_local = pvalid ? 1 : status_local;
            dj = 1.0;
          }
          const double ilj = rsqrt_d(dj);
          Li[j][j] = dj * ilj;
          iLd[j] = ilj;
#pragma unroll
          for (int i = j + 1; i < NADD; ++i)
          {
            double v = Li[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q)
              v -= Li[i][q] * Li[j][q];
            Li[i][j] = v * ilj;
          }
        }
        // columns of the inverse: L L^T x = e_c
#pragma unroll
        for (int c = 0; c < NADD; ++c)
        {
          double y[NADD];
#pragma unroll
          for (int i = 0; i < NADD; ++i)
          {
            double v = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < i; ++q)
              v -= Li[i][q] * y[q];
            y[i] = v * iLd[i];
          }
#pragma unroll
          for (int i = NADD - 1; i >= 0; --i)
          {
            double v = y[i];
#pragma unroll
            for (int q = i + 1; q < NADD; ++q)
              v -= Li[q][i] * Inv[q][c];
            Inv[i][c] = v * iLd[i];
          }
        }
#pragma unroll
        for (int q = 0; q < NADD; ++q)
        {
          double v = 0.0;
#pragma unroll
          for (int q2 = 0; q2 < NADD; ++q2)
            v += Inv[q][q2] * Lv[NC + q2];
          la[q] = v;
        }
#pragma unroll
        for (int h = 0; h < NC; ++h)
#pragma unroll
          for (int q = 0; q < NADD; ++q)
          {
            double v = 0.0;
#pragma unroll
            for (int q2 = 0; q2 < NADD; ++q2)
              v += T[h][NC + q2] * Inv[q2][q];
            ca[h][q] = v;
          }
#pragma unroll
        for (int h = 0; h < NC; ++h)
        {
#pragma unroll
          for (int q = 0; q < NADD; ++q)
            Lv[h] -= T[h][NC + q] * la[q];
#pragma unroll
          for (int g = 0; g <= h; ++g)
          {
            double v = T[h][g];
#pragma unroll
            for (int q = 0; q < NADD; ++q)
              v -= ca[h][q] * T[NC + q][g];
            T[h][g] = v;
            T[g][h] = v;
          }
        }
      }
      // (c) block row of facet E_sub: own minus-side blocks + plus-side blocks of the previous cell
      double Dg[KB][KB], bt[KB], rr[KB], Off[KB][KB];
      double alpha = group_sum_d<P>(T[0][0], gbase, sub), rd = group_sum_d<P>(Lv[0], gbase, sub);
      if (d_fixed || !pvalid)
      {
        alpha = 1.0;
        rd = 0.0;
      }
#pragma unroll
      for (int aa = 0; aa < KB; ++aa)
      {
        const double dp_prev = shfl_d(T[0][1 + KB + aa], gbase + prevl);
        const double lp_prev = shfl_d(Lv[1 + KB + aa], gbase + prevl);
        bt[aa] = T[0][1 + aa] + (has_prevcell ? dp_prev : 0.0);
        rr[aa] = Lv[1 + aa] + (has_prevcell ? lp_prev : 0.0);
#pragma unroll
        for (int bb = 0; bb < KB; ++bb)
        {
          const double pp_prev = shfl_d(T[1 + KB + aa][1 + KB + bb], gbase + prevl);
          Dg[aa][bb] = T[1 + aa][1 + bb] + (has_prevcell ? pp_prev : 0.0);
          Off[aa][bb] = T[1 + aa][1 + KB + bb];
        }
      }
      const bool row_valid = pvalid && sub < nf && !fx_m;
      if (!row_valid)
      {
#pragma unroll
        for (int aa = 0; aa < KB; ++aa)
        {
          bt[aa] = rr[aa] = 0.0;
#pragma unroll
          for (int bb = 0; bb < KB; ++bb)
          {
            Dg[aa][bb] = (aa == bb) ? 1.0 : 0.0;
            Off[aa][bb] = 0.0;
          }
        }
      }
      // (d) border data (lane 0) and the chain rows
      double Z[W][W], rz[W], Off0[KB][KB];
      Z[0][0] = alpha;
      rz[0] = rd;
#pragma unroll
      for (int aa = 0; aa < KB; ++aa)
      {
        const double b0 = shfl_d(bt[aa], gbase);
        Z[0][1 + aa] = Z[1 + aa][0] = b0;
        rz[1 + aa] = shfl_d(rr[aa], gbase);
#pragma unroll
        for (int bb = 0; bb < KB; ++bb)
        {
          Z[1 + aa][1 + bb] = shfl_d(Dg[aa][bb], gbase);
          Off0[aa][bb] = shfl_d(Off[aa][bb], gbase);
        }
      }
      const bool in_chain = pvalid && sub >= 1 && sub < nf;
      const bool wraps = interior_geo && sub == n - 1; // coupling of E_{n-1} to E_n == E_0
      double Dp[KB][KB], Rp[KB][1 + W], OffC[KB][KB];
#ifndef EQLB_CHAIN_MASKS
#define EQLB_CHAIN_MASKS 1 // full-patch instances of RT_3 / RT_4: 0/1 factors instead of selects (one multiply per double instead of two v_cndmask)
#endif
      if constexpr (FULLS && EQLB_CHAIN_MASKS && KB >= 2)
      {
        // (the values are this lane's own and finite: a factor 0 is an exact zero)
        const double mc = (sub >= 1) ? 1.0 : 0.0, mi = 1.0 - mc;          // chain row | border row (lane 0)
        const double mo = (sub >= 1 && sub != P - 1) ? 1.0 : 0.0;         // coupling to the next chain row
        const double m1 = (sub == 1) ? 1.0 : 0.0, mw = mc - mo;           // first | last (wrapping) chain row
#pragma unroll
        for (int aa = 0; aa < KB; ++aa)
        {
          Rp[aa][0] = mc * rr[aa];
          Rp[aa][1] = mc * bt[aa];
#pragma unroll
          for (int bb = 0; bb < KB; ++bb)
          {
            Dp[aa][bb] = (aa == bb) ? __builtin_fma(mc, Dg[aa][bb], mi) : mc * Dg[aa][bb];
            OffC[aa][bb] = mo * Off[aa][bb];
            Rp[aa][2 + bb] = __builtin_fma(mw, Off[aa][bb], m1 * Off0[bb][aa]);
          }
        }
      }
      else
      {
#pragma unroll
      for (int aa = 0; aa < KB; ++aa)
      {
        Rp[aa][0] = in_chain ? rr[aa] : 0.0;
        Rp[aa][1] = in_chain ? bt[aa] : 0.0;
#pragma unroll
        for (int bb = 0; bb < KB; ++bb)
        {
          Dp[aa][bb] = in_chain ? Dg[aa][bb] : ((aa == bb) ? 1.0 : 0.0);
          OffC[aa][bb] = (in_chain && !wraps) ? Off[aa][bb] : 0.0;
          double c0 = 0.0;
          if (in_chain && sub == 1)
            c0 += Off0[bb][aa];
          if (in_chain && wraps)
            c0 += Off[aa][bb];
          Rp[aa][2 + bb] = c0;
        }
      }
      }
      double zz[W], xs[KB];
      if constexpr (KB == 1 && P <= 16 && EQLB_CHAIN_PCR)
      {
        // (e') RT_2: the chain is a scalar symmetric tridiagonal system with three right-hand sides
        // [rr | column of d | column of x_0].  Parallel cyclic reduction: log2(P) levels in which
        // EVERY lane eliminates its couplings to the rows i - s and i + s (DPP row shifts), instead
        // of P - 2 sequential hand-offs down the chain and P - 2 back up.  a = coupling to row
        // i - s (the coupling to i + s is a of lane i + s: the level matrices stay symmetric);
        // rows outside the chain are identity rows, couplings across the ends are exact zeros, so
        // what a shift drags in from a neighbouring patch group is multiplied by zero.
        double b = Dp[0][0], r0 = Rp[0][0], r1 = Rp[0][1], r2 = Rp[0][2];
        const double B1 = Rp[0][1], B2 = Rp[0][2];
        double am = dpp_d<0x111>(OffC[0][0]);
        if (!(FULLS && EQLB_CHAIN_MASKS) && sub == 0) // (full patches: what lane 0 receives is the zero of a wrapping row)
          am = 0.0;
        bool posdef = true; // pivots of the levels, of the last level and of the border system
#define EQLB_PCR_LEVEL(S)                                                                          \
  if constexpr (P > S)                                                                             \
  {                                                                                                \
    posdef = posdef && (b > 0.0);                                                                  \
    const double ib = rcp_d(b);                                                                    \
    const double ib_lo = dpp_d<0x110 + S>(ib), a_lo = dpp_d<0x110 + S>(am);                        \
    const double r0_lo = dpp_d<0x110 + S>(r0), r1_lo = dpp_d<0x110 + S>(r1), r2_lo = dpp_d<0x110 + S>(r2); \
    const double ib_hi = dpp_d<0x100 + S>(ib), a_hi = dpp_d<0x100 + S>(am);                        \
    const double r0_hi = dpp_d<0x100 + S>(r0), r1_hi = dpp_d<0x100 + S>(r1), r2_hi = dpp_d<0x100 + S>(r2); \
    const double cp = EQLB_PCR_SELECTS ? ((sub + S < P) ? a_hi : 0.0) : a_hi; /* coupling to row i + S */ \
    const double al = am * ib_lo, ga = cp * ib_hi;                                                 \
    b = __builtin_fma(-ga, cp, __builtin_fma(-al, am, b));                                         \
    r0 = __builtin_fma(-ga, r0_hi, __builtin_fma(-al, r0_lo, r0));                                 \
    r1 = __builtin_fma(-ga, r1_hi, __builtin_fma(-al, r1_lo, r1));                                 \
    r2 = __builtin_fma(-ga, r2_hi, __builtin_fma(-al, r2_lo, r2));                                 \
    am = EQLB_PCR_SELECTS ? ((sub >= 2 * S) ? -al * a_lo : 0.0) : -al * a_lo;                      \
  }
        EQLB_PCR_LEVEL(1)
        EQLB_PCR_LEVEL(2)
        EQLB_PCR_LEVEL(4)
        EQLB_PCR_LEVEL(8)
#undef EQLB_PCR_LEVEL
        posdef = posdef && (b > 0.0);
        const double ibf = rcp_d(b);
        const double s0 = r0 * ibf, s1 = r1 * ibf, s2 = r2 * ibf; // A^-1 [rr | bt | c0]
        // (f') Schur complement of the chain on the border [d ; x_0]
        double tred[W], Sred[W][W];
        tred[0] = group_sum_d<P>(B1 * s0, gbase, sub);
        tred[1] = group_sum_d<P>(B2 * s0, gbase, sub);
        Sred[0][0] = group_sum_d<P>(B1 * s1, gbase, sub);
        Sred[1][0] = group_sum_d<P>(B2 * s1, gbase, sub);
        Sred[1][1] = group_sum_d<P>(B2 * s2, gbase, sub);
        {
          const double l00 = Z[0][0] - Sred[0][0], l10 = Z[1][0] - Sred[1][0];
          const double i0 = rcp_d(l00);
          const double m = l10 * i0;
          const double d1 = __builtin_fma(-m, l10, Z[1][1] - Sred[1][1]);
          posdef = posdef && (l00 > 0.0) && (d1 > 0.0);
          const double q0 = rz[0] - tred[0];
          const double q1 = __builtin_fma(-m, q0, rz[1] - tred[1]);
          zz[1] = q1 * rcp_d(d1);
          zz[0] = __builtin_fma(-l10, zz[1], q0) * i0;
        }
        if (!posdef && pvalid)
          status_local = 1;
        const double v = __builtin_fma(-s2, zz[1], __builtin_fma(-s1, zz[0], s0));
        xs[0] = in_chain ? v : 0.0;
      }
      else
      {
        // small SPD inverse (KB = 1, 2), X = Einv Rp, Y = Einv OffC
        double Ei[KB][KB], X[KB][1 + W], Y[KB][KB];
        auto finish_row = [&]() {
          if constexpr (KB == 1)
            Ei[0][0] = rcp_d(Dp[0][0]);
          else if constexpr (KB == 3)
          {
            // symmetric 3 x 3: adjugate / determinant
            const double c00 = Dp[1][1] * Dp[2][2] - Dp[1][2] * Dp[2][1];
            const double c01 = Dp[0][2] * Dp[2][1] - Dp[0][1] * Dp[2][2];
            const double c02 = Dp[0][1] * Dp[1][2] - Dp[0][2] * Dp[1][1];
            const double c11 = Dp[0][0] * Dp[2][2] - Dp[0][2] * Dp[2][0];
            const double c12 = Dp[0][2] * Dp[1][0] - Dp[0][0] * Dp[1][2];
            const double c22 = Dp[0][0] * Dp[1][1] - Dp[0][1] * Dp[1][0];
            const double det = Dp[0][0] * c00 + Dp[0][1] * c01 + Dp[0][2] * c02;
            const double id = rcp_d(det);
            Ei[0][0] = c00 * id;
            Ei[0][1] = Ei[1][0] = c01 * id;
            Ei[0][2] = Ei[2][0] = c02 * id;
            Ei[1][1] = c11 * id;
            Ei[1][2] = Ei[2][1] = c12 * id;
            Ei[2][2] = c22 * id;
          }
          else
          {
            const double det = Dp[0][0] * Dp[1][1] - Dp[0][1] * Dp[1][0];
            const double id = rcp_d(det);
            Ei[0][0] = Dp[1][1] * id;
            Ei[1][1] = Dp[0][0] * id;
            Ei[0][1] = -Dp[0][1] * id;
            Ei[1][0] = -Dp[1][0] * id;
          }
  #pragma unroll
          for (int aa = 0; aa < KB; ++aa)
          {
  #pragma unroll
            for (int c = 0; c < 1 + W; ++c)
            {
              double v = 0.0;
  #pragma unroll
              for (int e = 0; e < KB; ++e)
                v += Ei[aa][e] * Rp[e][c];
              X[aa][c] = v;
            }
  #pragma unroll
            for (int bb = 0; bb < KB; ++bb)
            {
              double v = 0.0;
  #pragma unroll
              for (int e = 0; e < KB; ++e)
                v += Ei[aa][e] * OffC[e][bb];
              Y[aa][bb] = v;
            }
          }
        };
        constexpr bool BLOCK_PCR = KB == 2 && P <= 16 && EQLB_CHAIN_PCR;
        if constexpr (BLOCK_PCR)
        {
          // (e') RT_3: block parallel cyclic reduction (2 x 2 blocks, right-hand sides
          // [rr | column of d | two columns of x_0]), log2(P) levels instead of P - 2 hand-offs down and
          // P - 2 up.  Row i:  A_i x_{i-s} + D_i x_i + A_{i+s}^T x_{i+s} = R_i, A_i = coupling to row
          // i - s (level matrices stay symmetric).  Rows outside the chain are identity rows, the
          // couplings across the ends are exact zeros (see the scalar variant above).
          double D00 = Dp[0][0], D01 = Dp[0][1], D11 = Dp[1][1];
          double A[KB][KB], R[KB][1 + W], Bo[KB][W];
  #pragma unroll
          for (int aa = 0; aa < KB; ++aa)
          {
  #pragma unroll
            for (int e = 0; e < KB; ++e)
            {
              const double v = dpp_d<0x111>(OffC[e][aa]); // A_i = OffC_{i-1}^T
              // (full-patch instance: the lane before lane 0 is the wrapping row of the previous patch of the wave-block,
              // whose OffC is zero, or lies outside the row)
              A[aa][e] = (FULLS && EQLB_CHAIN_MASKS) ? v : ((sub == 0) ? 0.0 : v);
            }
  #pragma unroll
            for (int c = 0; c < 1 + W; ++c)
              R[aa][c] = Rp[aa][c];
  #pragma unroll
            for (int c = 0; c < W; ++c)
              Bo[aa][c] = Rp[aa][1 + c];
          }
          bool posdef = true;
          double I00, I01, I11; // inverse of the diagonal block
          auto invert = [&]() {
            const double det = __builtin_fma(D00, D11, -D01 * D01);
            posdef = posdef && (D00 > 0.0) && (det > 0.0);
            const double id = rcp_d(det);
            I00 = D11 * id;
            I11 = D00 * id;
            I01 = -D01 * id;
          };
  #define FNMA4(a0, b0, a1, b1, a2, b2, a3, b3, x)                                                    \
    __builtin_fma(-(a3), (b3), __builtin_fma(-(a2), (b2), __builtin_fma(-(a1), (b1), __builtin_fma(-(a0), (b0), (x)))))
  #define EQLB_BPCR_LEVEL(S)                                                                          \
    if constexpr (P > S)                                                                              \
    {                                                                                                 \
      invert();                                                                                       \
      const double l00 = dpp_d<0x110 + S>(I00), l01 = dpp_d<0x110 + S>(I01), l11 = dpp_d<0x110 + S>(I11); \
      const double h00 = dpp_d<0x100 + S>(I00), h01 = dpp_d<0x100 + S>(I01), h11 = dpp_d<0x100 + S>(I11); \
      double Al[KB][KB], Ah[KB][KB], Rl[KB][1 + W], Rh[KB][1 + W];                                    \
      _Pragma("unroll") for (int aa = 0; aa < KB; ++aa)                                               \
      {                                                                                               \
        _Pragma("unroll") for (int e = 0; e < KB; ++e)                                                \
        {                                                                                             \
          Al[aa][e] = dpp_d<0x110 + S>(A[aa][e]);                                                     \
          const double t = dpp_d<0x100 + S>(A[aa][e]);                                                \
          Ah[aa][e] = (EQLB_PCR_SELECTS && P < 16 && !(sub + S < P)) ? 0.0 : t; /* row i + S of ANOTHER group */ \
        }                                                                                             \
        _Pragma("unroll") for (int c = 0; c < 1 + W; ++c)                                             \
        {                                                                                             \
          Rl[aa][c] = dpp_d<0x110 + S>(R[aa][c]);                                                     \
          Rh[aa][c] = dpp_d<0x100 + S>(R[aa][c]);                                                     \
        }                                                                                             \
      }                                                                                               \
      /* al = A Dinv(i-S), ga = A(i+S)^T Dinv(i+S) */                                                 \
      double al[KB][KB], ga[KB][KB];                                                                  \
      _Pragma("unroll") for (int aa = 0; aa < KB; ++aa)                                               \
      {                                                                                               \
        al[aa][0] = A[aa][0] * l00 + A[aa][1] * l01;                                                  \
        al[aa][1] = A[aa][0] * l01 + A[aa][1] * l11;                                                  \
        ga[aa][0] = Ah[0][aa] * h00 + Ah[1][aa] * h01;                                                \
        ga[aa][1] = Ah[0][aa] * h01 + Ah[1][aa] * h11;                                                \
      }                                                                                               \
      /* (chains of fused multiply-adds: "x -= a * b + c * d + ..." costs one instruction more) */  \
      D00 = FNMA4(al[0][0], A[0][0], al[0][1], A[0][1], ga[0][0], Ah[0][0], ga[0][1], Ah[1][0], D00); \
      D01 = FNMA4(al[0][0], A[1][0], al[0][1], A[1][1], ga[0][0], Ah[0][1], ga[0][1], Ah[1][1], D01); \
      D11 = FNMA4(al[1][0], A[1][0], al[1][1], A[1][1], ga[1][0], Ah[0][1], ga[1][1], Ah[1][1], D11); \
      _Pragma("unroll") for (int aa = 0; aa < KB; ++aa)                                               \
      {                                                                                               \
        _Pragma("unroll") for (int c = 0; c < 1 + W; ++c)                                             \
          R[aa][c] = FNMA4(al[aa][0], Rl[0][c], al[aa][1], Rl[1][c], ga[aa][0], Rh[0][c], ga[aa][1], Rh[1][c], R[aa][c]); \
        const double n0 = -(al[aa][0] * Al[0][0] + al[aa][1] * Al[1][0]);                             \
        const double n1 = -(al[aa][0] * Al[0][1] + al[aa][1] * Al[1][1]);                             \
        A[aa][0] = n0;                                                                                \
        A[aa][1] = n1;                                                                                \
      }                                                                                               \
    }
          EQLB_BPCR_LEVEL(1)
          EQLB_BPCR_LEVEL(2)
          EQLB_BPCR_LEVEL(4)
          EQLB_BPCR_LEVEL(8)
  #undef EQLB_BPCR_LEVEL
  #undef FNMA4
          invert();
          if (!posdef && pvalid)
            status_local = 1;
  #pragma unroll
          for (int c = 0; c < 1 + W; ++c)
          {
            X[0][c] = I00 * R[0][c] + I01 * R[1][c];
            X[1][c] = I01 * R[0][c] + I11 * R[1][c];
          }
          // the border reduction below pairs the ORIGINAL coupling columns with the solution
  #pragma unroll
          for (int aa = 0; aa < KB; ++aa)
  #pragma unroll
            for (int c = 0; c < W; ++c)
              Rp[aa][1 + c] = Bo[aa][c];
        }
        else
        {
          // (e) forward elimination down the chain: lane s receives OffC^T Einv [OffC | Rp] of lane s-1
    #ifdef EQLB_EXP_CHAIN3 // timing experiment (wrong results): log2(P) elimination steps, no back substitution
          constexpr int SEND = (P == 8) ? 5 : ((P == 16) ? 6 : P);
    #else
          constexpr int SEND = P;
    #endif
    #pragma unroll 1
          for (int s = 2; s < SEND; ++s)
          {
            finish_row();
            double P1[KB][KB], P2[KB][1 + W];
    #pragma unroll
            for (int aa = 0; aa < KB; ++aa)
            {
    #pragma unroll
              for (int bb = 0; bb < KB; ++bb)
              {
                double v = 0.0;
    #pragma unroll
                for (int e = 0; e < KB; ++e)
                  v += OffC[e][aa] * Y[e][bb];
                P1[aa][bb] = v;
              }
    #pragma unroll
              for (int c = 0; c < 1 + W; ++c)
              {
                double v = 0.0;
    #pragma unroll
                for (int e = 0; e < KB; ++e)
                  v += OffC[e][aa] * X[e][c];
                P2[aa][c] = v;
              }
            }
            const bool take = in_chain && sub == s;
    #pragma unroll
            for (int aa = 0; aa < KB; ++aa)
            {
    #pragma unroll
              for (int bb = 0; bb < KB; ++bb)
              {
                const double v = lane_down_d<P, 1>(P1[aa][bb], gbase, sub);
                if (take)
                  Dp[aa][bb] -= v;
              }
    #pragma unroll
              for (int c = 0; c < 1 + W; ++c)
              {
                const double v = lane_down_d<P, 1>(P2[aa][c], gbase, sub);
                if (take)
                  Rp[aa][c] -= v;
              }
            }
          }
          finish_row();
        }
        // (f) Schur complement of the chain on the border, W x W solve
        double Sred[W][W], tred[W];
  #pragma unroll
        for (int c = 0; c < W; ++c)
        {
          double v = 0.0;
  #pragma unroll
          for (int e = 0; e < KB; ++e)
            v += Rp[e][1 + c] * X[e][0];
          tred[c] = v;
  #pragma unroll
          for (int c2 = 0; c2 <= c; ++c2)
          {
            double u2 = 0.0;
  #pragma unroll
            for (int e = 0; e < KB; ++e)
              u2 += Rp[e][1 + c] * X[e][1 + c2];
            Sred[c][c2] = u2;
          }
        }
  #pragma unroll
        for (int c = 0; c < W; ++c)
        {
          tred[c] = group_sum_d<P>(tred[c], gbase, sub);
  #pragma unroll
          for (int c2 = 0; c2 <= c; ++c2)
            Sred[c][c2] = group_sum_d<P>(Sred[c][c2], gbase, sub);
        }
          {
          // Cholesky of Z - S (lower), then two triangular solves
          double Lz[W][W], iLz[W];
  #pragma unroll
          for (int c = 0; c < W; ++c)
          {
            rz[c] -= tred[c];
  #pragma unroll
            for (int c2 = 0; c2 <= c; ++c2)
              Lz[c][c2] = Z[c][c2] - Sred[c][c2];
          }
  #pragma unroll
          for (int j = 0; j < W; ++j)
          {
            double dj = Lz[j][j];
  #pragma unroll
            for (int q = 0; q < j; ++q)
              dj -= Lz[j][q] * Lz[j][q];
            if (!(dj > 0.0))
            {
              status_local = pvalid ? 1 : status_local;
              dj = 1.0;
            }
            const double ilj = rsqrt_d(dj), lj = dj * ilj;
            Lz[j][j] = lj;
            iLz[j] = ilj;
  #pragma unroll
            for (int i = j + 1; i < W; ++i)
            {
              double v = Lz[i][j];
  #pragma unroll
              for (int q = 0; q < j; ++q)
                v -= Lz[i][q] * Lz[j][q];
              Lz[i][j] = v * ilj;
            }
          }
  #pragma unroll
          for (int i = 0; i < W; ++i)
          {
            double v = rz[i];
  #pragma unroll
            for (int q = 0; q < i; ++q)
              v -= Lz[i][q] * zz[q];
            zz[i] = v * iLz[i];
          }
  #pragma unroll
          for (int i = W - 1; i >= 0; --i)
          {
            double v = zz[i];
  #pragma unroll
            for (int q = i + 1; q < W; ++q)
              v -= Lz[q][i] * zz[q];
            zz[i] = v * iLz[i];
          }
        }
        // (g) back substitution up the chain
  #pragma unroll
        for (int aa = 0; aa < KB; ++aa)
        {
          double v = X[aa][0];
  #pragma unroll
          for (int c = 0; c < W; ++c)
            v -= X[aa][1 + c] * zz[c];
          xs[aa] = in_chain ? v : 0.0;
        }
  #ifdef EQLB_EXP_CHAIN3
        constexpr int SBACK = (P == 8 || P == 16) ? 0 : P - 2;
  #else
        constexpr int SBACK = P - 2;
  #endif
  #pragma unroll 1
        for (int s = BLOCK_PCR ? 0 : SBACK; s >= 1; --s)
        {
          double xn[KB];
  #pragma unroll
          for (int aa = 0; aa < KB; ++aa)
            xn[aa] = lane_up1_d<P>(xs[aa], gbase, sub);
          if (in_chain && sub == s)
          {
  #pragma unroll
            for (int aa = 0; aa < KB; ++aa)
  #pragma unroll
              for (int bb = 0; bb < KB; ++bb)
                xs[aa] -= Y[aa][bb] * xn[bb];
          }
        }
      }
      // (h) local unknowns of the cell
#pragma unroll
      for (int aa = 0; aa < KB; ++aa)
        if (sub == 0)
          xs[aa] = zz[1 + aa];
      ul[0] = zz[0];
#pragma unroll
      for (int aa = 0; aa < KB; ++aa)
      {
        ul[1 + aa] = xs[aa];
        ul[1 + KB + aa] = shfl_d(xs[aa], gbase + ((fi_p < P) ? fi_p : 0));
      }
      if constexpr (NADD >= 1)
      {
#pragma unroll
        for (int q = 0; q < NADD; ++q)
        {
          double v = la[q];
#pragma unroll
          for (int h = 0; h < NC; ++h)
            v -= ca[h][q] * ul[h];
          ul[NC + q] = v;
        }
      }
      if (!active)
      {
#pragma unroll
        for (int h = 0; h < NH; ++h)
          ul[h] = 0.0;
      }
    }


    // ---- phase E: back-map to RT coefficients (se/solve_patch_semiexplt.hpp:1082-1153) ----
    if (active)
    {
      // own-frame moments: mu_m -= Bm [d; um], mu_p += [d; up]
      double ym[K], yp[K];
#if EQLB_REV_FMA
      double uk[K], ut[K];
#pragma unroll
      for (int c = 0; c < K; ++c)
        uk[c] = ul[c]; // ul[0] = d, ul[1..KB] = um
      reversal_apply<K>(uk, rho_m, ut);
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        ym[j] = mu_m[j] - ut[j];
        yp[j] = mu_p[j] + ((j == 0) ? ul[0] : ul[KB + j]);
      }
#else
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        double s = mu_m[j];
#pragma unroll
        for (int c = 0; c < K; ++c)
          s -= (rev_m ? bcoef(j, c) : ((j == c) ? 1.0 : 0.0)) * ul[c]; // ul[0] = d, ul[1..KB] = um
        ym[j] = s;
        yp[j] = mu_p[j] + ((j == 0) ? ul[0] : ul[KB + j]);
      }
#endif
      if constexpr (SCATTER == 2)
      {
        // tiled launch: the row goes to the LDS slot of the owned cell (halo lanes drop it); the two
        // patch facets are written at their offsets, the DOFs of the outer facet (always zero) are
        // not stored - no select chain
        const uint32_t loc = info >> INFO_LOCAL_SHIFT;
#ifdef EQLB_EXP_NOSTORE // timing experiment (wrong results): one LDS store per row instead of 8
        if (loc != 0u)
          tile_slots[loc] = ym[0] + yp[0] + ym[K - 1] + yp[K - 1] + Rq[NQ - 1] + ul[NH - 1];
        else
#endif
        if (loc != 0u)
        {
          // packed row: the two facets of the cell that touch the patch node (ascending facet id),
          // then the interior DOFs; facet f sits at position f - (f > ln)
          constexpr int NPK = NRT - K;
          double* o = tile_slots + ((int64_t)(loc - 1) * 3 + ln) * NPK;
          const int pm = fm - ((fm > ln) ? 1 : 0), pp = fp - ((fp > ln) ? 1 : 0);
#pragma unroll
          for (int j = 0; j < K; ++j)
          {
            o[pm * K + j] = pf_m * ym[j];
            o[pp * K + j] = pf_p * yp[j];
          }
#pragma unroll
          for (int q = 0; q < NDIV; ++q)
            o[2 * K + q] = Rq[1 + q];
#pragma unroll
          for (int q = 0; q < NADD; ++q)
            o[2 * K + NDIV + q] = sgn * ul[1 + 2 * KB + q];
        }
      }
      double cout[(SCATTER == 2) ? 1 : NRT];
      if constexpr (SCATTER != 2)
      {
#pragma unroll
        for (int e = 0; e < 3 * K; ++e)
        {
          const int fe = e / K, j = e % K;
          cout[e] = (fe == fm) ? pf_m * ym[j] : ((fe == fp) ? pf_p * yp[j] : 0.0);
        }
#pragma unroll
        for (int q = 0; q < NDIV; ++q)
          cout[3 * K + q] = Rq[1 + q];
#pragma unroll
        for (int q = 0; q < NADD; ++q)
          cout[3 * K + NDIV + q] = sgn * ul[1 + 2 * KB + q]; // interior unknowns are scaled by sign(detJ),
                                                            // so that the tensors TE/WQ carry no sign
      }
      (void)cout;
      if constexpr (SCATTER == 2)
      {
      }
      else if constexpr (SCATTER == 0)
      {
        // every (cell, vertex) row of the slot buffer is written exactly once
#ifdef EQLB_EXP_SLOTROW // timing experiment (wrong results): rows of consecutive lanes are contiguous
        double* o = a.out + (slot % ((int64_t)a.ncells * 3)) * NRT;
#else
        double* o = a.out + (((int64_t)a.rhs_out * a.ncells + cell) * 3 + ln) * NRT;
#endif
#pragma unroll
        for (int e = 0; e < NRT; ++e)
          o[e] = cout[e];
      }
      else
      {
        double* o = a.out + ((int64_t)a.rhs_out * a.ncells + cell) * NRT;
#pragma unroll
        for (int e = 0; e < NRT; ++e)
          unsafeAtomicAdd(o + e, cout[e]);
      }
    }
  }

  if (status_local)
    atomicOr(a.status, 1);
}

// one bin per launch (any solver)
#ifndef EQLB_K4_WAVES
#define EQLB_K4_WAVES 1 // waves per SIMD asked for the register-solver kernels at k = 4 (1: up to 512 registers)
#endif
template <int K, int DEG, int P, int SOLVER, int SCATTER, int MODE = 0>
__global__ void __launch_bounds__((Sizes<K, DEG, P>::block_of(SOLVER)), ((K >= 4 && SOLVER == 1) ? EQLB_K4_WAVES : 1))
k_se_patch(const SeArgs a)
{
  extern __shared__ __align__(16) double lds[];
  se_patch_body<K, DEG, P, SOLVER, SCATTER, Sizes<K, DEG, P>::block_of(SOLVER), MODE>(a, blockIdx.x, lds);
}

// all bins in ONE launch (register solver): blocks [start[b], start[b+1]) run the P = 4 << b body,
// so the bins overlap on the chip and there are no launch gaps / tails between them
template <int K, int DEG, int SOLVER, int SCATTER>
#ifndef EQLB_FUSED_WAVES
#define EQLB_FUSED_WAVES 4
#endif
#ifndef EQLB_PERSIST_OVERSUB
#define EQLB_PERSIST_OVERSUB 1
#endif
__global__ void __launch_bounds__(256, (K <= 2 ? EQLB_FUSED_WAVES : 1)) k_se_patch_fused(const SeArgs a0, const FusedBins fb)
{
  // PERSISTENT workgroups: the grid is sized to the chip (launch_fused_kd), every workgroup stages
  // the reference tensors in LDS once and then strides over the 256-lane work blocks of all bins;
  // its waves drift apart (no block barrier in the loop), so gathers of one wave overlap the
  // arithmetic of the others.
  extern __shared__ __align__(16) double lds[];
  using Z = Sizes<K, DEG, 8>;
  for (int i = threadIdx.x; i < Z::NTAB; i += 256)
    lds[i] = a0.tables[Z::NS + i];
  __syncthreads();
  const int64_t nwork = fb.block_start[MAX_BINS];
  for (int64_t bid = blockIdx.x; bid < nwork; bid += gridDim.x)
  {
    int b = 0;
#pragma unroll
    for (int i = 1; i < MAX_BINS; ++i)
      if (bid >= fb.block_start[i])
        b = i;
    SeArgs a = a0;
    a.npatch = fb.npatch[b];
    a.slot_offset = fb.slot_offset[b];
    a.patch_offset = fb.patch_offset[b];
    const int64_t lb = bid - fb.block_start[b];
    switch (b)
    {
    case 0:
      se_patch_body<K, DEG, 4, SOLVER, SCATTER, 256>(a, lb, lds, true);
      break;
    case 1:
      se_patch_body<K, DEG, 8, SOLVER, SCATTER, 256>(a, lb, lds, true);
      break;
    case 2:
      se_patch_body<K, DEG, 16, SOLVER, SCATTER, 256>(a, lb, lds, true);
      break;
    case 3:
      se_patch_body<K, DEG, 32, SOLVER, SCATTER, 256>(a, lb, lds, true);
      break;
    default:
      se_patch_body<K, DEG, 64, SOLVER, SCATTER, 256>(a, lb, lds, true);
      break;
    }
  }
}

// constrained-minimisation (EV) patch problems, all bins in one launch (MODE 1 of the body)
template <int K, int DEG>
__global__ void __launch_bounds__(256, (K <= 2 ? 2 : 1)) k_ev_patch_fused(const SeArgs a0, const FusedBins fb)
{
  extern __shared__ __align__(16) double lds[];
  const int64_t bid = blockIdx.x;
  int b = 0;
#pragma unroll
  for (int i = 1; i < MAX_BINS; ++i)
    if (bid >= fb.block_start[i])
      b = i;
  SeArgs a = a0;
  a.npatch = fb.npatch[b];
  a.slot_offset = fb.slot_offset[b];
  a.patch_offset = fb.patch_offset[b];
  const int64_t lb = bid - fb.block_start[b];
  switch (b)
  {
  case 0:
    se_patch_body<K, DEG, 4, 1, 0, 256, 1>(a, lb, lds);
    break;
  case 1:
    se_patch_body<K, DEG, 8, 1, 0, 256, 1>(a, lb, lds);
    break;
  case 2:
    se_patch_body<K, DEG, 16, 1, 0, 256, 1>(a, lb, lds);
    break;
  case 3:
    se_patch_body<K, DEG, 32, 1, 0, 256, 1>(a, lb, lds);
    break;
  default:
    se_patch_body<K, DEG, 64, 1, 0, 256, 1>(a, lb, lds);
    break;
  }
}

template <int K, int DEG>
static int launch_ev_fused_kd(const SeArgs& a, const FusedBins& fb, hipStream_t stream)
{
  const size_t lds_bytes = sizeof(double) * (size_t)Sizes<K, DEG, 8>::lds_doubles(256, 1, 1);
  const int64_t grid = fb.block_start[MAX_BINS];
  if (grid == 0)
    return 0;
  hipLaunchKernelGGL((k_ev_patch_fused<K, DEG>), dim3((unsigned)grid), dim3(256), lds_bytes, stream, a, fb);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_ev_patch_fused(int k, const SeArgs& a, const FusedBins& fb, hipStream_t stream)
{
  if (k == 1)
    return launch_ev_fused_kd<1, 0>(a, fb, stream);
  if (k == 2)
    return launch_ev_fused_kd<2, 1>(a, fb, stream);
  if (k == 3)
    return launch_ev_fused_kd<3, 2>(a, fb, stream);
  return EQLB_ERR_UNSUPPORTED;
}

// ---- tiled launch: no slot buffer, no reduction pass ----------------------------------------------
// One workgroup (8 waves) per tile of TC owned cells.  It solves every patch that touches an owned
// cell (patches on the tile rim are solved by each tile they touch), writes the (cell, vertex) rows
// of its OWN cells into LDS - every row exactly once, no atomics - and finally adds
// row(v0) + row(v1) + row(v2) in fixed order to flux_hdiv: bitwise reproducible from run to run.
#ifndef EQLB_XCD_REMAP
#define EQLB_XCD_REMAP 1
#endif
#ifndef EQLB_TILE_THREADS
#define EQLB_TILE_THREADS 512
#endif
#ifndef EQLB_FLUSH_NT
#define EQLB_FLUSH_NT 0 // nontemporal loads/stores of flux_hdiv in the tile flush
#endif
#ifndef EQLB_TILE_THREADS_K3
#define EQLB_TILE_THREADS_K3 512 // the k = 3 body runs at 2 waves/SIMD: ONE 8-wave workgroup per CU with the whole LDS
#endif
// k <= 2: 8 waves own EQLB_TILE_CELLS cells (two workgroups per CU); k = 3: 8 waves, EQLB_TILE_CELLS_K3 cells (one per CU)
constexpr int tile_threads_c(int k) { return (k >= 3) ? EQLB_TILE_THREADS_K3 : EQLB_TILE_THREADS; }
#ifndef EQLB_TILE_CELLS
#define EQLB_TILE_CELLS 480 // 480 cells x 18 packed values + tables: two workgroups per CU (SE and EV)
#endif
#ifndef EQLB_TILE_CELLS_K3
#define EQLB_TILE_CELLS_K3 492 // LARGEST tile: 492 cells x 36 packed values (141.7 KB) + 21.2 KB of tables (half of WQ) of the 160 KB: one workgroup per CU; the tile builder picks the size that fills whole rounds of the 256 slots
                               // (measured at 1M triangles: 256 threads / 160 cells 0.384 ms, 512 / 320 0.366, 512 / 440 0.339)
#endif
constexpr int tile_cells_c(int k) { return (k >= 3) ? EQLB_TILE_CELLS_K3 : EQLB_TILE_CELLS; }
#ifndef EQLB_TILE_CELLS_K3_EV
#define EQLB_TILE_CELLS_K3_EV 468 // EV mode stages 7.2 KB more tensors (HG, WG)
#endif
// largest tile the LDS budget of two workgroups per CU allows (k <= 2: 490 x 144 B + tensors <= 80 KB)
constexpr int tile_cells_max_c(int k) { return (k >= 3) ? EQLB_TILE_CELLS_K3 : (EQLB_TILE_CELLS > 490 ? EQLB_TILE_CELLS : 490); }
int tile_cells_of(int k) { return tile_cells_c(k); }
int tile_cells_ev_of(int k) { return (k >= 3) ? EQLB_TILE_CELLS_K3_EV : tile_cells_c(k); }
int tile_cells_max_of(int k) { return tile_cells_max_c(k); }

// facet-owner table of the EV flush: for the owned cell cl of a tile and its local facet lf the
// code 2 * facet + reversal bit if the cell is the FIRST cell of the facet (it writes the facet's
// conforming DOFs), else -1
__global__ void __launch_bounds__(256)
k_tile_facet_owner(int64_t n, const int32_t* __restrict__ tile_cells, const int32_t* __restrict__ cell_facets,
                   const int32_t* __restrict__ facet_cells_off, const int32_t* __restrict__ facet_cells,
                   const uint8_t* __restrict__ facet_perm, int32_t* __restrict__ code)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n)
    return;
  const int32_t c = tile_cells[e / 3];
  const int lf = (int)(e % 3);
  int32_t v = -1;
  if (c >= 0)
  {
    const int32_t fct = cell_facets[(int64_t)c * 3 + lf];
    if (facet_cells[facet_cells_off[fct]] == c)
      v = 2 * fct + (facet_perm[(int64_t)c * 3 + lf] ? 1 : 0);
  }
  code[e] = v;
}

void launch_tile_facet_owner(const DeviceMesh& m, int64_t n, const int32_t* tile_cells, int32_t* code,
                             hipStream_t stream)
{
  if (n > 0)
    hipLaunchKernelGGL(k_tile_facet_owner, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n,
                       tile_cells, m.cell_facets, m.facet_cells_off, m.facet_cells, m.facet_perm, code);
}

// MODE 0: semi-explicit flux, flux_hdiv in the broken layout.  MODE 1: EV patch problems; flush to
// the conforming DOFs (ta.facet_owner != nullptr) or to the broken layout ("output" = 1).
// bijective XCD swizzle (cdna_hip_programming.md T1): block b of n -> position of b in the order
// "all blocks of XCD label 0, then of label 1, ..."
__device__ __forceinline__ int xcd_remap(int b, int n)
{
#if EQLB_XCD_REMAP
  const int q = n / 8, r = n % 8, x = b % 8;
  return ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
#else
  return b;
#endif
}

// DOF i of a cell from its three packed (cell, vertex) rows in LDS: row(v0) + row(v1) + row(v2) in this
// order (the order of the slot path); a facet DOF gets nothing from the vertex opposite to its facet
template <int K, int NPK>
__device__ __forceinline__ double packed_sum(const double* rows, int i)
{
  if (i >= 3 * K) // interior DOFs: all three rows
  {
    const int q = i - K;
    return (rows[q] + rows[NPK + q]) + rows[2 * NPK + q];
  }
  const int f = i / K, j = i - f * K;
  // rows ln != f, ascending; position of facet f in row ln: f - (f > ln)
  const int la = (f == 0) ? 1 : 0, lb = (f == 2) ? 1 : 2;
  return rows[la * NPK + (f - ((f > la) ? 1 : 0)) * K + j] + rows[lb * NPK + (f - ((f > lb) ? 1 : 0)) * K + j];
}

// first half of a tile: the reference tensors into LDS (and the zeroing of the slots where a node mask leaves
// rows unwritten), ends with the workgroup barrier
template <int K, int DEG, int MODE>
__device__ __forceinline__ void tile_stage(const SeArgs& a0, const TileArgs& ta, const int tile, double* lds)
{
  constexpr int TILE_THREADS = tile_threads_c(K);
  using Z = Sizes<K, DEG, 8>;
  constexpr int NRT = Z::NRT;
  const int TC = ta.tc;
  constexpr int NPK = NRT - K; // packed (cell, vertex) row: without the facet opposite to the vertex
  constexpr bool HALFWQ = K == 3; // as in se_patch_body
  constexpr int NTABL = HALFWQ ? Z::NTAB_HALF : Z::NTAB;
  constexpr int NTABM = NTABL + (MODE ? Z::NEV : 0);
  double* sSlots = lds + NTABM;
  if constexpr (HALFWQ)
  {
    constexpr int NHEAD = Z::NF + Z::NHT + Z::NDT + Z::NTET; // F | H | D | TE
    constexpr int NCMB = Z::NCMB;                            // one combination of WQ (NCMBH: its padded stride in LDS)
    for (int i = threadIdx.x; i < NHEAD; i += TILE_THREADS)
      lds[i] = a0.tables[Z::NS + i];
    for (int i = threadIdx.x; i < (NCOMBO / 2) * NCMB; i += TILE_THREADS) // combinations 0, 2, 4, ...: no reversal
      lds[NHEAD + (i / NCMB) * Z::NCMBH + (i % NCMB)] = a0.tables[Z::NS + NHEAD + (i / NCMB) * 2 * NCMB + (i % NCMB)];
    for (int i = threadIdx.x; i < Z::NHB; i += TILE_THREADS)
      lds[NHEAD + Z::NWQH + i] = a0.tables[Z::NS + NHEAD + Z::NWQT + i];
  }
  else
    for (int i = threadIdx.x; i < Z::NTAB; i += TILE_THREADS)
      lds[i] = a0.tables[Z::NS + i];
  if constexpr (MODE == 1)
    for (int i = threadIdx.x; i < Z::NEV; i += TILE_THREADS)
      lds[NTABL + i] = a0.tables[Z::OFF_HG + i];
  // every packed row of an owned cell is written completely by the patch of its vertex; rows of
  // vertices that this rank does not equilibrate (node mask; flagged per tile) must read as zero in
  // the flush
  if (ta.tiles[tile].zero)
    for (int i = threadIdx.x; i < TC * 3 * NPK; i += TILE_THREADS)
      sSlots[i] = 0.0;
  __syncthreads();
}

// second half: every patch of the tile for ONE right-hand side (a0.rhs, a0.flux_dg, a0.rhs_dg, a0.out), then the
// flush of the tile's rows
template <int K, int DEG, int MODE>
__device__ __forceinline__ void tile_sweep_flush(const SeArgs& a0, const TileArgs& ta, const int tile, double* lds)
{
  constexpr int TILE_THREADS = tile_threads_c(K);
  using Z = Sizes<K, DEG, 8>;
  constexpr int NRT = Z::NRT;
  constexpr int TCMAX = tile_cells_max_c(K); // sizes the register arrays of the flush
  const int TC = ta.tc;                      // cells per tile of this SoA (<= TCMAX)
  constexpr int NPK = NRT - K;
  constexpr bool HALFWQ = K == 3;
  constexpr int NTABL = HALFWQ ? Z::NTAB_HALF : Z::NTAB;
  constexpr int NTABM = NTABL + (MODE ? Z::NEV : 0);
  double* sSlots = lds + NTABM;
  const TileDesc& td = ta.tiles[tile];
#ifndef EQLB_TILE_SOLVER
#define EQLB_TILE_SOLVER 1
#endif
#ifndef EQLB_EXP_FULLONLY
#define EQLB_EXP_FULLONLY 0
#endif
#ifndef EQLB_TILE_INTERIOR
#define EQLB_TILE_INTERIOR 1
#endif
#ifndef EQLB_TILE_INTERIOR_K3
#define EQLB_TILE_INTERIOR_K3 0 // the interior-patch instance for RT_3 as well: no gain on the Delaunay mesh (0.413 - 0.418 ms either way)
#endif
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = TILE_THREADS / 64;
  int u = wave;
  SeArgs a = a0;
#define EQLB_TILE_BIN(B, PP)                                                                        \
  {                                                                                                 \
    const int np = td.npatch[B];                                                                    \
    const int nwb = (np * PP + 63) >> 6;                                                            \
    a.npatch = np;                                                                                  \
    a.slot_offset = td.slot_start[B];                                                               \
    a.patch_offset = td.patch_start[B];                                                             \
    /* complete wave-blocks of full patches (RT_1: the body is too small for the second instance to pay) */ \
    constexpr bool SPEC = PP <= 8 && K >= 2;                                                        \
    const int nwb_full = SPEC ? ((td.nfull[B] * PP) >> 6) : 0;                                      \
    /* wave-blocks of interior patches of any size (the patches behind the full ones; K = 2, P = 8, 16) */ \
    constexpr bool SPECI = EQLB_TILE_INTERIOR && ((K == 2 && (PP == 8 || PP == 16)) || (EQLB_TILE_INTERIOR_K3 && K == 3 && PP == 8)); \
    const int nwb_int = SPECI ? ((td.nint[B] * PP) >> 6) : 0;                                       \
    for (; u < nwb; u += NW)                                                                        \
    {                                                                                               \
      if (u < nwb_full)                                                                             \
        se_patch_body<K, DEG, PP, EQLB_TILE_SOLVER, 2, 64, MODE, SPEC>(a, 0, lds, true, (int64_t)u * 64 + lane, sSlots); \
      else if (SPECI && u < nwb_int)                                                                \
        se_patch_body<K, DEG, PP, EQLB_TILE_SOLVER, 2, 64, MODE, false, SPECI>(a, 0, lds, true, (int64_t)u * 64 + lane, sSlots); \
      else if (!EQLB_EXP_FULLONLY) /* timing experiment: only the full-patch instance */            \
        se_patch_body<K, DEG, PP, EQLB_TILE_SOLVER, 2, 64, MODE>(a, 0, lds, true, (int64_t)u * 64 + lane, sSlots); \
    }                                                                                               \
    u -= nwb;                                                                                       \
  }
#ifndef EQLB_EXP_NOBODY
  EQLB_TILE_BIN(0, 4)
  EQLB_TILE_BIN(1, 8)
  EQLB_TILE_BIN(2, 16)
  EQLB_TILE_BIN(3, 32)
  EQLB_TILE_BIN(4, 64)
#else
  (void)u;
  (void)lane;
  (void)td;
#endif
#undef EQLB_TILE_BIN
  const int32_t* cells = ta.tile_cells + (int64_t)tile * TC;
  const bool conforming = MODE == 1 && ta.facet_owner != nullptr;
  // flush operands of the broken layout: the old values of flux_hdiv are fetched BEFORE the barrier
  // (only this tile writes them), so that the two dependent loads hide behind the waves still solving
  double* x = a0.out + (int64_t)a0.rhs_out * a0.ncells * NRT;
  // (two consecutive DOFs per thread where the row length is even: 16-byte loads and stores)
  constexpr int VW = (NRT % 2 == 0) ? 2 : 1;
  constexpr int NIT = (TCMAX * NRT / VW + TILE_THREADS - 1) / TILE_THREADS;
  double xv[NIT][VW];
  int64_t xi[NIT];
  if (!conforming)
  {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
    {
      const int e = (it * TILE_THREADS + threadIdx.x) * VW;
      const int cl = e / NRT, i = e - cl * NRT;
      const int32_t cell = (e < TC * NRT) ? cells[cl] : -1;
      xi[it] = (cell >= 0) ? (int64_t)cell * NRT + i : -1;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
    {
#ifdef EQLB_EXP_NOXREAD // timing experiment (wrong results)
      xv[it][0] = xv[it][VW - 1] = 0.0;
      continue;
#endif
      if (!ta.accumulate) // store instead of add: the old values are not read (wave-uniform)
      {
        xv[it][0] = xv[it][VW - 1] = 0.0;
        continue;
      }
      if constexpr (VW == 2)
      {
#if EQLB_FLUSH_NT
        xv[it][0] = (xi[it] >= 0) ? __builtin_nontemporal_load(x + xi[it]) : 0.0;
        xv[it][1] = (xi[it] >= 0) ? __builtin_nontemporal_load(x + xi[it] + 1) : 0.0;
#else
        const double2 t = (xi[it] >= 0) ? *reinterpret_cast<const double2*>(x + xi[it]) : make_double2(0.0, 0.0);
        xv[it][0] = t.x;
        xv[it][1] = t.y;
#endif
      }
      else
        xv[it][0] = (xi[it] >= 0) ? x[xi[it]] : 0.0;
    }
  }
  // conforming layout (hierarchic basis): the owner codes, the DOF numbers and the old values of this thread's
  // facet / interior entries are fetched BEFORE the barrier as well (three dependent loads otherwise sit
  // between the barrier and the stores of every tile)
  constexpr int NIC = K * K - K;
  constexpr int NFE = (TCMAX * 3 + TILE_THREADS - 1) / TILE_THREADS;
  constexpr int NIE = (NIC > 0) ? (TCMAX * NIC + TILE_THREADS - 1) / TILE_THREADS : 1;
  int32_t fcode[NFE];
  int64_t fdof[NFE];
  double fold[NFE][K];
  int64_t idof[NIE];
  double iold[NIE];
  const bool conf_plain = conforming && ta.basis_C == nullptr;
  if (conf_plain)
  {
    const double* xc0 = a0.out + (int64_t)a0.rhs_out * ta.ndofs;
    const int32_t* own0 = ta.facet_owner + (int64_t)tile * TC * 3;
#pragma unroll
    for (int it = 0; it < NFE; ++it)
    {
      const int e = it * TILE_THREADS + threadIdx.x;
      fcode[it] = (e < TC * 3) ? own0[e] : -1;
    }
#pragma unroll
    for (int it = 0; it < NFE; ++it)
    {
      const int e = it * TILE_THREADS + threadIdx.x;
      const int cl = e / 3, lf = e - 3 * cl;
      fdof[it] = -1;
      if (fcode[it] >= 0) // facet DOFs are numbered consecutively along j (default map) or looked up per j below
        fdof[it] = ta.cell_dofs ? (int64_t)cells[cl] * NRT + lf * K : (int64_t)(fcode[it] >> 1) * K;
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        fold[it][j] = 0.0;
        if (fcode[it] >= 0 && ta.accumulate)
          fold[it][j] = ta.cell_dofs ? xc0[ta.cell_dofs[fdof[it] + j]] : xc0[fdof[it] + j];
      }
    }
    if constexpr (NIC > 0)
    {
#pragma unroll
      for (int it = 0; it < NIE; ++it)
      {
        const int e = it * TILE_THREADS + threadIdx.x;
        const int cl = e / NIC, i = e - cl * NIC;
        const int32_t cell = (e < TC * NIC) ? cells[cl] : -1;
        idof[it] = -1;
        iold[it] = 0.0;
        if (cell >= 0)
        {
          idof[it] = ta.cell_dofs ? (int64_t)ta.cell_dofs[(int64_t)cell * NRT + 3 * K + i]
                                  : (int64_t)ta.nfacets * K + (int64_t)cell * NIC + i;
          if (ta.accumulate)
            iold[it] = xc0[idof[it]];
        }
      }
    }
  }
  __syncthreads();

  if (conforming && ta.basis_C != nullptr)
  {
    // conforming DOFs of ANOTHER element basis of RT_k (the Basix space of FluxEqlbEV.py:95-100 through a
    // DOLFINx-side adapter): cell coefficients = C x broken hierarchic coefficients, facet block through R
    // where the cell sees the facet reversed; facet DOFs by the first cell of the facet
    constexpr int NI = K * K - K;
    double* xc = a0.out + (int64_t)a0.rhs_out * ta.ndofs;
    const int32_t* own = ta.facet_owner + (int64_t)tile * TC * 3;
    for (int cl = threadIdx.x; cl < TC; cl += TILE_THREADS)
    {
      const int32_t cell = cells[cl];
      if (cell < 0)
        continue;
      double c[NRT];
#pragma unroll
      for (int i = 0; i < NRT; ++i)
        c[i] = packed_sum<K, NPK>(sSlots + (int64_t)cl * 3 * NPK, i);
      // rows of C are fetched per output row inside ROLLED loops: unrolled, the compiler hoists the whole matrix
      // (k(k+2)^2 doubles) out of the cell loop and the kernel pays for the scratch it then needs on EVERY path
      auto row = [&](const int i) {
        const double* Ci = ta.basis_C + (int64_t)i * NRT;
        double s_ = 0.0;
#pragma unroll
        for (int j = 0; j < NRT; ++j)
          s_ = __builtin_fma(Ci[j], c[j], s_);
        return s_;
      };
#pragma unroll 1
      for (int lf = 0; lf < 3; ++lf)
      {
        const int32_t code = own[cl * 3 + lf];
        if (code < 0)
          continue;
        const bool rev = (code & 1) != 0 && ta.basis_R != nullptr;
        double yf[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
          yf[j] = row(lf * K + j);
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          double g = yf[j];
          if (rev)
          {
            g = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i)
              g += ta.basis_R[j * K + i] * yf[i];
          }
          const int64_t dof = ta.cell_dofs ? (int64_t)ta.cell_dofs[(int64_t)cell * NRT + lf * K + j]
                                           : (int64_t)(code >> 1) * K + j;
          xc[dof] = ta.accumulate ? xc[dof] + g : g;
        }
      }
#pragma unroll 1
      for (int i = 0; i < NI; ++i)
      {
        const double yi = row(3 * K + i);
        const int64_t dof = ta.cell_dofs ? (int64_t)ta.cell_dofs[(int64_t)cell * NRT + 3 * K + i]
                                         : (int64_t)ta.nfacets * K + (int64_t)cell * NI + i;
        xc[dof] = ta.accumulate ? xc[dof] + yi : yi;
      }
    }
    return;
  }
  if (conforming)
  {
    // conforming DOFs (ev/solve_patch.hpp:223-227): facet DOFs by the first cell of the facet,
    // mapped to the global facet frame (T_f = -I / B), interior DOFs by their cell
    double* xc = a0.out + (int64_t)a0.rhs_out * ta.ndofs;
#pragma unroll
    for (int it = 0; it < NFE; ++it)
    {
      if (fcode[it] < 0)
        continue;
      const int e = it * TILE_THREADS + threadIdx.x;
      const int cl = e / 3, lf = e - 3 * cl;
      const bool rev = (fcode[it] & 1) != 0;
      double v[K];
#pragma unroll
      for (int j = 0; j < K; ++j)
        v[j] = packed_sum<K, NPK>(sSlots + (int64_t)cl * 3 * NPK, lf * K + j);
#pragma unroll
      for (int j = 0; j < K; ++j)
      {
        double g = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i)
          g += (rev ? bcoef(j, i) : ((i == j) ? -1.0 : 0.0)) * v[i];
        const int64_t dof = ta.cell_dofs ? (int64_t)ta.cell_dofs[fdof[it] + j] : fdof[it] + j;
        xc[dof] = fold[it][j] + g;
      }
    }
    if constexpr (NIC > 0)
    {
#pragma unroll
      for (int it = 0; it < NIE; ++it)
      {
        if (idof[it] < 0)
          continue;
        const int e = it * TILE_THREADS + threadIdx.x;
        const int cl = e / NIC, i = e - cl * NIC;
        xc[idof[it]] = iold[it] + packed_sum<K, NPK>(sSlots + (int64_t)cl * 3 * NPK, 3 * K + i);
      }
    }
    return;
  }

#ifdef EQLB_EXP_NOFLUSH // timing experiment (wrong results)
  if (ta.ntiles > 0)
    return;
#endif
#pragma unroll
  for (int it = 0; it < NIT; ++it)
  {
    const int e = (it * TILE_THREADS + threadIdx.x) * VW;
    const int cl = e / NRT, i = e - cl * NRT;
    if (xi[it] >= 0)
    {
      const double* sl = sSlots + (int64_t)cl * 3 * NPK;
      if constexpr (VW == 2)
      {
        double2 t;
        t.x = xv[it][0] + packed_sum<K, NPK>(sl, i);
        t.y = xv[it][1] + packed_sum<K, NPK>(sl, i + 1);
#ifdef EQLB_EXP_NOXSTORE // timing experiment (wrong results)
        if (t.x != 1.2345)
          continue;
#endif
#if EQLB_FLUSH_NT
        __builtin_nontemporal_store(t.x, x + xi[it]);
        __builtin_nontemporal_store(t.y, x + xi[it] + 1);
#else
        *reinterpret_cast<double2*>(x + xi[it]) = t;
#endif
      }
      else
        x[xi[it]] = xv[it][0] + packed_sum<K, NPK>(sl, i);
    }
  }
}

template <int K, int DEG, int MODE>
__global__ void __launch_bounds__(tile_threads_c(K), (K <= 2 ? 4 : 1)) k_se_patch_tiled(const SeArgs a0, const TileArgs ta)
{
  extern __shared__ __align__(16) double lds[];
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, tiles are numbered along the
  // bisection tree (neighbours in space are neighbours in index); give every XCD one contiguous
  // range of tiles so that the rim cells two tiles share are read through the same L2
  const int tile = ta.tile_first + xcd_remap(blockIdx.x, ta.ntiles);
  tile_stage<K, DEG, MODE>(a0, ta, tile, lds);
  tile_sweep_flush<K, DEG, MODE>(a0, ta, tile, lds);
}

// All right-hand sides of a call in ONE launch (se/solve_patch_semiexplt.hpp:1040-1075 loops the right-hand
// sides inside the patch): a workgroup per (tile, right-hand side); the workgroups of one tile are neighbours in
// the launch order of their XCD, so the tile's descriptors, geometry and tensors are read from HBM once and
// come from that XCD's L2 for the other right-hand sides.  A loop over the right-hand sides INSIDE the
// workgroup was built and measured first (tensors staged once per tile): the loop makes the compiler hoist
// lane predicates and table addresses out of it, 47 spilled registers, 0.565 ms for 4 right-hand sides at 1M
// triangles against 0.357 ms for four separate launches.  What the reference re-uses across the
// right-hand sides - the factorisation - is a handful of multipliers here; they are recomputed.
template <int K, int DEG, int MODE>
__global__ void __launch_bounds__(tile_threads_c(K), (K <= 2 ? 4 : 1))
k_se_patch_tiled_multi(const SeArgs a0, const TileArgs ta, const MultiRhs mr)
{
  extern __shared__ __align__(16) double lds[];
  const int pos = xcd_remap(blockIdx.x, ta.ntiles * mr.n);
  const int tile = ta.tile_first + pos / mr.n;
  const int r = pos - (pos / mr.n) * mr.n;
  SeArgs a = a0;
  a.rhs = mr.rhs0 + r;
  a.flux_dg = mr.g[r];
  a.rhs_dg = mr.f[r];
  a.out = mr.x[r];
  a.rhs_in = 0;
  a.rhs_out = 0;
  tile_stage<K, DEG, MODE>(a, ta, tile, lds);
  tile_sweep_flush<K, DEG, MODE>(a, ta, tile, lds);
}

template <int K, int DEG, int MODE>
static int launch_tiled_multi_kd(const SeArgs& a, const TileArgs& t, const MultiRhs& mr, hipStream_t stream)
{
  using Z = Sizes<K, DEG, 8>;
  const size_t lds_bytes
      = sizeof(double) * ((size_t)(K == 3 ? Z::NTAB_HALF : Z::NTAB) + (MODE ? (size_t)Z::NEV : 0) + (size_t)t.tc * 3 * (Z::NRT - K));
  if (lds_bytes > 160 * 1024 || t.tc < 1 || t.tc > tile_cells_max_c(K) || mr.n < 1 || mr.n > MULTI_RHS_MAX)
    return EQLB_ERR_UNSUPPORTED;
  auto kern = k_se_patch_tiled_multi<K, DEG, MODE>;
  if (lds_bytes > 64 * 1024)
  {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return EQLB_ERR_DEVICE;
  }
  if (t.ntiles == 0)
    return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)t.ntiles * (unsigned)mr.n), dim3(tile_threads_c(K)), lds_bytes, stream, a, t, mr);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_se_patch_tiled_multi(int k, int deg, int mode, const SeArgs& a, const TileArgs& t, const MultiRhs& mr,
                                hipStream_t stream)
{
  if (mode == 1)
  {
    if (k == 1)
      return launch_tiled_multi_kd<1, 0, 1>(a, t, mr, stream);
    if (k == 2)
      return launch_tiled_multi_kd<2, 1, 1>(a, t, mr, stream);
    if (k == 3)
      return launch_tiled_multi_kd<3, 2, 1>(a, t, mr, stream);
    return EQLB_ERR_UNSUPPORTED;
  }
  if (k == 1 && deg == 0)
    return launch_tiled_multi_kd<1, 0, 0>(a, t, mr, stream);
  if (k == 2 && deg == 1)
    return launch_tiled_multi_kd<2, 1, 0>(a, t, mr, stream);
  if (k == 3 && deg == 2)
    return launch_tiled_multi_kd<3, 2, 0>(a, t, mr, stream);
  return EQLB_ERR_UNSUPPORTED;
}

template <int K, int DEG, int MODE>
static int launch_tiled_kd(const SeArgs& a, const TileArgs& t, hipStream_t stream)
{
  using Z = Sizes<K, DEG, 8>;
  const size_t lds_bytes
      = sizeof(double) * ((size_t)(K == 3 ? Z::NTAB_HALF : Z::NTAB) + (MODE ? (size_t)Z::NEV : 0) + (size_t)t.tc * 3 * (Z::NRT - K));
  if (lds_bytes > 160 * 1024 || t.tc < 1 || t.tc > tile_cells_max_c(K))
    return EQLB_ERR_UNSUPPORTED;
  auto kern = k_se_patch_tiled<K, DEG, MODE>;
  if (lds_bytes > 64 * 1024)
  {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return EQLB_ERR_DEVICE;
  }
  if (t.ntiles == 0)
    return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)t.ntiles), dim3(tile_threads_c(K)), lds_bytes, stream, a, t);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_se_patch_tiled(int k, int deg, int mode, const SeArgs& a, const TileArgs& t, hipStream_t stream)
{
  if (mode == 1)
  {
    if (k == 1)
      return launch_tiled_kd<1, 0, 1>(a, t, stream);
    if (k == 2)
      return launch_tiled_kd<2, 1, 1>(a, t, stream);
    if (k == 3)
      return launch_tiled_kd<3, 2, 1>(a, t, stream);
    return EQLB_ERR_UNSUPPORTED;
  }
  if (k == 1 && deg == 0)
    return launch_tiled_kd<1, 0, 0>(a, t, stream);
  if (k == 2 && deg == 1)
    return launch_tiled_kd<2, 1, 0>(a, t, stream);
  if (k == 3 && deg == 2)
    return launch_tiled_kd<3, 2, 0>(a, t, stream);
  return EQLB_ERR_UNSUPPORTED;
}

// flux_hdiv[r][cell][i] += slot0 + slot1 + slot2  (fixed order -> bitwise reproducible)
template <int NRT>
__global__ void __launch_bounds__(256)
k_reduce_slots(int64_t ntotal, const double* __restrict__ slots, double* __restrict__ x, int accumulate)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ntotal)
    return;
  const int64_t c = e / NRT;
  const int i = (int)(e - c * NRT);
  const double* s = slots + c * 3 * NRT + i;
  const double v = (s[0] + s[NRT]) + s[2 * NRT];
  x[e] = accumulate ? x[e] + v : v;
}

int launch_reduce_slots(int nrt, int32_t ncells, int32_t nrhs, const double* slots, double* x, int accumulate,
                        hipStream_t stream)
{
  const int64_t ntotal = (int64_t)nrhs * ncells * nrt;
  const int block = 256;
  const int64_t grid = (ntotal + block - 1) / block;
  if (nrt == 3)
    hipLaunchKernelGGL(k_reduce_slots<3>, dim3(grid), dim3(block), 0, stream, ntotal, slots, x, accumulate);
  else if (nrt == 8)
    hipLaunchKernelGGL(k_reduce_slots<8>, dim3(grid), dim3(block), 0, stream, ntotal, slots, x, accumulate);
  else if (nrt == 15)
    hipLaunchKernelGGL(k_reduce_slots<15>, dim3(grid), dim3(block), 0, stream, ntotal, slots, x, accumulate);
  else if (nrt == 24)
    hipLaunchKernelGGL(k_reduce_slots<24>, dim3(grid), dim3(block), 0, stream, ntotal, slots, x, accumulate);
  else
    return EQLB_ERR_UNSUPPORTED;
  return 0;
}

// the same for a LIST of cells, always adding (the rest of a fused stress launch: the few cells that have a vertex
// whose patch ran on the generic kernels; the slot rows of the other vertices are zero)
template <int NRT>
__global__ void __launch_bounds__(256)
k_reduce_slots_cells(int64_t ntotal, const int32_t* __restrict__ cells, const double* __restrict__ slots,
                     double* __restrict__ x)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ntotal)
    return;
  const int64_t c = cells[e / NRT];
  const int i = (int)(e % NRT);
  const double* s = slots + c * 3 * NRT + i;
  x[c * NRT + i] += (s[0] + s[NRT]) + s[2 * NRT];
}

int launch_reduce_slots_cells(int nrt, int32_t ncells, int64_t nlist, const int32_t* cells, const double* slots,
                              double* x, hipStream_t stream)
{
  (void)ncells;
  const int64_t ntotal = nlist * nrt;
  if (ntotal == 0)
    return 0;
  const int block = 256;
  const int64_t grid = (ntotal + block - 1) / block;
  if (nrt == 8)
    hipLaunchKernelGGL(k_reduce_slots_cells<8>, dim3(grid), dim3(block), 0, stream, ntotal, cells, slots, x);
  else
    return EQLB_ERR_UNSUPPORTED;
  return 0;
}

// ---- dispatch -------------------------------------------------------------------------------------
template <int K, int DEG, int P, int SOLVER, int SCATTER, int MODE = 0>
static int launch_t(const SeArgs& a, hipStream_t stream)
{
  using Z = Sizes<K, DEG, P>;
  constexpr int BLOCK = Z::block_of(SOLVER);
  const size_t lds_bytes = sizeof(double) * (size_t)Z::lds_doubles(BLOCK, SOLVER, MODE);
  if (lds_bytes > 160 * 1024)
    return EQLB_ERR_UNSUPPORTED;
  auto kern = k_se_patch<K, DEG, P, SOLVER, SCATTER, MODE>;
  if (lds_bytes > 64 * 1024)
  {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
        != hipSuccess)
      return EQLB_ERR_DEVICE;
  }
  const int64_t nthreads = a.npatch * P;
  const int64_t grid = (nthreads + BLOCK - 1) / BLOCK;
  if (grid == 0)
    return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds_bytes, stream, a);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

template <int K, int DEG, int SOLVER, int SCATTER>
static int launch_p(int P, const SeArgs& a, hipStream_t stream)
{
  switch (P)
  {
  case 4:
    return launch_t<K, DEG, 4, SOLVER, SCATTER>(a, stream);
  case 8:
    return launch_t<K, DEG, 8, SOLVER, SCATTER>(a, stream);
  case 16:
    return launch_t<K, DEG, 16, SOLVER, SCATTER>(a, stream);
  case 32:
    return launch_t<K, DEG, 32, SOLVER, SCATTER>(a, stream);
  case 64:
    return launch_t<K, DEG, 64, SOLVER, SCATTER>(a, stream);
  }
  return EQLB_ERR_UNSUPPORTED;
}

template <int K, int DEG>
static int launch_kd(int P, int solver, int scatter, const SeArgs& a, hipStream_t stream)
{
#ifdef EQLB_EXP_SOLVER9
  if (solver == 9)
    return launch_p<K, DEG, 9, 0>(P, a, stream);
#endif
  if (solver == EQLB_SOLVER_SHUFFLE)
  {
    if (scatter == EQLB_SCATTER_SLOTS)
      return launch_p<K, DEG, 1, 0>(P, a, stream);
    return launch_p<K, DEG, 1, 1>(P, a, stream);
  }
  if (scatter == EQLB_SCATTER_SLOTS)
    return launch_p<K, DEG, 0, 0>(P, a, stream);
  return launch_p<K, DEG, 0, 1>(P, a, stream);
}

template <int K, int DEG>
static int launch_fused_kd(int scatter, const SeArgs& a, const FusedBins& fb, hipStream_t stream)
{
  const size_t lds_bytes = sizeof(double) * (size_t)Sizes<K, DEG, 8>::lds_doubles(256, 1);
  const int64_t nwork = fb.block_start[MAX_BINS];
  if (nwork == 0)
    return 0;
  // persistent grid: resident workgroups of the device (occupancy x CUs), at most the work
  static int64_t resident[2] = {0, 0};
  const int si = (scatter == EQLB_SCATTER_SLOTS) ? 0 : 1;
  if (resident[si] == 0)
  {
    int dev = 0, ncu = 0, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const void* fn = si == 0 ? reinterpret_cast<const void*>(k_se_patch_fused<K, DEG, 1, 0>)
                             : reinterpret_cast<const void*>(k_se_patch_fused<K, DEG, 1, 1>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes) != hipSuccess || per_cu < 1)
      per_cu = 1;
    resident[si] = (int64_t)std::max(ncu, 1) * per_cu * EQLB_PERSIST_OVERSUB;
  }
  const int64_t grid = std::min(nwork, resident[si]);
  if (scatter == EQLB_SCATTER_SLOTS)
    hipLaunchKernelGGL((k_se_patch_fused<K, DEG, 1, 0>), dim3((unsigned)grid), dim3(256), lds_bytes, stream, a, fb);
  else
    hipLaunchKernelGGL((k_se_patch_fused<K, DEG, 1, 1>), dim3((unsigned)grid), dim3(256), lds_bytes, stream, a, fb);
  return (hipGetLastError() == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_se_patch_fused(int k, int deg, int scatter, const SeArgs& a, const FusedBins& fb,
                          hipStream_t stream)
{
  if (k == 1 && deg == 0)
    return launch_fused_kd<1, 0>(scatter, a, fb, stream);
  if (k == 2 && deg == 1)
    return launch_fused_kd<2, 1>(scatter, a, fb, stream);
  if (k == 3 && deg == 2)
    return launch_fused_kd<3, 2>(scatter, a, fb, stream);
  return EQLB_ERR_UNSUPPORTED;
}

// k = 4 (three interior unknowns per cell).  Register solver (three interior unknowns condensed per cell, 3 x 3
// blocks handed down the chain): every lanes-per-patch bin, slots or atomics.  Dense LDS Cholesky: patches of up to
// 8 facets (its tile of 52 x 52 / 2 doubles per patch and the 80 KB of RT_4 tensors fill the LDS); the only path of
// the EV patch problems at k = 4.
static int launch_k4(int P, int solver, int scatter, const SeArgs& a, hipStream_t stream, int mode)
{
  if (solver == EQLB_SOLVER_SHUFFLE && mode == 0)
  {
    if (scatter == EQLB_SCATTER_SLOTS)
      return launch_p<4, 3, 1, 0>(P, a, stream);
    return launch_p<4, 3, 1, 1>(P, a, stream);
  }
  if (solver == EQLB_SOLVER_SHUFFLE && mode == 1 && scatter == EQLB_SCATTER_SLOTS)
  {
    // constrained-minimisation patch problems (MODE 1 of the body) on the register solver
    switch (P)
    {
    case 4:
      return launch_t<4, 3, 4, 1, 0, 1>(a, stream);
    case 8:
      return launch_t<4, 3, 8, 1, 0, 1>(a, stream);
    case 16:
      return launch_t<4, 3, 16, 1, 0, 1>(a, stream);
    case 32:
      return launch_t<4, 3, 32, 1, 0, 1>(a, stream);
    case 64:
      return launch_t<4, 3, 64, 1, 0, 1>(a, stream);
    }
    return EQLB_ERR_UNSUPPORTED;
  }
  if (solver != EQLB_SOLVER_LDS_CHOLESKY || (P != 4 && P != 8))
    return EQLB_ERR_UNSUPPORTED;
  if (mode == 1) // constrained-minimisation patch problems (slots only)
  {
    if (scatter != EQLB_SCATTER_SLOTS)
      return EQLB_ERR_UNSUPPORTED;
    return (P == 4) ? launch_t<4, 3, 4, 0, 0, 1>(a, stream) : launch_t<4, 3, 8, 0, 0, 1>(a, stream);
  }
  if (scatter == EQLB_SCATTER_SLOTS)
    return (P == 4) ? launch_t<4, 3, 4, 0, 0>(a, stream) : launch_t<4, 3, 8, 0, 0>(a, stream);
  return (P == 4) ? launch_t<4, 3, 4, 0, 1>(a, stream) : launch_t<4, 3, 8, 0, 1>(a, stream);
}

int launch_se_patch(int k, int deg, int P, int solver, int scatter, const SeArgs& a,
                    hipStream_t stream, int mode)
{
  if (k == 4 && deg == 3)
    return launch_k4(P, solver, scatter, a, stream, mode);
  if (mode != 0)
    return EQLB_ERR_UNSUPPORTED; // k <= 3: the EV patch problems run on the fused / tiled launches
  if (k == 1 && deg == 0)
    return launch_kd<1, 0>(P, solver, scatter, a, stream);
  if (k == 2 && deg == 1)
    return launch_kd<2, 1>(P, solver, scatter, a, stream);
  if (k == 3 && deg == 2)
    return launch_kd<3, 2>(P, solver, scatter, a, stream);
  return EQLB_ERR_UNSUPPORTED;
}

} // namespace eqlb
