// Device-side helpers shared by the patch kernels (sizes, wave-level primitives).
#pragma once

#include "eqlb_internal.h"

namespace eqlb
{

// ---- compile-time sizes -----------------------------------------------------------------------
__host__ __device__ constexpr int nd_of(int deg) { return (deg + 1) * (deg + 2) / 2; }
__host__ __device__ constexpr int nrt_of(int k) { return k * (k + 2); }
__host__ __device__ constexpr int nq_of(int k) { return k * (k + 1) / 2; }
__host__ __device__ constexpr int binom(int n, int r)
{
  int v = 1;
  for (int i = 0; i < r; ++i)
    v = v * (n - i) / (i + 1);
  return v;
}
// B_ji = C(j,i)(-1)^i : moments w.r.t. s of a trace known by its moments w.r.t. 1-s
__host__ __device__ constexpr double bcoef(int j, int i)
{
  return (i > j) ? 0.0 : ((i % 2 == 0) ? 1.0 : -1.0) * binom(j, i);
}

// ---- wave-level helpers -----------------------------------------------------------------------
// LDS traffic between lanes of ONE wave: DS operations of a wave execute in order, the fences
// keep the compiler from moving accesses across the hand-off.
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double shfl_d(double v, int src_lane) { return __shfl(v, src_lane, 64); }

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; } // i >= j

// ---- DPP lane exchange (VALU data path, no LDS round trip) --------------------------------------
// gfx9 dpp_ctrl codes: quad_perm 0x00-0xff, row_shl:n 0x100+n, row_shr:n 0x110+n, wave_shl:1 0x130,
// wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141.  Lanes without a source get 0
// (bound_ctrl: the destination needs no copy of the old value, one v_mov_dpp per dword).
#ifndef EQLB_USE_DPP
#define EQLB_USE_DPP 1
#endif
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// value of lane - OFF (OFF = 1, 2, 4, 8 within a group of P <= 16 lanes aligned to 16-lane rows;
// any P for OFF = 1); lanes whose source is outside the row / wave get 0 (DPP) or their own value
template <int P, int OFF>
__device__ __forceinline__ double lane_down_d(double v, int gbase, int sub)
{
  if constexpr (EQLB_USE_DPP && P <= 16)
    return dpp_d<0x110 + OFF>(v);
  else if constexpr (EQLB_USE_DPP && OFF == 1)
    return dpp_d<0x138>(v);
  else
    return __shfl(v, gbase + ((sub >= OFF) ? sub - OFF : sub), 64);
}
// value of lane + 1
template <int P>
__device__ __forceinline__ double lane_up1_d(double v, int gbase, int sub)
{
  if constexpr (EQLB_USE_DPP && P <= 16)
    return dpp_d<0x101>(v);
  else if constexpr (EQLB_USE_DPP)
    return dpp_d<0x130>(v);
  else
    return __shfl(v, gbase + ((sub + 1 < P) ? sub + 1 : sub), 64);
}
// sum over the P lanes of a group, result in every lane
// EQLB_GSUM_SWIZZLE: the butterfly of group_sum_d through ds_swizzle_b32 (LDS crossbar, no LDS memory, no address
// register) instead of DPP moves: the same pairs in the same order, bit-identical sums; 2 VALU issue slots less per
// step at the price of the LDS pipe's latency
#ifndef EQLB_GSUM_SWIZZLE
#define EQLB_GSUM_SWIZZLE 0
#endif
template <int XOR>
__device__ __forceinline__ double swizzle_xor_d(double v)
{
  constexpr int PAT = (XOR << 10) | 0x1f; // bit-mask mode: lane' = ((lane & 0x1f) | 0) ^ XOR within 32 lanes
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, PAT);
  hi = __builtin_amdgcn_ds_swizzle(hi, PAT);
  return __hiloint2double(hi, lo);
}
template <int P>
__device__ __forceinline__ double group_sum_d(double v, int gbase, int sub)
{
  if constexpr (EQLB_GSUM_SWIZZLE && P <= 16)
  {
    v += swizzle_xor_d<1>(v);
    v += swizzle_xor_d<2>(v);
    if constexpr (P >= 8)
      v += swizzle_xor_d<4>(v);
    if constexpr (P >= 16)
      v += swizzle_xor_d<8>(v);
    return v;
  }
  else if constexpr (EQLB_USE_DPP && P <= 16)
  {
    v += dpp_d<0xB1>(v); // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v); // quad_perm [2,3,0,1]
    if constexpr (P >= 8)
      v += dpp_d<0x141>(v); // row_half_mirror: lane i <-> 7 - i of each half row
    if constexpr (P >= 16)
      v += dpp_d<0x140>(v); // row_mirror: lane i <-> 15 - i
    return v;
  }
  else
  {
#pragma unroll
    for (int off = 1; off < P; off <<= 1)
      v += __shfl(v, gbase + (sub ^ off), 64);
    return v;
  }
}

// ---- reciprocal / reciprocal square root: hardware seed + 2 Newton steps (full fp64 accuracy for
// the well-scaled pivots of the patch systems; no denormal/overflow scaling as in the IEEE division)
#ifndef EQLB_FAST_RCP
#define EQLB_FAST_RCP 1
#endif
__device__ __forceinline__ double rcp_d(double x)
{
#if EQLB_FAST_RCP
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
__device__ __forceinline__ double rsqrt_d(double x)
{
#if EQLB_FAST_RCP
  double y = __builtin_amdgcn_rsq(x);
  // y <- y + y * (1 - x y^2) / 2
  y = __builtin_fma(y * 0.5, __builtin_fma(-x * y, y, 1.0), y);
  y = __builtin_fma(y * 0.5, __builtin_fma(-x * y, y, 1.0), y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}


// row of the reduced tensors TE / WQ / VQ / WG: the six ordered pairs (fm, fp) of distinct local
// facet ids of a patch cell x reversal flag of the minus facet (tools/gen_tables.py: combo)
constexpr int NCOMBO = 12;
__host__ __device__ constexpr int combo_index(int fm, int fp, bool rev)
{
  return (fm * 2 + ((fp < fm) ? fp : fp - 1)) * 2 + (rev ? 1 : 0);
}

// ---- compile-time sizes of a (K, DEG, P) patch kernel ------------------------------------------
#ifndef EQLB_WQ_PAD
#define EQLB_WQ_PAD 0 // 13 was measured: RT_3 0.300 -> 0.325 ms at 1M triangles, 2.30 -> 2.47 at 8M (see Sizes::NCMBH)
#endif
template <int K, int DEG, int P>
struct Sizes
{
  static constexpr int KB = K - 1;
  static constexpr int NADD = (K - 1) * (K - 2) / 2;
  static constexpr int NDIV = K * (K + 1) / 2 - 1;
  static constexpr int NRT = nrt_of(K), ND = nd_of(DEG), NQ = nq_of(K);
  static constexpr int NY = 2 * K + NADD;      // own-frame unknowns of a cell: mu_m, mu_p, add
  static constexpr int NCOL = 2 * K + NDIV;    // columns of the load tensor: mu_m, mu_p, div DOFs
  static constexpr int NH = 1 + 2 * KB + NADD; // local unknowns [d | um | up | ua]
  static constexpr int NTE = NH * (NH + 1) / 2;
  // row strides of TE / WQ in the table buffer, padded to an even number of doubles (16-byte rows)
  static constexpr int NTES = NTE + (NTE & 1), NCOLS = NCOL + (NCOL & 1);
  static constexpr int DIMMAX = 1 + KB * P + NADD * P;
  static constexpr int TRI = DIMMAX * (DIMMAX + 1) / 2;
  static constexpr int LDS_GROUP = TRI + DIMMAX; // doubles per patch for SOLVER 0
  // device table buffer: S | F | H | D | TE | WQ ; the kernel stages everything behind S in LDS
  // rows of H padded to an even number of doubles: with it every segment and every row of the k = 2
  // tensors starts on a 16-byte boundary in LDS (ds_read_b128 instead of the half-rate ds_read2_b64)
  static constexpr int HROW = ND * NQ + ((ND * NQ) & 1);
  static constexpr int NS = 3 * NRT * NRT, NF = 9 * ND * K, NHT = 3 * HROW, NDT = 6 * ND * NQ;
  static constexpr int NTET = NCOMBO * 3 * NTES, NWQT = NCOMBO * 3 * NH * NCOLS;
  static constexpr int NHB = 9 * K * K;                           // flux-BC tensor HB
  static constexpr int NTAB = NF + NHT + NDT + NTET + NWQT + NHB;
  static constexpr int NVT = 3 * NRT * 2, NVQT = NCOMBO * 2 * NH * 3; // weak symmetry: V, VQ (behind HB)
  static constexpr int OFF_TE = NS + NF + NHT + NDT, OFF_V = OFF_TE + NTET + NWQT + NHB, OFF_VQ = OFF_V + NVT;
  // constrained-minimisation (EV) mode: HG | WG behind VQ, staged in LDS behind HB
  static constexpr int NHG = HROW, NWG = NCOMBO * NH * ND * 2, NEV = NHG + NWG; // HG padded like a row of H
  static constexpr int OFF_HG = OFF_VQ + NVQT;
  // RT_3 tiles stage only the combinations WITHOUT the reversal flag of the load tensor (half of WQ: the LDS
  // decides their tile size); for a reversed minus facet the load follows from Q_rev = Q blockdiag(B, I):
  // load_h = sum_j B[j][h] load_j for the K unknowns [d | um] (se_patch_body, HALFWQ)
  // EQLB_WQ_PAD > 0 pads the combinations of the staged half: a wave reads the same (row, column) of different
  // combinations at once, and their natural stride of 3 NH NCOLS doubles (k = 3: 216 doubles = 432 dwords = 16 mod 32
  // banks) puts three of the six combinations on the same banks of a ds_read2_b64.  Built with 13 doubles more (458
  // dwords, banks of its own for every combination) and measured SLOWER (0.325 against 0.300 ms: the rows lose their
  // 16-byte alignment, the reads are no longer paired) - kept at 0
  static constexpr int NCMB = 3 * NH * NCOLS, NCMBH = NCMB + ((K == 3) ? EQLB_WQ_PAD : 0);
  static constexpr int NWQH = (NCOMBO / 2) * NCMBH, NTAB_HALF = NTAB - NWQT + NWQH;
  // workgroup size: as many waves as fit a 64 KiB LDS budget for the dense tiles (at least one)
  // k = 4 (dense LDS solver): the tables and the patch tiles leave no room for WG, it is read from global memory
  static constexpr int NEV_LDS = (K >= 4) ? NHG : NEV;
  static constexpr int lds_doubles(int block, int solver, int mode = 0)
  {
    return NTAB + (mode ? NEV_LDS : 0) + ((K > 1 && solver == 0) ? (block / P) * LDS_GROUP : 0);
  }
  static constexpr int block_of(int solver)
  {
    // register solver at k = 4: the 80 KB of tensors are staged once per workgroup - as many waves as possible
    // behind them (two workgroups of 256 threads share a CU's 160 KB)
    if (K >= 4 && solver != 0 && P < 32)
      return 256;
    return (P >= 32) ? 64
                     : ((lds_doubles(256, solver) * 8 <= 65536)
                            ? 256
                            : ((lds_doubles(128, solver) * 8 <= 65536) ? 128 : 64));
  }
};


} // namespace eqlb
