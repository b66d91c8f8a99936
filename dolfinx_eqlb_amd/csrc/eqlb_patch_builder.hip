// Patch builder on the device: oriented vertex-patch fans -> lane-contiguous SoA descriptors.
//
// Replaces OrientedPatch::initialize_patch + next_facet/get_fctid_local/node_local
// (cpp/dolfinx_eqlb/se/Patch.cpp:406-635,637-759) and the reversed-facet detection of
// equilibrate_flux_semiexplt (se/solve_patch_semiexplt.hpp:324-389).  One thread per mesh node
// walks the fan through the CSR connectivities; the walk order (start facet, direction) is the
// reference's, so the fans are bit-identical to the CPU restatement (tests/test_patch_builder).
//
// The reference finds "the other facet of the cell that belongs to the patch" by searching a
// sorted copy of the node's facet list (Patch.cpp:718-756); a facet belongs to the patch iff it
// contains the patch node, which is what is tested here (same result, no sort).
#include "eqlb_internal.h"

namespace eqlb
{

__device__ inline int local_facet(const int32_t* cell_facets, int32_t cell, int32_t fct)
{
  const int32_t* cf = cell_facets + 3 * (int64_t)cell;
  return (cf[0] == fct) ? 0 : ((cf[1] == fct) ? 1 : 2);
}

__device__ inline int local_node(const int32_t* cell_nodes, int32_t cell, int32_t node)
{
  const int32_t* cn = cell_nodes + 3 * (int64_t)cell;
  return (cn[0] == node) ? 0 : ((cn[1] == node) ? 1 : 2);
}

// of the two other facets of `cell`, the one containing `node`
__device__ inline int32_t next_facet(const BuildArgs& a, int32_t cell, int lf, int32_t node)
{
  const int32_t* cf = a.cell_facets + 3 * (int64_t)cell;
  const int32_t e0 = cf[(lf + 1) % 3], e1 = cf[(lf + 2) % 3];
  const int32_t* fn = a.facet_nodes + 2 * (int64_t)e0;
  return (fn[0] == node || fn[1] == node) ? e0 : e1;
}

__global__ void __launch_bounds__(256) k_build_patches(BuildArgs a)
{
  const int64_t tid_g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool inst = (a.inst_node != nullptr);
  if (tid_g >= (inst ? a.ninst : (int64_t)a.nnodes))
    return;
  const int32_t node = inst ? a.inst_node[tid_g] : (int32_t)tid_g;
  const int32_t tile = inst ? a.inst_tile[tid_g] : -1;
  // tiled SoA: 1 + local index of an owned cell in the upper bits of the descriptor
  const int ws_kind = (a.node_ws && !inst) ? a.node_ws[node] : 0;
  auto local_bits = [&](int32_t c) -> uint32_t {
    if (inst)
      return (a.cell_tile[c] != tile) ? 0u
                                      : (uint32_t)(a.cell_pos[c] - tile * a.tile_cells + 1) << INFO_LOCAL_SHIFT;
    if (ws_kind != 2)
      return 0u;
    // internal patch of a group: mark the vertices of the cell whose rows belong to "the stress accumulated so
    // far" (se/solve_patch_weaksym.hpp:100-131): the two-cell members of the own group and every patch of an
    // EARLIER group (smaller id = treated before this one by se/reconstruction.hpp:170-234)
    uint32_t bits = 0u;
    const int32_t g = a.node_group[node];
    for (int v = 0; v < 3; ++v)
    {
      const int32_t nd = a.cell_nodes[3 * (int64_t)c + v];
      const int32_t gn = a.node_group[nd];
      if (nd != node && gn >= 0 && ((gn == g && a.node_ws[nd] == 1) || gn < g))
        bits |= 1u << (INFO_GROUPROW_SHIFT + v);
    }
    return bits;
  };
  const int n = a.node_cells_off[node + 1] - a.node_cells_off[node];
  const int nf = a.node_facets_off[node + 1] - a.node_facets_off[node];
  const int32_t* nfcts = a.node_facets + a.node_facets_off[node];
  const bool interior = (nf == n);
  const int64_t slot0 = inst ? (int64_t)a.inst_slot[tid_g] : (a.node_slot ? a.node_slot[node] : -1);
  const int64_t patch = inst ? tid_g : ((slot0 >= 0) ? a.node_patch[node] : -1);
  const bool ex = (a.ex_ncells != nullptr) && !inst;
  const int64_t exo = (int64_t)node * a.stride;

  if (ex)
  {
    a.ex_ncells[node] = n;
    for (int i = 0; i < a.stride; ++i)
    {
      a.ex_cells[exo + i] = -1;
      a.ex_fcts[exo + i] = -1;
      a.ex_il[exo + i] = -1;
      a.ex_fl[2 * exo + 2 * i] = -1;
      a.ex_fl[2 * exo + 2 * i + 1] = -1;
      a.ex_rev[2 * exo + 2 * i] = -1;
      a.ex_rev[2 * exo + 2 * i + 1] = -1;
    }
  }
  if (n < 2 || n + 2 > 65 || (ex && n + 2 > a.stride))
    return; // rejected on the host (EQLB_ERR_PATCH_TOO_SMALL / _TOO_LARGE)

  // --- start facet (Patch.cpp:425-484): interior -> first facet of the node; boundary ->
  // first flux-BC facet of RHS 0 if any, else the first primal-Dirichlet facet
  int32_t fct_first = nfcts[0];
  if (!interior)
  {
    int32_t f_ep = -1, f_ef = -1;
    for (int i = 0; i < nf; ++i)
    {
      const int8_t t = a.facet_type[nfcts[i]];
      if (t == EQLB_FACET_ESSNT_PRIMAL && f_ep < 0)
        f_ep = nfcts[i];
      else if (t == EQLB_FACET_ESSNT_DUAL && f_ef < 0)
        f_ef = nfcts[i];
    }
    fct_first = (f_ef >= 0) ? f_ef : f_ep;
  }

  int32_t cell, fct; // current cell T_a and facet E_a
  int fm_carry = 0, revm_carry = 0;
  int32_t fct0 = -1;
  if (interior)
  {
    fct = fct_first; // E_1
    cell = a.facet_cells[a.facet_cells_off[fct] + 1];
  }
  else
  {
    fct0 = fct_first; // E_0
    cell = a.facet_cells[a.facet_cells_off[fct0]];
    fm_carry = local_facet(a.cell_facets, cell, fct0);
    fct = next_facet(a, cell, fm_carry, node);
    if (ex)
    {
      a.ex_fcts[exo + 0] = fct0;
      a.ex_fl[2 * exo + 0] = (int8_t)fm_carry;
      a.ex_fl[2 * exo + 1] = (int8_t)fm_carry;
    }
  }

  uint32_t info_lane0 = 0;
  int32_t cell_first = cell;
  const int nloop = interior ? n : n - 1;
  for (int aa = 1; aa <= nloop; ++aa)
  {
    // E_a = fct between T_a = cell and T_{a+1}
    const int lf_a = local_facet(a.cell_facets, cell, fct);
    const int ln_a = local_node(a.cell_nodes, cell, node);
    const int32_t* fc = a.facet_cells + a.facet_cells_off[fct];
    const int32_t cell_ap1 = (fc[0] == cell) ? fc[1] : fc[0];
    const int lf_ap1 = local_facet(a.cell_facets, cell_ap1, fct);
    const int rev = a.facet_perm[3 * (int64_t)cell + lf_a] != a.facet_perm[3 * (int64_t)cell_ap1 + lf_ap1];

    uint32_t info = ((uint32_t)fm_carry << INFO_FM_SHIFT) | ((uint32_t)lf_a << INFO_FP_SHIFT)
                    | ((uint32_t)ln_a << INFO_LN_SHIFT) | (revm_carry ? INFO_REV_M : 0u)
                    | (rev ? INFO_REV_P : 0u) | local_bits(cell);
    if (aa == 1)
      info_lane0 = info;
    if (slot0 >= 0)
    {
      a.slot_cell[slot0 + aa - 1] = cell;
      a.slot_info[slot0 + aa - 1] = info;
    }
    if (ex)
    {
      a.ex_cells[exo + aa] = cell;
      a.ex_fcts[exo + aa] = fct;
      a.ex_il[exo + aa] = (int8_t)ln_a;
      a.ex_fl[2 * exo + 2 * aa] = (int8_t)lf_a;
      a.ex_fl[2 * exo + 2 * aa + 1] = (int8_t)lf_ap1;
      a.ex_rev[2 * exo + 2 * (aa - 1) + 1] = (int8_t)rev;                              // E_a of T_a
      a.ex_rev[2 * exo + 2 * ((interior && aa == n) ? 0 : aa)] = (int8_t)rev;          // E_a of T_{a+1}
    }
    fm_carry = lf_ap1;
    revm_carry = rev;
    fct = next_facet(a, cell_ap1, lf_ap1, node);
    cell = cell_ap1;
  }

  if (interior)
  {
    // close the ring: T_{n+1} == T_1, E_0 == E_n; lane 0 learns its minus facet now
    info_lane0 = (info_lane0 & ~(3u << INFO_FM_SHIFT) & ~INFO_REV_M)
                 | ((uint32_t)fm_carry << INFO_FM_SHIFT) | (revm_carry ? INFO_REV_M : 0u);
    if (slot0 >= 0)
      a.slot_info[slot0] = info_lane0;
    if (ex)
    {
      const int32_t cell_n = a.ex_cells[exo + n];
      a.ex_cells[exo + 0] = cell_n;
      a.ex_cells[exo + n + 1] = cell_first;
      a.ex_il[exo + 0] = a.ex_il[exo + n];
      a.ex_il[exo + n + 1] = a.ex_il[exo + 1];
      a.ex_fcts[exo + 0] = a.ex_fcts[exo + n];
      a.ex_fl[2 * exo + 0] = a.ex_fl[2 * exo + 2 * n];
      a.ex_fl[2 * exo + 1] = a.ex_fl[2 * exo + 2 * n + 1];
    }
  }
  else
  {
    // last cell T_n with the boundary facet E_n = fct
    const int lf_n = local_facet(a.cell_facets, cell, fct);
    const int ln_n = local_node(a.cell_nodes, cell, node);
    const uint32_t info = ((uint32_t)fm_carry << INFO_FM_SHIFT) | ((uint32_t)lf_n << INFO_FP_SHIFT)
                          | ((uint32_t)ln_n << INFO_LN_SHIFT) | (revm_carry ? INFO_REV_M : 0u)
                          | local_bits(cell);
    if (slot0 >= 0)
    {
      a.slot_cell[slot0 + n - 1] = cell;
      a.slot_info[slot0 + n - 1] = info;
    }
    if (ex)
    {
      a.ex_cells[exo + n] = cell;
      a.ex_fcts[exo + n] = fct;
      a.ex_il[exo + n] = (int8_t)ln_n;
      a.ex_fl[2 * exo + 2 * n] = (int8_t)lf_n;
      a.ex_fl[2 * exo + 2 * n + 1] = (int8_t)lf_n;
      a.ex_rev[2 * exo + 0] = 0;
      a.ex_rev[2 * exo + 2 * (n - 1) + 1] = 0;
    }
  }

  if (patch >= 0)
  {
    a.pn[patch] = (uint8_t)n;
    const int32_t fct_n = fct; // boundary: E_n
    for (int r = 0; r < a.nrhs; ++r)
    {
      uint8_t fl = interior ? PFLAG_INTERIOR : 0;
      if (r == 0 && ws_kind == 1)
        fl |= PFLAG_WS_SKIP;
      if (r == 0 && ws_kind == 2)
        fl |= PFLAG_WS_GROUP | (uint8_t)(a.node_wslevel[node] << PFLAG_WS_LEVEL_SHIFT);
      if (!interior)
      {
        const int8_t* ft = a.facet_type + (int64_t)r * a.nfacets;
        if (ft[fct0] == EQLB_FACET_ESSNT_DUAL)
          fl |= PFLAG_BC0;
        if (ft[fct_n] == EQLB_FACET_ESSNT_DUAL)
          fl |= PFLAG_BCN;
      }
      a.pflag[(int64_t)r * a.npatch_total + patch] = fl;
    }
  }
}

// cached affine maps: J = [x1-x0 | x2-x0] per cell (base/KernelData.cpp:66-90)
__global__ void __launch_bounds__(256)
k_cell_geometry(int32_t ncells, const double* x, const int32_t* cell_nodes, double* cellJ)
{
  const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells)
    return;
  const int32_t* cn = cell_nodes + 3 * (int64_t)c;
  const double *x0 = x + 3 * (int64_t)cn[0], *x1 = x + 3 * (int64_t)cn[1], *x2 = x + 3 * (int64_t)cn[2];
  double* J = cellJ + 4 * (int64_t)c;
  J[0] = x1[0] - x0[0];
  J[1] = x2[0] - x0[0];
  J[2] = x1[1] - x0[1];
  J[3] = x2[1] - x0[1];
}

void launch_build_patches(const BuildArgs& a, hipStream_t stream)
{
  const int block = 256;
  const int64_t n = a.inst_node ? a.ninst : (int64_t)a.nnodes;
  const int grid = (int)((n + block - 1) / block);
  if (grid > 0)
    hipLaunchKernelGGL(k_build_patches, dim3(grid), dim3(block), 0, stream, a);
}

void launch_cell_geometry(int32_t ncells, const double* x, const int32_t* cell_nodes,
                          double* cellJ, hipStream_t stream)
{
  const int block = 256;
  const int grid = (ncells + block - 1) / block;
  hipLaunchKernelGGL(k_cell_geometry, dim3(grid), dim3(block), 0, stream, ncells, x, cell_nodes, cellJ);
}

} // namespace eqlb
