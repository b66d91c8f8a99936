// Internal structures shared by the host API and the HIP kernels of libeqlb_amd.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/eqlb.h"

namespace eqlb
{

// Packed per-(patch, cell) descriptor, one uint32 per lane slot (lane-contiguous SoA):
//   bits 0-1 fm  local id of E_{a-1} on T_a      bits 2-3 fp  local id of E_a on T_a
//   bits 4-5 ln  local id of the patch node      bit 6 rev_m  E_{a-1} reversed w.r.t. T_{a-1}
//   bit 7 rev_p  E_a reversed w.r.t. T_{a+1}
constexpr uint32_t INFO_FM_SHIFT = 0, INFO_FP_SHIFT = 2, INFO_LN_SHIFT = 4;
constexpr uint32_t INFO_REV_M = 1u << 6, INFO_REV_P = 1u << 7;
//   bits 8-31 (tiled SoA only): 1 + position of the cell among the cells owned by the lane's
//   tile, 0 if the cell belongs to another tile (halo lane: computed, not accumulated)
constexpr uint32_t INFO_LOCAL_SHIFT = 8;

// per-(rhs, patch) flags
constexpr uint8_t PFLAG_INTERIOR = 1, PFLAG_BC0 = 2, PFLAG_BCN = 4;
// grouped boundary patches of the stress path (se/reconstruction.hpp:170-234), in the flags of RHS 0:
// WS_SKIP: two-cell patch of a group, no weak-symmetry step of its own; WS_GROUP: internal patch of a
// group, its weak-symmetry step works on own rows + rows of the group's two-cell patches
constexpr uint8_t PFLAG_WS_SKIP = 8, PFLAG_WS_GROUP = 16;
// level of the patch's group among overlapping groups (bits 5, 6 of the flag of RHS 0): the weak-symmetry kernel
// runs once per level, se/reconstruction.hpp:170-234 treats the groups one after the other
constexpr uint8_t PFLAG_WS_LEVEL_SHIFT = 5;
constexpr int WS_MAX_LEVELS = 4;
// slot_info bits 8-10 (plain SoA): local vertex v of the cell belongs to a two-cell patch of the
// lane's group -> its slot row is added to the stress coefficients (bit 8 + v)
constexpr uint32_t INFO_GROUPROW_SHIFT = 8;

constexpr int MAX_BINS = 5;          // lanes per patch P = 4, 8, 16, 32, 64
constexpr int BIN_P[MAX_BINS] = {4, 8, 16, 32, 64};

struct DeviceMesh
{
  int32_t nnodes = 0, ncells = 0, nfacets = 0, ncells_max = 0;
  // device arrays
  double* x = nullptr;            // [nnodes][3]
  double* cellJ = nullptr;        // [ncells][4] J00 J01 J10 J11 (dx_i/dX_j), cached affine maps
  int32_t *cell_nodes = nullptr, *cell_facets = nullptr, *facet_nodes = nullptr;
  int32_t *facet_cells_off = nullptr, *facet_cells = nullptr;
  int32_t *node_cells_off = nullptr, *node_facets_off = nullptr, *node_facets = nullptr;
  int32_t* node_cells = nullptr;  // uploaded on first use (node-wise gathers of the estimator step)
  uint8_t* facet_perm = nullptr;
  // host copies needed for binning / tiling
  std::vector<int32_t> h_node_ncells, h_node_nfcts, h_cell_nodes;
  std::vector<int32_t> h_facet_nodes, h_node_facets_off, h_node_facets, h_node_cells_off, h_node_cells;
  std::vector<double> h_x;
};

struct Bin
{
  int P = 0;
  int64_t npatch = 0;
  int64_t slot_offset = 0;   // into slot arrays
  int64_t patch_offset = 0;  // into patch arrays
  int64_t nfull = 0;         // fused stress tiles: the leading patches of the bin are the FULL ones (interior, as many
                             // cells as lanes) that the fused kernel takes; the slot path takes [nfull, npatch)
};

// kernel arguments of the patch kernel (one launch per bin)
struct SeArgs
{
  const double* cellJ;
  const int32_t* slot_cell;   // [nslots] global cell id or -1
  const uint32_t* slot_info;  // [nslots]
  const uint8_t* pn;          // [npatch] cells per patch
  const uint8_t* pflag;       // [nrhs][npatch_total]
  const double* tables;       // S | F | H | D
  const double* flux_dg;      // [nrhs][ncells*ND*2]
  const double* rhs_dg;       // [nrhs][ncells*ND]
  const double* bvals;        // [nrhs][ncells*NRT] global flux-boundary DOFs or nullptr (homogeneous)
  double* out;                // slots [nrhs][ncells][3][NRT] or flux_hdiv [nrhs][ncells*NRT]
  int32_t* status;            // device error flag
  int64_t npatch;             // patches of this bin
  int64_t slot_offset, patch_offset, npatch_total;
  int32_t ncells, nrhs;
  int32_t rhs;                // index of the right-hand side handled by this launch (flags, bvals)
  int32_t rhs_in, rhs_out;    // block index of that right-hand side inside flux_dg / rhs_dg and inside out
                              // (0 when the pointers already address the block: lists of separate arrays)
  int32_t ws_level = 0;       // weak-symmetry kernels: the pass handles the patches of this group level
};

// bins of a fused launch: blocks [block_start[b], block_start[b+1]) of 256 threads handle bin b
struct FusedBins
{
  int64_t block_start[MAX_BINS + 1];
  int64_t npatch[MAX_BINS], slot_offset[MAX_BINS], patch_offset[MAX_BINS];
};

// Tiled launch (EQLB_SCATTER_TILED): a workgroup owns TC cells (a leaf of the recursive coordinate
// bisection of the cell centroids), solves every patch that touches one of them and accumulates the
// three vertex contributions of its cells in LDS, so neither the slot buffer nor the reduction pass
// exist.
struct TileDesc
{
  int32_t slot_start[MAX_BINS]; // first lane slot of the tile's patches of bin b (multiple of 64)
  int32_t patch_start[MAX_BINS];
  int32_t npatch[MAX_BINS];
  int32_t nfull[MAX_BINS]; // leading patches of the bin that are interior with exactly P cells
  int32_t nint[MAX_BINS];  // leading patches that are interior (no boundary facet), nfull of them full
  // behind the full ones the interior patches are ordered by their number of cells: nval[b][j] = end (in patches of the
  // bin) of those with P - 1 - j cells, j = 0, 1, 2; the other interior patches follow up to nint (bins 0, 1 only)
  int32_t nval[2][3];
  int32_t zero;            // 1: a vertex of an owned cell is not equilibrated here (node mask): its row
                           // is never written, the LDS slots of the tile are zeroed first
};

struct TileArgs
{
  const TileDesc* tiles;
  const int32_t* tile_cells; // [ntiles][tc] owned cells (-1: padding)
  int32_t ntiles, tc;
  // EV flush to the conforming DOFs (nullptr: broken layout)
  const int32_t* facet_owner; // [ntiles][tc][3] 2 * facet + reversal bit, or -1
  const int32_t* cell_dofs;   // caller's dofmap or nullptr (default numbering)
  int64_t ndofs;
  int32_t nfacets;
  int32_t tile_first; // this launch handles the tiles [tile_first, tile_first + ntiles)
  int32_t accumulate; // 1: flux_hdiv += result (reference semantics), 0: flux_hdiv = result (no read of the old values)
  // EV: change of basis of the conforming output (eqlb_ev_set_basis_transform) or nullptr (hierarchic RT_k):
  // target cell DOFs = basis_C x broken hierarchic cell DOFs; basis_R: facet block of a reversed facet
  const double* basis_C;
  const double* basis_R;
};

// right-hand sides of a multi-RHS tiled launch (k_se_patch_tiled_multi)
constexpr int MULTI_RHS_MAX = 8;
struct MultiRhs
{
  int32_t n, rhs0; // right-hand sides rhs0 .. rhs0 + n - 1 of the handle (boundary flags, boundary values)
  const double* g[MULTI_RHS_MAX];
  const double* f[MULTI_RHS_MAX];
  double* x[MULTI_RHS_MAX];
};

struct BuildArgs
{
  int32_t nnodes, nfacets, nrhs;
  const int32_t *cell_nodes, *cell_facets, *facet_nodes, *facet_cells_off, *facet_cells;
  const int32_t *node_cells_off, *node_facets_off, *node_facets;
  const uint8_t* facet_perm;
  const int8_t* facet_type;  // [nrhs][nfacets]
  const int64_t* node_slot;  // first lane slot of the node's patch, -1: not equilibrated
  const int64_t* node_patch; // patch index
  int64_t npatch_total;
  int32_t* slot_cell;
  uint32_t* slot_info;
  uint8_t* pn;
  uint8_t* pflag;
  // grouped stress patches (nullptr: none): per node 0 / 1 (two-cell member) / 2 (internal patch of
  // a group) and the group id
  const int8_t* node_ws;
  const int32_t* node_group;
  const int8_t* node_wslevel;
  // instance mode (tiled SoA): thread i builds the patch of node inst_node[i] at slot inst_slot[i]
  // as patch i of tile inst_tile[i]; nullptr: one patch per node (node_slot / node_patch)
  int64_t ninst;
  const int32_t* inst_node;
  const int32_t* inst_slot;
  const int32_t* inst_tile;
  const int32_t* cell_tile; // [ncells] owning tile
  const int32_t* cell_pos;  // [ncells] position in the tile-sorted cell list
  int32_t tile_cells;       // cells per tile
  // optional export in OrientedPatch layout (nullptr: off)
  int32_t stride;
  int32_t *ex_ncells, *ex_cells, *ex_fcts;
  int8_t *ex_fl, *ex_il, *ex_rev;
};

void launch_build_patches(const BuildArgs& a, hipStream_t stream);
void launch_cell_geometry(int32_t ncells, const double* x, const int32_t* cell_nodes,
                          double* cellJ, hipStream_t stream);
// returns 0 or EQLB_ERR_UNSUPPORTED
int launch_se_patch(int k, int deg, int P, int solver, int scatter, const SeArgs& a,
                    hipStream_t stream, int mode = 0);
int launch_se_patch_fused(int k, int deg, int scatter, const SeArgs& a, const FusedBins& fb,
                          hipStream_t stream);
int launch_se_patch_tiled(int k, int deg, int mode, const SeArgs& a, const TileArgs& t, hipStream_t stream);
int launch_se_patch_tiled_multi(int k, int deg, int mode, const SeArgs& a, const TileArgs& t, const MultiRhs& mr,
                                hipStream_t stream);
void launch_tile_facet_owner(const DeviceMesh& m, int64_t n, const int32_t* tile_cells, int32_t* code,
                             hipStream_t stream);
int tile_cells_of(int k);
int tile_cells_ev_of(int k);
int tile_cells_max_of(int k);
int launch_se_weaksym(int k, int P, bool no_flux_bcs, const SeArgs& a, hipStream_t stream);
// fused stress launch (RT_2, no stress flux BCs, patches of up to 8 facets): rows 0, 1 + weak symmetry
// mixed: tile lists with every patch of up to 8 lanes (full ones first; generic instance of the body for the others),
// else lists of full patches only
int launch_se_stress_tiled(const SeArgs& a, const TileArgs& t, const double* const* g, const double* const* f,
                           double* const* x, hipStream_t stream, bool mixed = false);
int stress_tile_cells();
int launch_ev_patch_fused(int k, const SeArgs& a, const FusedBins& fb, hipStream_t stream);
// conforming <-> broken layout of the EV equilibrator (eqlb_ev.hip); cell_dofs may be nullptr
// (default numbering: facet*k + j, then nfacets*k + cell*(k^2-k) + i)
// facet_maps: nullptr (hierarchic RT_k: -I / B) or [3][2][k][k]: broken facet DOFs = map[lf][reversed] x conforming ones
void launch_ev_boundary_to_broken(const DeviceMesh& m, int k, int nrhs, const int32_t* cell_dofs,
                                  int64_t ndofs, const double* bv_conf, double* bv_broken,
                                  const double* facet_maps, hipStream_t stream);
void launch_ev_reduce(const DeviceMesh& m, int k, int nrhs, const int32_t* cell_dofs, int64_t ndofs,
                      const double* slots, double* x, int accumulate, const double* basis_C, const double* basis_R,
                      hipStream_t stream);
int launch_reduce_slots(int nrt, int32_t ncells, int32_t nrhs, const double* slots, double* x, int accumulate,
                        hipStream_t stream);
int launch_reduce_slots_cells(int nrt, int32_t ncells, int64_t nlist, const int32_t* cells, const double* slots,
                              double* x, hipStream_t stream);
int projection_matrix_host(int degree, int nq, const double* pts, const double* wts,
                           std::vector<double>& Pm);
void launch_project_dg(int64_t ncells, int nd, int nq, int bs, const double* Pm, const double* qv,
                       double* out, hipStream_t stream);
void launch_korn(const DeviceMesh& m, const int64_t* node_slot, const int64_t* node_patch,
                 const int32_t* slot_cell, const uint32_t* slot_info, const uint8_t* pn,
                 const uint8_t* pflag, double* cks, double* korn, hipStream_t stream);
int launch_estimate(const DeviceMesh& m, int k, int nrhs, const double* x_eq, const double* flux_dg,
                    const double* rhs_dg, double* div2, double* sig2, double* jump, double alpha,
                    double beta, hipStream_t stream);
int launch_estimate_stress(const DeviceMesh& m, const int32_t* node_cells, int k, const double* x0,
                           const double* x1, const double* korn, double pi_1, double* energy, double* wsym,
                           double* node_asym, hipStream_t stream);
int launch_oscillation(const DeviceMesh& m, int k, int nrhs, const double* x_eq, const double* flux_dg, int nq,
                       const double* qpoints, const double* qweights, const double* fvalues, const double* korn,
                       double* out, hipStream_t stream);
int set_error(int code, const char* fmt, ...); // thread-local message of eqlb_last_error + the code back
void launch_halo_pack(int nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells, double* x,
                      double* buf, int clear, hipStream_t stream);
void launch_halo_unpack_add(int nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells,
                            double* x, const double* buf, hipStream_t stream);
// tiles by recursive coordinate bisection on the device (eqlb_tiling_device.hip): 0 ok, 1 stretched mesh (host
// bisection), < 0 device error
void device_tiling_prepare();
int device_tile_order(const DeviceMesh& m, int tc, int32_t ntiles, const double blo[2], const double bhi[2], double inv,
                      std::vector<int32_t>& order);
size_t table_doubles(int k, int deg);
int fill_tables_host(int k, int deg, std::vector<double>& out);

} // namespace eqlb

struct eqlb_mesh
{
  eqlb::DeviceMesh m;
  // cells in the order of the tile bisection (sorted by cell id inside every tile), per tile size: the tiling
  // depends on the mesh alone, so further handles on the mesh (SE + EV, a stress handle ...) and further
  // eqlb_se_set_boundary calls reuse it
  std::mutex tiling_mutex;
  std::map<int, std::vector<int32_t>> tiling_order;
};

struct eqlb_ev
{
  struct eqlb_se* se = nullptr; // patch topology, tables, slots, timing of the shared machinery
};

struct eqlb_se
{
  eqlb_mesh* mesh = nullptr;
  int k = 0, deg = 0, nrhs = 0, stress = 0;
  int nrt = 0, nd = 0;
  int solver = EQLB_SOLVER_SHUFFLE, scatter = EQLB_SCATTER_AUTO, timing = 0, fused = 1;
  int accumulate = 1;               // option "accumulate": 0 stores the result instead of adding it
  int multi_rhs = 1;                // option "multi_rhs": all right-hand sides of a tiled call in one launch
  int scatter_last = EQLB_SCATTER_SLOTS; // scatter mode the last equilibrate call resolved to
  int mode = 0;                     // 1: constrained-minimisation (EV) patch problems
  int ev_output = 0;                // EV: 0 conforming DOFs, 1 broken hierarchic RT_k layout
  int tile_cells_user = 0;          // option "tile_cells": cells per tile of the tiled launch (0 = automatic)
  int ev_bv_hier = 0;               // EV: boundary values in the hierarchic basis although a basis transform is set
  int32_t* ev_cell_dofs = nullptr;  // EV: device copy of the caller's dofmap or nullptr (default)
  int64_t ev_ndofs = 0;             // EV: number of conforming flux DOFs
  double* ev_basis = nullptr;       // EV: device copy of [C (nrt x nrt) | R (k x k) | facet maps 3 x 2 x k x k] or nullptr
  bool ev_basis_has_R = false;
  bool stress_flux_bcs = true;       // some facet of stress row 0 / 1 carries a flux BC
  bool boundary_set = false;
  int64_t npatch_total = 0, nslots = 0;
  eqlb::Bin bins[eqlb::MAX_BINS];
  // device
  double* tables = nullptr;
  int8_t* facet_type = nullptr;     // [nrhs][nfacets]
  int8_t* node_ws = nullptr;        // grouped stress patches (stress && k == 2 && groups exist)
  int32_t* node_group = nullptr;
  int8_t* node_wslevel = nullptr;   // level of the node's group among overlapping groups
  int ws_levels = 1;                // passes of the weak-symmetry kernel
  double* bvals = nullptr;          // [nrhs][ncells*nrt] global boundary DOFs (nullptr: homogeneous)
  int64_t* node_slot = nullptr;     // [nnodes] first slot of the node's patch or -1
  int64_t* node_patch = nullptr;    // [nnodes] patch index or -1
  int32_t* node_P = nullptr;        // [nnodes] lanes per patch
  int32_t* slot_cell = nullptr;
  uint32_t* slot_info = nullptr;
  uint8_t* pn = nullptr;
  uint8_t* pflag = nullptr;
  // tiled SoA (plain SE, EQLB_SCATTER_TILED)
  int32_t ntiles = 0, tile_tc = 0;
  bool t_stress = false;            // the tiles serve the fused stress launch (bins P <= 8 only)
  int64_t t_rest = 0;               // patches left to the generic kernels when t_stress (everything but full patches)
  hipStream_t side_stream = nullptr; // the rest's patch kernels run here, next to the fused kernel
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int32_t* rest_cells = nullptr;    // cells with a vertex whose patch runs on the generic kernels (compact reduction)
  int64_t nrest_cells = 0;
  int64_t t_nslots = 0, t_npatch = 0;
  eqlb::TileDesc* t_tiles = nullptr;
  int32_t *t_tile_cells = nullptr, *t_slot_cell = nullptr, *t_facet_owner = nullptr;
  uint32_t* t_slot_info = nullptr;
  uint8_t *t_pn = nullptr, *t_pflag = nullptr;
  // fused stress launch: the tiles list EVERY patch of the bins 0, 1 (full ones first), not the full ones only - where
  // the others are more than a few per cent of the patches (unstructured meshes); kernel with both instances
  bool t_mixed = false;
  // two-phase sweeps (multi-GPU overlap): tiles owning a priority cell are numbered first
  std::vector<int32_t> prio_cells;
  int32_t t_nprio = 0;              // number of priority tiles
  int32_t tile_first = 0, tile_count = -1; // options "tile_first" / "tile_count" (-1: to the end)
  double* slots = nullptr;          // [nrhs][ncells][3][nrt]
  int slots_first_bin = 0;          // the slot rows of the bins >= this one hold values of the last slot-path run
  int32_t* status = nullptr;
  // staging for host-memory calls
  double *d_flux_dg = nullptr, *d_rhs_dg = nullptr, *d_flux_hdiv = nullptr;
  double *d_cks = nullptr, *d_korn = nullptr; // Korn estimate: per node / staging per cell
  // timing ("timing" option): ring of event sets, one set per equilibrate call
  static constexpr int EV_RING = 64, EV_PER_SET = 2 * eqlb::MAX_BINS + 4;
  hipEvent_t* ev = nullptr; // [EV_RING][EV_PER_SET]: bin b start/end at 2b, 2b+1; reduce start/end; weak symmetry start/end
  int64_t ev_calls = 0;     // calls recorded since timing was (re)enabled
};
