/*
 * eqlb.h - C ABI of the MI355X-native patch-local flux equilibrator (libeqlb_amd.so).
 *
 * Drop-in boundary for the patch-wise equilibration hot path of dolfinx_eqlb v1.2.0.
 * The reference exposes this path through pybind11 on DOLFINx objects
 * (python/dolfinx_eqlb/wrappers.cpp:52-137: `local_solver_*`, `reconstruct_fluxes_minimisation`,
 * `reconstruct_fluxes_semiexplt[_with_kornconst]`; drivers cpp/dolfinx_eqlb/se/reconstruction.hpp:
 * 337-407, ev/reconstruction.hpp:32-176, base/local_solver.hpp:38-187); here the same calls take the
 * flat arrays those objects hold.  Every entry point cites the reference interface it replaces.
 * INTEGRATION.md shows the pybind11/DOLFINx-side adapter a maintainer would add.
 *
 * Conventions: all floating point is fp64, indices int32, flags int8/uint8.  Functions return
 * 0 on success and a negative EQLB_ERR_* code otherwise (the reference throws
 * std::runtime_error -> Python RuntimeError); eqlb_last_error() gives the message of the last
 * failure on the calling thread.  One handle per host thread / HIP stream; not re-entrant
 * (like the reference, se/reconstruction.hpp:275-283 shared scratch).
 */
#ifndef EQLB_H
#define EQLB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQLB_OK 0
#define EQLB_ERR_INVALID_ARGUMENT (-1) /* size / degree mismatch, se/reconstruction.hpp:358-388 */
#define EQLB_ERR_PATCH_TOO_SMALL (-2)  /* patch with one cell, se/Patch.cpp:353-359 */
#define EQLB_ERR_UNSUPPORTED (-3)      /* configuration outside this build */
#define EQLB_ERR_DEVICE (-4)           /* HIP runtime failure / no device */
#define EQLB_ERR_PATCH_TOO_LARGE (-5)  /* patch with more than 63 cells (one wavefront per patch) */
#define EQLB_ERR_SINGULAR (-6)         /* patch system not positive definite (incompatible data) */
#define EQLB_ERR_NO_MEMORY (-7)        /* host allocation failed during set-up */

/* memory space of the data pointers handed to eqlb_se_equilibrate */
#define EQLB_MEM_HOST 0
#define EQLB_MEM_DEVICE 1

/* facet types = base::PatchFacetType, cpp/dolfinx_eqlb/base/Patch.hpp:22-27 */
#define EQLB_FACET_INTERNAL 0
#define EQLB_FACET_ESSNT_PRIMAL 1
#define EQLB_FACET_ESSNT_DUAL 2

/* variants of the patch kernel (eqlb_se_set_option, key "solver" / "scatter") */
#define EQLB_SOLVER_LDS_CHOLESKY 0 /* dense Cholesky of the patch tile in LDS */
#define EQLB_SOLVER_SHUFFLE 1      /* block-tridiagonal elimination in registers, wave shuffles */
#define EQLB_SCATTER_SLOTS 0       /* per-(cell, vertex) slots + deterministic reduction */
#define EQLB_SCATTER_ATOMIC 1      /* fp64 global atomic add into the RT coefficient vector */
#define EQLB_SCATTER_AUTO (-1)     /* default: TILED where it applies (k <= 3 with the shuffle solver; stress:
                                      RT_2 without flux BCs on the stress rows), else SLOTS */
#define EQLB_SCATTER_TILED 2       /* one workgroup per tile of cells: vertex contributions summed in
                                      LDS in fixed order, no slot buffer (plain flux equilibration) */

typedef struct eqlb_mesh eqlb_mesh_t;
typedef struct eqlb_se eqlb_se_t;

/* Message of the last error on this thread ("" if none). */
const char* eqlb_last_error(void);

/* Number of HIP devices visible (0 if none / runtime unavailable). */
int eqlb_device_count(void);

/*
 * Mesh topology/geometry, copied to the current HIP device.  Replaces what the reference reads
 * from dolfinx::mesh::Mesh after FluxEquilibrator.initialise_mesh_info
 * (python/dolfinx_eqlb/eqlb/FluxEquilibrator.py:52-67; se/Patch.cpp:20-26 connectivities,
 * se/reconstruction.hpp:83-84 facet permutations):
 *   x            [nnodes][3]   geometry().x()
 *   cell_nodes   [ncells][3]   topology 2->0 (== geometry dofmap for affine P1 meshes)
 *   cell_facets  [ncells][3]   topology 2->1, local facet f opposite local vertex f
 *   facet_nodes  [nfacets][2]  topology 1->0
 *   facet_cells  CSR           topology 1->2
 *   node_cells   CSR           topology 0->2
 *   node_facets  CSR           topology 0->1
 *   facet_perm   [ncells][3]   get_facet_permutations(): reflection bit of each cell facet
 * All pointers are host pointers; nothing is retained.
 */
int eqlb_mesh_create(int32_t nnodes, int32_t ncells, int32_t nfacets, const double* x,
                     const int32_t* cell_nodes, const int32_t* cell_facets,
                     const int32_t* facet_nodes, const int32_t* facet_cells_offsets,
                     const int32_t* facet_cells, const int32_t* node_cells_offsets,
                     const int32_t* node_cells, const int32_t* node_facets_offsets,
                     const int32_t* node_facets, const uint8_t* facet_perm,
                     eqlb_mesh_t** mesh);
void eqlb_mesh_destroy(eqlb_mesh_t* mesh);

/*
 * Semi-explicit equilibrator for RT_k fluxes with projected flux / RHS in DG_{degree_dg}
 * (degree_dg <= k-1; the reference requires deg(flux_dg) == deg(rhs_dg) <= k-1,
 * se/reconstruction.hpp:363-373) and nrhs simultaneously equilibrated fluxes.
 * Replaces the per-call setup of se::reconstruction<T,k> (se/reconstruction.hpp:62-163:
 * KernelData tabulation, kernel generation, Patch/PatchData allocation) - done once here and
 * cached on the device.  reconstruct_stress / korn are the flags of
 * reconstruct_fluxes_semiexplt[_with_kornconst] (wrappers.cpp:97-137).  reconstruct_stress != 0:
 * the first two RHS are the rows of a stress tensor and the weak symmetry condition is imposed
 * patch-wise after the row-wise equilibration (se/solve_patch_weaksym.hpp:59-233), including the
 * grouped boundary patches for RT_2 with flux BCs on the stress (se/reconstruction.hpp:170-234;
 * groups that overlap are treated in the reference's node order: one pass of the weak-symmetry kernel per
 * level of the conflict graph, at most 4 levels).  k <= 4 (k = 4 is the upper end of the reference's test
 * range: register solver with the three interior unknowns of a cell condensed, every lanes-per-patch bin, slot
 * path; the weak-symmetry step and the EV patch problems at k = 4 run on the dense LDS solver, patches of up to 8
 * facets).  estimate_korn is accepted for symmetry with the reference constructor (the estimate itself is
 * requested per call, see below).
 */
int eqlb_se_create(eqlb_mesh_t* mesh, int32_t k, int32_t degree_dg, int32_t nrhs,
                   int32_t reconstruct_stress, int32_t estimate_korn, eqlb_se_t** handle);
void eqlb_se_destroy(eqlb_se_t* handle);

/* Integer options: "solver" (EQLB_SOLVER_*; default SHUFFLE; LDS_CHOLESKY for the EV problems at k = 4), "scatter"
 * (EQLB_SCATTER_*; default AUTO), "fused" (1: all patch-size bins of the slot path in one launch,
 * default), "timing" (1: record HIP events around the kernels, see eqlb_se_last_kernel_ms),
 * "tile_first" / "tile_count" (range of tiles swept by the next tiled launches, default 0 / -1 = all;
 * see eqlb_se_set_priority_cells), "accumulate" (1, default: flux_hdiv += result as the reference does,
 * se/solve_patch_semiexplt.hpp:1157-1160, which assumes a zero-initialised output; 0: flux_hdiv = result,
 * the old values are neither read nor uploaded - every DOF of every cell is written; not with the
 * atomic scatter), "tile_cells" (cells per tile of the tiled launch, 0 = automatic; capped by the LDS of a
 * workgroup; applies to the next eqlb_se_set_boundary - a tuning knob), "multi_rhs" (1, default: the tiled
 * launch sweeps all right-hand sides of a call - the reference loops them inside the patch,
 * se/solve_patch_semiexplt.hpp:1040-1075; 0: one launch per right-hand side). */
int eqlb_se_set_option(eqlb_se_t* handle, const char* key, int32_t value);

/*
 * Boundary information = the tables base::BoundaryData hands to the patch loop
 * (base/BoundaryData.hpp: facet_type(), boundary_values(); built by
 * base/BoundaryData.cpp:279-633 from the FluxBC lists):
 *   facet_type       [nrhs][nfacets] int8, EQLB_FACET_*
 *   boundary_values  [nrhs][ncells*k(k+2)] GLOBAL boundary DOFs of the flux (the boundary
 *                    functions BoundaryData fills from the FluxBC lists: facet DOFs
 *                    int (detJ K g).N_f s^j on the flux-BC facets, zero elsewhere), or NULL for
 *                    homogeneous flux BCs.  The per-patch values hat_a * g of
 *                    BoundaryData::calculate_patch_bc (base/BoundaryData.cpp:687-745) are formed
 *                    in the kernel.  With stress equilibration the rows carry the tractions; the
 *                    weak-symmetry corrections have zero normal flux on those facets.
 *   node_mask        [nnodes] uint8 or NULL: equilibrate only patches of nodes with mask != 0
 *                    (node ownership of a partitioned run; the reference loops
 *                    index_map(0)->size_local() owned nodes, se/reconstruction.hpp:90,286).
 * Builds the oriented patch fans (OrientedPatch::initialize_patch, se/Patch.cpp:406-635, and
 * the reversal flags of se/solve_patch_semiexplt.hpp:324-389) with a HIP kernel into
 * lane-contiguous SoA buffers, binned by patch size.  Host pointers.
 */
int eqlb_se_set_boundary(eqlb_se_t* handle, const int8_t* facet_type,
                         const double* boundary_values, const uint8_t* node_mask);

/*
 * The hot path: se::reconstruction<T,k> node loop (se/reconstruction.hpp:286-313) =
 * for every patch: explicit step, patch assembly, small dense factorise/solve, back-map and
 * scatter (se/solve_patch_semiexplt.hpp:212-1163).  Replaces the body of
 * reconstruct_fluxes_semiexplt (wrappers.cpp:97-115).
 *   flux_dg    [nrhs][ncells*nd*2]   projected fluxes, DG_{degree_dg}^2 blocked (x,y per node),
 *                                    = flux_dg[i]->x()->array()  (solve_patch_semiexplt.hpp:456)
 *   rhs_dg     [nrhs][ncells*nd]     projected right-hand sides   (:462)
 *   flux_hdiv  [nrhs][ncells*k(k+2)] equilibrated correctors in the discontinuous hierarchic
 *                                    RT_k space, global DOF = cell*k(k+2)+local
 *                                    (se/Patch.hpp:480); ACCUMULATED (+=) like the reference
 *                                    (solve_patch_semiexplt.hpp:1157-1160)
 *   memspace   EQLB_MEM_HOST: pointers are host memory (copied in and out, synchronous);
 *              EQLB_MEM_DEVICE: device pointers, work is enqueued on `stream` (hipStream_t,
 *              NULL = default stream) and the call returns without synchronising.
 */
int eqlb_se_equilibrate(eqlb_se_t* handle, const double* flux_dg, const double* rhs_dg,
                        double* flux_hdiv, int32_t memspace, void* stream);

/* The same call on one array per right-hand side - what the reference's binding receives: lists of
 * dolfinx Functions (wrappers.cpp:97-115: flux_hdiv, flux_dg, rhs_dg), each with its own vector.
 * flux_dg[r], rhs_dg[r], flux_hdiv[r] (r < nrhs) are the blocks of eqlb_se_equilibrate; all in the
 * same memory space. */
int eqlb_se_equilibrate_lists(eqlb_se_t* handle, const double* const* flux_dg, const double* const* rhs_dg,
                              double* const* flux_hdiv, int32_t memspace, void* stream);

/*
 * Same as eqlb_se_equilibrate plus the upper bounds of the cells' squared Korn constants:
 * reconstruct_fluxes_semiexplt_with_kornconst (wrappers.cpp:117-137) =
 * se/reconstruction.hpp:291-304 with OrientedPatch::estimate_squared_korn_constant
 * (se/Patch.cpp:130-334).  cells_kornconst [ncells] is ACCUMULATED: every patch adds
 * (gdim+1) c_K^2 to its cells; the Python caller takes the square root (FluxEqlbSE.py:165).
 */
int eqlb_se_equilibrate_with_kornconst(eqlb_se_t* handle, const double* flux_dg,
                                       const double* rhs_dg, double* flux_hdiv,
                                       double* cells_kornconst, int32_t memspace, void* stream);

/* The Korn part of that call alone (cells_kornconst [ncells] += (gdim+1) c_K^2 per patch cell), for
 * callers that equilibrate through eqlb_se_equilibrate_lists. */
int eqlb_se_kornconst(eqlb_se_t* handle, double* cells_kornconst, int32_t memspace, void* stream);

/* Number of patches equilibrated per call (nodes selected by node_mask). */
int64_t eqlb_se_num_patches(const eqlb_se_t* handle);

/*
 * Test/diagnostic export of the device-built patch fans in the layout of
 * OrientedPatch (_cells, _fcts, _fcts_local, _inodes_local; se/Patch.hpp:371-376), one row of
 * `stride` (>= max cells per patch + 2) entries per mesh node, unused entries -1:
 *   ncells [nnodes], cells [nnodes][stride], fcts [nnodes][stride],
 *   fcts_local [nnodes][2*stride], inodes_local [nnodes][stride], reversed [nnodes][2*stride]
 *   ([2a], [2a+1] = E_{a-1} / E_a of cell T_a reversed, 0-based cell a).  Host pointers.
 */
int eqlb_se_export_patches(eqlb_se_t* handle, int32_t stride, int32_t* ncells, int32_t* cells,
                           int32_t* fcts, int8_t* fcts_local, int8_t* inodes_local,
                           int8_t* reversed);

/*
 * Cell-local L2 projection into DG_degree (scalar: bs = 1, blocked vector: bs = 2, ...), the loop of
 * base::local_solver_cholesky (cpp/dolfinx_eqlb/base/local_solver.hpp:38-187,214-224) as used by
 * local_projection (python/dolfinx_eqlb/lsolver/projection.py:17-77; forms a = (u,v), l_i = (f_i,v)).
 * The reference evaluates f_i inside JIT-compiled FFCx kernels; here the caller supplies its point
 * values at the images of a reference-cell quadrature rule of its choice (exact for
 * deg(f) + degree on affine cells):
 *   qpoints [nq][2], qweights [nq]   rule on the reference triangle (weights sum to 1/2), host
 *   qvalues [nrhs][ncells][nq][bs]   f_i at x_c(qpoints)
 *   out     [nrhs][ncells][nd][bs]   DOFs (= x[bs*dof + cb], cell-major DG numbering); OVERWRITTEN
 *                                    like the reference (:163-182), nd = (degree+1)(degree+2)/2
 * degree <= 3, nq <= 64.  memspace / stream as in eqlb_se_equilibrate (qvalues and out).
 */
int eqlb_project_dg(eqlb_mesh_t* mesh, int32_t degree, int32_t bs, int32_t nrhs, int32_t nq,
                    const double* qpoints, const double* qweights, const double* qvalues,
                    double* out, int32_t memspace, void* stream);

/* Largest number of cells of a patch of the mesh (OrientedPatch::ncells_max). */
int32_t eqlb_mesh_max_patch_cells(const eqlb_mesh_t* mesh);

/*
 * Constant reference-cell tensors compiled into the library (tools/gen_tables.py), for tests:
 * name in {"S","F","H","D"}; returns the number of doubles copied (<= capacity), or a
 * negative error.
 */
int eqlb_get_reference_table(int32_t k, int32_t degree_dg, const char* name, double* out,
                             int32_t capacity);

/* Two-phase sweeps of the tiled launch (multi-GPU: the reference has no distributed equilibration,
 * SURVEY 8e).  Cells listed here before eqlb_se_set_boundary - the ghost cells whose rows a
 * neighbour rank waits for - make their tiles the FIRST tiles; eqlb_se_num_priority_tiles returns how
 * many there are.  With the options "tile_first" / "tile_count" (eqlb_se_set_option; count -1 = to the
 * end) an equilibrate call sweeps a range of tiles only: first the priority tiles, then - while the
 * halo exchange of their rows is in flight - the rest.  Tiled scatter only.  Stress equilibration: the patches the
 * fused stress kernel does not take (boundary patches, patches that are not full) are equilibrated by the call whose
 * range starts at tile 0, so that the ghost rows are complete behind the first range (option "accumulate" = 1, the
 * default; with accumulate = 0 the tiled launches store and those patches follow the last range). */
int eqlb_se_set_priority_cells(eqlb_se_t* handle, const int32_t* cells, int32_t n);
int32_t eqlb_se_num_priority_tiles(const eqlb_se_t* handle);
/* eqlb_se_equilibrate on device memory for the tiles [tile_first, tile_first + tile_count) only
 * (count -1 = to the end), without touching the "tile_first" / "tile_count" options */
int eqlb_se_equilibrate_tiles(eqlb_se_t* handle, const double* flux_dg, const double* rhs_dg,
                              double* flux_hdiv, int32_t tile_first, int32_t tile_count, void* stream);

/* Device-memory calls (EQLB_MEM_DEVICE) return without synchronising, so a patch system that is not
 * positive definite (degenerate cell geometry; the matrix does not depend on the data) cannot be
 * reported by the call itself: the kernels raise a flag on the device.  eqlb_se_check_status waits for `stream`, reads and clears
 * the flag: EQLB_OK or EQLB_ERR_SINGULAR (the reference has no such check: Eigen's LLT / LU results
 * are used unchecked, se/PatchData.hpp:576-663).  Host-memory calls check it themselves. */
int eqlb_se_check_status(eqlb_se_t* handle, void* stream);

/* With option "timing" = 1 every equilibrate call records HIP events on the launch stream around
 * each kernel (ring of the last 64 calls).  Returns the average device time in ms per launch of
 * kernel `which` over the recorded calls: which = b in 0..4: patch kernel of the bin with
 * P = 4 << b lanes per patch (single-launch paths - tiled and fused - report in slot 0);
 * which = 5: slot-reduction kernel (0 on the tiled path); which = 6: the weak-symmetry kernels of a
 * stress equilibration (all bins together).  Synchronises with the events;
 * 0 if nothing was recorded.  Setting the option again resets the ring. */
double eqlb_se_last_kernel_ms(const eqlb_se_t* handle, int32_t which);

/* Acceptance predicates and estimator quantities of a semi-explicit result, on the device - the
 * step after the equilibration in the reference's workflows:
 *   cell_div2  [nrhs][ncells]  || Pi f - div(sigma_eq + G) ||^2_L2(T)   (divergence condition,
 *                              python/dolfinx_eqlb/eqlb/check_eqlb_conditions.py:183-291)
 *   cell_sig2  [nrhs][ncells]  || sigma_eq ||^2_L2(T)                   (flux indicator err_sig of
 *                              demo/poisson/demo_error_estimation.py:93-100 for the SE flux)
 *   facet_jump [nrhs][nfacets] max_j | j-th moment of [(sigma_eq + G).n] | on interior facets, 0 on
 *                              boundary facets (H(div) conformity, check_eqlb_conditions.py:294-359)
 * Any output may be NULL.  Arrays in the layouts of eqlb_se_equilibrate; memspace as there. */
int eqlb_se_estimate(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux_hdiv,
                     const double* flux_dg, const double* rhs_dg, double* cell_div2,
                     double* cell_sig2, double* facet_jump, int32_t memspace, void* stream);

/* The same quantities for a conforming (EV) flux handed over in the broken layout
 * (eqlb_ev_set_option "output" = 1): the total flux is sigma_eq itself, so
 *   cell_div2 = || Pi f - div sigma_eq ||^2_T,  cell_sig2 = || sigma_eq - G ||^2_T  (err_sig =
 *   grad(u_h) + sigma_eqlb of demo_error_estimation.py:97-100),  facet_jump = jump moments of sigma_eq. */
int eqlb_ev_estimate(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux_broken,
                     const double* flux_dg, const double* rhs_dg, double* cell_div2,
                     double* cell_sig2, double* facet_jump, int32_t memspace, void* stream);

/* Stress estimator terms of demo/elasticity/demo_error_estimation.py:49-148 for an equilibrated stress
 * delta_sigma = (row 0; row 1), flux_hdiv [2][ncells*k(k+2)] as eqlb_se_equilibrate writes it, per cell:
 *   cell_energy [ncells]  int_T delta_sigma : A delta_sigma,  A tau = (tau - pi_1/(2 + 2 pi_1) tr(tau) I)/2
 *                         (:100-102, 109; pi_1 = lambda / mu)
 *   cell_wsym   [ncells]  int_T (C_K (delta_sigma_01 - delta_sigma_10) / 2)^2            (:108, 121)
 *   node_asym   [nnodes]  (delta_sigma_01 - delta_sigma_10, hat_n): the weak symmetry condition
 *                         (python/dolfinx_eqlb/eqlb/check_eqlb_conditions.py:476-521), assembled over the
 *                         cells of every node in the order of the node -> cell list
 * korn [ncells]: the cell-wise Korn constants C_K as FluxEqlbSE hands them out (square root taken,
 * FluxEqlbSE.py:165) or NULL (C_K = 1).  Any output may be NULL.  Quadrature-free (exact). */
int eqlb_se_estimate_stress(eqlb_mesh_t* mesh, int32_t k, const double* flux_hdiv, const double* korn,
                            double pi_1, double* cell_energy, double* cell_wsym, double* node_asym,
                            int32_t memspace, void* stream);

/* Data oscillation per cell,  out [nrhs][ncells] = C_K^2 (h_T / pi)^2 || f - div(sigma) ||^2_L2(T)
 * (err_osc of demo/poisson/demo_error_estimation.py:96-98 and, with the Korn constant, of
 * demo/elasticity/demo_error_estimation.py:104-106; h_T = longest edge as dolfinx::mesh::h).
 *   flux     [nrhs][ncells*k(k+2)]  RT_k coefficients in the broken hierarchic layout
 *   flux_dg  [nrhs][ncells*k(k+1)]  sigma = flux + flux_dg (semi-explicit result), or NULL: sigma = flux
 *                                   (a conforming flux, eqlb_ev_set_option "output" = 1)
 *   qpoints [nq][2], qweights [nq]  rule on the reference triangle (weights sum to 1/2), HOST arrays
 *   fvalues  [nrhs][ncells][nq]     the un-projected f at the images of the points (as eqlb_project_dg takes)
 *   korn     [ncells] or NULL       as eqlb_se_estimate_stress
 * div(sigma) is evaluated exactly (polynomial of P_{k-1} per cell); the rule only integrates f. nq <= 128. */
int eqlb_oscillation(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux, const double* flux_dg,
                     int32_t nq, const double* qpoints, const double* qweights, const double* fvalues,
                     const double* korn, double* out, int32_t memspace, void* stream);

/* Multi-GPU decomposition by node ownership (the reference has no distributed
 * equilibration, se/reconstruction.hpp:90 loops the owned nodes only): after the local sweep the
 * partial sums of the ghost-cell rows are sent to the owning rank and added there.  DEVICE pointers:
 *   eqlb_halo_pack        buf[r][i][:] = x[r][cells[i]][:]  (i < nlist), rows cleared if clear != 0
 *   eqlb_halo_unpack_add  x[r][cells[i]][:] += buf[r][i][:]
 * x [nrhs][ncells][nrt], cells [nlist] int64, buf [nrhs][nlist][nrt]; asynchronous on `stream`.  The
 * transport between the two calls: eqlb_halo_exchange below, or the caller's own (torch.distributed send / recv in
 * dolfinx_eqlb_amd/distributed.py). */
int eqlb_halo_pack(int32_t nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells,
                   double* x, double* buf, int32_t clear, void* stream);
int eqlb_halo_unpack_add(int32_t nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells,
                         double* x, const double* buf, void* stream);

/* The transport of the reverse halo in the C++ host itself ("ncclGroupStart; ncclSend / ncclRecv per
 * neighbour; ncclGroupEnd" - new design, the reference's node loop cpp/dolfinx_eqlb/se/reconstruction.hpp:90 has
 * no counterpart): grouped point-to-point sends / receives over RCCL (xGMI) on the CALLER's communicator
 * (`comm` = ncclComm_t) and stream.  RCCL is resolved at run time - first among the libraries the process has
 * already loaded (the caller's own RCCL), then librccl.so of the ROCm installation; EQLB_ERR_UNSUPPORTED if
 * there is none.
 *   eqlb_halo_exchange  peers [npeers] ranks; send_buf[i] / recv_buf[i] DEVICE buffers of send_count[i] /
 *                       recv_count[i] doubles (0 = nothing in that direction); one ncclGroup for all of them.
 *   eqlb_halo_reduce    the whole reduction in one call: eqlb_halo_pack (with clear) of the rows send_idx[i]
 *                       [nsend[i]] of x [nrhs][nentries][nrt] into send_buf[i], the grouped exchange,
 *                       eqlb_halo_unpack_add of recv_buf[i] onto the rows recv_idx[i] [nrecv[i]]; the index
 *                       lists are DEVICE arrays (int64), the arrays of pointers / counts HOST arrays; buffers
 *                       of nrhs * n * nrt doubles.  Asynchronous on `stream`.
 * Communicator helpers for hosts that do not link RCCL themselves (id128 = ncclUniqueId, 128 bytes, made on one
 * rank and distributed by the caller - MPI_Bcast where DOLFINx runs):
 *   eqlb_rccl_get_unique_id, eqlb_rccl_comm_create (ncclCommInitRank), eqlb_rccl_comm_destroy. */
int eqlb_halo_exchange(void* comm, int32_t npeers, const int32_t* peers, const double* const* send_buf,
                       const int64_t* send_count, double* const* recv_buf, const int64_t* recv_count,
                       void* stream);
int eqlb_halo_reduce(void* comm, int32_t nrhs, int32_t nrt, int64_t nentries, double* x, int32_t npeers,
                     const int32_t* peers, const int64_t* const* send_idx, const int64_t* nsend,
                     double* const* send_buf, const int64_t* const* recv_idx, const int64_t* nrecv,
                     double* const* recv_buf, void* stream);
/* A halo plan keeps the index lists on the device and owns the staging buffers (what a C++ host would otherwise
 * allocate itself): send_idx[i] / recv_idx[i] are HOST arrays here, copied once.
 *   eqlb_halo_reduce_plan  = eqlb_halo_reduce with the plan's lists and buffers
 *   eqlb_halo_bytes        bytes sent / received by this rank per reduction */
typedef struct eqlb_halo eqlb_halo_t;
int eqlb_halo_create(int32_t nrhs, int32_t nrt, int64_t nentries, int32_t npeers, const int32_t* peers,
                     const int64_t* const* send_idx, const int64_t* nsend, const int64_t* const* recv_idx,
                     const int64_t* nrecv, eqlb_halo_t** handle);
void eqlb_halo_destroy(eqlb_halo_t* handle);
int eqlb_halo_bytes(const eqlb_halo_t* handle, int64_t* bytes_sent, int64_t* bytes_received);
int eqlb_halo_reduce_plan(eqlb_halo_t* handle, void* comm, double* x, void* stream);
int eqlb_rccl_get_unique_id(void* id128);
int eqlb_rccl_comm_create(const void* id128, int32_t nranks, int32_t rank, void** comm);
void eqlb_rccl_comm_destroy(void* comm);

/* Tiling of the EQLB_SCATTER_TILED launch (built by eqlb_se_set_boundary for plain flux
 * equilibration): number of tiles, owned cells per tile, patch instances (a patch on a tile rim is
 * solved once per tile it touches; compare with eqlb_se_num_patches) and lane slots. */
int eqlb_se_tiling_info(const eqlb_se_t* handle, int64_t* ntiles, int64_t* cells_per_tile,
                        int64_t* npatch_instances, int64_t* nlane_slots);

/* ---------------------------------------------------------------------------------------------
 * Constrained-minimisation equilibrator (Ern & Vohralik) - replaces
 * `reconstruct_fluxes_minimisation(a, l_pen, l, flux_hdiv, boundary_data)`
 * (python/dolfinx_eqlb/wrappers.cpp:85-95 -> ev/reconstruction.hpp:32-176,
 * ev/solve_patch.hpp:58-238) behind `FluxEqlbEV.equilibrate_fluxes` (eqlb/FluxEqlbEV.py:167-176).
 *
 * The forms of FluxEqlbEV.py:113-134 are fixed (a = (sig,v) - (r,div v) + (div sig,q),
 * l = hat G.v + (hat f + grad hat . G) q with G = list_proj_flux, f = list_rhs), so the UFL/FFCx
 * form objects of the reference signature are replaced by the flat arrays G, f.  Each patch
 * problem has the unique solution of the reference's (ndof+1)^2 saddle-point LU; it is computed in
 * the reduced unknowns of the semi-explicit kernel.
 *
 * Output space: H(div)-conforming RT_k.  Without Basix the conforming version of the hierarchic
 * RT_k of create_hierarchic_rt is used: k facet DOFs per facet in the global facet frame (parameter
 * from the lower to the higher node id, normal n_E = (t_y, -t_x), t = x_hi - x_lo), then k^2-k
 * interior DOFs per cell.  Default numbering: facet*k + j, then nfacets*k + cell*(k^2-k) + i;
 * `eqlb_ev_set_dofmap` installs the caller's cell->dof table instead (the conforming dofmap
 * `V_flux.dofmap.list` of ev/Patch.cpp:497-501, local order of the hierarchic element).
 * ------------------------------------------------------------------------------------------- */
typedef struct eqlb_ev eqlb_ev_t;

int eqlb_ev_create(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, eqlb_ev_t** handle);
void eqlb_ev_destroy(eqlb_ev_t* handle);

/* "output": 0 conforming DOFs (default), 1 broken hierarchic RT_k layout [ncells*k(k+2)] as
 * eqlb_se_equilibrate writes it; "timing", "scatter" (EQLB_SCATTER_AUTO / _SLOTS / _TILED),
 * "accumulate": as eqlb_se_set_option; "boundary_basis": 0 (default) the boundary values of
 * eqlb_ev_set_boundary are DOFs of the output basis (eqlb_ev_set_basis_transform), 1 they are DOFs of the
 * conforming hierarchic RT_k whatever the output basis (what a caller has who computes the facet moments
 * int_E g s^j itself; set before eqlb_ev_set_boundary). */
int eqlb_ev_set_option(eqlb_ev_t* handle, const char* key, int32_t value);

/* cell_dofs [ncells][k(k+2)] host array (NULL restores the default numbering), ndofs = size of the
 * conforming space.  Call before eqlb_ev_set_boundary. */
int eqlb_ev_set_dofmap(eqlb_ev_t* handle, const int32_t* cell_dofs, int64_t ndofs);
int64_t eqlb_ev_num_dofs(const eqlb_ev_t* handle);

/* Element basis of the conforming output.  The reference scatters the EV flux into the Basix RT_k space
 * through V_flux.dofmap (ev/solve_patch.hpp:223-227, FluxEqlbEV.py:95-100); without Basix the library
 * writes the conforming hierarchic RT_k (above).  An adapter installs the change of basis here:
 *   C [k(k+2)][k(k+2)] row-major: coefficients of a cell in the target element = C x its coefficients in the
 *                      broken (cell-frame) hierarchic RT_k, C[i][j] = l_i^target(phi_j^hierarchic) on the
 *                      reference cell; the facet rows may only involve the DOFs of their own facet;
 *   R [k][k] or NULL   the target element's base transformation of a reflected edge: applied to the facet
 *                      block of a cell whose facet_perm bit is set (NULL: none).
 * Facet DOFs are written by the first cell of the facet; numbering by eqlb_ev_set_dofmap (or the default).
 * Boundary values handed to eqlb_ev_set_boundary are then target-element DOFs as well.  C = NULL restores the
 * hierarchic basis (equivalent to C = diag(-I facets, I interior), R = -B).  Call before eqlb_ev_set_boundary. */
int eqlb_ev_set_basis_transform(eqlb_ev_t* handle, const double* C, const double* R);

/* facet_type as eqlb_se_set_boundary; boundary_values [nrhs][ndofs] conforming boundary DOFs
 * (facet DOFs of the prescribed normal flux on the flux-BC facets, zero elsewhere) or NULL; the
 * per-patch values hat_a * g (base/BoundaryData.cpp:687-745) are formed in the kernel.
 * node_mask as eqlb_se_set_boundary. */
int eqlb_ev_set_boundary(eqlb_ev_t* handle, const int8_t* facet_type,
                         const double* boundary_values, const uint8_t* node_mask);

/* flux_dg [nrhs][ncells*k(k+1)], rhs_dg [nrhs][ncells*k(k+1)/2] as eqlb_se_equilibrate;
 * flux_hdiv [nrhs][ndofs] (or [nrhs][ncells*k(k+2)] with "output" = 1) is ACCUMULATED (+=),
 * ev/solve_patch.hpp:223-227. */
int eqlb_ev_equilibrate(eqlb_ev_t* handle, const double* flux_dg, const double* rhs_dg,
                        double* flux_hdiv, int32_t memspace, void* stream);
/* one array per right-hand side, as eqlb_se_equilibrate_lists (wrappers.cpp:85-95: list of flux_hdiv) */
int eqlb_ev_equilibrate_lists(eqlb_ev_t* handle, const double* const* flux_dg, const double* const* rhs_dg,
                              double* const* flux_hdiv, int32_t memspace, void* stream);
int64_t eqlb_ev_num_patches(const eqlb_ev_t* handle);
/* which = 0: patch kernel (all bins in one launch), 5: reduction to the conforming DOFs */
double eqlb_ev_last_kernel_ms(const eqlb_ev_t* handle, int32_t which);
int eqlb_ev_check_status(eqlb_ev_t* handle, void* stream); /* as eqlb_se_check_status */

#ifdef __cplusplus
}
#endif
#endif /* EQLB_H */
