/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement (plain C, single thread) of the reference's semi-explicit flux
 * equilibration hot path, dolfinx_eqlb v1.2.0.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED BY EXECUTION: the reference needs DOLFINx 0.6 / Basix 0.6 / Eigen /
 * FFCx and can be neither compiled nor imported in this pipeline, and it ships no golden
 * vectors.  The restatement is pinned by mathematics instead (tests/test_oracle_*.py):
 * uniqueness of the patch-wise constrained minimiser (checked against an independent
 * dense KKT solve), the reference's acceptance predicates (divergence, H(div) jump,
 * flux BC), convergence rates and multi-RHS == single-RHS.
 */
#ifndef EQLB_ORACLE_H
#define EQLB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
  int32_t nnodes, ncells, nfacets;
  const double* x;              /* [nnodes][3] */
  const int32_t* cell_nodes;    /* [ncells][3] */
  const int32_t* cell_facets;   /* [ncells][3] */
  const int32_t* facet_nodes;   /* [nfacets][2] */
  const int32_t* facet_cells_off; /* CSR */
  const int32_t* facet_cells;
  const int32_t* node_cells_off;
  const int32_t* node_cells;
  const int32_t* node_facets_off;
  const int32_t* node_facets;
  const uint8_t* facet_perm;    /* [ncells][3] */
} oracle_mesh_t;

/* Element tabulations = what se::KernelData / base::KernelData hold
 * (se/KernelData.cpp:13-197, base/KernelData.cpp:13-62,191-268). */
typedef struct
{
  int32_t k;       /* RT degree (reference convention, lowest = 1) */
  int32_t ndofs;   /* k(k+2) */
  int32_t nd;      /* DOFs of DG_{deg} (projected flux component / RHS) */
  int32_t ndf;     /* facet-closure DOFs of that element */
  int32_t nq;      /* cell quadrature points */
  int32_t nqf;     /* facet interpolation points */
  const double* qpoints;     /* [nq][2] */
  const double* qweights;    /* [nq] */
  const double* flux_basis;  /* [nq][ndofs][2] reference values */
  const double* rhs_cell;    /* [3][nq][nd] value, d/dX, d/dY */
  const double* rhs_fct;     /* [3*nqf][nd]  facet-major */
  const double* hat_cell;    /* [nq][3] */
  const double* hat_fct;     /* [3*nqf][3] */
  const double* M;           /* [3][k][2][nqf] */
  const double* doftrafo;    /* [k][k] */
  const uint8_t* fct_normal_out; /* [3] */
  const int32_t* fct_dofs;   /* [3][ndf] */
  const double* flux_basis_fct; /* [3*nqf][ndofs][2] RT basis at the facet interpolation points */
} oracle_tables_t;

/* Patch fans of nodes [node_begin, node_end) as built by OrientedPatch::initialize_patch
 * (se/Patch.cpp:406-635), flattened with a fixed stride >= ncells_max + 2; unused entries -1:
 *   cells [.][stride] (a = 0..n+1; ends only for interior patches), fcts [.][stride] (a = 0..n),
 *   fcts_local [.][2*stride] ([2a] = E_a on T_a, [2a+1] = E_a on T_{a+1}),
 *   inodes_local [.][stride], types [.][nrhs].  Used to test the device patch builder. */
int oracle_build_patches(const oracle_mesh_t* mesh, int nrhs, const int8_t* facet_type,
                         int32_t node_begin, int32_t node_end, int32_t stride,
                         int32_t* ncells, int32_t* cells, int32_t* fcts, int8_t* fcts_local,
                         int8_t* inodes_local, int8_t* types);

/* se::reconstruction<T,k> without stress: loops all nodes (se/reconstruction.hpp:286-313).
 *   facet_type      [nrhs][nfacets]   0 internal, 1 essnt_primal, 2 essnt_dual
 *   boundary_values [nrhs][ncells*ndofs] GLOBAL boundary DOFs of the flux (the boundary functions of
 *                   BoundaryData, base/BoundaryData.cpp:414-623) or NULL (homogeneous flux BCs);
 *                   the per-patch values hat_a * g are computed as in calculate_patch_bc (:687-745)
 *   flux_dg         [nrhs][ncells*nd*2], rhs_dg [nrhs][ncells*nd]
 *   flux_hdiv       [nrhs][ncells*ndofs]  accumulated (+=) like the reference
 *   node_begin/end  sub-range of nodes (for partitioned runs); pass 0, nnodes for all.
 * returns 0, or a negative error code (-1: patch with one cell, -2: singular patch matrix). */
int oracle_se_reconstruct(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                          const int8_t* facet_type, const double* boundary_values,
                          const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                          int32_t node_begin, int32_t node_end);

/* As oracle_se_reconstruct with reconstruct_stress = true: rows 0 and 1 are the rows of a stress
 * tensor; after the row-wise equilibration each patch imposes the weak symmetry condition
 * (se/solve_patch_weaksym.hpp:59-233).  returns -4 if nrhs < 2 or k < 2. */
int oracle_se_reconstruct_stress(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                                 const int8_t* facet_type, const double* boundary_values,
                                 const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                                 int32_t node_begin, int32_t node_end);

/* Same, but only for the listed nodes and writing the per-patch result of the explicit
 * step (sigma-tilde) and of the full patch solve, for debugging/tests:
 *   out_patch [nrhs][ncells_patch][ndofs]  (ncells_patch of that node)               */
int oracle_se_patch(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                    const int8_t* facet_type, const double* boundary_values,
                    const double* flux_dg, const double* rhs_dg, int32_t node,
                    double* out_sigma_tilde, double* out_patch, int32_t* out_cells,
                    double* out_u);

/* Squared Korn constants: node loop part of se/reconstruction.hpp:291-304 with
 * OrientedPatch::estimate_squared_korn_constant (se/Patch.cpp:130-334); korn [ncells] accumulated
 * (the Python caller takes the square root, FluxEqlbSE.py:165). */
int oracle_se_korn(const oracle_mesh_t* mesh, int nrhs, const int8_t* facet_type, double* korn,
                   int32_t node_begin, int32_t node_end);

/* ev::reconstruction (ev/reconstruction.hpp:32-176, ev/solve_patch.hpp:58-238): constrained
 * minimisation per patch in the mixed RT_k x DG_{k-1} space, dense partial-pivot LU; the flux lives
 * in the conforming version of the hierarchic RT_k (see eqlb_oracle_ev.c for the conventions).
 *   flux_div        [nq][k(k+2)] reference divergence of the RT basis at the cell quadrature points
 *   cell_dofs       [ncells][k(k+2)] conforming dofmap, ndofs_glob global flux DOFs
 *   boundary_values [nrhs][ndofs_glob] or NULL; flux_hdiv [nrhs][ndofs_glob] accumulated */
int oracle_ev_reconstruct(const oracle_mesh_t* mesh, const oracle_tables_t* tab,
                          const double* flux_div, int nrhs, const int8_t* facet_type,
                          const int32_t* cell_dofs, int64_t ndofs_glob,
                          const double* boundary_values, const double* flux_dg,
                          const double* rhs_dg, double* flux_hdiv, int32_t node_begin,
                          int32_t node_end);
int oracle_ev_patch(const oracle_mesh_t* mesh, const oracle_tables_t* tab, const double* flux_div,
                    int nrhs, const int8_t* facet_type, const int32_t* cell_dofs,
                    int64_t ndofs_glob, const double* boundary_values, const double* flux_dg,
                    const double* rhs_dg, int32_t node, int32_t* out_cells, double* out_u,
                    int32_t* out_ndof_max);

#ifdef __cplusplus
}
#endif
#endif
