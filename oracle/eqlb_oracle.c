/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH (see eqlb_oracle.h).
 *
 * Single-thread CPU restatement of the reference's semi-explicit equilibration loop.
 * Same structure as the reference: one patch at a time, quadrature loops over the
 * tabulated bases, Piola re-mapping of the RT basis per (patch, cell), dense Cholesky.
 * All `file:line` citations are relative to /root/reference/cpp/dolfinx_eqlb/.
 */
#include "eqlb_oracle.h"

#define _USE_MATH_DEFINES
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

enum
{
  PT_INTERNAL = 0,
  PT_ESSNT_DUAL = 1,
  PT_ESSNT_PRIMAL = 2,
  PT_MIXED = 3
}; /* base/Patch.hpp:14-20 */
enum
{
  FT_INTERNAL = 0,
  FT_ESSNT_PRIMAL = 1,
  FT_ESSNT_DUAL = 2
}; /* base/Patch.hpp:22-27 */

/* ------------------------------------------------------------------------------------ */
/* Patch (se/Patch.hpp:36-381 OrientedPatch, :383-1109 Patch<T,k>)                      */
/* ------------------------------------------------------------------------------------ */
typedef struct
{
  const oracle_mesh_t* m;
  const int8_t* ftype; /* [nrhs][nfacets] */
  int nrhs, k, ndofs, nd, ndf, nadd, ndiv, ndofs_pc;
  int ncells_max;
  int node, ncells, nfcts;
  int8_t* type;        /* [nrhs] */
  int32_t *cells, *fcts, *fcts_sorted;
  int8_t *fcts_local, *inodes_local;
  int32_t* dofmap;     /* [4][ncells_max+2][ndofs_pc] */
  int32_t* fctdofs_dg; /* [ncells_max+1][2*ndf] */
  int offs[5];
  int ndof_min_flux;
} patch_t;

#define DM(p, id, a, i) ((p)->dofmap[((size_t)(id) * ((p)->ncells_max + 2) + (a)) * (p)->ndofs_pc + (i)])
#define FDG(p, a, i) ((p)->fctdofs_dg[(size_t)(a) * 2 * (p)->ndf + (i)])

static int cmp_i32(const void* a, const void* b)
{
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

static int ftype_at(const patch_t* p, int rhs, int32_t fct)
{
  return p->ftype[(size_t)rhs * p->m->nfacets + fct];
}

/* se/Patch.cpp:652-670 */
static int8_t fctid_local_cell(const patch_t* p, int32_t fct, int32_t cell)
{
  const int32_t* cf = p->m->cell_facets + 3 * (size_t)cell;
  int8_t l = 0;
  while (l < 3 && cf[l] != fct)
    ++l;
  return l;
}

/* se/Patch.cpp:672-688 */
static int8_t node_local(const patch_t* p, int32_t cell, int32_t node)
{
  const int32_t* cn = p->m->cell_nodes + 3 * (size_t)cell;
  int8_t l = 0;
  while (cn[l] != node)
    ++l;
  return l;
}

/* se/Patch.cpp:690-759: of the two other facets of the cell take the one on the patch */
static int32_t next_facet(const patch_t* p, int32_t cell, int8_t lf)
{
  const int32_t* cf = p->m->cell_facets + 3 * (size_t)cell;
  int32_t e0 = cf[(lf + 1) % 3], e1 = cf[(lf + 2) % 3];
  if (e0 > e1)
  {
    int32_t t = e0;
    e0 = e1;
    e1 = t;
  }
  if (e0 < p->fcts_sorted[0])
    return e1;
  if (e1 > p->fcts_sorted[p->nfcts - 1])
    return e0;
  for (int i = 0; i < p->nfcts; ++i)
    if (p->fcts_sorted[i] == e0)
      return e0;
  return e1;
}

/* se/Patch.cpp:406-635 */
static void initialize_patch(patch_t* p, int32_t node)
{
  const oracle_mesh_t* m = p->m;
  p->node = node;
  const int32_t* cells = m->node_cells + m->node_cells_off[node];
  const int32_t* fcts = m->node_facets + m->node_facets_off[node];
  (void)cells;
  p->ncells = m->node_cells_off[node + 1] - m->node_cells_off[node];
  p->nfcts = m->node_facets_off[node + 1] - m->node_facets_off[node];
  const int ncells = p->ncells, nfcts = p->nfcts;

  memcpy(p->fcts_sorted, fcts, sizeof(int32_t) * nfcts);
  qsort(p->fcts_sorted, nfcts, sizeof(int32_t), cmp_i32);

  for (int i = 0; i < p->nrhs; ++i)
    p->type[i] = PT_INTERNAL;

  int32_t fct_first = fcts[0];
  if (nfcts > ncells)
  {
    /* :427-484 type for rhs 0, start on a flux-BC facet if there is one */
    int32_t fct_ef[2] = {-1, -1}, fct_ep[2] = {-1, -1};
    for (int i = 0; i < nfcts; ++i)
    {
      int32_t f = fcts[i];
      int t = ftype_at(p, 0, f);
      if (t == FT_ESSNT_PRIMAL)
      {
        if (fct_ep[0] < 0)
          fct_ep[0] = f;
        else
          fct_ep[1] = f;
      }
      else if (t == FT_ESSNT_DUAL)
      {
        if (fct_ef[0] < 0)
          fct_ef[0] = f;
        else
          fct_ef[1] = f;
      }
    }
    if (fct_ef[0] < 0)
    {
      p->type[0] = PT_ESSNT_PRIMAL;
      fct_first = fct_ep[0];
    }
    else
    {
      p->type[0] = (fct_ep[0] < 0) ? PT_ESSNT_DUAL : PT_MIXED;
      fct_first = fct_ef[0];
    }
    /* :487-526 types of the following RHS from the same two end facets */
    for (int r = 1; r < p->nrhs; ++r)
    {
      int32_t f0, fn;
      if (p->type[0] == PT_ESSNT_PRIMAL)
      {
        f0 = fct_ep[0];
        fn = fct_ep[1];
      }
      else if (p->type[0] == PT_ESSNT_DUAL)
      {
        f0 = fct_ef[0];
        fn = fct_ef[1];
      }
      else
      {
        f0 = fct_ef[0];
        fn = fct_ep[0];
      }
      if (ftype_at(p, r, f0) == ftype_at(p, r, fn))
        p->type[r] = (ftype_at(p, r, f0) == FT_ESSNT_PRIMAL) ? PT_ESSNT_PRIMAL : PT_ESSNT_DUAL;
      else
        p->type[r] = PT_MIXED;
    }
  }

  const int internal = (p->type[0] == PT_INTERNAL);
  if (internal)
    p->fcts[1] = fct_first;
  else
    p->fcts[0] = fct_first;

  /* :546-635 walk the fan */
  int lloop = ncells + 1;
  if (internal)
  {
    p->cells[1] = m->facet_cells[m->facet_cells_off[p->fcts[1]] + 1];
  }
  else
  {
    p->cells[1] = m->facet_cells[m->facet_cells_off[p->fcts[0]]];
    int8_t lf = fctid_local_cell(p, fct_first, p->cells[1]);
    p->fcts_local[0] = lf;
    p->fcts_local[1] = lf;
    p->fcts[1] = next_facet(p, p->cells[1], lf);
    lloop = ncells;
  }
  for (int a = 1; a < lloop; ++a)
  {
    int32_t fct_a = p->fcts[a], cell_a = p->cells[a];
    const int32_t* cf = m->facet_cells + m->facet_cells_off[fct_a];
    int32_t cell_ap1 = (cf[0] == cell_a) ? cf[1] : cf[0];
    p->cells[a + 1] = cell_ap1;
    int8_t lf_ap1 = fctid_local_cell(p, fct_a, cell_ap1);
    p->fcts_local[2 * a] = fctid_local_cell(p, fct_a, cell_a);
    p->fcts_local[2 * a + 1] = lf_ap1;
    p->inodes_local[a] = node_local(p, cell_a, node);
    p->fcts[a + 1] = next_facet(p, cell_ap1, lf_ap1);
  }
  if (!internal)
  {
    p->inodes_local[ncells] = node_local(p, p->cells[ncells], node);
    int8_t lf = fctid_local_cell(p, p->fcts[ncells], p->cells[ncells]);
    p->fcts_local[2 * ncells] = lf;
    p->fcts_local[2 * ncells + 1] = lf;
  }
  else
  {
    p->cells[0] = p->cells[ncells];
    p->cells[ncells + 1] = p->cells[1];
    p->inodes_local[0] = p->inodes_local[ncells];
    p->inodes_local[ncells + 1] = p->inodes_local[1];
    p->fcts[0] = p->fcts[nfcts];
    p->fcts_local[0] = p->fcts_local[2 * nfcts];
    p->fcts_local[1] = p->fcts_local[2 * nfcts + 1];
  }
}

/* se/Patch.hpp:921-996 (offset selection collapses to this for valid arguments) */
static int8_t fctid_local(const patch_t* p, int fct_i, int cell_i)
{
  int offst;
  if (p->type[0] == PT_INTERNAL && (fct_i == 0 || fct_i == p->ncells))
    offst = (cell_i == 1 || cell_i == p->ncells + 1) ? 1 : 0;
  else
    offst = (cell_i == fct_i) ? 0 : 1;
  if (p->type[0] != PT_INTERNAL && fct_i == 0)
    offst = 0;
  return p->fcts_local[2 * fct_i + offst];
}

/* se/Patch.hpp:1001-1007 */
static void fctid_local_pair(const patch_t* p, int a, int8_t* eam1, int8_t* ea)
{
  *eam1 = p->fcts_local[2 * a - 1];
  *ea = p->fcts_local[2 * a];
}

/* se/Patch.cpp:106-128 */
static int requires_flux_bcs_rhs(const patch_t* p, int r)
{
  return p->type[r] == PT_ESSNT_DUAL || p->type[r] == PT_MIXED;
}
static int reversion_required(const patch_t* p, int r)
{
  if (r > 0 && requires_flux_bcs_rhs(p, r))
  {
    if (p->type[r] != p->type[r - 1] || p->type[r] == PT_MIXED)
      if (ftype_at(p, r, p->fcts[0]) != FT_ESSNT_DUAL)
        return 1;
  }
  return 0;
}

/* se/Patch.hpp:468-593 (without the weak-symmetry constraint DOFs) */
static void flux_dofmap_cell(patch_t* p, int a)
{
  const int k = p->k;
  const int32_t cell = p->cells[a];
  int8_t fl_eam1, fl_ea;
  fctid_local_pair(p, a, &fl_eam1, &fl_ea);
  const int32_t gdof = cell * p->ndofs;
  const int internal = (p->type[0] == PT_INTERNAL);

  int pdof_eam1 = (a - 1) * (k - 1);
  int pdof_ea = (internal && a == p->ncells) ? 0 : pdof_eam1 + k - 1;
  for (int ii = 0; ii < k; ++ii)
  {
    int l_eam1 = fl_eam1 * k + ii, l_ea = fl_ea * k + ii, o = p->offs[1] + ii;
    DM(p, 0, a, ii) = l_eam1;
    DM(p, 0, a, o) = l_ea;
    DM(p, 1, a, ii) = gdof + l_eam1;
    DM(p, 1, a, o) = gdof + l_ea;
    DM(p, 2, a, ii) = (ii == 0) ? 0 : pdof_eam1 + ii;
    DM(p, 2, a, o) = (ii == 0) ? 0 : pdof_ea + ii;
  }
  /* additional (interior, divergence-free) cell DOFs */
  {
    int o = p->offs[2], ldof = 3 * k + p->ndiv;
    int pdof = p->nfcts * (k - 1) + 1 + (a - 1) * p->nadd;
    for (int ii = 0; ii < p->nadd; ++ii, ++o, ++ldof)
    {
      DM(p, 0, a, o) = ldof;
      DM(p, 1, a, o) = gdof + ldof;
      DM(p, 2, a, o) = pdof + ii;
      DM(p, 3, a, o) = 1;
    }
  }
  /* divergence DOFs */
  {
    int o = p->offs[4], ldof = 3 * k;
    for (int ii = 0; ii < p->ndiv; ++ii, ++o, ++ldof)
    {
      DM(p, 0, a, o) = ldof;
      DM(p, 1, a, o) = gdof + ldof;
      DM(p, 3, a, o) = 0;
    }
  }
}

/* DOFs of the patch-wise P1 multiplier space, se/Patch.hpp:595-708: per cell the slots
 * offs[3] + {0: patch node, 1: outer node of E_a, 2: outer node of E_{a-1}}; patch-local ids:
 * centre 0, outer node of E_a -> a; boundary patches: outer node of E_0 -> nfcts-1, of E_n -> nfcts */
static int32_t facet_outer_node(const patch_t* p, int fct_i)
{
  const int32_t* fn = p->m->facet_nodes + 2 * (size_t)p->fcts[fct_i];
  return (fn[0] == p->node) ? fn[1] : fn[0];
}

static void constraint_dofmap(patch_t* p)
{
  const int n = p->ncells, o = p->offs[3];
  const int internal = (p->type[0] == PT_INTERNAL);
  for (int a = 1; a <= n; ++a)
  {
    const int32_t cell = p->cells[a];
    /* centre */
    DM(p, 0, a, o) = p->inodes_local[a];
    DM(p, 2, a, o) = 0;
    DM(p, 3, a, o) = 1;
    /* outer node of E_a (slot 1) and of E_{a-1} (slot 2) */
    const int32_t node_ea = facet_outer_node(p, a);
    const int32_t node_eam1 = facet_outer_node(p, a - 1);
    DM(p, 0, a, o + 1) = node_local(p, cell, node_ea);
    DM(p, 0, a, o + 2) = node_local(p, cell, node_eam1);
    DM(p, 3, a, o + 1) = 1;
    DM(p, 3, a, o + 2) = 1;
    if (internal)
    {
      DM(p, 2, a, o + 1) = a;
      DM(p, 2, a, o + 2) = (a == 1) ? n : a - 1;
    }
    else
    {
      DM(p, 2, a, o + 1) = (a == n) ? p->nfcts : a;
      DM(p, 2, a, o + 2) = (a == 1) ? p->nfcts - 1 : a - 1;
    }
  }
}

/* se/Patch.hpp:792-898 */
static void create_subdofmap(patch_t* p, int32_t node)
{
  initialize_patch(p, node);
  const int n = p->ncells;
  p->ndof_min_flux = 1 + (p->k - 1) * p->nfcts + p->nadd * n;
  for (int a = 1; a <= n; ++a)
    flux_dofmap_cell(p, a);
  constraint_dofmap(p);
}

/* se/Patch.hpp:836-897: facet DOFs of the projected flux; needs the element table */
static void set_fctdofs_dg(patch_t* p, const oracle_tables_t* tab)
{
  const int n = p->ncells, ndf = p->ndf;
  for (int a = 1; a <= n; ++a)
  {
    int8_t l_eam1, l_ea;
    fctid_local_pair(p, a, &l_eam1, &l_ea);
    for (int i = 0; i < ndf; ++i)
    {
      FDG(p, a - 1, i) = tab->fct_dofs[l_eam1 * ndf + i];
      FDG(p, a, ndf + i) = tab->fct_dofs[l_ea * ndf + i];
    }
  }
  if (p->type[0] == PT_INTERNAL)
  {
    for (int id = 0; id < 3; ++id)
      for (int ii = 0; ii < p->ndofs_pc; ++ii)
      {
        DM(p, id, 0, ii) = DM(p, id, n, ii);
        DM(p, id, n + 1, ii) = DM(p, id, 1, ii);
      }
    for (int ii = 0; ii < ndf; ++ii)
    {
      FDG(p, n, ii) = FDG(p, 0, ii);
      FDG(p, 0, ndf + ii) = FDG(p, n, ndf + ii);
    }
  }
  else
  {
    for (int ii = 0; ii < ndf; ++ii)
    {
      FDG(p, 0, ndf + ii) = FDG(p, 0, ii);
      FDG(p, n, ii) = FDG(p, n, ndf + ii);
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* PatchData (se/PatchData.hpp:26-157) + scratch of KernelData                          */
/* ------------------------------------------------------------------------------------ */
typedef struct
{
  int dim_max, nh; /* nh = 2k + nadd - 1 functions per cell */
  double *J, *K, *detJ, *prefactor;
  uint8_t* reversed;
  double* Mm;        /* [n][k+1][2][nqf] */
  double* coeffs;    /* [nrhs][ncells_max][ndofs] */
  double* jumpG;     /* [nqf][2][2] */
  double *c_ta_div, *cj_ta_ea;
  double *A, *Lc, *L, *u, *Te;
  int8_t* bmarkers;
  /* weak symmetry (se/PatchData.hpp:118-157) */
  int npnt_max, dim_c;
  double *A_rec, *Bm, *Cm, *Lfull, *u_c, *AinvB, *u_sig2, *cstress, *Be, *Ce, *Le2;
  int8_t* bmarkers2;
  int meanvalue_required;
  double *G_Ta, *G_Tap1, *f_Ta;
  double* phi;       /* mapped RT basis [nq][ndofs][2] */
  double* rhs_cur;   /* mapped gradients [2][nq][nd] */
  double* gphi;      /* [k][2] */
} pdata_t;

static void* xcalloc(size_t n, size_t s) { return calloc(n ? n : 1, s); }

static void patch_alloc(patch_t* p, pdata_t* d, const oracle_mesh_t* m,
                        const oracle_tables_t* tab, int nrhs, const int8_t* ftype)
{
  memset(p, 0, sizeof(*p));
  memset(d, 0, sizeof(*d));
  p->m = m;
  p->ftype = ftype;
  p->nrhs = nrhs;
  const int k = p->k = tab->k;
  p->ndofs = tab->ndofs;
  p->nd = tab->nd;
  p->ndf = tab->ndf;
  p->nadd = (k - 1) * (k - 2) / 2;
  p->ndiv = k * (k + 1) / 2 - 1;
  p->ndofs_pc = 2 * k + p->nadd + 3 + p->ndiv; /* [E_am1 | E_a | add | constr (3) | div] */
  p->offs[0] = 0;
  p->offs[1] = k;
  p->offs[2] = 2 * k;
  p->offs[3] = 2 * k + p->nadd;
  p->offs[4] = p->offs[3] + 3;
  int nmax = 0;
  for (int i = 0; i < m->nnodes; ++i)
  {
    int c = m->node_cells_off[i + 1] - m->node_cells_off[i];
    if (c > nmax)
      nmax = c;
  }
  p->ncells_max = nmax;
  const int sp1 = nmax + 1, sp2 = nmax + 2;
  p->type = xcalloc(nrhs, 1);
  p->cells = xcalloc(sp2, sizeof(int32_t));
  p->fcts = xcalloc(sp2, sizeof(int32_t));
  p->fcts_sorted = xcalloc(sp1, sizeof(int32_t));
  p->fcts_local = xcalloc(2 * sp1 + 2, 1);
  p->inodes_local = xcalloc(sp2, 1);
  p->dofmap = xcalloc((size_t)4 * sp2 * p->ndofs_pc, sizeof(int32_t));
  p->fctdofs_dg = xcalloc((size_t)2 * sp1 * p->ndf, sizeof(int32_t));

  d->nh = 2 * k + p->nadd - 1;
  d->dim_max = 1 + (k - 1) * sp1 + p->nadd * nmax;
  d->J = xcalloc(4 * nmax, sizeof(double));
  d->K = xcalloc(4 * nmax, sizeof(double));
  d->detJ = xcalloc(nmax, sizeof(double));
  d->prefactor = xcalloc(2 * nmax, sizeof(double));
  d->reversed = xcalloc(2 * nmax, 1);
  d->Mm = xcalloc((size_t)nmax * (k + 1) * 2 * tab->nqf, sizeof(double));
  d->coeffs = xcalloc((size_t)nrhs * nmax * p->ndofs, sizeof(double));
  d->jumpG = xcalloc((size_t)tab->nqf * 4, sizeof(double));
  d->c_ta_div = xcalloc(p->ndiv, sizeof(double));
  d->cj_ta_ea = xcalloc(k, sizeof(double));
  d->A = xcalloc((size_t)d->dim_max * d->dim_max, sizeof(double));
  d->Lc = xcalloc((size_t)d->dim_max * d->dim_max, sizeof(double));
  d->L = xcalloc(d->dim_max, sizeof(double));
  d->u = xcalloc(d->dim_max, sizeof(double));
  d->Te = xcalloc((size_t)(d->nh + 1) * d->nh, sizeof(double));
  d->bmarkers = xcalloc(d->dim_max, 1);
  d->npnt_max = nmax + 3;
  d->A_rec = xcalloc((size_t)d->dim_max * d->dim_max, sizeof(double));
  d->Bm = xcalloc((size_t)d->dim_max * 2 * d->npnt_max, sizeof(double));
  d->Cm = xcalloc((size_t)(d->npnt_max + 1) * (d->npnt_max + 1), sizeof(double));
  d->Lfull = xcalloc((size_t)2 * d->dim_max + d->npnt_max + 1, sizeof(double));
  d->u_c = xcalloc(d->npnt_max + 1, sizeof(double));
  d->AinvB = xcalloc((size_t)d->dim_max * d->npnt_max, sizeof(double));
  d->u_sig2 = xcalloc((size_t)2 * d->dim_max, sizeof(double));
  d->cstress = xcalloc((size_t)nmax * 2 * p->ndofs, sizeof(double));
  d->Be = xcalloc((size_t)d->nh * 6, sizeof(double));
  d->Ce = xcalloc(3, sizeof(double));
  d->Le2 = xcalloc((size_t)2 * d->nh + 3, sizeof(double));
  d->bmarkers2 = xcalloc((size_t)2 * d->dim_max, 1);
  d->G_Ta = xcalloc(2 * p->nd, sizeof(double));
  d->G_Tap1 = xcalloc(2 * p->nd, sizeof(double));
  d->f_Ta = xcalloc(p->nd, sizeof(double));
  d->phi = xcalloc((size_t)tab->nq * p->ndofs * 2, sizeof(double));
  d->rhs_cur = xcalloc((size_t)2 * tab->nq * p->nd, sizeof(double));
  d->gphi = xcalloc(2 * k, sizeof(double));
}

static void patch_free(patch_t* p, pdata_t* d)
{
  free(p->type);
  free(p->cells);
  free(p->fcts);
  free(p->fcts_sorted);
  free(p->fcts_local);
  free(p->inodes_local);
  free(p->dofmap);
  free(p->fctdofs_dg);
  free(d->J);
  free(d->K);
  free(d->detJ);
  free(d->prefactor);
  free(d->reversed);
  free(d->Mm);
  free(d->coeffs);
  free(d->jumpG);
  free(d->c_ta_div);
  free(d->cj_ta_ea);
  free(d->A);
  free(d->Lc);
  free(d->L);
  free(d->u);
  free(d->Te);
  free(d->bmarkers);
  free(d->A_rec);
  free(d->Bm);
  free(d->Cm);
  free(d->Lfull);
  free(d->u_c);
  free(d->AinvB);
  free(d->u_sig2);
  free(d->cstress);
  free(d->Be);
  free(d->Ce);
  free(d->Le2);
  free(d->bmarkers2);
  free(d->G_Ta);
  free(d->G_Tap1);
  free(d->f_Ta);
  free(d->phi);
  free(d->rhs_cur);
  free(d->gphi);
}

#define MM(d, tab, a, i, c, q) ((d)->Mm[(((size_t)(a) * ((tab)->k + 1) + (i)) * 2 + (c)) * (tab)->nqf + (q)])
#define MREF(tab, f, j, c, q) ((tab)->M[(((size_t)(f) * (tab)->k + (j)) * 2 + (c)) * (tab)->nqf + (q)])
#define COEF(p, d, r, ida, i) ((d)->coeffs[((size_t)(r) * (p)->ncells_max + (ida)) * (p)->ndofs + (i)])
#define JG(d, n, s, c) ((d)->jumpG[((n) * 2 + (s)) * 2 + (c)])

/* se/Patch.hpp:710-789 */
static void set_assembly_informations(patch_t* p, const pdata_t* d, const uint8_t* orient)
{
  const int k = p->k, n = p->ncells;
  for (int a = 1; a <= n; ++a)
  {
    const int id_a = a - 1;
    int8_t l_eam1, l_ea;
    fctid_local_pair(p, a, &l_eam1, &l_ea);
    int pf_eam1, pf_ea;
    if (d->detJ[id_a] < 0)
    {
      pf_eam1 = d->reversed[2 * id_a] ? DM(p, 3, a - 1, k) : (orient[l_eam1] ? -1 : 1);
      pf_ea = orient[l_ea] ? 1 : -1;
    }
    else
    {
      pf_eam1 = d->reversed[2 * id_a] ? DM(p, 3, a - 1, k) : (orient[l_eam1] ? 1 : -1);
      pf_ea = orient[l_ea] ? -1 : 1;
    }
    for (int i = 0; i < k; ++i)
    {
      DM(p, 3, a, i) = pf_eam1;
      DM(p, 3, a, k + i) = pf_ea;
    }
  }
  if (p->type[0] == PT_INTERNAL)
  {
    if (d->reversed[0])
      for (int i = 0; i < k; ++i)
        DM(p, 3, 1, i) = DM(p, 3, n, k + i);
    for (int ii = 0; ii < p->ndofs_pc; ++ii)
    {
      DM(p, 3, 0, ii) = DM(p, 3, n, ii);
      DM(p, 3, n + 1, ii) = DM(p, 3, 1, ii);
    }
  }
}

/* se/assembly.hpp:46-98 */
static void set_boundary_markers(int8_t* bm, int dim, int type, int reversion, int ncells, int k)
{
  memset(bm, 0, dim);
  const int offset_En = ncells * (k - 1);
  if (type != PT_ESSNT_PRIMAL)
  {
    bm[0] = 1;
    for (int j = 1; j < k; ++j)
    {
      if (type == PT_ESSNT_DUAL)
      {
        bm[j] = 1;
        bm[j + offset_En] = 1;
      }
      else if (reversion)
        bm[offset_En + j] = 1;
      else
        bm[j] = 1;
    }
  }
}

/* se/fluxmin_kernel.hpp:60-190 with KernelData::shapefunctions_flux
 * (se/KernelData.hpp:111-127, se/KernelData.cpp:201-221) */
static void fluxmin_kernel(const patch_t* p, pdata_t* d, const oracle_tables_t* tab, int a,
                           const double* coefficients, int asmbl_matrix)
{
  const int k = p->k, ndofs = p->ndofs, nh = d->nh, nq = tab->nq;
  const int id_a = a - 1;
  const double detJ = d->detJ[id_a];
  const double* J = d->J + 4 * id_a;
  const uint8_t eam1_reversed = d->reversed[2 * id_a];
  double* Te = d->Te;
  memset(Te, 0, sizeof(double) * (nh + 1) * nh);

  /* contravariant Piola map of ALL basis values (re-done per (patch, cell)) */
  const double inv = 1.0 / detJ;
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < ndofs; ++i)
    {
      const double* r = tab->flux_basis + ((size_t)q * ndofs + i) * 2;
      double* c = d->phi + ((size_t)q * ndofs + i) * 2;
      c[0] = inv * J[0] * r[0] + inv * J[1] * r[1];
      c[1] = inv * J[2] * r[0] + inv * J[3] * r[1];
    }

  const int ld0_eam1 = DM(p, 0, a, 0), ld0_ea = DM(p, 0, a, k);
  const int p_eam1 = DM(p, 3, a, 0), p_ea = DM(p, 3, a, k);

  for (int q = 0; q < nq; ++q)
  {
    double* phi = d->phi + (size_t)q * ndofs * 2;
    double sig[2] = {0, 0};
    for (int i = 0; i < ndofs; ++i)
    {
      sig[0] += coefficients[i] * phi[2 * i];
      sig[1] += coefficients[i] * phi[2 * i + 1];
    }
    if (eam1_reversed)
    {
      memset(d->gphi, 0, sizeof(double) * 2 * k);
      for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
        {
          int ldj = DM(p, 0, a, j);
          d->gphi[2 * i] += tab->doftrafo[i * k + j] * phi[2 * ldj];
          d->gphi[2 * i + 1] += tab->doftrafo[i * k + j] * phi[2 * ldj + 1];
        }
      for (int i = 0; i < k; ++i)
      {
        int ldi = DM(p, 0, a, i);
        phi[2 * ldi] = d->gphi[2 * i];
        phi[2 * ldi + 1] = d->gphi[2 * i + 1];
      }
    }
    /* the d0 function, stored in the slot of the zero-order function of E_a */
    phi[2 * ld0_ea] = p_ea * (p_eam1 * phi[2 * ld0_eam1] + p_ea * phi[2 * ld0_ea]);
    phi[2 * ld0_ea + 1] = p_ea * (p_eam1 * phi[2 * ld0_eam1 + 1] + p_ea * phi[2 * ld0_ea + 1]);

    const double dvol = tab->qweights[q] * fabs(detJ);
    for (int i = 0; i < nh; ++i)
    {
      const int ip1 = i + 1;
      const double alpha = DM(p, 3, a, ip1) * dvol;
      const double phi_i0 = phi[2 * DM(p, 0, a, ip1)] * alpha;
      const double phi_i1 = phi[2 * DM(p, 0, a, ip1) + 1] * alpha;
      Te[nh * nh + i] -= phi_i0 * sig[0] + phi_i1 * sig[1];
      if (asmbl_matrix)
        for (int j = i; j < nh; ++j)
        {
          const int jp1 = j + 1;
          const double phi_j0 = phi[2 * DM(p, 0, a, jp1)] * DM(p, 3, a, jp1);
          const double phi_j1 = phi[2 * DM(p, 0, a, jp1) + 1] * DM(p, 3, a, jp1);
          Te[i * nh + j] += phi_i0 * phi_j0 + phi_i1 * phi_j1;
        }
    }
  }
  if (asmbl_matrix)
    for (int i = 1; i < nh; ++i)
      for (int j = 0; j < i; ++j)
        Te[i * nh + j] = Te[j * nh + i];
}

/* se/assembly.hpp:119-274 */
static void assemble_fluxminimiser(const patch_t* p, pdata_t* d, const oracle_tables_t* tab,
                                   int i_rhs, int requires_flux_bc, int asmbl_matrix)
{
  const int dim = p->ndof_min_flux, nh = d->nh, dm = d->dim_max;
  if (asmbl_matrix)
    memset(d->A, 0, sizeof(double) * dm * dm);
  memset(d->L, 0, sizeof(double) * dm);
  (void)dim;
  for (int a = 1; a <= p->ncells; ++a)
  {
    fluxmin_kernel(p, d, tab, a, &COEF(p, d, i_rhs, a - 1, 0), asmbl_matrix);
    const double* Te = d->Te;
    if (p->k == 1)
    {
      if (requires_flux_bc)
      {
        d->L[0] = 0;
        if (asmbl_matrix)
          d->A[0] = 1;
      }
      else
      {
        d->L[0] += Te[nh * nh];
        if (asmbl_matrix)
          d->A[0] += Te[0];
      }
      continue;
    }
    for (int i = 0; i < nh; ++i)
    {
      const int dof_i = DM(p, 2, a, i + 1);
      const int bm_i = requires_flux_bc ? d->bmarkers[dof_i] : 0;
      if (bm_i)
        d->L[dof_i] = 0;
      else
        d->L[dof_i] += Te[nh * nh + i];
      if (!asmbl_matrix)
        continue;
      if (bm_i)
        d->A[dof_i * dm + dof_i] = 1;
      else
        for (int j = 0; j < nh; ++j)
        {
          const int dof_j = DM(p, 2, a, j + 1);
          const int bm_j = requires_flux_bc ? d->bmarkers[dof_j] : 0;
          if (bm_j)
            d->A[dof_i * dm + dof_j] = 0;
          else
            d->A[dof_i * dm + dof_j] += Te[i * nh + j];
        }
    }
  }
}

/* Eigen::LLT stand-in (se/PatchData.hpp:576-595): plain right-looking Cholesky */
static int factorise_A(pdata_t* d, int dim)
{
  const int dm = d->dim_max;
  double* Lc = d->Lc;
  for (int i = 0; i < dim; ++i)
    for (int j = 0; j <= i; ++j)
      Lc[i * dm + j] = d->A[i * dm + j];
  for (int j = 0; j < dim; ++j)
  {
    double s = Lc[j * dm + j];
    for (int q = 0; q < j; ++q)
      s -= Lc[j * dm + q] * Lc[j * dm + q];
    if (!(s > 0.0))
      return -2;
    const double ljj = sqrt(s);
    Lc[j * dm + j] = ljj;
    for (int i = j + 1; i < dim; ++i)
    {
      double t = Lc[i * dm + j];
      for (int q = 0; q < j; ++q)
        t -= Lc[i * dm + q] * Lc[j * dm + q];
      Lc[i * dm + j] = t / ljj;
    }
  }
  return 0;
}

static void solve_A(pdata_t* d, int dim)
{
  const int dm = d->dim_max;
  const double* Lc = d->Lc;
  double* u = d->u;
  for (int i = 0; i < dim; ++i)
  {
    double t = d->L[i];
    for (int q = 0; q < i; ++q)
      t -= Lc[i * dm + q] * u[q];
    u[i] = t / Lc[i * dm + i];
  }
  for (int i = dim - 1; i >= 0; --i)
  {
    double t = u[i];
    for (int q = i + 1; q < dim; ++q)
      t -= Lc[q * dm + i] * u[q];
    u[i] = t / Lc[i * dm + i];
  }
}

static void copy_G(const patch_t* p, const double* x_flux_proj, int32_t cell, double* out)
{
  /* DG space: dofmap.links(c) = c*nd + j, block size 2 (se/solve_patch_semiexplt.hpp:119-146) */
  memcpy(out, x_flux_proj + (size_t)cell * p->nd * 2, sizeof(double) * 2 * p->nd);
}

/* calculate_jump, se/solve_patch_semiexplt.hpp:64-111 */
static void calculate_jump(const patch_t* p, const oracle_tables_t* tab, double GtHat_Ea[2][2],
                           int iq_Ta, int Ea_reversed, const int32_t* dofs_G_Ea,
                           const double* G_Tap1, int fl_Tap1Ea, int node_Tap1, const double* G_Ta,
                           int fl_TaEa, int node_Ta)
{
  const int nqf = tab->nqf, nd = tab->nd, ndf = p->ndf;
  const int iq_Tap1 = Ea_reversed ? nqf - iq_Ta - 1 : iq_Ta;
  const double* shp_Ta = tab->rhs_fct + (size_t)(fl_TaEa * nqf + iq_Ta) * nd;
  const double* shp_Tap1 = tab->rhs_fct + (size_t)(fl_Tap1Ea * nqf + iq_Tap1) * nd;
  GtHat_Ea[0][0] = GtHat_Ea[0][1] = GtHat_Ea[1][0] = GtHat_Ea[1][1] = 0.0;
  for (int i = 0; i < ndf; ++i)
  {
    const int id_Ta = dofs_G_Ea[i + ndf], id_Tap1 = dofs_G_Ea[i];
    GtHat_Ea[0][0] += G_Ta[2 * id_Ta] * shp_Ta[id_Ta];
    GtHat_Ea[0][1] += G_Ta[2 * id_Ta + 1] * shp_Ta[id_Ta];
    GtHat_Ea[1][0] += G_Tap1[2 * id_Tap1] * shp_Tap1[id_Tap1];
    GtHat_Ea[1][1] += G_Tap1[2 * id_Tap1 + 1] * shp_Tap1[id_Tap1];
  }
  const double hat_Ta = tab->hat_fct[(size_t)(fl_TaEa * nqf + iq_Ta) * 3 + node_Ta];
  const double hat_Tap1 = tab->hat_fct[(size_t)(fl_Tap1Ea * nqf + iq_Tap1) * 3 + node_Tap1];
  GtHat_Ea[0][0] *= hat_Ta;
  GtHat_Ea[0][1] *= hat_Ta;
  GtHat_Ea[1][0] *= hat_Tap1;
  GtHat_Ea[1][1] *= hat_Tap1;
}

/* BoundaryData::calculate_patch_bc + KernelDataBC::interpolate_flux
 * (base/BoundaryData.cpp:687-745, :171-250): DOFs on the boundary facet lfct of `cell` of
 * hat_{hat_id} * g, g = sum_j bdofs[j] phi_{lfct,j}: evaluate at the facet interpolation points,
 * multiply by the hat function, pull back, apply the interpolation matrix.  All DOFs smaller
 * than 1e-7 in magnitude -> the facet is skipped (values stay zero), :714-725. */
static void calculate_patch_bc(const oracle_tables_t* tab, const double* bglob /* [k] */, int lfct,
                               int hat_id, const double* J, double detJ, const double* K,
                               double* bpatch /* [k] */)
{
  const int k = tab->k, nqf = tab->nqf, ndofs = tab->ndofs;
  int nzero = 0;
  for (int i = 0; i < k; ++i)
  {
    bpatch[i] = 0.0;
    if (fabs(bglob[i]) < 1e-7)
      ++nzero;
  }
  if (nzero == k)
    return;
  for (int i = 0; i < k; ++i)
    bpatch[i] = 0.0;
  for (int q = 0; q < nqf; ++q)
  {
    const double* phi = tab->flux_basis_fct + ((size_t)(lfct * nqf + q) * ndofs + lfct * k) * 2;
    double vr[2] = {0, 0};
    for (int j = 0; j < k; ++j)
    {
      vr[0] += bglob[j] * phi[2 * j];
      vr[1] += bglob[j] * phi[2 * j + 1];
    }
    /* push forward (J v / detJ), times hat, pull back (detJ K v): the maps cancel */
    double v[2] = {(J[0] * vr[0] + J[1] * vr[1]) / detJ, (J[2] * vr[0] + J[3] * vr[1]) / detJ};
    const double hat = tab->hat_fct[(size_t)(lfct * nqf + q) * 3 + hat_id];
    v[0] *= hat;
    v[1] *= hat;
    const double m0 = detJ * (K[0] * v[0] + K[1] * v[1]), m1 = detJ * (K[2] * v[0] + K[3] * v[1]);
    for (int i = 0; i < k; ++i)
      bpatch[i] += MREF(tab, lfct, i, 0, q) * m0 + MREF(tab, lfct, i, 1, q) * m1;
  }
}

/* pull-back of a flux value to the reference cell: detJ K v (se/KernelData.hpp:82-88) */
static void pull_back_flux(double out[2], const double v[2], double detJ, const double* K)
{
  out[0] = detJ * (K[0] * v[0] + K[1] * v[1]);
  out[1] = detJ * (K[2] * v[0] + K[3] * v[1]);
}

/* equilibrate_flux_semiexplt, se/solve_patch_semiexplt.hpp:212-1163 */
static int equilibrate_patch(patch_t* p, pdata_t* d, const oracle_tables_t* tab,
                             const double* boundary_values, const double* flux_dg,
                             const double* rhs_dg, double* flux_hdiv, double* out_sigma_tilde,
                             double* out_patch, double* out_u)
{
  const oracle_mesh_t* m = p->m;
  const int k = p->k, n = p->ncells, ndofs = p->ndofs, nd = p->nd, ndf = p->ndf;
  const int nqf = tab->nqf, nq = tab->nq;
  const int nrhs = p->nrhs;
  const int on_boundary = (p->type[0] != PT_INTERNAL);
  const size_t ncells_mesh = m->ncells;

  /* PatchData::reinitialisation, se/PatchData.hpp:168-223 */
  memset(d->reversed, 0, 2 * n);
  memset(d->coeffs, 0, sizeof(double) * nrhs * p->ncells_max * ndofs);
  memset(d->jumpG, 0, sizeof(double) * nqf * 4);

  /* --- pre-evaluation, :297-424 --- */
  for (int a = 1; a <= n; ++a)
  {
    const int id_a = a - 1;
    const int32_t c = p->cells[a];
    const int32_t* cn = m->cell_nodes + 3 * (size_t)c;
    const double *x0 = m->x + 3 * (size_t)cn[0], *x1 = m->x + 3 * (size_t)cn[1],
                 *x2 = m->x + 3 * (size_t)cn[2];
    /* affine Jacobian (base/KernelData.cpp:66-90): J(i,j) = dx_i/dX_j */
    double* J = d->J + 4 * id_a;
    double* K = d->K + 4 * id_a;
    J[0] = x1[0] - x0[0];
    J[1] = x2[0] - x0[0];
    J[2] = x1[1] - x0[1];
    J[3] = x2[1] - x0[1];
    const double detJ = J[0] * J[3] - J[1] * J[2];
    K[0] = J[3] / detJ;
    K[1] = -J[1] / detJ;
    K[2] = -J[2] / detJ;
    K[3] = J[0] / detJ;
    d->detJ[id_a] = detJ;

    int8_t fl_eam1, fl_ea;
    fctid_local_pair(p, a, &fl_eam1, &fl_ea);
    const int nout_eam1 = tab->fct_normal_out[fl_eam1], nout_ea = tab->fct_normal_out[fl_ea];

    /* reversed facets :324-389 */
    if (on_boundary && (a == 1 || a == n))
    {
      if (a == 1)
      {
        if (n > 1)
        {
          const int32_t c_ap1 = p->cells[a + 1];
          const int8_t fl_tap1_ea = fctid_local(p, a, a + 1);
          if (m->facet_perm[3 * (size_t)c + fl_ea] != m->facet_perm[3 * (size_t)c_ap1 + fl_tap1_ea])
            d->reversed[2 * id_a + 1] = 1;
        }
      }
      else
      {
        const int32_t c_am1 = p->cells[a - 1];
        const int8_t fl_tam1_eam1 = fctid_local(p, a - 1, a - 1);
        if (m->facet_perm[3 * (size_t)c_am1 + fl_tam1_eam1] != m->facet_perm[3 * (size_t)c + fl_eam1])
          d->reversed[2 * id_a] = 1;
      }
    }
    else
    {
      const int32_t c_am1 = p->cells[a - 1], c_ap1 = p->cells[a + 1];
      const int8_t fl_tam1_eam1 = fctid_local(p, a - 1, a - 1);
      const int8_t fl_tap1_ea = fctid_local(p, a, a + 1);
      if (m->facet_perm[3 * (size_t)c_am1 + fl_tam1_eam1] != m->facet_perm[3 * (size_t)c + fl_eam1])
        d->reversed[2 * id_a] = 1;
      if (m->facet_perm[3 * (size_t)c + fl_ea] != m->facet_perm[3 * (size_t)c_ap1 + fl_tap1_ea])
        d->reversed[2 * id_a + 1] = 1;
    }

    const double sgn = detJ / fabs(detJ);
    d->prefactor[2 * id_a] = nout_eam1 ? sgn : -sgn;
    d->prefactor[2 * id_a + 1] = nout_ea ? sgn : -sgn;

    /* push-back of the interpolation matrix :399-423 */
    for (int i = 0; i < k + 1; ++i)
      for (int q = 0; q < nqf; ++q)
      {
        const int fctid = (i == 0) ? fl_eam1 : fl_ea;
        const int ii = (i < 2) ? 0 : i - 1;
        const double m0 = MREF(tab, fctid, ii, 0, q), m1 = MREF(tab, fctid, ii, 1, q);
        MM(d, tab, id_a, i, 0, q) = detJ * (m0 * K[0] + m1 * K[2]);
        MM(d, tab, id_a, i, 1, q) = detJ * (m0 * K[1] + m1 * K[3]);
        if (ii > 0 && d->reversed[2 * id_a + 1] && a != n)
        {
          MM(d, tab, id_a, i, 0, q) -= MM(d, tab, id_a, 1, 0, q);
          MM(d, tab, id_a, i, 1, q) -= MM(d, tab, id_a, 1, 1, q);
        }
      }
  }

  set_assembly_informations(p, d, tab->fct_normal_out);

  const int offs_ffEa = p->offs[1], offs_fcdiv = p->offs[4];
  const int ndofs_hdivz = p->ndof_min_flux;
  const int ndofs_hdivz_per_cell = 2 * k + p->nadd;
  int status = 0;

  for (int r = 0; r < nrhs; ++r)
  {
    const int type_patch = p->type[r];
    const int reversion = reversion_required(p, r);
    const double* x_flux_proj = flux_dg + (size_t)r * ncells_mesh * nd * 2;
    const double* x_rhs_proj = rhs_dg + (size_t)r * ncells_mesh * nd;
    const double* bvals = boundary_values ? boundary_values + (size_t)r * ncells_mesh * ndofs : NULL;

    double c_ta_ea = 0, c_ta_eam1 = 0, c_tam1_eam1 = 0, c_t1_e0 = 0;
    double* G_Ta = d->G_Ta;
    double* G_Tap1 = d->G_Tap1;

    /* --- Step 1: sigma-tilde, :468-992 --- */
    copy_G(p, x_flux_proj, p->cells[1], G_Tap1);
    memset(d->jumpG, 0, sizeof(double) * nqf * 4);

    for (int a = 1; a <= n; ++a)
    {
      const int id_a = a - 1;
      const int32_t c_a = p->cells[a];
      const int node_Ta = p->inodes_local[a];
      const int node_Tap1 = p->inodes_local[a + 1];
      const int fct_on_boundary = on_boundary && (a == 1 || a == n);
      int fct_has_bc = 0;
      if (fct_on_boundary)
      {
        if (type_patch == PT_ESSNT_DUAL)
          fct_has_bc = 1;
        else if (type_patch == PT_MIXED)
        {
          if (a == 1)
            fct_has_bc = (ftype_at(p, r, p->fcts[0]) == FT_ESSNT_DUAL);
          else if (a == n)
            fct_has_bc = (ftype_at(p, r, p->fcts[n]) == FT_ESSNT_DUAL);
        }
      }
      /* NOTE reference quirk: for a one-cell... n == 1 never happens (error), and for a == 1 == n
       * both branches cannot apply; a boundary patch has n >= 2 here. */

      int8_t fl_TaEam1, fl_TaEa;
      fctid_local_pair(p, a, &fl_TaEam1, &fl_TaEa);
      const int8_t fl_Tap1Ea = (fct_on_boundary && a == n) ? fl_TaEa : fctid_local(p, a, a + 1);

      const double detJ = d->detJ[id_a];
      const double sign_detJ = (detJ > 0.0) ? 1.0 : -1.0;
      const double* K = d->K + 4 * id_a;

      {
        double* t = G_Ta;
        G_Ta = G_Tap1;
        G_Tap1 = t;
      }
      if (!(on_boundary && a == n))
        copy_G(p, x_flux_proj, p->cells[a + 1], G_Tap1);
      memcpy(d->f_Ta, x_rhs_proj + (size_t)c_a * nd, sizeof(double) * nd);

      const int e0_special = (a == 1) && (fct_has_bc || type_patch == PT_MIXED);
      const int32_t* pflux_ldofs_E0 = &FDG(p, 0, 0);

      c_ta_eam1 = -c_tam1_eam1;
      for (int j = 0; j < k; ++j)
        d->cj_ta_ea[j] = 0.0;

      /* flux BCs :585-638: per-patch boundary DOFs hat_a * g from the global boundary DOFs */
      if (fct_has_bc && bvals)
      {
        const int offs_bdofs = (a == 1) ? 0 : k;
        const int lfct_b = (a == 1) ? fl_TaEam1 : fl_TaEa;
        double bpatch[8];
        calculate_patch_bc(tab, bvals + DM(p, 1, a, offs_bdofs), lfct_b, node_Ta, d->J + 4 * id_a,
                           detJ, K, bpatch);
        if (a == 1)
          c_ta_eam1 += d->prefactor[2 * id_a] * bpatch[0];
        for (int j = 1; j < k; ++j)
          COEF(p, d, r, id_a, DM(p, 0, a, offs_bdofs + j)) += bpatch[j];
        if (reversion)
          c_t1_e0 -= d->prefactor[2 * id_a + 1] * bpatch[0];
      }

      double surfint_c_ta_eam1 = 0.0;
      for (int q = 0; q < nqf; ++q)
      {
        double GtHat_Ea[2][2] = {{0, 0}, {0, 0}};
        if (fct_on_boundary)
        {
          if (a == 1)
          {
            if (e0_special)
            {
              /* jump on the boundary facet E0: (0 - hat*G) :657-672 */
              const double* shp = tab->rhs_fct + (size_t)(fl_TaEam1 * nqf + q) * nd;
              for (int i = 0; i < ndf; ++i)
              {
                const int id_Ta = pflux_ldofs_E0[i + ndf];
                JG(d, q, 0, 0) -= G_Ta[2 * id_Ta] * shp[id_Ta];
                JG(d, q, 0, 1) -= G_Ta[2 * id_Ta + 1] * shp[id_Ta];
              }
              const double hat = tab->hat_fct[(size_t)(fl_TaEam1 * nqf + q) * 3 + node_Ta];
              JG(d, q, 0, 0) *= hat;
              JG(d, q, 0, 1) *= hat;
              if (k > 1)
              {
                /* higher-order DOFs on E0 :675-719 */
                double g[2], gm[2];
                const double sg = (type_patch == PT_MIXED && !fct_has_bc) ? -1.0 : 1.0;
                g[0] = sg * JG(d, q, 0, 0);
                g[1] = sg * JG(d, q, 0, 1);
                pull_back_flux(gm, g, detJ, K);
                for (int j = 1; j < k; ++j)
                  COEF(p, d, r, id_a, DM(p, 0, a, j))
                      += MREF(tab, fl_TaEam1, j, 0, q) * gm[0] + MREF(tab, fl_TaEam1, j, 1, q) * gm[1];
              }
              if (!fct_has_bc)
              {
                JG(d, q, 0, 0) = 0.0;
                JG(d, q, 0, 1) = 0.0;
              }
            }
            calculate_jump(p, tab, GtHat_Ea, q, d->reversed[2 * id_a + 1], &FDG(p, a, 0), G_Tap1,
                           fl_Tap1Ea, node_Tap1, G_Ta, fl_TaEa, node_Ta);
          }
          else
          {
            /* last boundary facet En :738-785 */
            const int32_t* dofs_G_Ea = &FDG(p, a, 0);
            const double pfctr = fct_has_bc ? -1.0 : 1.0;
            const double* shp = tab->rhs_fct + (size_t)(fl_TaEa * nqf + q) * nd;
            for (int i = 0; i < ndf; ++i)
            {
              const int id_Ta = dofs_G_Ea[i + ndf];
              GtHat_Ea[1][0] += G_Ta[2 * id_Ta] * pfctr * shp[id_Ta];
              GtHat_Ea[1][1] += G_Ta[2 * id_Ta + 1] * pfctr * shp[id_Ta];
            }
            const double hat = tab->hat_fct[(size_t)(fl_TaEa * nqf + q) * 3 + node_Ta];
            GtHat_Ea[1][0] *= hat;
            GtHat_Ea[1][1] *= hat;
            if (reversion)
            {
              double gm[2];
              pull_back_flux(gm, GtHat_Ea[1], detJ, K);
              const double aux = MREF(tab, fl_TaEa, 0, 0, q) * gm[0] + MREF(tab, fl_TaEa, 0, 1, q) * gm[1];
              c_t1_e0 -= d->prefactor[2 * id_a + 1] * aux;
            }
          }
        }
        else
        {
          calculate_jump(p, tab, GtHat_Ea, q, d->reversed[2 * id_a + 1], &FDG(p, a, 0), G_Tap1,
                         fl_Tap1Ea, node_Tap1, G_Ta, fl_TaEa, node_Ta);
        }

        /* zero-order moment on E_{a-1} from the stored jump :796-800 */
        double jG[2];
        jG[0] = JG(d, q, 1, 0) - JG(d, q, 0, 0);
        jG[1] = JG(d, q, 1, 1) - JG(d, q, 0, 1);
        surfint_c_ta_eam1 -= MM(d, tab, id_a, 0, 0, q) * jG[0] + MM(d, tab, id_a, 0, 1, q) * jG[1];

        /* higher-order moments on E_a :803-821 */
        jG[0] = GtHat_Ea[1][0] - GtHat_Ea[0][0];
        jG[1] = GtHat_Ea[1][1] - GtHat_Ea[0][1];
        for (int j = 2; j < k + 1; ++j)
          d->cj_ta_ea[j - 2] += MM(d, tab, id_a, j, 0, q) * jG[0] + MM(d, tab, id_a, j, 1, q) * jG[1];

        JG(d, q, 0, 0) = GtHat_Ea[0][0];
        JG(d, q, 0, 1) = GtHat_Ea[0][1];
        JG(d, q, 1, 0) = GtHat_Ea[1][0];
        JG(d, q, 1, 1) = GtHat_Ea[1][1];
      }
      /* (:831-842 swaps one component of the stored jump on reversed facets; only the
       *  symmetric-weight zero-order moment consumes it, so the swap has no effect: omitted) */

      c_ta_eam1 += d->prefactor[2 * id_a] * surfint_c_ta_eam1;
      c_t1_e0 -= d->prefactor[2 * id_a] * surfint_c_ta_eam1;

      /* cell integrals :848-934 */
      if (k == 1)
      {
        const double vol_int = d->f_Ta[0] * (fabs(detJ) / 6);
        c_ta_ea = vol_int - c_ta_eam1;
        c_t1_e0 += vol_int;
      }
      else
      {
        /* mapped gradients of the DG basis (se/KernelData.cpp:225-247) */
        for (int q = 0; q < nq; ++q)
          for (int i = 0; i < nd; ++i)
          {
            const double dX = tab->rhs_cell[((size_t)1 * nq + q) * nd + i];
            const double dY = tab->rhs_cell[((size_t)2 * nq + q) * nd + i];
            d->rhs_cur[((size_t)0 * nq + q) * nd + i] = K[0] * dX + K[2] * dY;
            d->rhs_cur[((size_t)1 * nq + q) * nd + i] = K[1] * dX + K[3] * dY;
          }
        c_ta_ea = -c_ta_eam1;
        for (int i = 0; i < p->ndiv; ++i)
          d->c_ta_div[i] = 0.0;
        for (int q = 0; q < nq; ++q)
        {
          double f = 0.0, div_g = 0.0;
          for (int i = 0; i < nd; ++i)
          {
            f += d->f_Ta[i] * tab->rhs_cell[(size_t)q * nd + i];
            div_g += G_Ta[2 * i] * d->rhs_cur[((size_t)0 * nq + q) * nd + i]
                     + G_Ta[2 * i + 1] * d->rhs_cur[((size_t)1 * nq + q) * nd + i];
          }
          const double aux = (f - div_g) * tab->hat_cell[(size_t)q * 3 + node_Ta] * tab->qweights[q] * detJ;
          const double vol_int = aux * sign_detJ;
          c_ta_ea += vol_int;
          c_t1_e0 += vol_int;
          int count = 0;
          for (int l = 0; l < k; ++l)
            for (int mm = 0; mm < k - l; ++mm)
              if (l + mm > 0)
              {
                d->c_ta_div[count] += aux * pow(tab->qpoints[2 * q], l) * pow(tab->qpoints[2 * q + 1], mm);
                ++count;
              }
        }
      }

      /* correction of higher-order facet moments on reversed facets :937-951 */
      if (k > 1 && d->reversed[2 * id_a + 1] && a != n)
        for (int i = 1; i < k; ++i)
          d->cj_ta_ea[i - 1] += d->prefactor[2 * (id_a + 1)] * c_ta_ea;

      /* store :953-988 */
      COEF(p, d, r, id_a, DM(p, 0, a, 0)) += d->prefactor[2 * id_a] * c_ta_eam1;
      COEF(p, d, r, id_a, DM(p, 0, a, offs_ffEa)) += d->prefactor[2 * id_a + 1] * c_ta_ea;
      for (int i = 1; i < k; ++i)
        COEF(p, d, r, id_a, DM(p, 0, a, offs_ffEa + i)) += d->cj_ta_ea[i - 1];
      for (int i = 0; i < p->ndiv; ++i)
        COEF(p, d, r, id_a, DM(p, 0, a, offs_fcdiv + i)) += d->c_ta_div[i];

      c_tam1_eam1 = c_ta_ea;
    }

    /* reversed (mixed) patch :995-1027 */
    if (reversion)
      for (int a = 1; a <= n; ++a)
      {
        const int id_a = a - 1;
        COEF(p, d, r, id_a, DM(p, 0, a, 0)) += d->prefactor[2 * id_a] * c_t1_e0;
        COEF(p, d, r, id_a, DM(p, 0, a, offs_ffEa)) -= d->prefactor[2 * id_a + 1] * c_t1_e0;
        if (k > 1 && d->reversed[2 * id_a + 1] && a != n)
          for (int i = 1; i < k; ++i)
            COEF(p, d, r, id_a, DM(p, 0, a, offs_ffEa + i)) -= d->prefactor[2 * (id_a + 1)] * c_t1_e0;
      }

    if (out_sigma_tilde)
      for (int a = 0; a < n; ++a)
        memcpy(out_sigma_tilde + ((size_t)r * n + a) * ndofs, &COEF(p, d, r, a, 0), sizeof(double) * ndofs);

    /* --- Step 2: minimisation :1029-1078 --- */
    const int req_bc = requires_flux_bcs_rhs(p, r);
    if (req_bc)
      set_boundary_markers(d->bmarkers, ndofs_hdivz, type_patch, reversion, n, k);

    int assemble_entire_system = (r == 0);
    if (r > 0 && on_boundary && (p->type[r] != p->type[r - 1] || p->type[r] == PT_MIXED))
      assemble_entire_system = 1;

    assemble_fluxminimiser(p, d, tab, r, req_bc, assemble_entire_system);
    if (assemble_entire_system && k > 1)
    {
      int st = factorise_A(d, ndofs_hdivz);
      if (st)
        status = st;
    }
    if (k == 1)
      d->u[0] = d->L[0] / d->A[0];
    else
      solve_A(d, ndofs_hdivz);
    if (out_u)
      memcpy(out_u + (size_t)r * d->dim_max, d->u, sizeof(double) * ndofs_hdivz);

    /* --- back-map H(div=0) -> RT and scatter :1080-1161 --- */
    double* x_flux = flux_hdiv ? flux_hdiv + (size_t)r * ncells_mesh * ndofs : NULL;
    for (int a = 1; a <= n; ++a)
    {
      const int id_a = a - 1;
      int start_i = 0;
      if (d->reversed[2 * id_a])
      {
        for (int i = 0; i < k; ++i)
        {
          double local_value = 0.0;
          for (int j = 0; j < k; ++j)
            local_value += tab->doftrafo[j * k + i] * DM(p, 3, a, j) * d->u[DM(p, 2, a, j)];
          COEF(p, d, r, id_a, DM(p, 0, a, i)) += local_value;
        }
        start_i = k;
      }
      for (int i = start_i; i < ndofs_hdivz_per_cell; ++i)
        COEF(p, d, r, id_a, DM(p, 0, a, i)) += DM(p, 3, a, i) * d->u[DM(p, 2, a, i)];

      if (x_flux)
      {
        const size_t g0 = (size_t)p->cells[a] * ndofs;
        for (int i = 0; i < ndofs; ++i)
          x_flux[g0 + i] += COEF(p, d, r, id_a, i);
      }
      if (out_patch)
        memcpy(out_patch + ((size_t)r * n + id_a) * ndofs, &COEF(p, d, r, id_a, 0), sizeof(double) * ndofs);
    }
  }
  return status;
}

/* ------------------------------------------------------------------------------------ */
/* Weak symmetry of the stress (rows 0 and 1), se/solve_patch_weaksym.hpp:59-233          */
/* ------------------------------------------------------------------------------------ */

/* generate_stress_minimisation_kernel, se/stressmin_kernel.hpp:76-248.  Te (as in the flux
 * kernel) is written to d->Te when assemble_A; Be [nh][6], Ce [3], Le2 [2 nh + 3]. */
static void stressmin_kernel(const patch_t* p, pdata_t* d, const oracle_tables_t* tab, int a,
                             const double* coefficients /* [2][ndofs] */, int assemble_A)
{
  const int k = p->k, ndofs = p->ndofs, nh = d->nh, nq = tab->nq;
  const int id_a = a - 1;
  const double detJ = d->detJ[id_a];
  const double* J = d->J + 4 * id_a;
  const uint8_t eam1_reversed = d->reversed[2 * id_a];
  const int offs_c = p->offs[3];
  const int offs_Lc = 2 * nh;
  if (assemble_A)
    memset(d->Te, 0, sizeof(double) * (nh + 1) * nh);
  memset(d->Be, 0, sizeof(double) * nh * 6);
  memset(d->Ce, 0, sizeof(double) * 3);
  memset(d->Le2, 0, sizeof(double) * (2 * nh + 3));

  const double inv = 1.0 / detJ;
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < ndofs; ++i)
    {
      const double* r = tab->flux_basis + ((size_t)q * ndofs + i) * 2;
      double* c = d->phi + ((size_t)q * ndofs + i) * 2;
      c[0] = inv * J[0] * r[0] + inv * J[1] * r[1];
      c[1] = inv * J[2] * r[0] + inv * J[3] * r[1];
    }
  const int ld0_eam1 = DM(p, 0, a, 0), ld0_ea = DM(p, 0, a, k);
  const int p_eam1 = DM(p, 3, a, 0), p_ea = DM(p, 3, a, k);

  for (int q = 0; q < nq; ++q)
  {
    double* phi = d->phi + (size_t)q * ndofs * 2;
    double sig_r0[2] = {0, 0}, sig_r1[2] = {0, 0};
    for (int i = 0; i < ndofs; ++i)
    {
      sig_r0[0] += coefficients[i] * phi[2 * i];
      sig_r0[1] += coefficients[i] * phi[2 * i + 1];
      sig_r1[0] += coefficients[ndofs + i] * phi[2 * i];
      sig_r1[1] += coefficients[ndofs + i] * phi[2 * i + 1];
    }
    if (eam1_reversed)
    {
      memset(d->gphi, 0, sizeof(double) * 2 * k);
      for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
        {
          int ldj = DM(p, 0, a, j);
          d->gphi[2 * i] += tab->doftrafo[i * k + j] * phi[2 * ldj];
          d->gphi[2 * i + 1] += tab->doftrafo[i * k + j] * phi[2 * ldj + 1];
        }
      for (int i = 0; i < k; ++i)
      {
        int ldi = DM(p, 0, a, i);
        phi[2 * ldi] = d->gphi[2 * i];
        phi[2 * ldi + 1] = d->gphi[2 * i + 1];
      }
    }
    phi[2 * ld0_ea] = p_ea * (p_eam1 * phi[2 * ld0_eam1] + p_ea * phi[2 * ld0_ea]);
    phi[2 * ld0_ea + 1] = p_ea * (p_eam1 * phi[2 * ld0_eam1 + 1] + p_ea * phi[2 * ld0_ea + 1]);

    const double dvol = tab->qweights[q] * fabs(detJ);
    for (int i = 0; i < nh; ++i)
    {
      const int ip1 = i + 1;
      const int dl_i = DM(p, 0, a, ip1);
      const double alpha = DM(p, 3, a, ip1) * dvol;
      const double phi_i0 = phi[2 * dl_i] * alpha, phi_i1 = phi[2 * dl_i + 1] * alpha;
      if (assemble_A)
      {
        d->Le2[i] -= sig_r0[0] * phi_i0 + sig_r0[1] * phi_i1;
        d->Le2[nh + i] -= sig_r1[0] * phi_i0 + sig_r1[1] * phi_i1;
        for (int j = i; j < nh; ++j)
        {
          const int jp1 = j + 1;
          const int dl_j = DM(p, 0, a, jp1);
          d->Te[i * nh + j] += phi_i0 * phi[2 * dl_j] * DM(p, 3, a, jp1)
                               + phi_i1 * phi[2 * dl_j + 1] * DM(p, 3, a, jp1);
        }
      }
      for (int j = 0; j < 3; ++j)
      {
        const double phi_j = tab->hat_cell[(size_t)q * 3 + DM(p, 0, a, offs_c + j)];
        d->Be[i * 6 + j] += phi_i1 * phi_j;
        d->Be[i * 6 + 3 + j] -= phi_i0 * phi_j;
      }
    }
    for (int i = 0; i < 3; ++i)
    {
      const double phi_i = tab->hat_cell[(size_t)q * 3 + DM(p, 0, a, offs_c + i)] * dvol;
      d->Ce[i] += phi_i;
      d->Le2[offs_Lc + i] -= phi_i * (sig_r0[1] - sig_r1[0]);
    }
  }
  if (assemble_A)
    for (int i = 1; i < nh; ++i)
      for (int j = 0; j < i; ++j)
        d->Te[i * nh + j] = d->Te[j * nh + i];
}

/* Eigen::PartialPivLU stand-in: solve M x = b in place, M [n][ld] destroyed */
static int lu_solve(double* M, int ld, int n, double* b)
{
  for (int c = 0; c < n; ++c)
  {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabs(M[r * ld + c]) > fabs(M[piv * ld + c]))
        piv = r;
    if (M[piv * ld + c] == 0.0)
      return -2;
    if (piv != c)
    {
      for (int j = 0; j < n; ++j)
      {
        double t = M[c * ld + j];
        M[c * ld + j] = M[piv * ld + j];
        M[piv * ld + j] = t;
      }
      double t = b[c];
      b[c] = b[piv];
      b[piv] = t;
    }
    for (int r = c + 1; r < n; ++r)
    {
      const double f = M[r * ld + c] / M[c * ld + c];
      for (int j = c; j < n; ++j)
        M[r * ld + j] -= f * M[c * ld + j];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r)
  {
    double t = b[r];
    for (int j = r + 1; j < n; ++j)
      t -= M[r * ld + j] * b[j];
    b[r] = t / M[r * ld + r];
  }
  return 0;
}

/* impose_weak_symmetry<T,k,false>: assembly (se/assembly.hpp:292-472), Schur solve
 * (se/PatchData.hpp:598-663), scatter (se/solve_patch_weaksym.hpp:189-232).
 * Requires equilibrate_patch to have run for this patch (coefficients of rows 0, 1, mapping
 * data, and - without flux BCs - the Cholesky factor of A). */
static int impose_weak_symmetry(patch_t* p, pdata_t* d, const oracle_tables_t* tab,
                                double* flux_hdiv, int modified_patch)
{
  const int k = p->k, n = p->ncells, ndofs = p->ndofs, nh = d->nh, dm = d->dim_max;
  const int gdim = 2;
  const int dim = p->ndof_min_flux;
  const int on_boundary = (p->type[0] != PT_INTERNAL);
  const int npnt = p->nfcts + 1;
  const int ldB = 2 * d->npnt_max, ldC = d->npnt_max + 1;
  const size_t ncells_mesh = p->m->ncells;
  int status = 0;

  /* PatchData::reinitialisation :175-206 */
  d->meanvalue_required = 1;
  if (on_boundary)
  {
    int cnt = 0;
    for (int i = 0; i < gdim; ++i)
      if (p->type[i] == PT_ESSNT_PRIMAL || p->type[i] == PT_MIXED)
        ++cnt;
    if (cnt > 0)
      d->meanvalue_required = 0;
  }
  int requires_bcs = 0;
  int types[2] = {PT_INTERNAL, PT_INTERNAL}, revs[2] = {0, 0};
  if (on_boundary)
    for (int i = 0; i < gdim; ++i)
    {
      types[i] = p->type[i];
      revs[i] = reversion_required(p, i);
      if (types[i] == PT_ESSNT_DUAL || types[i] == PT_MIXED)
        requires_bcs = 1;
    }
  /* stress coefficients = patch-local result of steps 1+2 :134-142; grouped patches
   * (modified_patch, :100-131): the values accumulated in the global stress so far */
  for (int r = 0; r < gdim; ++r)
    for (int a = 1; a <= n; ++a)
    {
      if (modified_patch)
        memcpy(d->cstress + ((size_t)(a - 1) * gdim + r) * ndofs,
               flux_hdiv + ((size_t)r * ncells_mesh + p->cells[a]) * ndofs, sizeof(double) * ndofs);
      else
        memcpy(d->cstress + ((size_t)(a - 1) * gdim + r) * ndofs, &COEF(p, d, r, a - 1, 0),
               sizeof(double) * ndofs);
    }
  memset(d->bmarkers2, 0, 2 * dm);
  if (on_boundary)
    for (int i = 0; i < gdim; ++i)
      set_boundary_markers(d->bmarkers2 + i * dim, dim, types[i], revs[i], n, k);
  /* NB: set_boundary_markers of the reference handles all rows in one call with offset
   * i*ndofs_hdivz (se/assembly.hpp:59-97); the two calls above are the same thing. */

  if (requires_bcs)
    memset(d->A_rec, 0, sizeof(double) * dm * dm);
  memset(d->Bm, 0, sizeof(double) * dm * ldB);
  memset(d->Cm, 0, sizeof(double) * ldC * ldC);
  memset(d->Lfull, 0, sizeof(double) * (2 * dm + d->npnt_max + 1));
  const int offs_c = p->offs[3];
  for (int a = 1; a <= n; ++a)
  {
    stressmin_kernel(p, d, tab, a, d->cstress + (size_t)(a - 1) * gdim * ndofs, requires_bcs);
    for (int kk = 0; kk < gdim; ++kk)
    {
      for (int i = 0; i < nh; ++i)
      {
        const int dof_i = DM(p, 2, a, i + 1);
        if (requires_bcs)
        {
          d->Lfull[kk * dm + dof_i] += d->Le2[kk * nh + i];
          if (kk == 0)
            for (int j = 0; j < nh; ++j)
              d->A_rec[dof_i * dm + DM(p, 2, a, j + 1)] += d->Te[i * nh + j];
        }
        for (int j = 0; j < 3; ++j)
        {
          const int dof_j = DM(p, 2, a, offs_c + j);
          if (!requires_bcs || !d->bmarkers2[kk * dim + dof_i])
            d->Bm[dof_i * ldB + kk * d->npnt_max + dof_j] += d->Be[i * 6 + kk * 3 + j];
        }
      }
      if (kk == 0)
        for (int i = 0; i < 3; ++i)
        {
          const int dof_i = DM(p, 2, a, offs_c + i);
          if (d->meanvalue_required)
          {
            d->Cm[dof_i * ldC + npnt] += d->Ce[i];
            d->Cm[npnt * ldC + dof_i] += d->Ce[i];
          }
          d->Lfull[2 * dm + dof_i] += d->Le2[2 * nh + i];
        }
    }
  }

  /* solve_constrained_minimisation, se/PatchData.hpp:598-663 */
  for (int kk = 0; kk < gdim; ++kk)
  {
    if (requires_bcs)
    {
      /* apply_bcs_on_A :735-768 + factorise */
      memset(d->A, 0, sizeof(double) * dm * dm);
      for (int i = 0; i < dim; ++i)
      {
        if (d->bmarkers2[kk * dim + i])
        {
          d->Lfull[kk * dm + i] = 0.0;
          d->A[i * dm + i] = 1.0;
        }
        else
          for (int j = 0; j < dim; ++j)
            d->A[i * dm + j] = d->bmarkers2[kk * dim + j] ? 0.0 : d->A_rec[i * dm + j];
      }
      if (k > 1 && factorise_A(d, dim))
        status = -2;
    }
    for (int c = 0; c < npnt; ++c)
    {
      for (int i = 0; i < dim; ++i)
        d->L[i] = d->Bm[i * ldB + kk * d->npnt_max + c];
      if (k == 1)
        d->u[0] = d->L[0] / d->A[0];
      else
        solve_A(d, dim);
      for (int i = 0; i < dim; ++i)
        d->AinvB[i * d->npnt_max + c] = d->u[i];
    }
    for (int r = 0; r < npnt; ++r)
      for (int c = 0; c < npnt; ++c)
      {
        double t = 0.0;
        for (int i = 0; i < dim; ++i)
          t += d->Bm[i * ldB + kk * d->npnt_max + r] * d->AinvB[i * d->npnt_max + c];
        d->Cm[r * ldC + c] -= t;
      }
  }
  const int dim_c = d->meanvalue_required ? npnt + 1 : npnt;
  for (int i = 0; i < dim_c; ++i)
    d->u_c[i] = d->Lfull[2 * dm + i];
  if (lu_solve(d->Cm, ldC, dim_c, d->u_c))
    status = -2;
  for (int kk = gdim - 1; kk >= 0; --kk)
  {
    if (requires_bcs && kk != gdim - 1)
    {
      memset(d->A, 0, sizeof(double) * dm * dm);
      for (int i = 0; i < dim; ++i)
      {
        if (d->bmarkers2[kk * dim + i])
          d->A[i * dm + i] = 1.0;
        else
          for (int j = 0; j < dim; ++j)
            d->A[i * dm + j] = d->bmarkers2[kk * dim + j] ? 0.0 : d->A_rec[i * dm + j];
      }
      if (k > 1 && factorise_A(d, dim))
        status = -2;
    }
    for (int i = 0; i < dim; ++i)
    {
      double t = 0.0;
      for (int c = 0; c < npnt; ++c)
        t -= d->Bm[i * ldB + kk * d->npnt_max + c] * d->u_c[c];
      d->L[i] = t;
    }
    if (k == 1)
      d->u[0] = d->L[0] / d->A[0];
    else
      solve_A(d, dim);
    memcpy(d->u_sig2 + (size_t)kk * dm, d->u, sizeof(double) * dim);
  }

  /* scatter :189-232 */
  const int ndofs_hdivz_per_cell = 2 * k + p->nadd;
  for (int r = 0; r < gdim; ++r)
  {
    double* x = flux_hdiv + (size_t)r * ncells_mesh * ndofs;
    const double* u = d->u_sig2 + (size_t)r * dm;
    for (int a = 1; a <= n; ++a)
    {
      int start_j = 0;
      if (d->reversed[2 * (a - 1)])
      {
        for (int j = 0; j < k; ++j)
        {
          double v = 0.0;
          for (int kk = 0; kk < k; ++kk)
            v += tab->doftrafo[kk * k + j] * DM(p, 3, a, kk) * u[DM(p, 2, a, kk)];
          x[DM(p, 1, a, j)] += v;
        }
        start_j = k;
      }
      for (int j = start_j; j < ndofs_hdivz_per_cell; ++j)
        x[DM(p, 1, a, j)] += DM(p, 3, a, j) * u[DM(p, 2, a, j)];
    }
  }
  return status;
}

/* OrientedPatch::estimate_squared_korn_constant, se/Patch.cpp:130-334 (Kim's bound
 * 2 / sin^2(theta_min / 2) for star-shaped patches) */
static double estimate_squared_korn_constant(const patch_t* p)
{
  const oracle_mesh_t* m = p->m;
  const double* x = m->x;
  const int node_patch = p->node;
  const int n = p->ncells, nf = p->nfcts;
  double theta_min;
  if (p->type[0] == PT_INTERNAL)
  {
    const double xi0 = x[3 * (size_t)node_patch], xi1 = x[3 * (size_t)node_patch + 1];
    theta_min = 0.5 * M_PI;
    /* cells() of an interior patch = _cells[0 .. n+1] (se/Patch.hpp:222-232): T_n, T_1..T_n, T_1 */
    for (int a = 0; a < n + 2; ++a)
    {
      const int32_t cell = p->cells[a];
      const int32_t* cn = m->cell_nodes + 3 * (size_t)cell;
      int32_t b[2];
      int cnt = 0;
      for (int j = 0; j < 3; ++j)
        if (cn[j] != node_patch)
          b[cnt++] = 3 * cn[j];
      double v2[2] = {x[b[1]] - x[b[0]], x[b[1] + 1] - x[b[0] + 1]};
      const double abs_v2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1]);
      double v1[2] = {xi0 - x[b[0]], xi1 - x[b[0] + 1]};
      double abs_v1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1]);
      double v1_t_v2 = v1[0] * v2[0] + v1[1] * v2[1];
      theta_min = fmin(theta_min, acos(v1_t_v2 / (abs_v1 * abs_v2)));
      v1[0] = xi0 - x[b[1]];
      v1[1] = xi1 - x[b[1] + 1];
      abs_v1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1]);
      v1_t_v2 = v1[0] * v2[0] + v1[1] * v2[1];
      theta_min = fmin(theta_min, acos(-v1_t_v2 / (abs_v1 * abs_v2)));
    }
  }
  else
  {
    double cnodes[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    double phi_min[3] = {M_PI, M_PI, M_PI};
    if (n % 2 == 0)
    {
      const int h = n / 2;
      for (int i = 0; i < 2; ++i)
      {
        const int32_t* en = m->cell_nodes + 3 * (size_t)p->cells[h + i];
        for (int j = 0; j < 3; ++j)
        {
          cnodes[i][0] += x[3 * (size_t)en[j]] / 3;
          cnodes[i][1] += x[3 * (size_t)en[j] + 1] / 3;
        }
      }
      const int32_t* en = m->facet_nodes + 2 * (size_t)p->fcts[h];
      for (int j = 0; j < 2; ++j)
      {
        cnodes[2][0] += 0.5 * x[3 * (size_t)en[j]];
        cnodes[2][1] += 0.5 * x[3 * (size_t)en[j] + 1];
      }
    }
    else
    {
      const int h = nf / 2;
      for (int i = 0; i < 2; ++i)
      {
        const int32_t* en = m->facet_nodes + 2 * (size_t)p->fcts[h - i];
        for (int j = 0; j < 2; ++j)
        {
          cnodes[i][0] += 0.5 * x[3 * (size_t)en[j]];
          cnodes[i][1] += 0.5 * x[3 * (size_t)en[j] + 1];
        }
      }
      const int32_t* en = m->cell_nodes + 3 * (size_t)p->cells[h];
      for (int j = 0; j < 3; ++j)
      {
        cnodes[2][0] += x[3 * (size_t)en[j]] / 3;
        cnodes[2][1] += x[3 * (size_t)en[j] + 1] / 3;
      }
    }
    /* walk the patch boundary starting at the patch node (:273-321) */
    int32_t node_i = node_patch, idn_i = 3 * node_i;
    const int32_t* en = m->facet_nodes + 2 * (size_t)p->fcts[n];
    int32_t node_im1 = (en[0] == node_i) ? en[1] : en[0];
    int32_t idn_im1 = 3 * node_im1;
    double v2[2] = {x[idn_im1] - x[idn_i], x[idn_im1 + 1] - x[idn_i + 1]};
    double abs_v2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1]);
    for (int i = 0; i < nf; ++i)
    {
      const int32_t* e2 = m->facet_nodes + 2 * (size_t)p->fcts[i];
      const int32_t node_ip1 = (e2[0] == node_patch) ? e2[1] : e2[0];
      const int32_t idn_ip1 = 3 * node_ip1;
      double v3[2] = {x[idn_ip1] - x[idn_i], x[idn_ip1 + 1] - x[idn_i + 1]};
      const double abs_v3 = sqrt(v3[0] * v3[0] + v3[1] * v3[1]);
      for (int j = 0; j < 3; ++j)
      {
        const double v1[2] = {cnodes[j][0] - x[idn_i], cnodes[j][1] - x[idn_i + 1]};
        const double abs_v1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1]);
        double d = v1[0] * v2[0] + v1[1] * v2[1];
        phi_min[j] = fmin(phi_min[j], acos(d / (abs_v1 * abs_v2)));
        d = v1[0] * v3[0] + v1[1] * v3[1];
        phi_min[j] = fmin(phi_min[j], acos(d / (abs_v1 * abs_v3)));
      }
      node_i = node_ip1;
      idn_i = idn_ip1;
      v2[0] = -v3[0];
      v2[1] = -v3[1];
      abs_v2 = abs_v3;
    }
    theta_min = 0.0;
    for (int j = 0; j < 3; ++j)
      theta_min = fmax(theta_min, phi_min[j]);
  }
  return 2 * pow(sin(theta_min / 2), -2);
}

/* Korn part of the node loop, se/reconstruction.hpp:291-304: every patch adds
 * (gdim + 1) * c_K^2 to all its cells; korn [ncells] is accumulated. */
int oracle_se_korn(const oracle_mesh_t* mesh, int nrhs, const int8_t* facet_type, double* korn,
                   int32_t node_begin, int32_t node_end)
{
  patch_t p;
  pdata_t d;
  oracle_tables_t tab;
  memset(&tab, 0, sizeof(tab));
  tab.k = 1;
  tab.ndofs = 3;
  tab.nd = 1;
  tab.ndf = 1;
  tab.nq = 1;
  tab.nqf = 1;
  patch_alloc(&p, &d, mesh, &tab, nrhs, facet_type);
  for (int32_t node = node_begin; node < node_end; ++node)
  {
    initialize_patch(&p, node);
    const double cks = estimate_squared_korn_constant(&p) * 3;
    for (int a = 1; a <= p.ncells; ++a)
      korn[p.cells[a]] += cks;
  }
  patch_free(&p, &d);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* public entry points                                                                  */
/* ------------------------------------------------------------------------------------ */
int oracle_build_patches(const oracle_mesh_t* mesh, int nrhs, const int8_t* facet_type,
                         int32_t node_begin, int32_t node_end, int32_t stride,
                         int32_t* ncells, int32_t* cells, int32_t* fcts, int8_t* fcts_local,
                         int8_t* inodes_local, int8_t* types)
{
  patch_t p;
  pdata_t d;
  /* minimal k=1 tables are enough for the topology part */
  oracle_tables_t tab;
  memset(&tab, 0, sizeof(tab));
  tab.k = 1;
  tab.ndofs = 3;
  tab.nd = 1;
  tab.ndf = 1;
  tab.nq = 1;
  tab.nqf = 1;
  patch_alloc(&p, &d, mesh, &tab, nrhs, facet_type);
  if (stride < p.ncells_max + 2)
  {
    patch_free(&p, &d);
    return -3;
  }
  for (int32_t node = node_begin; node < node_end; ++node)
  {
    const size_t o = (size_t)(node - node_begin);
    initialize_patch(&p, node);
    const int n = p.ncells;
    const int internal = (p.type[0] == PT_INTERNAL);
    ncells[o] = n;
    int32_t* c = cells + o * stride;
    int32_t* f = fcts + o * stride;
    int8_t* fl = fcts_local + o * 2 * stride;
    int8_t* il = inodes_local + o * stride;
    for (int i = 0; i < stride; ++i)
    {
      c[i] = -1;
      f[i] = -1;
      il[i] = -1;
      fl[2 * i] = fl[2 * i + 1] = -1;
    }
    for (int a = internal ? 0 : 1; a <= (internal ? n + 1 : n); ++a)
    {
      c[a] = p.cells[a];
      il[a] = p.inodes_local[a];
    }
    for (int a = 0; a <= n; ++a)
    {
      f[a] = p.fcts[a];
      fl[2 * a] = p.fcts_local[2 * a];
      fl[2 * a + 1] = p.fcts_local[2 * a + 1];
    }
    memcpy(types + o * nrhs, p.type, nrhs);
  }
  patch_free(&p, &d);
  return 0;
}

static int se_reconstruct_impl(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                               const int8_t* facet_type, const double* boundary_values,
                               const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                               int32_t node_begin, int32_t node_end, int stress);

int oracle_se_reconstruct(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                          const int8_t* facet_type, const double* boundary_values,
                          const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                          int32_t node_begin, int32_t node_end)
{
  return se_reconstruct_impl(mesh, tab, nrhs, facet_type, boundary_values, flux_dg, rhs_dg,
                             flux_hdiv, node_begin, node_end, 0);
}

/* se::reconstruction with reconstruct_stress = true: grouped boundary patches
 * (se/reconstruction.hpp:170-234, only with flux BCs on the stress and RT_2), then "all other
 * patches" (:237-270). */
int oracle_se_reconstruct_stress(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                                 const int8_t* facet_type, const double* boundary_values,
                                 const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                                 int32_t node_begin, int32_t node_end)
{
  if (nrhs < 2 || tab->k < 2)
    return -4; /* se/reconstruction.hpp:376-388 */
  return se_reconstruct_impl(mesh, tab, nrhs, facet_type, boundary_values, flux_dg, rhs_dg,
                             flux_hdiv, node_begin, node_end, 1);
}

static int se_reconstruct_impl(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                               const int8_t* facet_type, const double* boundary_values,
                               const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                               int32_t node_begin, int32_t node_end, int stress)
{
  /* OrientedPatch::set_max_patch_size, se/Patch.cpp:337-404 (ncells_min = 1): checks the
   * nodes the loop visits (size_local owned nodes in the reference) */
  for (int i = node_begin; i < node_end; ++i)
    if (mesh->node_cells_off[i + 1] - mesh->node_cells_off[i] == 1)
      return -1;
  patch_t p;
  pdata_t d;
  patch_alloc(&p, &d, mesh, tab, nrhs, facet_type);
  int status = 0;
  const int nn = mesh->nnodes;
  uint8_t* perform = (uint8_t*)malloc((size_t)nn);
  memset(perform, 1, (size_t)nn);
  /* grouped boundary patches, se/reconstruction.hpp:170-234 (RT_2 only): a node whose two boundary
   * facets carry flux BCs on both stress rows (base/BoundaryData.cpp:611-631) and that has only two
   * cells is equilibrated together with the adjacent internal patch (se/Patch.cpp:60-104,762-784);
   * the weak symmetry is imposed once, on the internal patch, with the stress accumulated so far */
  if (stress && tab->k == 2)
  {
    int8_t* on_bnd = (int8_t*)calloc((size_t)nn, 1);
    for (int r = 0; r < 2; ++r)
      for (int32_t f = 0; f < mesh->nfacets; ++f)
        if (facet_type[(size_t)r * mesh->nfacets + f] == FT_ESSNT_DUAL)
        {
          on_bnd[mesh->facet_nodes[2 * (size_t)f]] += 1;
          on_bnd[mesh->facet_nodes[2 * (size_t)f + 1]] += 1;
        }
    for (int i = 0; i < nn; ++i)
      on_bnd[i] = (on_bnd[i] == 4);
    int32_t group[64];
    for (int32_t node = node_begin; node < node_end; ++node)
    {
      if (!(on_bnd[node] && perform[node]))
        continue;
      if (mesh->node_cells_off[node + 1] - mesh->node_cells_off[node] != 2)
        continue;
      /* adjacent internal patch: other node of the first internal facet of the node */
      int32_t inner = -1;
      for (int32_t q = mesh->node_facets_off[node]; q < mesh->node_facets_off[node + 1]; ++q)
      {
        const int32_t f = mesh->node_facets[q];
        if (facet_type[f] == FT_INTERNAL)
        {
          const int32_t* fn = mesh->facet_nodes + 2 * (size_t)f;
          inner = (fn[0] == node) ? fn[1] : fn[0];
          break;
        }
      }
      int ng = 0;
      group[ng++] = inner;
      for (int32_t q = mesh->node_cells_off[inner]; q < mesh->node_cells_off[inner + 1]; ++q)
      {
        const int32_t* cn = mesh->cell_nodes + 3 * (size_t)mesh->node_cells[q];
        for (int v = 0; v < 3; ++v)
        {
          const int32_t pnt = cn[v];
          if (!on_bnd[pnt])
            continue;
          int seen = 0;
          for (int g = 0; g < ng; ++g)
            seen |= (group[g] == pnt);
          if (!seen && mesh->node_cells_off[pnt + 1] - mesh->node_cells_off[pnt] == 2 && ng < 64)
            group[ng++] = pnt;
        }
      }
      if (ng < 2)
        continue;
      for (int g = ng - 1; g >= 0; --g)
      {
        const int32_t node_i = group[g];
        if (!perform[node_i])
        {
          status = -5; /* "Incompatible mesh! To many patches with 2 cells on neumann boundary." */
          continue;
        }
        perform[node_i] = 0;
        create_subdofmap(&p, node_i);
        set_fctdofs_dg(&p, tab);
        const int st = equilibrate_patch(&p, &d, tab, boundary_values, flux_dg, rhs_dg, flux_hdiv, NULL,
                                         NULL, NULL);
        if (st)
          status = st;
      }
      const int st = impose_weak_symmetry(&p, &d, tab, flux_hdiv, 1);
      if (st)
        status = st;
    }
    free(on_bnd);
  }
  /* se/reconstruction.hpp:237-270 / 286-313: all other patches */
  for (int32_t node = node_begin; node < node_end; ++node)
  {
    if (!perform[node])
      continue;
    create_subdofmap(&p, node);
    set_fctdofs_dg(&p, tab);
    int st = equilibrate_patch(&p, &d, tab, boundary_values, flux_dg, rhs_dg, flux_hdiv, NULL,
                               NULL, NULL);
    if (st)
      status = st;
    if (stress)
    {
      st = impose_weak_symmetry(&p, &d, tab, flux_hdiv, 0);
      if (st)
        status = st;
    }
  }
  free(perform);
  patch_free(&p, &d);
  return status;
}

int oracle_se_patch(const oracle_mesh_t* mesh, const oracle_tables_t* tab, int nrhs,
                    const int8_t* facet_type, const double* boundary_values,
                    const double* flux_dg, const double* rhs_dg, int32_t node,
                    double* out_sigma_tilde, double* out_patch, int32_t* out_cells,
                    double* out_u)
{
  patch_t p;
  pdata_t d;
  patch_alloc(&p, &d, mesh, tab, nrhs, facet_type);
  create_subdofmap(&p, node);
  set_fctdofs_dg(&p, tab);
  int st = equilibrate_patch(&p, &d, tab, boundary_values, flux_dg, rhs_dg, NULL,
                             out_sigma_tilde, out_patch, out_u);
  if (out_cells)
    memcpy(out_cells, p.cells + 1, sizeof(int32_t) * p.ncells);
  int n = p.ncells;
  patch_free(&p, &d);
  return st ? st : n;
}

/* constrained-minimisation (EV) equilibrator, shares the helpers above */
#include "eqlb_oracle_ev.c"
