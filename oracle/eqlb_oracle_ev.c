/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT PATH (see eqlb_oracle.h).
 *
 * CPU restatement of the reference's constrained-minimisation equilibration (Ern & Vohralik):
 * per patch a mixed RT_k x DG_{k-1} saddle-point system with one Lagrange row, factorised by
 * a dense partial-pivot LU, exactly the structure of
 *   ev/reconstruction.hpp:32-176 (node loop), ev/Patch.cpp:482-676 (patch DOF lists),
 *   ev/assembly.hpp:53-87 (lifting), :121-307 (assembly, mean-value row), ev/solve_patch.hpp:58-238
 *   (LU + scatter), eqlb/FluxEqlbEV.py:113-134 (the forms).
 * This file is #included at the end of eqlb_oracle.c and shares its patch fan, geometry and
 * tabulation helpers.  `file:line` citations are relative to /root/reference/cpp/dolfinx_eqlb/.
 *
 * What replaces the third-party pieces of the reference (PARITY UNPINNED BY EXECUTION):
 *  - FFCx tabulate_tensor kernels of the three forms -> the quadrature loops of ev_element();
 *  - Basix RT_k + DOLFINx DOF transformations -> the CONFORMING version of the hierarchic RT_k of
 *    a18 (same functionals; facet moments taken w.r.t. the global facet frame: parameter s from
 *    the lower to the higher node id, normal n_E = (t_y, -t_x) of the tangent t = x_hi - x_lo).
 *    A cell sees its facet DOFs through T_f = -I (cell traverses the facet low->high) or
 *    T_f = B (against; B_ji = C(j,i)(-1)^i): c_local = T_f g_global.  Both are involutions, so the
 *    "transform" and "transform to transpose" steps of ev/assembly.hpp:185-186 are T^T Ae T.
 * The patch-local numbering is [facet DOFs | cell-interior flux DOFs | DG DOFs | multiplier]
 * instead of the cell-interleaved order of ev/Patch.cpp:505-640; the LU solution does not
 * depend on it.
 */

typedef struct
{
  int k, nrt, ni, nd, ne; /* ne = nrt + nd DOFs of the mixed element */
  int nmax, ndof_max;
  double *Ae, *Pe, *Le;   /* per patch cell: [nmax][ne*ne], [nmax][nd], [ne] */
  double *A, *Acopy, *L;  /* patch system */
  int32_t* pdof;          /* [nmax][ne] patch-local DOF or -1 (outer facet) */
  int32_t* gdof;          /* [nmax][nrt] global conforming flux DOF */
  int8_t* marked;         /* [ndof_max] */
  double* bval;           /* [ndof_max] */
  double* Bmat;           /* [k][k] */
} ev_data_t;

static void ev_alloc(ev_data_t* e, const oracle_tables_t* tab, int nmax)
{
  memset(e, 0, sizeof(*e));
  const int k = e->k = tab->k;
  e->nrt = tab->ndofs;
  e->ni = k * k - k;
  e->nd = tab->nd;
  e->ne = e->nrt + e->nd;
  e->nmax = nmax;
  e->ndof_max = k * (nmax + 1) + nmax * (e->ni + e->nd) + 1;
  e->Ae = xcalloc((size_t)nmax * e->ne * e->ne, sizeof(double));
  e->Pe = xcalloc((size_t)nmax * e->nd, sizeof(double));
  e->Le = xcalloc(e->ne, sizeof(double));
  e->A = xcalloc((size_t)e->ndof_max * e->ndof_max, sizeof(double));
  e->Acopy = xcalloc((size_t)e->ndof_max * e->ndof_max, sizeof(double));
  e->L = xcalloc(e->ndof_max, sizeof(double));
  e->pdof = xcalloc((size_t)nmax * e->ne, sizeof(int32_t));
  e->gdof = xcalloc((size_t)nmax * e->nrt, sizeof(int32_t));
  e->marked = xcalloc(e->ndof_max, 1);
  e->bval = xcalloc(e->ndof_max, sizeof(double));
  e->Bmat = xcalloc((size_t)k * k, sizeof(double));
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i)
    {
      double c = 1.0; /* C(j,i) */
      for (int t = 0; t < i; ++t)
        c = c * (j - t) / (t + 1);
      e->Bmat[j * k + i] = (i <= j) ? ((i % 2) ? -c : c) : 0.0;
    }
}

static void ev_free(ev_data_t* e)
{
  free(e->Ae);
  free(e->Pe);
  free(e->Le);
  free(e->A);
  free(e->Acopy);
  free(e->L);
  free(e->pdof);
  free(e->gdof);
  free(e->marked);
  free(e->bval);
  free(e->Bmat);
}

/* T_f of the header: out = T_f in (k values) */
static void ev_facet_map(const ev_data_t* e, int rev, const double* in, double* out)
{
  const int k = e->k;
  for (int j = 0; j < k; ++j)
  {
    if (!rev)
      out[j] = -in[j];
    else
    {
      double s = 0.0;
      for (int i = 0; i < k; ++i)
        s += e->Bmat[j * k + i] * in[i];
      out[j] = s;
    }
  }
}

/* Cell tensors of a = (sig,v) - (r,div v) + (div sig,q), l_pen = (1,q) and of
 * l = hat G.v + (hat f + grad hat . G) q  (FluxEqlbEV.py:113-134 with list_proj_flux = G), then
 * T^T . T on the facet DOFs (ev/assembly.hpp:163-196). */
static void ev_element(const ev_data_t* e, const oracle_mesh_t* m, const oracle_tables_t* tab,
                       int32_t cell, int hat_id, const double* G /* [nd][2] */,
                       const double* f /* [nd] */, double* Ae, double* Pe, double* Le,
                       const double* flux_div)
{
  const int nrt = e->nrt, nd = e->nd, ne = e->ne, nq = tab->nq, k = e->k;
  const int32_t* cn = m->cell_nodes + 3 * (size_t)cell;
  const double *x0 = m->x + 3 * (size_t)cn[0], *x1 = m->x + 3 * (size_t)cn[1],
               *x2 = m->x + 3 * (size_t)cn[2];
  const double J[4] = {x1[0] - x0[0], x2[0] - x0[0], x1[1] - x0[1], x2[1] - x0[1]};
  const double detJ = J[0] * J[3] - J[1] * J[2];
  const double K[4] = {J[3] / detJ, -J[1] / detJ, -J[2] / detJ, J[0] / detJ};
  static const double ghat_ref[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  const double ghat[2] = {K[0] * ghat_ref[hat_id][0] + K[2] * ghat_ref[hat_id][1],
                          K[1] * ghat_ref[hat_id][0] + K[3] * ghat_ref[hat_id][1]};
  if (Ae)
  {
    memset(Ae, 0, sizeof(double) * ne * ne);
    memset(Pe, 0, sizeof(double) * nd);
  }
  memset(Le, 0, sizeof(double) * ne);
  double phi[64][2], dphi[64];
  for (int q = 0; q < nq; ++q)
  {
    const double dvol = tab->qweights[q] * fabs(detJ);
    const double* psi = tab->rhs_cell + (size_t)q * nd;
    for (int i = 0; i < nrt; ++i)
    {
      const double* r = tab->flux_basis + ((size_t)q * nrt + i) * 2;
      phi[i][0] = (J[0] * r[0] + J[1] * r[1]) / detJ;
      phi[i][1] = (J[2] * r[0] + J[3] * r[1]) / detJ;
      dphi[i] = flux_div[(size_t)q * nrt + i] / detJ;
    }
    double Gq[2] = {0, 0}, fq = 0.0;
    for (int i = 0; i < nd; ++i)
    {
      Gq[0] += psi[i] * G[2 * i];
      Gq[1] += psi[i] * G[2 * i + 1];
      fq += psi[i] * f[i];
    }
    const double hat = tab->hat_cell[(size_t)q * 3 + hat_id];
    for (int i = 0; i < nrt; ++i)
      Le[i] += hat * (Gq[0] * phi[i][0] + Gq[1] * phi[i][1]) * dvol;
    for (int mm = 0; mm < nd; ++mm)
      Le[nrt + mm] += (hat * fq + ghat[0] * Gq[0] + ghat[1] * Gq[1]) * psi[mm] * dvol;
    if (Ae)
    {
      for (int i = 0; i < nrt; ++i)
      {
        for (int j = 0; j < nrt; ++j)
          Ae[i * ne + j] += (phi[i][0] * phi[j][0] + phi[i][1] * phi[j][1]) * dvol;
        for (int mm = 0; mm < nd; ++mm)
        {
          Ae[i * ne + nrt + mm] -= psi[mm] * dphi[i] * dvol;
          Ae[(nrt + mm) * ne + i] += psi[mm] * dphi[i] * dvol;
        }
      }
      for (int mm = 0; mm < nd; ++mm)
        Pe[mm] += psi[mm] * dvol;
    }
  }
  /* DOF transformation T^T Ae T / T^T Le on the three facet blocks */
  double tmp[8], out[8];
  for (int fl = 0; fl < 3; ++fl)
  {
    const int rev = m->facet_perm[3 * (size_t)cell + fl];
    /* T^T acting on a row-block vector: (T^T w)_i = sum_j T_ji w_j */
    for (int pass = 0; pass < (Ae ? 1 + 2 * ne : 1); ++pass)
    {
      /* pass 0: Le; 1..ne: columns (rows transformed); ne+1..2ne: rows (columns transformed) */
      for (int j = 0; j < k; ++j)
      {
        if (pass == 0)
          tmp[j] = Le[fl * k + j];
        else if (pass <= ne)
          tmp[j] = Ae[(fl * k + j) * ne + (pass - 1)];
        else
          tmp[j] = Ae[(pass - 1 - ne) * ne + fl * k + j];
      }
      for (int i = 0; i < k; ++i)
      {
        double s = 0.0;
        for (int j = 0; j < k; ++j)
          s += (rev ? e->Bmat[j * k + i] : ((i == j) ? -1.0 : 0.0)) * tmp[j];
        out[i] = s;
      }
      for (int j = 0; j < k; ++j)
      {
        if (pass == 0)
          Le[fl * k + j] = out[j];
        else if (pass <= ne)
          Ae[(fl * k + j) * ne + (pass - 1)] = out[j];
        else
          Ae[(pass - 1 - ne) * ne + fl * k + j] = out[j];
      }
    }
  }
}

/* One patch: ev/solve_patch.hpp:58-238.  cell_dofs [ncells][nrt]: conforming dofmap. */
static int ev_equilibrate_patch(patch_t* p, ev_data_t* e, const oracle_tables_t* tab,
                                const double* flux_div, const int32_t* cell_dofs, int64_t ndofs_glob,
                                const double* boundary_values, const double* flux_dg,
                                const double* rhs_dg, double* flux_hdiv, double* out_u)
{
  const oracle_mesh_t* m = p->m;
  const int k = e->k, n = p->ncells, nrt = e->nrt, ni = e->ni, nd = e->nd, ne = e->ne;
  const int internal = (p->type[0] == PT_INTERNAL);
  const int nf = p->nfcts;
  const int ndof_patch = nf * k + n * (ni + nd); /* ev/Patch.cpp:492 */
  const int N = ndof_patch + 1;
  const size_t ncm = m->ncells;

  /* patch DOF lists (ev/Patch.cpp:505-676) */
  for (int a = 1; a <= n; ++a)
  {
    const int32_t c = p->cells[a];
    int8_t fm, fp;
    fctid_local_pair(p, a, &fm, &fp);
    const int jm = a - 1, jp = (internal && a == n) ? 0 : a;
    int32_t* pd = e->pdof + (size_t)(a - 1) * ne;
    for (int i = 0; i < ne; ++i)
      pd[i] = -1; /* DOFs of the facet opposite to the patch node stay zero */
    for (int i = 0; i < k; ++i)
    {
      pd[fm * k + i] = jm * k + i;
      pd[fp * k + i] = jp * k + i;
    }
    for (int i = 0; i < ni; ++i)
      pd[3 * k + i] = nf * k + (a - 1) * ni + i;
    for (int i = 0; i < nd; ++i)
      pd[nrt + i] = nf * k + n * ni + (a - 1) * nd + i;
    for (int i = 0; i < nrt; ++i)
      e->gdof[(size_t)(a - 1) * nrt + i] = cell_dofs[(size_t)c * nrt + i];
  }

  int status = 0;
  for (int r = 0; r < p->nrhs; ++r)
  {
    const double* G = flux_dg + (size_t)r * ncm * nd * 2;
    const double* f = rhs_dg + (size_t)r * ncm * nd;
    const double* bglob = boundary_values ? boundary_values + (size_t)r * ndofs_glob : NULL;
    const int type = p->type[r];
    const int req_bc = (type == PT_ESSNT_DUAL || type == PT_MIXED);

    /* per-patch boundary values hat_a * g on the flux-BC end facets
     * (ev/solve_patch.hpp:124-136, base/BoundaryData.cpp:687-745) */
    memset(e->marked, 0, N);
    memset(e->bval, 0, sizeof(double) * N);
    if (req_bc)
    {
      for (int side = 0; side < 2; ++side)
      {
        const int fj = side ? n : 0, a = side ? n : 1;
        const int32_t fct = p->fcts[fj];
        if (ftype_at(p, r, fct) != FT_ESSNT_DUAL)
          continue;
        const int32_t c = p->cells[a];
        int8_t fm, fp;
        fctid_local_pair(p, a, &fm, &fp);
        const int lf = side ? fp : fm;
        const int rev = m->facet_perm[3 * (size_t)c + lf];
        double gl[8] = {0}, cl[8], cp[8], gp[8];
        if (bglob)
          for (int i = 0; i < k; ++i)
            gl[i] = bglob[cell_dofs[(size_t)c * nrt + lf * k + i]];
        ev_facet_map(e, rev, gl, cl);
        const int32_t* cn = m->cell_nodes + 3 * (size_t)c;
        const double *x0 = m->x + 3 * (size_t)cn[0], *x1 = m->x + 3 * (size_t)cn[1],
                     *x2 = m->x + 3 * (size_t)cn[2];
        const double J[4] = {x1[0] - x0[0], x2[0] - x0[0], x1[1] - x0[1], x2[1] - x0[1]};
        const double detJ = J[0] * J[3] - J[1] * J[2];
        const double K[4] = {J[3] / detJ, -J[1] / detJ, -J[2] / detJ, J[0] / detJ};
        calculate_patch_bc(tab, cl, lf, p->inodes_local[a], J, detJ, K, cp);
        ev_facet_map(e, rev, cp, gp);
        for (int i = 0; i < k; ++i)
        {
          e->marked[fj * k + i] = 1;
          e->bval[fj * k + i] = gp[i];
        }
      }
    }

    /* assemble (ev/assembly.hpp:121-307); the matrix is rebuilt whenever the reference would
     * re-assemble it (ev/solve_patch.hpp:159-176), else the stored copy is reused */
    int assemble_A = (r == 0);
    if (r > 0 && !internal && (type != p->type[r - 1] || type == PT_MIXED))
      assemble_A = 1;
    if (assemble_A)
      memset(e->Acopy, 0, sizeof(double) * N * N);
    memset(e->L, 0, sizeof(double) * N);
    for (int a = 1; a <= n; ++a)
    {
      const int32_t c = p->cells[a];
      double* Ae = e->Ae + (size_t)(a - 1) * ne * ne;
      double* Pe = e->Pe + (size_t)(a - 1) * nd;
      ev_element(e, m, tab, c, p->inodes_local[a], G + (size_t)c * nd * 2, f + (size_t)c * nd,
                 (r == 0) ? Ae : NULL, Pe, e->Le, flux_div);
      const int32_t* pd = e->pdof + (size_t)(a - 1) * ne;
      double* Le = e->Le;
      /* lifting, ev/assembly.hpp:53-87 */
      if (req_bc)
        for (int kk = 0; kk < ne; ++kk)
        {
          if (pd[kk] < 0 || e->marked[pd[kk]])
            continue;
          for (int l = 0; l < ne; ++l)
            if (pd[l] >= 0 && e->marked[pd[l]])
              Le[kk] -= Ae[kk * ne + l] * e->bval[pd[l]];
        }
      for (int kk = 0; kk < ne; ++kk)
      {
        const int pk = pd[kk];
        if (pk < 0)
          continue;
        if (req_bc && e->marked[pk])
        {
          if (assemble_A)
            e->Acopy[(size_t)pk * N + pk] = 1.0;
          e->L[pk] = e->bval[pk];
          continue;
        }
        e->L[pk] += Le[kk];
        if (assemble_A)
          for (int l = 0; l < ne; ++l)
          {
            const int pl = pd[l];
            if (pl < 0 || (req_bc && e->marked[pl]))
              continue;
            e->Acopy[(size_t)pk * N + pl] += Ae[kk * ne + l];
          }
      }
      /* mean-value constraint, ev/assembly.hpp:281-305 */
      if (assemble_A)
      {
        if (type == PT_INTERNAL || type == PT_ESSNT_DUAL)
          for (int i = 0; i < nd; ++i)
          {
            e->Acopy[(size_t)pd[nrt + i] * N + (N - 1)] += Pe[i];
            e->Acopy[(size_t)(N - 1) * N + pd[nrt + i]] += Pe[i];
          }
        else
          e->Acopy[(size_t)(N - 1) * N + (N - 1)] = 1.0;
      }
    }
    /* PartialPivLU + solve, ev/solve_patch.hpp:197,213 (refactorised per RHS here) */
    memcpy(e->A, e->Acopy, sizeof(double) * N * N);
    const int st = lu_solve(e->A, N, N, e->L);
    if (st)
      status = st;
    /* scatter, ev/solve_patch.hpp:223-227: every patch flux DOF once */
    if (flux_hdiv)
    {
      double* x = flux_hdiv + (size_t)r * ndofs_glob;
      for (int a = 1; a <= n; ++a)
      {
        const int32_t* pd = e->pdof + (size_t)(a - 1) * ne;
        const int32_t* gd = e->gdof + (size_t)(a - 1) * nrt;
        int8_t fm, fp;
        fctid_local_pair(p, a, &fm, &fp);
        for (int i = 0; i < k; ++i)
        {
          /* a facet is shared by two cells: add it from the cell that has it as E_{a-1};
           * the last facet of a boundary patch from its only cell */
          x[gd[fm * k + i]] += e->L[pd[fm * k + i]];
          if (!internal && a == n)
            x[gd[fp * k + i]] += e->L[pd[fp * k + i]];
        }
        for (int i = 0; i < ni; ++i)
          x[gd[3 * k + i]] += e->L[pd[3 * k + i]];
      }
    }
    if (out_u)
      memcpy(out_u + (size_t)r * e->ndof_max, e->L, sizeof(double) * N);
  }
  return status;
}

/* ev::reconstruction (ev/reconstruction.hpp:32-176): loop over the nodes.
 *   cell_dofs       [ncells][k(k+2)] conforming dofmap (local hierarchic order), ndofs_glob DOFs
 *   boundary_values [nrhs][ndofs_glob] global boundary DOFs (facet DOFs on flux-BC facets) or NULL
 *   flux_div        [nq][k(k+2)] reference divergence of the RT basis at the cell points
 *   flux_hdiv       [nrhs][ndofs_glob] accumulated */
int oracle_ev_reconstruct(const oracle_mesh_t* mesh, const oracle_tables_t* tab,
                          const double* flux_div, int nrhs, const int8_t* facet_type,
                          const int32_t* cell_dofs, int64_t ndofs_glob,
                          const double* boundary_values, const double* flux_dg,
                          const double* rhs_dg, double* flux_hdiv, int32_t node_begin,
                          int32_t node_end)
{
  for (int i = node_begin; i < node_end; ++i)
    if (mesh->node_cells_off[i + 1] - mesh->node_cells_off[i] == 1)
      return -1;
  patch_t p;
  pdata_t d;
  ev_data_t e;
  patch_alloc(&p, &d, mesh, tab, nrhs, facet_type);
  ev_alloc(&e, tab, p.ncells_max);
  int status = 0;
  for (int32_t node = node_begin; node < node_end; ++node)
  {
    initialize_patch(&p, node);
    const int st = ev_equilibrate_patch(&p, &e, tab, flux_div, cell_dofs, ndofs_glob,
                                        boundary_values, flux_dg, rhs_dg, flux_hdiv, NULL);
    if (st)
      status = st;
  }
  ev_free(&e);
  patch_free(&p, &d);
  return status;
}

/* single patch, for tests: u [nrhs][ndof_max] in the patch-local numbering of the header,
 * returns the number of patch cells (cells written to out_cells) */
int oracle_ev_patch(const oracle_mesh_t* mesh, const oracle_tables_t* tab, const double* flux_div,
                    int nrhs, const int8_t* facet_type, const int32_t* cell_dofs,
                    int64_t ndofs_glob, const double* boundary_values, const double* flux_dg,
                    const double* rhs_dg, int32_t node, int32_t* out_cells, double* out_u,
                    int32_t* out_ndof_max)
{
  patch_t p;
  pdata_t d;
  ev_data_t e;
  patch_alloc(&p, &d, mesh, tab, nrhs, facet_type);
  ev_alloc(&e, tab, p.ncells_max);
  initialize_patch(&p, node);
  const int st = ev_equilibrate_patch(&p, &e, tab, flux_div, cell_dofs, ndofs_glob,
                                      boundary_values, flux_dg, rhs_dg, NULL, out_u);
  if (out_cells)
    memcpy(out_cells, p.cells + 1, sizeof(int32_t) * p.ncells);
  if (out_ndof_max)
    *out_ndof_max = e.ndof_max;
  const int n = p.ncells;
  ev_free(&e);
  patch_free(&p, &d);
  return st ? st : n;
}
