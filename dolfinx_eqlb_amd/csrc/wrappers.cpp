// pybind11 module `dolfinx_eqlb_amd._cpp`: the names, argument order and error behaviour of the
// reference's binding `dolfinx_eqlb.cpp` (python/dolfinx_eqlb/wrappers.cpp:52-272) over the C ABI of
// libeqlb_amd.so (include/eqlb.h).
//
//   reference (DOLFINx objects)                          here
//   ------------------------------------------------    -------------------------------------------------
//   dolfinx::mesh::Mesh                                  Mesh(x, cell_nodes, ..., facet_perm)  flat arrays
//   dolfinx::fem::FunctionSpace                          FunctionSpace(mesh, family, degree, bs, discontinuous)
//   dolfinx::fem::Function<double>                       Function(V[, array]) / Function.from_device(V, ptr)
//   dolfinx::fem::Constant<double>                       Constant(values)
//   dolfinx::fem::Form<double>                           Form(coefficients) / Form.from_point_values(...)
//   FluxBC(function_space, facets, pointer_boundary_kernel, nevals_per_fct[, quadrature_degree],
//          coefficients, position_of_coefficients, constants)              wrappers.cpp:144-232  same
//   BoundaryData(list_of_bcs, list_of_boundary_fluxes, V_flux_hdiv, rtflux_is_custom, quadrature_degree,
//                list_bfcts_prime, reconstruct_stress)                     wrappers.cpp:235-256  same
//   reconstruct_fluxes_semiexplt[_with_kornconst], reconstruct_fluxes_minimisation,
//   local_solver_lu / _cholesky / _cg                                      wrappers.cpp:52-137   same
//
// `pointer_boundary_kernel` is the address of a function with the signature of
// ufcx_expression::tabulate_tensor_float64 (the reference passes the ufcx_expression and takes that
// member, wrappers.cpp:164-172; FFCx' header is not part of this build).  The DOLFINx-object overloads
// (zero-copy from Function.x.array) belong in dolfinx_adapter.h, compiled only where <dolfinx/...> exists.
// C++ exceptions (std::runtime_error) arrive in Python as RuntimeError, as in the reference.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/eqlb.h"

namespace py = pybind11;

namespace
{
using darray = py::array_t<double, py::array::c_style | py::array::forcecast>;
template <typename T>
using carray = py::array_t<T, py::array::c_style | py::array::forcecast>;

[[noreturn]] void raise_last(int status)
{
  const char* msg = eqlb_last_error();
  throw std::runtime_error((msg && *msg) ? std::string(msg) : "eqlb error " + std::to_string(status));
}
inline void check(int status)
{
  if (status != EQLB_OK)
    raise_last(status);
}

void* g_stream = nullptr; // hipStream_t of device-memory calls (set_stream)

// ---- stand-ins of the DOLFINx objects -------------------------------------------------------------
struct Mesh
{
  eqlb_mesh_t* h = nullptr;
  int32_t nnodes = 0, ncells = 0, nfacets = 0;
  std::vector<double> x;
  std::vector<int32_t> cell_nodes, cell_facets, facet_nodes, facet_cells_off, facet_cells;
  std::vector<uint8_t> facet_perm;

  Mesh(carray<double> x_, carray<int32_t> cn, carray<int32_t> cf, carray<int32_t> fn, carray<int32_t> fco,
       carray<int32_t> fc, carray<int32_t> nco, carray<int32_t> nc, carray<int32_t> nfo, carray<int32_t> nf,
       carray<uint8_t> fp)
  {
    if (x_.ndim() != 2 || x_.shape(1) != 3 || cn.ndim() != 2 || cn.shape(1) != 3 || fn.ndim() != 2)
      throw std::runtime_error("Mesh: x [nnodes, 3], cell_nodes [ncells, 3], facet_nodes [nfacets, 2] expected");
    nnodes = (int32_t)x_.shape(0);
    ncells = (int32_t)cn.shape(0);
    nfacets = (int32_t)fn.shape(0);
    if (cf.size() != (py::ssize_t)ncells * 3 || fp.size() != (py::ssize_t)ncells * 3 || fco.size() != nfacets + 1
        || nco.size() != nnodes + 1 || nfo.size() != nnodes + 1)
      throw std::runtime_error("Mesh: inconsistent array sizes");
    x.assign(x_.data(), x_.data() + x_.size());
    cell_nodes.assign(cn.data(), cn.data() + cn.size());
    cell_facets.assign(cf.data(), cf.data() + cf.size());
    facet_nodes.assign(fn.data(), fn.data() + fn.size());
    facet_cells_off.assign(fco.data(), fco.data() + fco.size());
    facet_cells.assign(fc.data(), fc.data() + fc.size());
    facet_perm.assign(fp.data(), fp.data() + fp.size());
    check(eqlb_mesh_create(nnodes, ncells, nfacets, x.data(), cell_nodes.data(), cell_facets.data(),
                           facet_nodes.data(), fco.data(), fc.data(), nco.data(), nc.data(), nfo.data(),
                           nf.data(), fp.data(), &h));
  }
  ~Mesh() { eqlb_mesh_destroy(h); }
  Mesh(const Mesh&) = delete;
  Mesh& operator=(const Mesh&) = delete;

  // affine map of a cell: J (dx_i/dX_j), detJ
  double jacobian(int32_t c, double J[2][2]) const
  {
    const int32_t* v = &cell_nodes[3 * (size_t)c];
    for (int i = 0; i < 2; ++i)
    {
      J[i][0] = x[3 * (size_t)v[1] + i] - x[3 * (size_t)v[0] + i];
      J[i][1] = x[3 * (size_t)v[2] + i] - x[3 * (size_t)v[0] + i];
    }
    return J[0][0] * J[1][1] - J[0][1] * J[1][0];
  }
};

// family "DG": discontinuous Lagrange P_degree (block size bs); family "RT": the hierarchic RT_degree of
// create_hierarchic_rt (elmtlib/e_raviart_thomas.py:14-196) - discontinuous: global DOF = cell * k(k+2) +
// local (the semi-explicit flux space, FluxEqlbSE.py:98-101); conforming: k DOFs per facet in the global
// facet frame, then k^2 - k per cell (stand-in of the Basix RT_k space of FluxEqlbEV.py:100), numbered
// facet * k + j, nfacets * k + cell * (k^2 - k) + i unless a cell -> DOF table is given
struct FunctionSpace
{
  std::shared_ptr<Mesh> mesh;
  std::string family;
  int degree, bs;
  bool discontinuous;
  std::vector<int32_t> cell_dofs; // conforming RT only, optional
  int64_t ndofs_user = 0;

  FunctionSpace(std::shared_ptr<Mesh> m, std::string fam, int deg, int bs_, bool disc)
      : mesh(std::move(m)), family(std::move(fam)), degree(deg), bs(bs_), discontinuous(disc)
  {
    if (!mesh)
      throw std::runtime_error("FunctionSpace: mesh is None");
    if (family != "DG" && family != "RT")
      throw std::runtime_error("FunctionSpace: family must be 'DG' or 'RT'");
    if (family == "DG" && (degree < 0 || !discontinuous))
      throw std::runtime_error("FunctionSpace: DG spaces are discontinuous, degree >= 0");
    if (family == "RT" && (degree < 1 || bs != 1))
      throw std::runtime_error("FunctionSpace: RT_k needs k >= 1 and block size 1");
  }
  int ndofs_cell() const
  {
    return family == "DG" ? (degree + 1) * (degree + 2) / 2 : degree * (degree + 2);
  }
  int64_t ndofs() const // scalar DOFs x block size
  {
    if (family == "DG" || discontinuous)
      return (int64_t)mesh->ncells * ndofs_cell() * bs;
    if (!cell_dofs.empty())
      return ndofs_user;
    return (int64_t)mesh->nfacets * degree + (int64_t)mesh->ncells * (degree * degree - degree);
  }
  void set_dofmap(carray<int32_t> cd, int64_t n)
  {
    if (family != "RT" || discontinuous)
      throw std::runtime_error("FunctionSpace.set_dofmap: conforming RT spaces only");
    if (cd.size() != (py::ssize_t)mesh->ncells * ndofs_cell())
      throw std::runtime_error("FunctionSpace.set_dofmap: cell_dofs [ncells, k(k+2)] expected");
    cell_dofs.assign(cd.data(), cd.data() + cd.size());
    ndofs_user = n;
  }
};

struct Function
{
  std::shared_ptr<FunctionSpace> V;
  py::array_t<double> host; // owned or caller's array (zero copy)
  uintptr_t dev = 0;        // device pointer (from_device)
  bool on_device = false;

  explicit Function(std::shared_ptr<FunctionSpace> V_) : V(std::move(V_))
  {
    if (!V)
      throw std::runtime_error("Function: function space is None");
    host = py::array_t<double>(V->ndofs());
    std::fill_n(host.mutable_data(), host.size(), 0.0);
  }
  Function(std::shared_ptr<FunctionSpace> V_, py::array_t<double> a) : V(std::move(V_)), host(std::move(a))
  {
    if (!V)
      throw std::runtime_error("Function: function space is None");
    if (!(host.flags() & py::array::c_style) || !host.writeable() || host.size() != V->ndofs())
      throw std::runtime_error("Function: a writeable C-contiguous float64 array of the size of the space is required");
  }
  static std::shared_ptr<Function> from_device(std::shared_ptr<FunctionSpace> V, uintptr_t ptr)
  {
    if (!ptr)
      throw std::runtime_error("Function.from_device: null pointer");
    auto f = std::shared_ptr<Function>(new Function());
    f->V = std::move(V);
    f->dev = ptr;
    f->on_device = true;
    return f;
  }
  double* data() { return on_device ? reinterpret_cast<double*>(dev) : host.mutable_data(); }
  int64_t size() const { return V->ndofs(); }

private:
  Function() = default;
};

struct Constant
{
  std::vector<double> value;
  explicit Constant(carray<double> v) : value(v.data(), v.data() + v.size()) {}
};

// A compiled form of the reference is fixed by the call it is handed to (FluxEqlbEV.py:113-134: a, l_pen,
// l_i; lsolver/projection.py:54-66: a = (u, v), l_i = (f_i, v)); what varies is its data:
//   Form(coefficients)                       the Functions the form depends on (l_i of EV: [G_i, f_i])
//   Form.from_point_values(qp, qw, values)   l_i of the projector: f_i at the images of a reference rule
struct Form
{
  std::vector<std::shared_ptr<Function>> coefficients;
  darray qpoints, qweights, qvalues;
  bool has_points = false;
  explicit Form(std::vector<std::shared_ptr<Function>> c) : coefficients(std::move(c)) {}
  static std::shared_ptr<Form> from_point_values(darray qp, darray qw, darray qv)
  {
    auto f = std::make_shared<Form>(std::vector<std::shared_ptr<Function>>{});
    if (qp.ndim() != 2 || qp.shape(1) != 2 || qw.size() != qp.shape(0))
      throw std::runtime_error("Form.from_point_values: qpoints [nq, 2], qweights [nq] expected");
    f->qpoints = std::move(qp);
    f->qweights = std::move(qw);
    f->qvalues = std::move(qv);
    f->has_points = true;
    return f;
  }
};

// Gauss-Legendre rule on [0, 1] exact for `degree` (basix.make_quadrature(interval, degree): m = (degree+2)/2)
void facet_rule(int degree, std::vector<double>& s, std::vector<double>& w)
{
  const int m = std::max(1, (degree + 2) / 2);
  s.resize(m);
  w.resize(m);
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < m; ++i)
  {
    double z = std::cos(pi * (i + 0.75) / (m + 0.5)), pp = 1.0;
    for (int it = 0; it < 100; ++it)
    {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 1; j <= m; ++j)
      {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pp = m * (z * p1 - p2) / (z * z - 1.0);
      const double dz = p1 / pp;
      z -= dz;
      if (std::fabs(dz) < 1e-15)
        break;
    }
    s[m - 1 - i] = 0.5 * (1.0 + z);
    w[m - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
  }
}

// interpolation rule of the hierarchic RT_k facet functionals (e_raviart_thomas.py:63-71)
inline int interpolation_degree(int k) { return k == 1 ? 1 : 2 * k; }

using bkernel_t = void (*)(double*, const double*, const double*, const double*, const int*, const uint8_t*);

struct FluxBC
{
  std::shared_ptr<FunctionSpace> V;
  std::vector<int32_t> facets;
  bkernel_t kernel;
  int nevals, qdegree;
  bool projection;
  std::vector<std::shared_ptr<Function>> coefficients;
  std::vector<int> positions;
  std::vector<std::shared_ptr<Constant>> constants;

  FluxBC(std::shared_ptr<FunctionSpace> V_, std::vector<int32_t> fcts, uintptr_t kptr, int nev, int qdeg,
         bool proj, std::vector<std::shared_ptr<Function>> coeffs, std::vector<int> pos,
         std::vector<std::shared_ptr<Constant>> consts)
      : V(std::move(V_)), facets(std::move(fcts)), kernel(reinterpret_cast<bkernel_t>(kptr)), nevals(nev),
        qdegree(qdeg), projection(proj), coefficients(std::move(coeffs)), positions(std::move(pos)),
        constants(std::move(consts))
  {
    if (!V)
      throw std::runtime_error("FluxBC: function space is None");
    for (const auto& c : coefficients)
      if (!c || c->on_device || c->V->family != "DG")
        throw std::runtime_error("FluxBC: coefficients are host Functions of DG spaces");
  }
};

inline double binom(int n, int r)
{
  double v = 1.0;
  for (int i = 0; i < r; ++i)
    v = v * (n - i) / (i + 1);
  return v;
}
// T_f of the conforming hierarchic RT_k: -I (facet_perm 0) or B_ji = C(j,i)(-1)^i; an involution
inline double facet_map(bool rev, int j, int i)
{
  if (!rev)
    return (i == j) ? -1.0 : 0.0;
  return (i > j) ? 0.0 : ((i % 2 == 0) ? 1.0 : -1.0) * binom(j, i);
}

struct BoundaryData
{
  std::shared_ptr<FunctionSpace> V;
  std::shared_ptr<Mesh> mesh;
  int k, nrhs;
  bool custom, stress;
  std::vector<int8_t> facet_type;      // [nrhs][nfacets]
  std::vector<double> boundary_values; // [nrhs][ndofs], empty: homogeneous
  std::vector<std::shared_ptr<Function>> boundary_flux;
  // device handles, created with the first equilibration call (the degree of the projected data is
  // known only then, se/reconstruction.hpp:363-373) and kept
  eqlb_se_t* se = nullptr;
  eqlb_ev_t* ev = nullptr;
  std::vector<double> basis_C, basis_R;
  int se_degree_dg = -1;

  BoundaryData(std::vector<std::vector<std::shared_ptr<FluxBC>>>& list_bcs,
               std::vector<std::shared_ptr<Function>>& bflux, std::shared_ptr<FunctionSpace> V_, bool rt_custom,
               int quadrature_degree, const std::vector<std::vector<int32_t>>& fct_esntbound_prime, bool rstress)
      : V(std::move(V_)), custom(rt_custom), stress(rstress), boundary_flux(bflux)
  {
    if (!V || V->family != "RT")
      throw std::runtime_error("BoundaryData: V_flux_hdiv must be an RT space");
    if (custom != V->discontinuous)
      throw std::runtime_error("BoundaryData: rtflux_is_custom must match the flux space (discontinuous "
                               "hierarchic RT_k for the semi-explicit equilibrator)");
    mesh = V->mesh;
    k = V->degree;
    nrhs = (int)list_bcs.size();
    if ((int)bflux.size() != nrhs || (int)fct_esntbound_prime.size() != nrhs)
      throw std::runtime_error("Size of input data does not match!");
    const int64_t ndofs = V->ndofs();
    const int nrt = k * (k + 2);
    facet_type.assign((size_t)nrhs * mesh->nfacets, (int8_t)EQLB_FACET_INTERNAL);
    bool inhomogeneous = false;
    std::vector<double> sq, wq, vals, cdata, coefs;
    for (int r = 0; r < nrhs; ++r)
    {
      if (!bflux[r] || bflux[r]->on_device || bflux[r]->size() != ndofs)
        throw std::runtime_error("BoundaryData: boundary functions are host Functions of V_flux_hdiv");
      int8_t* ft = &facet_type[(size_t)r * mesh->nfacets];
      for (int32_t f : fct_esntbound_prime[r])
      {
        if (f < 0 || f >= mesh->nfacets)
          throw std::runtime_error("BoundaryData: facet index out of range");
        ft[f] = EQLB_FACET_ESSNT_PRIMAL;
      }
      double* xb = bflux[r]->data();
      for (const auto& bc : list_bcs[r])
      {
        // base/BoundaryData.cpp:437-445: the number of evaluation points must fit the rule in use
        facet_rule(bc->projection ? quadrature_degree : interpolation_degree(k), sq, wq);
        const int nq = (int)sq.size();
        if (nq != bc->nevals)
          throw std::runtime_error("BoundaryData: Number of evaluation points (FluxBC) does not match!");
        cdata.clear();
        for (const auto& c : bc->constants)
          cdata.insert(cdata.end(), c->value.begin(), c->value.end());
        vals.assign((size_t)3 * nq, 0.0);
        for (int32_t fct : bc->facets)
        {
          if (fct < 0 || fct >= mesh->nfacets
              || mesh->facet_cells_off[fct + 1] - mesh->facet_cells_off[fct] != 1)
            throw std::runtime_error("BoundaryData: flux BCs live on boundary facets");
          const int32_t cell = mesh->facet_cells[mesh->facet_cells_off[fct]];
          int lf = 0;
          for (int l = 1; l < 3; ++l)
            if (mesh->cell_facets[3 * (size_t)cell + l] == fct)
              lf = l;
          ft[fct] = EQLB_FACET_ESSNT_DUAL;
          if (!bc->kernel)
            continue; // homogeneous condition (no kernel: zero flux)
          double coords[9];
          for (int v = 0; v < 3; ++v)
            for (int d = 0; d < 3; ++d)
              coords[3 * v + d] = mesh->x[3 * (size_t)mesh->cell_nodes[3 * (size_t)cell + v] + d];
          coefs.clear(); // cell DOFs of the coefficients (FluxBC::extract_coefficients)
          for (const auto& c : bc->coefficients)
          {
            const int n = c->V->ndofs_cell() * c->V->bs;
            const double* p = c->data() + (size_t)cell * n;
            coefs.insert(coefs.end(), p, p + n);
          }
          std::fill(vals.begin(), vals.end(), 0.0);
          bc->kernel(vals.data(), coefs.data(), cdata.data(), coords, nullptr, nullptr);
          // DOF_j = pf_f sign(detJ) |E| int_0^1 g s^j ds : the functional int (detJ K w) . N_f s^j ds of
          // the element for a field with w . n = g (L2 projection of the normal trace and interpolation
          // coincide on the moments, base/BoundaryData.cpp:470-575)
          double J[2][2];
          const double detJ = mesh->jacobian(cell, J);
          const int32_t va = mesh->cell_nodes[3 * (size_t)cell + (lf == 0 ? 1 : 0)];
          const int32_t vb = mesh->cell_nodes[3 * (size_t)cell + (lf == 2 ? 1 : 2)];
          const double ex = mesh->x[3 * (size_t)vb] - mesh->x[3 * (size_t)va];
          const double ey = mesh->x[3 * (size_t)vb + 1] - mesh->x[3 * (size_t)va + 1];
          const double scale = ((lf == 1) ? 1.0 : -1.0) * ((detJ > 0.0) ? 1.0 : -1.0) * std::sqrt(ex * ex + ey * ey);
          double dof[16];
          for (int j = 0; j < k; ++j)
          {
            double acc = 0.0;
            for (int q = 0; q < nq; ++q)
              acc += wq[q] * vals[(size_t)lf * nq + q] * std::pow(sq[q], j);
            dof[j] = scale * acc;
            inhomogeneous = inhomogeneous || dof[j] != 0.0;
          }
          if (custom)
            for (int j = 0; j < k; ++j)
              xb[(size_t)cell * nrt + lf * k + j] = dof[j];
          else
          {
            const bool rev = mesh->facet_perm[3 * (size_t)cell + lf] != 0;
            for (int j = 0; j < k; ++j)
            {
              double g = 0.0;
              for (int i = 0; i < k; ++i)
                g += facet_map(rev, j, i) * dof[i];
              const int64_t d = V->cell_dofs.empty() ? (int64_t)fct * k + j
                                                     : (int64_t)V->cell_dofs[(size_t)cell * nrt + lf * k + j];
              xb[d] = g;
            }
          }
        }
      }
    }
    if (inhomogeneous)
    {
      boundary_values.resize((size_t)nrhs * ndofs);
      for (int r = 0; r < nrhs; ++r)
        std::copy_n(bflux[r]->data(), ndofs, &boundary_values[(size_t)r * ndofs]);
    }
  }
  ~BoundaryData()
  {
    eqlb_se_destroy(se);
    eqlb_ev_destroy(ev);
  }
  BoundaryData(const BoundaryData&) = delete;
  BoundaryData& operator=(const BoundaryData&) = delete;

  eqlb_se_t* se_handle(int degree_dg, bool rstress)
  {
    if (!custom)
      throw std::runtime_error("reconstruct_fluxes_semiexplt: the boundary data belongs to a conforming flux space");
    if (rstress != stress)
      throw std::runtime_error("reconstruct_stress does not match the BoundaryData");
    if (se && se_degree_dg == degree_dg)
      return se;
    eqlb_se_destroy(se);
    se = nullptr;
    check(eqlb_se_create(mesh->h, k, degree_dg, nrhs, stress ? 1 : 0, 0, &se));
    se_degree_dg = degree_dg;
    check(eqlb_se_set_boundary(se, facet_type.data(), boundary_values.empty() ? nullptr : boundary_values.data(),
                               nullptr));
    return se;
  }
  eqlb_ev_t* ev_handle()
  {
    if (custom)
      throw std::runtime_error("reconstruct_fluxes_minimisation: the boundary data belongs to the discontinuous "
                               "(semi-explicit) flux space");
    if (ev)
      return ev;
    check(eqlb_ev_create(mesh->h, k, nrhs, &ev));
    if (!V->cell_dofs.empty())
      check(eqlb_ev_set_dofmap(ev, V->cell_dofs.data(), V->ndofs_user));
    if (!basis_C.empty())
    {
      // the boundary DOFs of this object are facet moments, i.e. hierarchic DOFs: the library converts them
      check(eqlb_ev_set_basis_transform(ev, basis_C.data(), basis_R.empty() ? nullptr : basis_R.data()));
      check(eqlb_ev_set_option(ev, "boundary_basis", 1));
    }
    check(eqlb_ev_set_boundary(ev, facet_type.data(), boundary_values.empty() ? nullptr : boundary_values.data(),
                               nullptr));
    return ev;
  }
  // Output basis of the conforming flux (eqlb_ev_set_basis_transform): C [k(k+2)]^2 from the hierarchic
  // reference coefficients to the target element's, R [k]^2 for the facet block of reflected facets.
  void set_basis_transform(const py::array_t<double, py::array::c_style | py::array::forcecast>& C,
                           const py::object& R)
  {
    if (custom)
      throw std::runtime_error("set_basis_transform: only for the conforming (minimisation) flux space");
    const int nrt = k * (k + 2);
    if (C.ndim() != 2 || C.shape(0) != nrt || C.shape(1) != nrt)
      throw std::runtime_error("set_basis_transform: C must be [k(k+2), k(k+2)]");
    basis_C.assign(C.data(), C.data() + (size_t)nrt * nrt);
    basis_R.clear();
    if (!R.is_none())
    {
      auto r = R.cast<py::array_t<double, py::array::c_style | py::array::forcecast>>();
      if (r.ndim() != 2 || r.shape(0) != k || r.shape(1) != k)
        throw std::runtime_error("set_basis_transform: R must be [k, k]");
      basis_R.assign(r.data(), r.data() + (size_t)k * k);
    }
    eqlb_ev_destroy(ev); // rebuilt with the new basis on the next call
    ev = nullptr;
  }
  void set_option(const std::string& key, int value)
  {
    if (custom)
    {
      if (!se)
        throw std::runtime_error("BoundaryData.set_option: no handle yet (options apply after the first call)");
      check(eqlb_se_set_option(se, key.c_str(), value));
    }
    else
      check(eqlb_ev_set_option(ev_handle(), key.c_str(), value));
  }
};

int memspace_of(const std::vector<std::shared_ptr<Function>>& a, const std::vector<std::shared_ptr<Function>>& b,
                const std::vector<std::shared_ptr<Function>>& c)
{
  int ndev = 0, n = 0;
  for (const auto* l : {&a, &b, &c})
    for (const auto& f : *l)
    {
      if (!f)
        throw std::runtime_error("Equilibration: Input sizes does not match");
      ndev += f->on_device ? 1 : 0;
      ++n;
    }
  if (ndev != 0 && ndev != n)
    throw std::runtime_error("Equilibration: all Functions of a call must live in the same memory space");
  return ndev ? EQLB_MEM_DEVICE : EQLB_MEM_HOST;
}

void semiexplt(std::vector<std::shared_ptr<Function>>& flux_hdiv, std::vector<std::shared_ptr<Function>>& flux_dg,
               std::vector<std::shared_ptr<Function>>& rhs_dg, std::shared_ptr<BoundaryData> bd,
               bool reconstruct_stress, std::shared_ptr<Function> korn)
{
  // se/reconstruction.hpp:345-388
  if (!bd)
    throw std::runtime_error("Equilibration: Input sizes does not match");
  const size_t n = flux_hdiv.size();
  if (n == 0 || flux_dg.size() != n || rhs_dg.size() != n || (int)n != bd->nrhs)
    throw std::runtime_error("Equilibration: Input sizes does not match");
  const int mem = memspace_of(flux_hdiv, flux_dg, rhs_dg);
  const int k = bd->k;
  int degree_dg = -1;
  for (size_t i = 0; i < n; ++i)
  {
    const auto &Vh = flux_hdiv[i]->V, &Vg = flux_dg[i]->V, &Vf = rhs_dg[i]->V;
    if (Vh->family != "RT" || !Vh->discontinuous || Vh->degree != k || Vh->mesh != bd->mesh)
      throw std::runtime_error("Equilibration: flux_hdiv must live in the flux space of the boundary data");
    if (Vg->family != "DG" || Vf->family != "DG" || Vg->bs != 2 || Vf->bs != 1 || Vg->mesh != bd->mesh
        || Vf->mesh != bd->mesh)
      throw std::runtime_error("Equilibration: Input sizes does not match");
    if (Vg->degree != Vf->degree || Vg->degree > k - 1 || (degree_dg >= 0 && Vg->degree != degree_dg))
      throw std::runtime_error("Equilibration: Wrong polynomial degree of the projected RHS");
    degree_dg = Vg->degree;
  }
  if (reconstruct_stress)
  {
    if (n < 2)
      throw std::runtime_error("Stress equilibration: Specify all rows of stress tensor");
    if (k < 2)
      throw std::runtime_error("Stress equilibration: RT_k with k>1 required!");
  }
  if (degree_dg != k - 1)
    throw std::runtime_error("Equilibration: projected data of degree < k-1 has to be embedded into DG_{k-1} first "
                             "(dolfinx_eqlb_amd.lsolver.embed_dg; the FluxEqlbSE class does it)");
  eqlb_se_t* h = bd->se_handle(degree_dg, reconstruct_stress);
  std::vector<const double*> g(n), f(n);
  std::vector<double*> x(n);
  for (size_t i = 0; i < n; ++i)
  {
    g[i] = flux_dg[i]->data();
    f[i] = rhs_dg[i]->data();
    x[i] = flux_hdiv[i]->data();
  }
  {
    py::gil_scoped_release nogil;
    check(eqlb_se_equilibrate_lists(h, g.data(), f.data(), x.data(), mem, g_stream));
  }
  if (korn)
  {
    if (korn->V->family != "DG" || korn->V->degree != 0 || korn->V->bs != 1 || korn->on_device != (mem == EQLB_MEM_DEVICE))
      throw std::runtime_error("Equilibration: cells_kornconst must be a DG0 Function in the memory space of the call");
    py::gil_scoped_release nogil;
    check(eqlb_se_kornconst(h, korn->data(), mem, g_stream));
  }
}

void minimisation(const Form&, const Form&, const std::vector<std::shared_ptr<Form>>& l,
                  std::vector<std::shared_ptr<Function>>& flux_hdiv, std::shared_ptr<BoundaryData> bd)
{
  if (!bd || l.size() != flux_hdiv.size() || (int)l.size() != bd->nrhs || l.empty())
    throw std::runtime_error("Equilibration: Input sizes does not match");
  const size_t n = l.size();
  std::vector<std::shared_ptr<Function>> gs, fs;
  for (const auto& li : l)
  {
    // l_i = hat G_i . v + (hat f_i + grad hat . G_i) q  (FluxEqlbEV.py:129-134): coefficients [G_i, f_i]
    if (!li || li->coefficients.size() != 2)
      throw std::runtime_error("reconstruct_fluxes_minimisation: l[i] carries the coefficients [flux_dg_i, rhs_dg_i]");
    gs.push_back(li->coefficients[0]);
    fs.push_back(li->coefficients[1]);
  }
  const int mem = memspace_of(flux_hdiv, gs, fs);
  const int k = bd->k;
  for (size_t i = 0; i < n; ++i)
  {
    const auto &Vh = flux_hdiv[i]->V, &Vg = gs[i]->V, &Vf = fs[i]->V;
    if (Vh != bd->V && !(Vh->family == "RT" && !Vh->discontinuous && Vh->degree == k && Vh->mesh == bd->mesh))
      throw std::runtime_error("Equilibration: flux_hdiv must live in the flux space of the boundary data");
    if (Vg->family != "DG" || Vf->family != "DG" || Vg->bs != 2 || Vf->bs != 1 || Vg->degree != k - 1
        || Vf->degree != k - 1 || Vg->mesh != bd->mesh || Vf->mesh != bd->mesh)
      throw std::runtime_error("Equilibration: Input sizes does not match");
  }
  eqlb_ev_t* h = bd->ev_handle();
  std::vector<const double*> g(n), f(n);
  std::vector<double*> x(n);
  for (size_t i = 0; i < n; ++i)
  {
    g[i] = gs[i]->data();
    f[i] = fs[i]->data();
    x[i] = flux_hdiv[i]->data();
  }
  py::gil_scoped_release nogil;
  check(eqlb_ev_equilibrate_lists(h, g.data(), f.data(), x.data(), mem, g_stream));
}

// base::local_solver (base/local_solver.hpp:38-187): a = (u, v) on the space of the solutions, l_i = (f_i, v)
void local_solver(std::vector<std::shared_ptr<Function>>& sol, const Form&, const std::vector<std::shared_ptr<Form>>& l)
{
  if (sol.empty() || sol.size() != l.size())
    throw std::runtime_error("Local solver: Input sizes does not match");
  for (size_t i = 0; i < sol.size(); ++i)
  {
    const auto& u = sol[i];
    if (!u || !l[i] || !l[i]->has_points || u->V->family != "DG")
      throw std::runtime_error("Local solver: DG solutions and forms with point values expected");
    const auto& m = u->V->mesh;
    const int nq = (int)l[i]->qweights.size(), bs = u->V->bs;
    if (l[i]->qvalues.size() != (py::ssize_t)m->ncells * nq * bs)
      throw std::runtime_error("Local solver: Input sizes does not match");
    double* out = u->data();
    const double *qp = l[i]->qpoints.data(), *qw = l[i]->qweights.data(), *qv = l[i]->qvalues.data();
    if (u->on_device)
      throw std::runtime_error("Local solver: host Functions expected (point values are host arrays)");
    py::gil_scoped_release nogil;
    check(eqlb_project_dg(m->h, u->V->degree, bs, 1, nq, qp, qw, qv, out, EQLB_MEM_HOST, g_stream));
  }
}

// ---- multi-GPU: reverse halo over RCCL through the C ABI (include/eqlb.h: eqlb_halo_*, eqlb_rccl_*) --------------
// No counterpart in the reference module (its node loop covers the owned nodes of one process,
// cpp/dolfinx_eqlb/se/reconstruction.hpp:90); this is the C++ host's own multi-GPU path: one process per GPU, a
// communicator made from a unique id the caller distributes (MPI where DOLFINx runs), one call per sweep.
struct RcclComm
{
  void* comm = nullptr;
  int nranks = 0, rank = 0;
  RcclComm(py::bytes unique_id, int nranks_, int rank_) : nranks(nranks_), rank(rank_)
  {
    const std::string id = unique_id;
    if (id.size() != 128)
      throw std::runtime_error("RcclComm: the unique id has 128 bytes");
    check(eqlb_rccl_comm_create(id.data(), nranks, rank, &comm));
  }
  ~RcclComm() { eqlb_rccl_comm_destroy(comm); }
  RcclComm(const RcclComm&) = delete;
  RcclComm& operator=(const RcclComm&) = delete;
};

struct HaloExchange
{
  eqlb_halo_t* h = nullptr;
  int64_t nentries;
  int nrhs, width;
  // send / recv: {peer rank: int64 index array} - rows (of `width` doubles) that leave to / arrive from the peer
  HaloExchange(int nrhs_, int width_, int64_t nentries_, std::map<int, carray<int64_t>> send,
               std::map<int, carray<int64_t>> recv)
      : nentries(nentries_), nrhs(nrhs_), width(width_)
  {
    std::vector<int32_t> peers;
    for (auto& kv : send)
      peers.push_back(kv.first);
    for (auto& kv : recv)
      if (!send.count(kv.first))
        peers.push_back(kv.first);
    std::sort(peers.begin(), peers.end());
    std::vector<const int64_t*> si, ri;
    std::vector<int64_t> ns, nr;
    for (int32_t q : peers)
    {
      auto s_ = send.find(q);
      auto r_ = recv.find(q);
      si.push_back(s_ != send.end() ? s_->second.data() : nullptr);
      ns.push_back(s_ != send.end() ? (int64_t)s_->second.size() : 0);
      ri.push_back(r_ != recv.end() ? r_->second.data() : nullptr);
      nr.push_back(r_ != recv.end() ? (int64_t)r_->second.size() : 0);
    }
    check(eqlb_halo_create(nrhs, width, nentries, (int32_t)peers.size(), peers.data(), si.data(), ns.data(),
                           ri.data(), nr.data(), &h));
  }
  ~HaloExchange() { eqlb_halo_destroy(h); }
  HaloExchange(const HaloExchange&) = delete;
  HaloExchange& operator=(const HaloExchange&) = delete;
  void reduce_ptr(RcclComm& c, uintptr_t x)
  {
    if (!x)
      throw std::runtime_error("HaloExchange.reduce: null device pointer");
    check(eqlb_halo_reduce_plan(h, c.comm, reinterpret_cast<double*>(x), g_stream));
  }
  void reduce(RcclComm& c, std::vector<std::shared_ptr<Function>> fs)
  {
    // the right-hand sides of one call share a halo: consecutive device blocks of one array
    if ((int)fs.size() != nrhs || !fs[0] || !fs[0]->on_device)
      throw std::runtime_error("HaloExchange.reduce: nrhs device-memory Functions (Function.from_device) expected");
    for (int r = 0; r < nrhs; ++r)
      if (!fs[r] || !fs[r]->on_device || fs[r]->size() != nentries * width
          || fs[r]->dev != fs[0]->dev + (uintptr_t)r * nentries * width * sizeof(double))
        throw std::runtime_error("HaloExchange.reduce: the Functions must be consecutive blocks of one device array");
    reduce_ptr(c, fs[0]->dev);
  }
  py::tuple bytes() const
  {
    int64_t s = 0, r = 0;
    check(eqlb_halo_bytes(h, &s, &r));
    return py::make_tuple(s, r);
  }
};

} // namespace

PYBIND11_MODULE(_cpp, m)
{
  m.doc() = "dolfinx_eqlb_amd: MI355X-native patch-local flux equilibration - interface of dolfinx_eqlb.cpp";

  py::class_<Mesh, std::shared_ptr<Mesh>>(m, "Mesh", "Flat triangle mesh on the device (stand-in of dolfinx.mesh.Mesh)")
      .def(py::init<carray<double>, carray<int32_t>, carray<int32_t>, carray<int32_t>, carray<int32_t>,
                    carray<int32_t>, carray<int32_t>, carray<int32_t>, carray<int32_t>, carray<int32_t>,
                    carray<uint8_t>>(),
           py::arg("x"), py::arg("cell_nodes"), py::arg("cell_facets"), py::arg("facet_nodes"),
           py::arg("facet_cells_offsets"), py::arg("facet_cells"), py::arg("node_cells_offsets"),
           py::arg("node_cells"), py::arg("node_facets_offsets"), py::arg("node_facets"), py::arg("facet_perm"))
      .def_readonly("nnodes", &Mesh::nnodes)
      .def_readonly("ncells", &Mesh::ncells)
      .def_readonly("nfacets", &Mesh::nfacets)
      .def_property_readonly("max_patch_cells", [](const Mesh& s) { return eqlb_mesh_max_patch_cells(s.h); });

  py::class_<FunctionSpace, std::shared_ptr<FunctionSpace>>(m, "FunctionSpace")
      .def(py::init<std::shared_ptr<Mesh>, std::string, int, int, bool>(), py::arg("mesh"), py::arg("family"),
           py::arg("degree"), py::arg("bs") = 1, py::arg("discontinuous") = true)
      .def("set_dofmap", &FunctionSpace::set_dofmap, py::arg("cell_dofs"), py::arg("ndofs"))
      .def_readonly("mesh", &FunctionSpace::mesh)
      .def_readonly("family", &FunctionSpace::family)
      .def_readonly("degree", &FunctionSpace::degree)
      .def_readonly("bs", &FunctionSpace::bs)
      .def_readonly("discontinuous", &FunctionSpace::discontinuous)
      .def_property_readonly("ndofs", &FunctionSpace::ndofs);

  py::class_<Function, std::shared_ptr<Function>>(m, "Function")
      .def(py::init<std::shared_ptr<FunctionSpace>>(), py::arg("V"))
      .def(py::init<std::shared_ptr<FunctionSpace>, py::array_t<double>>(), py::arg("V"), py::arg("array"))
      .def_static("from_device", &Function::from_device, py::arg("V"), py::arg("device_ptr"))
      .def_readonly("function_space", &Function::V)
      .def_readonly("on_device", &Function::on_device)
      .def_property_readonly("device_ptr", [](const Function& f) { return f.dev; })
      .def_property_readonly("array", [](Function& f) -> py::object {
        if (f.on_device)
          throw std::runtime_error("Function.array: the values live on the device");
        return f.host;
      });

  py::class_<Constant, std::shared_ptr<Constant>>(m, "Constant").def(py::init<carray<double>>(), py::arg("value"));

  py::class_<Form, std::shared_ptr<Form>>(m, "Form")
      .def(py::init<std::vector<std::shared_ptr<Function>>>(), py::arg("coefficients") = std::vector<std::shared_ptr<Function>>{})
      .def_static("from_point_values", &Form::from_point_values, py::arg("qpoints"), py::arg("qweights"),
                  py::arg("qvalues"));

  // ---- boundary conditions (wrappers.cpp:140-257) ----
  py::class_<FluxBC, std::shared_ptr<FluxBC>>(m, "FluxBC", "FluxBC object")
      .def(py::init([](std::shared_ptr<FunctionSpace> V, const std::vector<int32_t>& facets, uintptr_t kernel_ptr,
                       int n_bceval_per_fct, std::vector<std::shared_ptr<Function>> coefficients,
                       std::vector<int> positions, std::vector<std::shared_ptr<Constant>> constants) {
             return std::make_shared<FluxBC>(V, facets, kernel_ptr, n_bceval_per_fct, 0, false, coefficients,
                                             positions, constants);
           }),
           py::arg("function_space"), py::arg("facets"), py::arg("pointer_boundary_kernel"),
           py::arg("nevals_per_fct"), py::arg("coefficients"), py::arg("position_of_coefficients"),
           py::arg("constants"))
      .def(py::init([](std::shared_ptr<FunctionSpace> V, const std::vector<int32_t>& facets, uintptr_t kernel_ptr,
                       int n_bceval_per_fct, int quadrature_degree,
                       std::vector<std::shared_ptr<Function>> coefficients, std::vector<int> positions,
                       std::vector<std::shared_ptr<Constant>> constants) {
             return std::make_shared<FluxBC>(V, facets, kernel_ptr, n_bceval_per_fct, quadrature_degree, true,
                                             coefficients, positions, constants);
           }),
           py::arg("function_space"), py::arg("facets"), py::arg("pointer_boundary_kernel"),
           py::arg("nevals_per_fct"), py::arg("quadrature_degree"), py::arg("coefficients"),
           py::arg("position_of_coefficients"), py::arg("constants"))
      .def_property_readonly("quadrature_degree", [](const FluxBC& b) { return b.qdegree; });

  py::class_<BoundaryData, std::shared_ptr<BoundaryData>>(m, "BoundaryData", py::dynamic_attr(), "BoundaryData object")
      .def(py::init([](std::vector<std::vector<std::shared_ptr<FluxBC>>>& list_bcs,
                       std::vector<std::shared_ptr<Function>>& boundary_flux, std::shared_ptr<FunctionSpace> V,
                       bool rtflux_is_custom, int quadrature_degree,
                       const std::vector<std::vector<int32_t>>& fct_esntbound_prime, bool reconstruct_stress) {
             return std::make_shared<BoundaryData>(list_bcs, boundary_flux, V, rtflux_is_custom, quadrature_degree,
                                                   fct_esntbound_prime, reconstruct_stress);
           }),
           py::arg("list_of_bcs"), py::arg("list_of_boundary_fluxes"), py::arg("V_flux_hdiv"),
           py::arg("rtflux_is_custom"), py::arg("quadrature_degree"), py::arg("list_bfcts_prime"),
           py::arg("reconstruct_stress"))
      .def("set_basis_transform", &BoundaryData::set_basis_transform, py::arg("C"), py::arg("R") = py::none(),
           "Output basis of the conforming flux: y_cell = C c_cell, R on the facet block of reflected facets")
      .def("set_option", &BoundaryData::set_option, py::arg("key"), py::arg("value"),
           "Integer options of the device handle (eqlb_se_set_option / eqlb_ev_set_option)")
      .def_property_readonly("facet_type", [](const BoundaryData& b) {
        py::array_t<int8_t> a({(py::ssize_t)b.nrhs, (py::ssize_t)b.mesh->nfacets});
        std::copy(b.facet_type.begin(), b.facet_type.end(), a.mutable_data());
        return a;
      });

  // ---- local solvers (wrappers.cpp:52-80): one kernel, the exact inverse of the reference mass matrix ----
  for (const char* name : {"local_solver_lu", "local_solver_cholesky", "local_solver_cg"})
    m.def(name, &local_solver, py::arg("solution"), py::arg("a"), py::arg("l"),
          "Cell-local projection (base/local_solver.hpp:38-187) on the device");

  // ---- equilibration (wrappers.cpp:82-137) ----
  m.def("reconstruct_fluxes_minimisation", &minimisation, py::arg("a"), py::arg("l_pen"), py::arg("l"),
        py::arg("flux_hdiv"), py::arg("boundary_data"),
        "Local equilibration of H(div) conforming fluxes, solving patch-wise, constrained minimisation problems.");
  m.def(
      "reconstruct_fluxes_semiexplt",
      [](std::vector<std::shared_ptr<Function>>& flux_hdiv, std::vector<std::shared_ptr<Function>>& flux_dg,
         std::vector<std::shared_ptr<Function>>& rhs_dg, std::shared_ptr<BoundaryData> bd, bool reconstruct_stress)
      { semiexplt(flux_hdiv, flux_dg, rhs_dg, bd, reconstruct_stress, nullptr); },
      py::arg("flux_hdiv"), py::arg("flux_dg"), py::arg("rhs_dg"), py::arg("boundary_data"),
      py::arg("reconstruct_stress"),
      "Local equilibration of H(div) conforming fluxes, using an explicit determination of the fluxes followed by "
      "a minimisation on a reduced space; reconstruct_stress: weak symmetry of the first gdim rows.");
  m.def(
      "reconstruct_fluxes_semiexplt_with_kornconst",
      [](std::vector<std::shared_ptr<Function>>& flux_hdiv, std::vector<std::shared_ptr<Function>>& flux_dg,
         std::vector<std::shared_ptr<Function>>& rhs_dg, std::shared_ptr<BoundaryData> bd, bool reconstruct_stress,
         std::shared_ptr<Function> cells_kornconst)
      {
        if (!cells_kornconst)
          throw std::runtime_error("Equilibration: Input sizes does not match");
        semiexplt(flux_hdiv, flux_dg, rhs_dg, bd, reconstruct_stress, cells_kornconst);
      },
      py::arg("flux_hdiv"), py::arg("flux_dg"), py::arg("rhs_dg"), py::arg("boundary_data"),
      py::arg("reconstruct_stress"), py::arg("cells_kornconst"),
      "As reconstruct_fluxes_semiexplt; upper bounds of the cells' squared Korn constants are accumulated.");

  // ---- helpers without a counterpart in the reference module ----
  m.def("facet_quadrature", [](int degree) {
    std::vector<double> s, w;
    facet_rule(degree, s, w);
    return py::make_tuple(py::array_t<double>(s.size(), s.data()), py::array_t<double>(w.size(), w.data()));
  }, py::arg("degree"), "Gauss rule on [0, 1] used for flux BCs (requires_projection: `quadrature_degree`)");
  m.def("interpolation_quadrature_degree", &interpolation_degree, py::arg("degree_flux"),
        "Degree of the facet rule that stands for the element's interpolation points (no projection)");
  // ---- multi-GPU (no counterpart in the reference module) ----
  m.def("rccl_unique_id", []() {
    char id[128];
    check(eqlb_rccl_get_unique_id(id));
    return py::bytes(id, 128);
  }, "ncclUniqueId (128 bytes) for RcclComm: made on one rank, distributed by the caller");
  py::class_<RcclComm, std::shared_ptr<RcclComm>>(m, "RcclComm", "RCCL communicator (ncclCommInitRank through the C ABI)")
      .def(py::init<py::bytes, int, int>(), py::arg("unique_id"), py::arg("nranks"), py::arg("rank"))
      .def_readonly("nranks", &RcclComm::nranks)
      .def_readonly("rank", &RcclComm::rank);
  py::class_<HaloExchange, std::shared_ptr<HaloExchange>>(
      m, "HaloExchange", "Reverse halo of the node-ownership decomposition: ghost rows -> owner, added there")
      .def(py::init<int, int, int64_t, std::map<int, carray<int64_t>>, std::map<int, carray<int64_t>>>(),
           py::arg("nrhs"), py::arg("width"), py::arg("nentries"), py::arg("send"), py::arg("recv"))
      .def("reduce", &HaloExchange::reduce, py::arg("comm"), py::arg("flux_hdiv"),
           "pack + clear, grouped ncclSend / ncclRecv, unpack-add on the stream of set_stream")
      .def("reduce_ptr", &HaloExchange::reduce_ptr, py::arg("comm"), py::arg("device_pointer"))
      .def("bytes", &HaloExchange::bytes, "(bytes sent, bytes received) per reduction");
  m.def("set_stream", [](uintptr_t s) { g_stream = reinterpret_cast<void*>(s); }, py::arg("stream"),
        "hipStream_t used by calls on device-memory Functions (0: default stream)");
  m.def("device_count", &eqlb_device_count);
  m.def("synchronize_and_check", [](std::shared_ptr<BoundaryData> bd) {
    if (bd->se)
      check(eqlb_se_check_status(bd->se, g_stream));
    if (bd->ev)
      check(eqlb_ev_check_status(bd->ev, g_stream));
  }, py::arg("boundary_data"), "Wait for the stream of device-memory calls and raise if a patch system was singular");
}
