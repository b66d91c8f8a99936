// DOLFINx 0.6 glue for include/eqlb.h: flat arrays out of dolfinx::mesh::Mesh / fem::Function objects.
//
// Compiled only where DOLFINx is installed (the guard below); in the pipeline that builds this repository
// DOLFINx is absent, so THIS HEADER HAS NEVER BEEN COMPILED OR RUN - it states, in code instead of prose, the
// ~100 lines a dolfinx_eqlb maintainer adds next to python/dolfinx_eqlb/wrappers.cpp:85-137 to route the
// reference's entry points through libeqlb_amd.so.  Everything it needs from the
// library is the C ABI; everything it needs from DOLFINx is the public 0.6 API the reference itself uses
// (FluxEquilibrator.py:52-67, se/Patch.cpp:20-26, se/reconstruction.hpp:83-90).
#pragma once
#if defined(__has_include)
#if __has_include(<dolfinx/mesh/Mesh.h>)
#define EQLB_HAVE_DOLFINX 1
#endif
#endif

#ifdef EQLB_HAVE_DOLFINX
#include "eqlb.h"
#include <dolfinx/fem/Function.h>
#include <dolfinx/mesh/Mesh.h>
#include <memory>
#include <stdexcept>
#include <vector>

namespace eqlb_dolfinx
{
inline void check(int status)
{
  if (status != EQLB_OK)
    throw std::runtime_error(eqlb_last_error());
}

/// eqlb_mesh_create from a triangle mesh: topology indices (owned + ghosts) of the process, vertex coordinates
/// looked up through the geometry dofmap (topology vertex ids and geometry node ids differ in general).
inline eqlb_mesh_t* create_mesh(dolfinx::mesh::Mesh& msh)
{
  auto& topo = msh.topology_mutable();
  const int tdim = topo.dim();
  if (tdim != 2)
    throw std::runtime_error("Equilibration only possible on triangles");
  // what FluxEquilibrator.initialise_mesh_info creates (FluxEquilibrator.py:52-67)
  topo.create_entities(1);
  topo.create_connectivity(2, 1);
  topo.create_connectivity(1, 2);
  topo.create_connectivity(1, 0);
  topo.create_connectivity(0, 1);
  topo.create_connectivity(0, 2);
  topo.create_entity_permutations();
  auto c2n = topo.connectivity(2, 0), c2f = topo.connectivity(2, 1);
  auto f2n = topo.connectivity(1, 0), f2c = topo.connectivity(1, 2);
  auto n2c = topo.connectivity(0, 2), n2f = topo.connectivity(0, 1);
  const std::vector<std::uint8_t>& perms = topo.get_facet_permutations(); // [cell * 3 + local facet]
  auto count = [&](int d) { return topo.index_map(d)->size_local() + topo.index_map(d)->num_ghosts(); };
  const std::int32_t nnodes = count(0), nfcts = count(1), ncells = count(2);
  // coordinates per topology vertex
  const auto& geo = msh.geometry();
  const auto& xdofs = geo.dofmap();
  std::span<const double> xg = geo.x();
  std::vector<double> x(3 * (std::size_t)nnodes, 0.0);
  for (std::int32_t c = 0; c < ncells; ++c)
  {
    auto vs = c2n->links(c);
    auto gs = xdofs.links(c);
    for (int v = 0; v < 3; ++v)
      for (int d = 0; d < 3; ++d)
        x[3 * (std::size_t)vs[v] + d] = xg[3 * (std::size_t)gs[v] + d];
  }
  eqlb_mesh_t* mesh = nullptr;
  check(eqlb_mesh_create(nnodes, ncells, nfcts, x.data(), c2n->array().data(), c2f->array().data(),
                         f2n->array().data(), f2c->offsets().data(), f2c->array().data(), n2c->offsets().data(),
                         n2c->array().data(), n2f->offsets().data(), n2f->array().data(), perms.data(), &mesh));
  return mesh;
}

/// owned nodes only, like the reference loop over index_map(0)->size_local() (se/reconstruction.hpp:90)
inline std::vector<std::uint8_t> owned_node_mask(const dolfinx::mesh::Mesh& msh)
{
  auto im = msh.topology().index_map(0);
  std::vector<std::uint8_t> mask(im->size_local() + im->num_ghosts(), 0);
  std::fill_n(mask.begin(), im->size_local(), 1);
  return mask;
}

using FunctionList = std::vector<std::shared_ptr<dolfinx::fem::Function<double>>>;

/// body of reconstruct_fluxes_semiexplt (wrappers.cpp:97-115) on a handle created with eqlb_se_create and
/// eqlb_se_set_boundary(handle, boundary_data.facet_type(), boundary_data.boundary_values(), owned_node_mask):
/// no staging copies, the Functions' own vectors are handed over (host memory)
inline void reconstruct_fluxes_semiexplt(eqlb_se_t* handle, FunctionList& flux_hdiv, FunctionList& flux_dg,
                                         FunctionList& rhs_dg)
{
  const std::size_t n = flux_hdiv.size();
  if (flux_dg.size() != n || rhs_dg.size() != n)
    throw std::runtime_error("Equilibration: Input sizes does not match");
  std::vector<const double*> g(n), f(n);
  std::vector<double*> xo(n);
  for (std::size_t i = 0; i < n; ++i)
  {
    g[i] = flux_dg[i]->x()->array().data();
    f[i] = rhs_dg[i]->x()->array().data();
    xo[i] = flux_hdiv[i]->x()->mutable_array().data();
  }
  check(eqlb_se_equilibrate_lists(handle, g.data(), f.data(), xo.data(), EQLB_MEM_HOST, nullptr));
}

/// body of reconstruct_fluxes_minimisation (wrappers.cpp:85-95); the handle carries the dofmap of V_flux
/// (eqlb_ev_set_dofmap(handle, V_flux->dofmap()->list().array().data(), ndofs)) and, for Basix' RT_k basis, the
/// reference matrices of eqlb_ev_set_basis_transform
inline void reconstruct_fluxes_minimisation(eqlb_ev_t* handle, FunctionList& flux_hdiv, FunctionList& flux_dg,
                                            FunctionList& rhs_dg)
{
  const std::size_t n = flux_hdiv.size();
  if (flux_dg.size() != n || rhs_dg.size() != n)
    throw std::runtime_error("Equilibration: Input sizes does not match");
  std::vector<const double*> g(n), f(n);
  std::vector<double*> xo(n);
  for (std::size_t i = 0; i < n; ++i)
  {
    g[i] = flux_dg[i]->x()->array().data();
    f[i] = rhs_dg[i]->x()->array().data();
    xo[i] = flux_hdiv[i]->x()->mutable_array().data();
  }
  check(eqlb_ev_equilibrate_lists(handle, g.data(), f.data(), xo.data(), EQLB_MEM_HOST, nullptr));
}
} // namespace eqlb_dolfinx
#endif // EQLB_HAVE_DOLFINX
