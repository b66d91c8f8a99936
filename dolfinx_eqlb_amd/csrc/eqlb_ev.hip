// Layout kernels of the constrained-minimisation (EV) equilibrator: the patch kernel
// (eqlb_se_kernels.hip, MODE 1) produces patch-local coefficients in the broken hierarchic RT_k
// layout [cell][vertex slot][k(k+2)]; the flux of FluxEqlbEV lives in an H(div)-conforming space
// (python/dolfinx_eqlb/eqlb/FluxEqlbEV.py:94-101, scatter ev/solve_patch.hpp:223-227).  Here the
// conforming version of the hierarchic RT_k is used: facet DOFs in the global facet frame, a cell
// sees them through T_f = -I (facet_perm 0) or T_f = B, B_ji = C(j,i)(-1)^i (both involutions).
#include "eqlb_device_common.h"

namespace eqlb
{

template <int K>
__device__ __forceinline__ void facet_map(bool rev, const double* in, double* out)
{
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      s += (rev ? bcoef(j, i) : ((i == j) ? -1.0 : 0.0)) * in[i];
    out[j] = s;
  }
}

template <int K>
__device__ __forceinline__ int64_t conf_dof(const int32_t* cell_dofs, const int32_t* cell_facets,
                                            int32_t nfacets, int32_t cell, int i)
{
  constexpr int NRT = K * (K + 2), NI = K * K - K;
  if (cell_dofs)
    return cell_dofs[(int64_t)cell * NRT + i];
  if (i < 3 * K)
    return (int64_t)cell_facets[(int64_t)cell * 3 + i / K] * K + i % K;
  return (int64_t)nfacets * K + (int64_t)cell * NI + (i - 3 * K);
}

// bv_broken[r][cell][f*K + j] = (T_f g)_j, g = conforming boundary DOFs of the cell's facet f
template <int K>
__global__ void __launch_bounds__(256)
k_ev_boundary_to_broken(int32_t ncells, int32_t nfacets, int nrhs, const int32_t* cell_facets,
                        const uint8_t* facet_perm, const int32_t* cell_dofs, int64_t ndofs,
                        const double* __restrict__ bv_conf, double* __restrict__ bv_broken,
                        const double* __restrict__ facet_maps)
{
  constexpr int NRT = K * (K + 2);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nrhs * ncells * 3)
    return;
  const int f = (int)(t % 3);
  const int64_t rc = t / 3;
  const int32_t c = (int32_t)(rc % ncells);
  const int r = (int)(rc / ncells);
  double g[K], o[K];
#pragma unroll
  for (int j = 0; j < K; ++j)
    g[j] = bv_conf[(int64_t)r * ndofs + conf_dof<K>(cell_dofs, cell_facets, nfacets, c, f * K + j)];
  const bool rev = facet_perm[(int64_t)c * 3 + f] != 0;
  if (facet_maps)
  {
    const double* M = facet_maps + ((f * 2 + (rev ? 1 : 0)) * K) * K;
#pragma unroll
    for (int j = 0; j < K; ++j)
    {
      double s_ = 0.0;
#pragma unroll
      for (int i = 0; i < K; ++i)
        s_ += M[j * K + i] * g[i];
      o[j] = s_;
    }
  }
  else
    facet_map<K>(rev, g, o);
#pragma unroll
  for (int j = 0; j < K; ++j)
    bv_broken[((int64_t)r * ncells + c) * NRT + f * K + j] = o[j];
}

// x[r][dof] += sum of the three vertex slots, facet DOFs read from the first cell of the facet
// (the patch solutions are conforming, so both sides agree) and mapped to the global frame
template <int K>
__global__ void __launch_bounds__(256)
k_ev_reduce(int32_t ncells, int32_t nfacets, int nrhs, const int32_t* cell_facets,
            const uint8_t* facet_perm, const int32_t* facet_cells_off, const int32_t* facet_cells,
            const int32_t* cell_dofs, int64_t ndofs, const double* __restrict__ slots,
            double* __restrict__ x, int accumulate)
{
  constexpr int NRT = K * (K + 2), NI = K * K - K;
  const int64_t per_rhs = (int64_t)nfacets + (int64_t)ncells * NI;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per_rhs * nrhs)
    return;
  const int r = (int)(t / per_rhs);
  const int64_t e = t - (int64_t)r * per_rhs;
  const double* sl = slots + (int64_t)r * ncells * 3 * NRT;
  double* xr = x + (int64_t)r * ndofs;
  if (e < nfacets)
  {
    const int32_t fct = (int32_t)e;
    const int32_t c = facet_cells[facet_cells_off[fct]];
    int lf = 0;
#pragma unroll
    for (int l = 1; l < 3; ++l)
      if (cell_facets[(int64_t)c * 3 + l] == fct)
        lf = l;
    const double* s = sl + (int64_t)c * 3 * NRT + lf * K;
    double v[K], g[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
      v[j] = (s[j] + s[NRT + j]) + s[2 * NRT + j];
    facet_map<K>(facet_perm[(int64_t)c * 3 + lf] != 0, v, g);
#pragma unroll
    for (int j = 0; j < K; ++j)
    {
      double* px = xr + conf_dof<K>(cell_dofs, cell_facets, nfacets, c, lf * K + j);
      *px = accumulate ? *px + g[j] : g[j];
    }
  }
  else if constexpr (NI > 0)
  {
    const int64_t q = e - nfacets;
    const int32_t c = (int32_t)(q / NI);
    const int i = (int)(q - (int64_t)c * NI);
    const double* s = sl + (int64_t)c * 3 * NRT + 3 * K + i;
    double* px = xr + conf_dof<K>(cell_dofs, cell_facets, nfacets, c, 3 * K + i);
    const double v = (s[0] + s[NRT]) + s[2 * NRT];
    *px = accumulate ? *px + v : v;
  }
}

// the same reduction into another element basis of RT_k (eqlb_ev_set_basis_transform): one thread per cell
template <int K>
__global__ void __launch_bounds__(256)
k_ev_reduce_basis(int32_t ncells, int32_t nfacets, int nrhs, const int32_t* cell_facets,
                  const uint8_t* facet_perm, const int32_t* facet_cells_off, const int32_t* facet_cells,
                  const int32_t* cell_dofs, int64_t ndofs, const double* __restrict__ slots,
                  double* __restrict__ x, int accumulate, const double* __restrict__ C,
                  const double* __restrict__ R)
{
  constexpr int NRT = K * (K + 2), NI = K * K - K;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)ncells * nrhs)
    return;
  const int r = (int)(t / ncells);
  const int32_t c = (int32_t)(t - (int64_t)r * ncells);
  const double* s = slots + ((int64_t)r * ncells + c) * 3 * NRT;
  double* xr = x + (int64_t)r * ndofs;
  double v[NRT], y[NRT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
    v[i] = (s[i] + s[NRT + i]) + s[2 * NRT + i];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
  {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NRT; ++j)
      a += C[i * NRT + j] * v[j];
    y[i] = a;
  }
#pragma unroll
  for (int lf = 0; lf < 3; ++lf)
  {
    const int32_t fct = cell_facets[(int64_t)c * 3 + lf];
    if (facet_cells[facet_cells_off[fct]] != c)
      continue; // the first cell of the facet writes its DOFs
    const bool rev = facet_perm[(int64_t)c * 3 + lf] != 0 && R != nullptr;
#pragma unroll
    for (int j = 0; j < K; ++j)
    {
      double g = y[lf * K + j];
      if (rev)
      {
        g = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i)
          g += R[j * K + i] * y[lf * K + i];
      }
      double* px = xr + conf_dof<K>(cell_dofs, cell_facets, nfacets, c, lf * K + j);
      *px = accumulate ? *px + g : g;
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
  {
    double* px = xr + conf_dof<K>(cell_dofs, cell_facets, nfacets, c, 3 * K + i);
    *px = accumulate ? *px + y[3 * K + i] : y[3 * K + i];
  }
}

void launch_ev_boundary_to_broken(const DeviceMesh& m, int k, int nrhs, const int32_t* cell_dofs,
                                  int64_t ndofs, const double* bv_conf, double* bv_broken,
                                  const double* facet_maps, hipStream_t stream)
{
  const int64_t n = (int64_t)nrhs * m.ncells * 3;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (k == 1)
    hipLaunchKernelGGL(k_ev_boundary_to_broken<1>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs,
                       m.cell_facets, m.facet_perm, cell_dofs, ndofs, bv_conf, bv_broken, facet_maps);
  else if (k == 2)
    hipLaunchKernelGGL(k_ev_boundary_to_broken<2>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs,
                       m.cell_facets, m.facet_perm, cell_dofs, ndofs, bv_conf, bv_broken, facet_maps);
  else if (k == 3)
    hipLaunchKernelGGL(k_ev_boundary_to_broken<3>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs,
                       m.cell_facets, m.facet_perm, cell_dofs, ndofs, bv_conf, bv_broken, facet_maps);
  else
    hipLaunchKernelGGL(k_ev_boundary_to_broken<4>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs,
                       m.cell_facets, m.facet_perm, cell_dofs, ndofs, bv_conf, bv_broken, facet_maps);
}

void launch_ev_reduce(const DeviceMesh& m, int k, int nrhs, const int32_t* cell_dofs, int64_t ndofs,
                      const double* slots, double* x, int accumulate, const double* basis_C, const double* basis_R,
                      hipStream_t stream)
{
  if (basis_C)
  {
    const int64_t nt = (int64_t)m.ncells * nrhs;
    const dim3 g2((unsigned)((nt + 255) / 256)), b2(256);
    if (k == 1)
      hipLaunchKernelGGL(k_ev_reduce_basis<1>, g2, b2, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets, m.facet_perm,
                         m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate, basis_C, basis_R);
    else if (k == 2)
      hipLaunchKernelGGL(k_ev_reduce_basis<2>, g2, b2, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets, m.facet_perm,
                         m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate, basis_C, basis_R);
    else if (k == 3)
      hipLaunchKernelGGL(k_ev_reduce_basis<3>, g2, b2, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets, m.facet_perm,
                         m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate, basis_C, basis_R);
    else
      hipLaunchKernelGGL(k_ev_reduce_basis<4>, g2, b2, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets, m.facet_perm,
                         m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate, basis_C, basis_R);
    return;
  }
  const int ni = k * k - k;
  const int64_t n = ((int64_t)m.nfacets + (int64_t)m.ncells * ni) * nrhs;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (k == 1)
    hipLaunchKernelGGL(k_ev_reduce<1>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets,
                       m.facet_perm, m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate);
  else if (k == 2)
    hipLaunchKernelGGL(k_ev_reduce<2>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets,
                       m.facet_perm, m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate);
  else if (k == 3)
    hipLaunchKernelGGL(k_ev_reduce<3>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets,
                       m.facet_perm, m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate);
  else
    hipLaunchKernelGGL(k_ev_reduce<4>, grid, block, 0, stream, m.ncells, m.nfacets, nrhs, m.cell_facets,
                       m.facet_perm, m.facet_cells_off, m.facet_cells, cell_dofs, ndofs, slots, x, accumulate);
}

// ---- halo rows of the multi-GPU decomposition -----------------------------------------------------
// pack: buf[r][i][:] = x[r][cells[i]][:], then the ghost rows are cleared (their content moves to the
// owner); unpack: x[r][cells[i]][:] += buf[r][i][:].  One launch each instead of a chain of
// framework indexing kernels in the timed step.
__global__ void __launch_bounds__(256)
k_halo_pack(int64_t n, int32_t ncells_list, int32_t nrt, int64_t ncells, const int64_t* __restrict__ cells,
            double* __restrict__ x, double* __restrict__ buf, int clear)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n)
    return;
  const int64_t per = (int64_t)ncells_list * nrt;
  const int64_t r = e / per, q = e - r * per;
  const int64_t i = q / nrt, j = q - i * nrt;
  double* px = x + (r * ncells + cells[i]) * nrt + j;
  buf[e] = *px;
  if (clear)
    *px = 0.0;
}

__global__ void __launch_bounds__(256)
k_halo_unpack_add(int64_t n, int32_t ncells_list, int32_t nrt, int64_t ncells,
                  const int64_t* __restrict__ cells, double* __restrict__ x, const double* __restrict__ buf)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n)
    return;
  const int64_t per = (int64_t)ncells_list * nrt;
  const int64_t r = e / per, q = e - r * per;
  const int64_t i = q / nrt, j = q - i * nrt;
  x[(r * ncells + cells[i]) * nrt + j] += buf[e];
}

void launch_halo_pack(int nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells, double* x,
                      double* buf, int clear, hipStream_t stream)
{
  const int64_t n = (int64_t)nrhs * nlist * nrt;
  if (n > 0)
    hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, nlist, nrt,
                       ncells, cells, x, buf, clear);
}

void launch_halo_unpack_add(int nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells,
                            double* x, const double* buf, hipStream_t stream)
{
  const int64_t n = (int64_t)nrhs * nlist * nrt;
  if (n > 0)
    hipLaunchKernelGGL(k_halo_unpack_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, nlist,
                       nrt, ncells, cells, x, buf);
}

} // namespace eqlb
