// Cell-local L2 projection into DG_d (scalar or blocked): the hot loop of base::local_solver
// (cpp/dolfinx_eqlb/base/local_solver.hpp:38-187) as called by local_projection
// (python/dolfinx_eqlb/lsolver/projection.py:17-77) with a = (u, v), l_i = (f_i, v).
//
// On affine cells the element mass matrix is |detJ| * Mref and the load is |detJ| * Psi^T W f_q,
// so the cell solve A_e u = L_e (reference: FFCx tabulate_tensor + Eigen::LLT per cell, :141-160)
// collapses to ONE constant matrix for all cells: u = (Mref^-1 Psi^T W) f_q =: Pm f_q.  The data is
// handed over as point values at the images of a quadrature rule chosen by the caller (stand-in
// for the FFCx kernel's embedded rule); the kernel is a streaming batched small GEMV, HBM-bound.
#include "eqlb_internal.h"
#include "eqlb_tables_gen.h"

namespace eqlb
{

constexpr int PROJ_MAX_ND = 10, PROJ_MAX_NQ = 64;

// out[(cell*nd + i)*bs + cb] = sum_q Pm[i][q] * qv[(cell*nq + q)*bs + cb]   (pure write, :163-182)
__global__ void __launch_bounds__(256)
k_project_dg(int64_t nitems, int nd, int nq, int bs, const double* __restrict__ Pm,
             const double* __restrict__ qv, double* __restrict__ out)
{
  __shared__ double sP[PROJ_MAX_ND * PROJ_MAX_NQ];
  for (int i = threadIdx.x; i < nd * nq; i += blockDim.x)
    sP[i] = Pm[i];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // (cell, component)
  if (t >= nitems)
    return;
  const int64_t cell = t / bs;
  const int cb = (int)(t - cell * bs);
  double acc[PROJ_MAX_ND];
#pragma unroll
  for (int i = 0; i < PROJ_MAX_ND; ++i)
    acc[i] = 0.0;
  const double* v = qv + cell * nq * bs + cb;
  for (int q = 0; q < nq; ++q)
  {
    const double f = v[(int64_t)q * bs];
#pragma unroll
    for (int i = 0; i < PROJ_MAX_ND; ++i)
      if (i < nd)
        acc[i] += sP[i * nq + q] * f;
  }
  double* o = out + cell * nd * bs + cb;
#pragma unroll
  for (int i = 0; i < PROJ_MAX_ND; ++i)
    if (i < nd)
      o[(int64_t)i * bs] = acc[i];
}

template <int DEG>
static void proj_matrix_t(int nq, const double* pts, const double* wts, std::vector<double>& Pm)
{
  using L = eqlb_tables::Lag<DEG>;
  constexpr int nd = L::ND;
  // Psi[q][i] = psi_i(x_q) from the monomial coefficients
  std::vector<double> psi((size_t)nq * nd);
  for (int q = 0; q < nq; ++q)
  {
    double mono[nd];
    int m = 0;
    for (int d = 0; d <= DEG; ++d)
      for (int a = d; a >= 0; --a, ++m)
      {
        double v = 1.0;
        for (int e = 0; e < a; ++e)
          v *= pts[2 * q];
        for (int e = 0; e < d - a; ++e)
          v *= pts[2 * q + 1];
        mono[m] = v;
      }
    for (int i = 0; i < nd; ++i)
    {
      double s = 0.0;
      for (int mm = 0; mm < nd; ++mm)
        s += L::COEF[i * nd + mm] * mono[mm];
      psi[(size_t)q * nd + i] = s;
    }
  }
  Pm.assign((size_t)nd * nq, 0.0);
  for (int i = 0; i < nd; ++i)
    for (int q = 0; q < nq; ++q)
    {
      double s = 0.0;
      for (int j = 0; j < nd; ++j)
        s += L::MINV[i * nd + j] * psi[(size_t)q * nd + j];
      Pm[(size_t)i * nq + q] = s * wts[q];
    }
}

int projection_matrix_host(int degree, int nq, const double* pts, const double* wts,
                           std::vector<double>& Pm)
{
  if (nq < 1 || nq > PROJ_MAX_NQ)
    return EQLB_ERR_INVALID_ARGUMENT;
  switch (degree)
  {
  case 0:
    proj_matrix_t<0>(nq, pts, wts, Pm);
    return 0;
  case 1:
    proj_matrix_t<1>(nq, pts, wts, Pm);
    return 0;
  case 2:
    proj_matrix_t<2>(nq, pts, wts, Pm);
    return 0;
  case 3:
    proj_matrix_t<3>(nq, pts, wts, Pm);
    return 0;
  }
  return EQLB_ERR_UNSUPPORTED;
}

void launch_project_dg(int64_t ncells, int nd, int nq, int bs, const double* Pm, const double* qv,
                       double* out, hipStream_t stream)
{
  const int64_t nitems = ncells * bs;
  const int block = 256;
  const int64_t grid = (nitems + block - 1) / block;
  hipLaunchKernelGGL(k_project_dg, dim3((unsigned)grid), dim3(block), 0, stream, nitems, nd, nq, bs, Pm,
                     qv, out);
}

} // namespace eqlb
