// Acceptance predicates and estimator quantities on the device - the step right after the
// equilibration in the reference's workflows: the divergence / jump checks of
// python/dolfinx_eqlb/eqlb/check_eqlb_conditions.py:183-359 and the cell-wise flux indicator
// of demo/poisson/demo_error_estimation.py:52-124: || sigma_eq ||^2_T for the discontinuous SE flux
// (alpha = 0, beta = 1: the total flux is sigma_eq + G), || sigma_eq - G ||^2_T for a conforming (EV)
// flux given in the broken layout (alpha = -1, beta = 0: the total flux is sigma_eq itself).  Quadrature-free like the patch kernel: with rho = detJ * (Pi f - div(sigma_eq + G)) in
// P_{k-1}(ref) and m_q = int rho mono_q,
//   || Pi f - div(sigma_eq + G) ||^2_T = m^T GMI m / |detJ|,   || sigma_eq ||^2_T = c^T (sum_x g_x S_x) c,
// and the normal-flux jump of sigma_eq + G on an interior facet from the outward moments of both
// cells (reversal matrix B when their facet parameters run against each other).
#include "eqlb_device_common.h"
#include <cmath>
#include <vector>
#include "eqlb_tables_gen.h"

namespace eqlb
{

template <int K>
struct EstTables
{
  static constexpr int DEG = K - 1;
  using R = eqlb_tables::Ref<K, DEG>;
  static constexpr int NRT = R::NRT, ND = R::ND, NQ = R::NQ;
  static constexpr int OFF_S = 0, OFF_HG = OFF_S + R::S_SIZE, OFF_DM = OFF_HG + R::HG_SIZE,
                       OFF_GMI = OFF_DM + R::DM_SIZE, OFF_F0 = OFF_GMI + R::GMI_SIZE,
                       OFF_MRD = OFF_F0 + R::F0_SIZE, OFF_MPS = OFF_MRD + R::MRD_SIZE,
                       TOTAL = OFF_MPS + R::MPS_SIZE;
  static void fill(std::vector<double>& t)
  {
    t.clear();
    t.insert(t.end(), R::S, R::S + R::S_SIZE);
    t.insert(t.end(), R::HG, R::HG + R::HG_SIZE);
    t.insert(t.end(), R::DM, R::DM + R::DM_SIZE);
    t.insert(t.end(), R::GMI, R::GMI + R::GMI_SIZE);
    t.insert(t.end(), R::F0, R::F0 + R::F0_SIZE);
    t.insert(t.end(), R::MRD, R::MRD + R::MRD_SIZE);
    t.insert(t.end(), R::MPS, R::MPS + R::MPS_SIZE);
  }
};

// one thread per cell: divergence residual and flux norm
template <int K>
__global__ void __launch_bounds__(256)
k_estimate_cells(int32_t ncells, const double* __restrict__ tab, const double* __restrict__ cellJ,
                 const double* __restrict__ x_eq, const double* __restrict__ flux_dg,
                 const double* __restrict__ rhs_dg, double* __restrict__ div2, double* __restrict__ sig2,
                 const double alpha, const double beta)
{
  using E = EstTables<K>;
  constexpr int NRT = E::NRT, ND = E::ND, NQ = E::NQ;
  extern __shared__ double st[];
  for (int i = threadIdx.x; i < E::TOTAL; i += 256)
    st[i] = tab[i];
  __syncthreads();
  const int32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells)
    return;
  const double* J = cellJ + 4 * (int64_t)c;
  const double J00 = J[0], J01 = J[1], J10 = J[2], J11 = J[3];
  const double detJ = J00 * J11 - J01 * J10, ia = 1.0 / fabs(detJ);
  const double a00 = J11, a01 = -J01, a10 = -J10, a11 = J00; // adj = detJ * K
  const double* co = x_eq + (int64_t)c * NRT;
  double cf[NRT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
    cf[i] = co[i];
  if (div2)
  {
    double m[NQ];
    // moments of the reference divergence: q = 0 from the zero-order facet DOFs (facet 1 measures the
    // outward flux, facets 0 and 2 the inward one), q >= 1 are the div DOFs themselves
    m[0] = -(-cf[0] + cf[K] - cf[2 * K]);
#pragma unroll
    for (int q = 1; q < NQ; ++q)
      m[q] = -cf[3 * K + q - 1];
    const double* G = flux_dg + (int64_t)c * ND * 2;
    const double* f = rhs_dg + (int64_t)c * ND;
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const double gx = G[2 * i], gy = G[2 * i + 1];
      const double h0 = beta * (a00 * gx + a01 * gy), h1 = beta * (a10 * gx + a11 * gy), fd = detJ * f[i];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        m[q] += fd * st[E::OFF_HG + i * NQ + q] - h0 * st[E::OFF_DM + (i * 2 + 0) * NQ + q]
                - h1 * st[E::OFF_DM + (i * 2 + 1) * NQ + q];
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      double r = 0.0;
#pragma unroll
      for (int p = 0; p < NQ; ++p)
        r += st[E::OFF_GMI + q * NQ + p] * m[p];
      s += m[q] * r;
    }
    div2[c] = s * ia;
  }
  if (sig2)
  {
    const double g0 = (J00 * J00 + J10 * J10) * ia, g1 = (J00 * J01 + J10 * J11) * ia,
                 g2 = (J01 * J01 + J11 * J11) * ia;
    double s = 0.0;
    for (int i = 0; i < NRT; ++i)
    {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < NRT; ++j)
        r += (g0 * st[E::OFF_S + i * NRT + j] + g1 * st[E::OFF_S + (NRT + i) * NRT + j]
              + g2 * st[E::OFF_S + (2 * NRT + i) * NRT + j])
             * cf[j];
      s += cf[i] * r;
    }
    // S1 holds phi_i^x phi_j^y + phi_i^y phi_j^x, so c^T S1 c counts the mixed term twice as needed
    if (alpha != 0.0)
    {
      // || sigma + alpha G ||^2 = || sigma ||^2 + 2 alpha (sigma, G) + alpha^2 (G, G)
      const double sg = (detJ > 0.0) ? 1.0 : -1.0;
      const double* G = flux_dg + (int64_t)c * ND * 2;
      double sgm = 0.0, gg = 0.0;
#pragma unroll
      for (int d = 0; d < ND; ++d)
      {
        const double gx = G[2 * d], gy = G[2 * d + 1];
        const double t0 = J00 * gx + J10 * gy, t1 = J01 * gx + J11 * gy; // J^T G_d
        for (int i = 0; i < NRT; ++i)
          sgm += cf[i] * (st[E::OFF_MRD + (i * ND + d) * 2] * t0 + st[E::OFF_MRD + (i * ND + d) * 2 + 1] * t1);
#pragma unroll
        for (int e = 0; e < ND; ++e)
          gg += st[E::OFF_MPS + d * ND + e] * (gx * G[2 * e] + gy * G[2 * e + 1]);
      }
      s += 2.0 * alpha * sg * sgm + alpha * alpha * fabs(detJ) * gg;
    }
    sig2[c] = s;
  }
}

// outward moments of (sigma_eq + G) on local facet lf of cell c
template <int K>
__device__ __forceinline__ void facet_moments(const double* st, const double* cellJ, const double* x_eq,
                                              const double* flux_dg, int32_t c, int lf, double beta,
                                              double* mu)
{
  using E = EstTables<K>;
  constexpr int NRT = E::NRT, ND = E::ND;
  const double* J = cellJ + 4 * (int64_t)c;
  const double detJ = J[0] * J[3] - J[1] * J[2];
  const double sgn = (detJ > 0.0) ? 1.0 : -1.0, pf = (lf == 1) ? sgn : -sgn;
  const double a00 = J[3], a01 = -J[1], a10 = -J[2], a11 = J[0];
  const double nx = (lf == 2) ? 0.0 : -1.0, ny = (lf == 0) ? -1.0 : ((lf == 1) ? 0.0 : 1.0);
  const double nu0 = a00 * nx + a10 * ny, nu1 = a01 * nx + a11 * ny; // adj^T N_f
#pragma unroll
  for (int j = 0; j < K; ++j)
    mu[j] = x_eq[(int64_t)c * NRT + lf * K + j];
  const double* G = flux_dg + (int64_t)c * ND * 2;
#pragma unroll
  for (int i = 0; i < ND; ++i)
  {
    const double gn = beta * (G[2 * i] * nu0 + G[2 * i + 1] * nu1);
#pragma unroll
    for (int j = 0; j < K; ++j)
      mu[j] += st[E::OFF_F0 + (lf * ND + i) * K + j] * gn;
  }
#pragma unroll
  for (int j = 0; j < K; ++j)
    mu[j] *= pf;
}

// one thread per facet: max_j |moment_j of the normal-flux jump| (0 on boundary facets)
template <int K>
__global__ void __launch_bounds__(256)
k_estimate_facets(int32_t nfacets, const double* __restrict__ tab, const double* __restrict__ cellJ,
                  const int32_t* __restrict__ cell_facets, const uint8_t* __restrict__ facet_perm,
                  const int32_t* __restrict__ facet_cells_off, const int32_t* __restrict__ facet_cells,
                  const double* __restrict__ x_eq, const double* __restrict__ flux_dg,
                  double* __restrict__ jump, const double beta)
{
  using E = EstTables<K>;
  extern __shared__ double st[];
  for (int i = threadIdx.x; i < E::TOTAL; i += 256)
    st[i] = tab[i];
  __syncthreads();
  const int32_t fct = blockIdx.x * 256 + threadIdx.x;
  if (fct >= nfacets)
    return;
  const int32_t o = facet_cells_off[fct];
  if (facet_cells_off[fct + 1] - o < 2)
  {
    jump[fct] = 0.0;
    return;
  }
  const int32_t c0 = facet_cells[o], c1 = facet_cells[o + 1];
  int l0 = 0, l1 = 0;
#pragma unroll
  for (int l = 1; l < 3; ++l)
  {
    if (cell_facets[(int64_t)c0 * 3 + l] == fct)
      l0 = l;
    if (cell_facets[(int64_t)c1 * 3 + l] == fct)
      l1 = l;
  }
  double m0[K], m1[K];
  facet_moments<K>(st, cellJ, x_eq, flux_dg, c0, l0, beta, m0);
  facet_moments<K>(st, cellJ, x_eq, flux_dg, c1, l1, beta, m1);
  const bool rev = facet_perm[(int64_t)c0 * 3 + l0] != facet_perm[(int64_t)c1 * 3 + l1];
  double worst = 0.0;
#pragma unroll
  for (int j = 0; j < K; ++j)
  {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      t += (rev ? bcoef(j, i) : ((i == j) ? 1.0 : 0.0)) * m1[i];
    worst = fmax(worst, fabs(m0[j] + t));
  }
  jump[fct] = worst;
}

template <int K>
static int launch_estimate_k(const DeviceMesh& m, int nrhs, const double* x_eq, const double* flux_dg,
                             const double* rhs_dg, double* div2, double* sig2, double* jump, double alpha,
                             double beta, hipStream_t stream)
{
  using E = EstTables<K>;
  std::vector<double> t;
  E::fill(t);
  double* d_t = nullptr;
  if (hipMalloc(&d_t, t.size() * sizeof(double)) != hipSuccess)
    return EQLB_ERR_DEVICE;
  hipError_t e = hipMemcpyAsync(d_t, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, stream);
  const size_t lds = t.size() * sizeof(double);
  const int64_t nx = (int64_t)m.ncells * E::NRT, ng = (int64_t)m.ncells * E::ND * 2, nf = (int64_t)m.ncells * E::ND;
  for (int r = 0; r < nrhs && e == hipSuccess; ++r)
  {
    if (div2 || sig2)
      hipLaunchKernelGGL(k_estimate_cells<K>, dim3((m.ncells + 255) / 256), dim3(256), lds, stream, m.ncells,
                         d_t, m.cellJ, x_eq + r * nx, flux_dg + r * ng, rhs_dg + r * nf,
                         div2 ? div2 + (int64_t)r * m.ncells : nullptr,
                         sig2 ? sig2 + (int64_t)r * m.ncells : nullptr, alpha, beta);
    if (jump)
      hipLaunchKernelGGL(k_estimate_facets<K>, dim3((m.nfacets + 255) / 256), dim3(256), lds, stream,
                         m.nfacets, d_t, m.cellJ, m.cell_facets, m.facet_perm, m.facet_cells_off,
                         m.facet_cells, x_eq + r * nx, flux_dg + r * ng, jump + (int64_t)r * m.nfacets, beta);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(stream); // the table buffer is freed below
  (void)hipFree(d_t);
  return (e == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_estimate(const DeviceMesh& m, int k, int nrhs, const double* x_eq, const double* flux_dg,
                    const double* rhs_dg, double* div2, double* sig2, double* jump, double alpha,
                    double beta, hipStream_t stream)
{
  if (k == 1)
    return launch_estimate_k<1>(m, nrhs, x_eq, flux_dg, rhs_dg, div2, sig2, jump, alpha, beta, stream);
  if (k == 2)
    return launch_estimate_k<2>(m, nrhs, x_eq, flux_dg, rhs_dg, div2, sig2, jump, alpha, beta, stream);
  if (k == 3)
    return launch_estimate_k<3>(m, nrhs, x_eq, flux_dg, rhs_dg, div2, sig2, jump, alpha, beta, stream);
  if (k == 4)
    return launch_estimate_k<4>(m, nrhs, x_eq, flux_dg, rhs_dg, div2, sig2, jump, alpha, beta, stream);
  return EQLB_ERR_UNSUPPORTED;
}

// ---- stress estimator (demo/elasticity/demo_error_estimation.py:49-148) --------------------------------
// delta_sigma = (row 0; row 1) of an equilibrated stress.  Per cell, with W^{ab}_{rs} = c_r^T SU^{ab} c_s on the
// reference cell and int_T sigma_r^x sigma_s^y = (1/|detJ|) sum_ab J_xa J_yb W^{ab}_{rs}:
//   energy = int delta_sigma : A delta_sigma,  A tau = (tau - pi_1/(2 + 2 pi_1) tr(tau) I) / 2        (:100-102,109)
//   wsym   = int (C_K (dsig_01 - dsig_10) / 2)^2                                                         (:108,121)
//   asym3  = int (dsig_01 - dsig_10) hat_v  per vertex (the weak symmetry condition,
//            python/dolfinx_eqlb/eqlb/check_eqlb_conditions.py:476-521, before assembly over the nodes)
template <int K>
__global__ void __launch_bounds__(256)
k_estimate_stress_cells(int32_t ncells, const double* __restrict__ tab, const double* __restrict__ cellJ,
                        const double* __restrict__ x0, const double* __restrict__ x1,
                        const double* __restrict__ korn, const double pi_1, double* __restrict__ energy,
                        double* __restrict__ wsym, double* __restrict__ asym3)
{
  using R = eqlb_tables::Ref<K, K - 1>;
  constexpr int NRT = R::NRT, NSU = R::SU_SIZE, NV = R::V_SIZE;
  extern __shared__ double st[];
  for (int i = threadIdx.x; i < NSU + NV; i += 256)
    st[i] = tab[i];
  __syncthreads();
  const int32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells)
    return;
  const double* J = cellJ + 4 * (int64_t)c;
  const double Jm[2][2] = {{J[0], J[1]}, {J[2], J[3]}};
  const double detJ = J[0] * J[3] - J[1] * J[2], ia = 1.0 / fabs(detJ);
  double c0[NRT], c1[NRT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
  {
    c0[i] = x0[(int64_t)c * NRT + i];
    c1[i] = x1[(int64_t)c * NRT + i];
  }
  // W[t][r][s] = c_r^T SU_t c_s, t = (xx, xy, yy)
  double W[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
  {
    double w00 = 0.0, w01 = 0.0, w10 = 0.0, w11 = 0.0;
    for (int i = 0; i < NRT; ++i)
    {
      double u0 = 0.0, u1 = 0.0;
#pragma unroll
      for (int j = 0; j < NRT; ++j)
      {
        const double a = st[(t * NRT + i) * NRT + j];
        u0 += a * c0[j];
        u1 += a * c1[j];
      }
      w00 += c0[i] * u0;
      w01 += c0[i] * u1;
      w10 += c1[i] * u0;
      w11 += c1[i] * u1;
    }
    W[t][0][0] = w00;
    W[t][0][1] = w01;
    W[t][1][0] = w10;
    W[t][1][1] = w11;
  }
  // int_T sigma_r^x sigma_s^y
  auto prod = [&](int x, int y, int r, int s) {
    return ia
           * (Jm[x][0] * Jm[y][0] * W[0][r][s] + Jm[x][0] * Jm[y][1] * W[1][r][s] + Jm[x][1] * Jm[y][0] * W[1][s][r]
              + Jm[x][1] * Jm[y][1] * W[2][r][s]);
  };
  const double fro = prod(0, 0, 0, 0) + prod(1, 1, 0, 0) + prod(0, 0, 1, 1) + prod(1, 1, 1, 1);
  const double tr2 = prod(0, 0, 0, 0) + 2.0 * prod(0, 1, 0, 1) + prod(1, 1, 1, 1);
  const double as2 = prod(1, 1, 0, 0) - 2.0 * prod(1, 0, 0, 1) + prod(0, 0, 1, 1);
  const double ck = korn ? korn[c] : 1.0;
  if (energy)
    energy[c] = 0.5 * (fro - pi_1 / (2.0 + 2.0 * pi_1) * tr2);
  if (wsym)
    wsym[c] = 0.25 * ck * ck * as2;
  if (asym3)
  {
    const double sg = (detJ > 0.0) ? 1.0 : -1.0;
    const double* V = st + NSU; // V[v][i][a] = int hat_v phi_i^a
#pragma unroll
    for (int v = 0; v < 3; ++v)
    {
      double r = 0.0;
      for (int i = 0; i < NRT; ++i)
      {
        const double v0 = V[(v * NRT + i) * 2], v1 = V[(v * NRT + i) * 2 + 1];
        r += c0[i] * (Jm[1][0] * v0 + Jm[1][1] * v1) - c1[i] * (Jm[0][0] * v0 + Jm[0][1] * v1);
      }
      asym3[(int64_t)c * 3 + v] = sg * r;
    }
  }
}

// node value = sum of the vertex values of the cells around the node, in the order of the CSR list
__global__ void __launch_bounds__(256)
k_gather_vertex_values(int32_t nnodes, const int32_t* __restrict__ node_cells_off,
                       const int32_t* __restrict__ node_cells, const int32_t* __restrict__ cell_nodes,
                       const double* __restrict__ val3, double* __restrict__ out)
{
  const int32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= nnodes)
    return;
  double s = 0.0;
  for (int32_t o = node_cells_off[n]; o < node_cells_off[n + 1]; ++o)
  {
    const int32_t c = node_cells[o];
    const int v = (cell_nodes[(int64_t)c * 3 + 1] == n) ? 1 : ((cell_nodes[(int64_t)c * 3 + 2] == n) ? 2 : 0);
    s += val3[(int64_t)c * 3 + v];
  }
  out[n] = s;
}

template <int K>
static int launch_estimate_stress_k(const DeviceMesh& m, const int32_t* node_cells, const double* x0,
                                    const double* x1, const double* korn, double pi_1, double* energy,
                                    double* wsym, double* node_asym, hipStream_t stream)
{
  using R = eqlb_tables::Ref<K, K - 1>;
  std::vector<double> t(R::SU, R::SU + R::SU_SIZE);
  t.insert(t.end(), R::V, R::V + R::V_SIZE);
  double *d_t = nullptr, *d_a = nullptr;
  if (hipMalloc(&d_t, t.size() * sizeof(double)) != hipSuccess)
    return EQLB_ERR_DEVICE;
  if (node_asym && hipMalloc(&d_a, sizeof(double) * 3 * (size_t)m.ncells) != hipSuccess)
  {
    (void)hipFree(d_t);
    return EQLB_ERR_DEVICE;
  }
  hipError_t e = hipMemcpyAsync(d_t, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess)
  {
    hipLaunchKernelGGL(k_estimate_stress_cells<K>, dim3((m.ncells + 255) / 256), dim3(256),
                       t.size() * sizeof(double), stream, m.ncells, d_t, m.cellJ, x0, x1, korn, pi_1, energy, wsym,
                       d_a);
    if (node_asym)
      hipLaunchKernelGGL(k_gather_vertex_values, dim3((m.nnodes + 255) / 256), dim3(256), 0, stream, m.nnodes,
                         m.node_cells_off, node_cells, m.cell_nodes, d_a, node_asym);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(stream); // the buffers are freed below
  (void)hipFree(d_t);
  (void)hipFree(d_a);
  return (e == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_estimate_stress(const DeviceMesh& m, const int32_t* node_cells, int k, const double* x0,
                           const double* x1, const double* korn, double pi_1, double* energy, double* wsym,
                           double* node_asym, hipStream_t stream)
{
  if (k == 1)
    return launch_estimate_stress_k<1>(m, node_cells, x0, x1, korn, pi_1, energy, wsym, node_asym, stream);
  if (k == 2)
    return launch_estimate_stress_k<2>(m, node_cells, x0, x1, korn, pi_1, energy, wsym, node_asym, stream);
  if (k == 3)
    return launch_estimate_stress_k<3>(m, node_cells, x0, x1, korn, pi_1, energy, wsym, node_asym, stream);
  if (k == 4)
    return launch_estimate_stress_k<4>(m, node_cells, x0, x1, korn, pi_1, energy, wsym, node_asym, stream);
  return EQLB_ERR_UNSUPPORTED;
}

// ---- data oscillation ((h_T / pi) || f - div sigma ||_T of demo/poisson/demo_error_estimation.py:93-100,
//      with the Korn constant in front for stresses, demo/elasticity/demo_error_estimation.py:104-106) -------
// detJ * div(sigma_eq + beta G) is a polynomial of P_{k-1}(ref): its monomial coefficients are GMI * (moments),
// the moments as in k_estimate_cells.  f comes as point values at the images of a reference rule.
// qtab: [nq][NQ] monomial values at the points, then [nq] weights.
template <int K>
__global__ void __launch_bounds__(256)
k_oscillation_cells(int32_t ncells, const double* __restrict__ tab, const double* __restrict__ qtab, int nq,
                    const double* __restrict__ cellJ, const double* __restrict__ x_eq,
                    const double* __restrict__ flux_dg, const double* __restrict__ fvalues,
                    const double* __restrict__ korn, double* __restrict__ out)
{
  using E = EstTables<K>;
  constexpr int NRT = E::NRT, ND = E::ND, NQ = E::NQ;
  extern __shared__ double st[];
  double* sq = st + E::TOTAL;
  for (int i = threadIdx.x; i < E::TOTAL; i += 256)
    st[i] = tab[i];
  for (int i = threadIdx.x; i < nq * (NQ + 1); i += 256)
    sq[i] = qtab[i];
  __syncthreads();
  const int32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells)
    return;
  const double* J = cellJ + 4 * (int64_t)c;
  const double J00 = J[0], J01 = J[1], J10 = J[2], J11 = J[3];
  const double detJ = J00 * J11 - J01 * J10;
  const double a00 = J11, a01 = -J01, a10 = -J10, a11 = J00;
  const double* cf = x_eq + (int64_t)c * NRT;
  double m[NQ];
  m[0] = -cf[0] + cf[K] - cf[2 * K];
#pragma unroll
  for (int q = 1; q < NQ; ++q)
    m[q] = cf[3 * K + q - 1];
  if (flux_dg)
  {
    const double* G = flux_dg + (int64_t)c * ND * 2;
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      const double gx = G[2 * i], gy = G[2 * i + 1];
      const double h0 = a00 * gx + a01 * gy, h1 = a10 * gx + a11 * gy;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        m[q] += h0 * st[E::OFF_DM + (i * 2 + 0) * NQ + q] + h1 * st[E::OFF_DM + (i * 2 + 1) * NQ + q];
    }
  }
  double a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    double r = 0.0;
#pragma unroll
    for (int p = 0; p < NQ; ++p)
      r += st[E::OFF_GMI + q * NQ + p] * m[p];
    a[q] = r / detJ;
  }
  const double* fv = fvalues + (int64_t)c * nq;
  double s = 0.0;
  for (int q = 0; q < nq; ++q)
  {
    double d = fv[q];
#pragma unroll
    for (int p = 0; p < NQ; ++p)
      d -= a[p] * sq[q * NQ + p];
    s += sq[nq * NQ + q] * d * d;
  }
  // cell diameter as dolfinx::mesh::h: the longest edge of the triangle
  const double e1 = J00 * J00 + J10 * J10, e2 = J01 * J01 + J11 * J11,
               e3 = (J01 - J00) * (J01 - J00) + (J11 - J10) * (J11 - J10);
  const double h2 = fmax(e1, fmax(e2, e3));
  const double ck = korn ? korn[c] : 1.0;
  constexpr double PI = 3.14159265358979323846;
  out[c] = ck * ck * h2 / (PI * PI) * fabs(detJ) * s;
}

template <int K>
static int launch_oscillation_k(const DeviceMesh& m, int nrhs, const double* x_eq, const double* flux_dg, int nq,
                                const double* qpoints, const double* qweights, const double* fvalues,
                                const double* korn, double* out, hipStream_t stream)
{
  using E = EstTables<K>;
  using R = eqlb_tables::Ref<K, K - 1>;
  std::vector<double> t;
  E::fill(t);
  std::vector<double> qt((size_t)nq * (E::NQ + 1));
  for (int q = 0; q < nq; ++q)
  {
    for (int p = 0; p < E::NQ; ++p)
      qt[(size_t)q * E::NQ + p] = std::pow(qpoints[2 * q], R::MONO_X[p]) * std::pow(qpoints[2 * q + 1], R::MONO_Y[p]);
    qt[(size_t)nq * E::NQ + q] = qweights[q];
  }
  double *d_t = nullptr, *d_q = nullptr;
  if (hipMalloc(&d_t, t.size() * sizeof(double)) != hipSuccess)
    return EQLB_ERR_DEVICE;
  if (hipMalloc(&d_q, qt.size() * sizeof(double)) != hipSuccess)
  {
    (void)hipFree(d_t);
    return EQLB_ERR_DEVICE;
  }
  hipError_t e = hipMemcpyAsync(d_t, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_q, qt.data(), qt.size() * sizeof(double), hipMemcpyHostToDevice, stream);
  const size_t lds = (t.size() + qt.size()) * sizeof(double);
  const int64_t nx = (int64_t)m.ncells * E::NRT, ng = (int64_t)m.ncells * E::ND * 2;
  for (int r = 0; r < nrhs && e == hipSuccess; ++r)
  {
    hipLaunchKernelGGL(k_oscillation_cells<K>, dim3((m.ncells + 255) / 256), dim3(256), lds, stream, m.ncells, d_t,
                       d_q, nq, m.cellJ, x_eq + r * nx, flux_dg ? flux_dg + r * ng : nullptr,
                       fvalues + (int64_t)r * m.ncells * nq, korn, out + (int64_t)r * m.ncells);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(stream);
  (void)hipFree(d_t);
  (void)hipFree(d_q);
  return (e == hipSuccess) ? 0 : EQLB_ERR_DEVICE;
}

int launch_oscillation(const DeviceMesh& m, int k, int nrhs, const double* x_eq, const double* flux_dg, int nq,
                       const double* qpoints, const double* qweights, const double* fvalues, const double* korn,
                       double* out, hipStream_t stream)
{
  if (k == 1)
    return launch_oscillation_k<1>(m, nrhs, x_eq, flux_dg, nq, qpoints, qweights, fvalues, korn, out, stream);
  if (k == 2)
    return launch_oscillation_k<2>(m, nrhs, x_eq, flux_dg, nq, qpoints, qweights, fvalues, korn, out, stream);
  if (k == 3)
    return launch_oscillation_k<3>(m, nrhs, x_eq, flux_dg, nq, qpoints, qweights, fvalues, korn, out, stream);
  if (k == 4)
    return launch_oscillation_k<4>(m, nrhs, x_eq, flux_dg, nq, qpoints, qweights, fvalues, korn, out, stream);
  return EQLB_ERR_UNSUPPORTED;
}


} // namespace eqlb
