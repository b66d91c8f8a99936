// Upper bound of the patch-wise squared Korn constant (Kim's formula for star-shaped domains),
// OrientedPatch::estimate_squared_korn_constant (cpp/dolfinx_eqlb/se/Patch.cpp:130-334), and its
// accumulation over the node loop (se/reconstruction.hpp:291-304): every patch adds
// (gdim+1) c_K^2 to all of its cells.  One thread per patch walks the cached fan (slot arrays of
// the patch builder) and writes c_K^2 of its node; a second streaming kernel gathers the three
// vertex values per cell (no atomics, fixed order -> bitwise reproducible).
#include "eqlb_internal.h"

namespace eqlb
{

struct KornArgs
{
  int32_t nnodes, ncells;
  const double* x;            // [nnodes][3]
  const int32_t* cell_nodes;  // [ncells][3]
  const int64_t* node_slot;   // first lane slot of the node's patch or -1
  const int64_t* node_patch;
  const int32_t* slot_cell;
  const uint32_t* slot_info;
  const uint8_t* pn;
  const uint8_t* pflag;       // rhs 0
  double* cks;                // [nnodes]
};

__device__ inline void xy(const double* x, int32_t node, double& a, double& b)
{
  a = x[3 * (int64_t)node];
  b = x[3 * (int64_t)node + 1];
}

__global__ void __launch_bounds__(256) k_korn_patch(KornArgs a)
{
  const int32_t node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= a.nnodes)
    return;
  const int64_t slot0 = a.node_slot[node];
  if (slot0 < 0)
  {
    a.cks[node] = 0.0;
    return;
  }
  const int64_t patch = a.node_patch[node];
  const int n = a.pn[patch];
  const bool interior = (a.pflag[patch] & PFLAG_INTERIOR) != 0;
  const double pi = 3.14159265358979323846;
  double xi0, xi1;
  xy(a.x, node, xi0, xi1);

  auto cell_of = [&](int aa) { return a.slot_cell[slot0 + aa - 1]; }; // patch cell T_aa, aa = 1..n
  // outer node (the one that is not the patch node) of patch facet E_aa, aa = 0..n
  auto outer_node = [&](int aa) {
    const int lane = (aa == 0) ? 0 : aa - 1;
    const uint32_t info = a.slot_info[slot0 + lane];
    const int f = (aa == 0) ? ((info >> INFO_FM_SHIFT) & 3) : ((info >> INFO_FP_SHIFT) & 3);
    const int ln = (info >> INFO_LN_SHIFT) & 3;
    return a.cell_nodes[3 * (int64_t)a.slot_cell[slot0 + lane] + (3 - f - ln)];
  };

  double theta_min;
  if (interior)
  {
    theta_min = 0.5 * pi;
    for (int aa = 1; aa <= n; ++aa)
    {
      const int32_t* cn = a.cell_nodes + 3 * (int64_t)cell_of(aa);
      int32_t b[2];
      int cnt = 0;
      for (int j = 0; j < 3; ++j)
        if (cn[j] != node)
          b[cnt++] = cn[j];
      double b00, b01, b10, b11;
      xy(a.x, b[0], b00, b01);
      xy(a.x, b[1], b10, b11);
      const double v20 = b10 - b00, v21 = b11 - b01;
      const double abs_v2 = sqrt(v20 * v20 + v21 * v21);
      double v10 = xi0 - b00, v11 = xi1 - b01;
      double abs_v1 = sqrt(v10 * v10 + v11 * v11);
      theta_min = fmin(theta_min, acos((v10 * v20 + v11 * v21) / (abs_v1 * abs_v2)));
      v10 = xi0 - b10;
      v11 = xi1 - b11;
      abs_v1 = sqrt(v10 * v10 + v11 * v11);
      theta_min = fmin(theta_min, acos(-(v10 * v20 + v11 * v21) / (abs_v1 * abs_v2)));
    }
  }
  else
  {
    const int nf = n + 1;
    double cn0[3] = {0, 0, 0}, cn1[3] = {0, 0, 0};
    auto add_centroid = [&](int j, int aa) {
      const int32_t* en = a.cell_nodes + 3 * (int64_t)cell_of(aa);
      for (int q = 0; q < 3; ++q)
      {
        double p0, p1;
        xy(a.x, en[q], p0, p1);
        cn0[j] += p0 / 3;
        cn1[j] += p1 / 3;
      }
    };
    auto add_midpoint = [&](int j, int aa) {
      double p0, p1;
      xy(a.x, outer_node(aa), p0, p1);
      cn0[j] += 0.5 * xi0 + 0.5 * p0; // facet nodes: patch node and the outer node (order-free sum)
      cn1[j] += 0.5 * xi1 + 0.5 * p1;
    };
    if (n % 2 == 0)
    {
      const int h = n / 2;
      add_centroid(0, h);
      add_centroid(1, h + 1);
      add_midpoint(2, h);
    }
    else
    {
      const int h = nf / 2;
      add_midpoint(0, h);
      add_midpoint(1, h - 1);
      add_centroid(2, h);
    }
    double phi_min[3] = {pi, pi, pi};
    int32_t node_i = node;
    double xc0 = xi0, xc1 = xi1;
    double p0, p1;
    xy(a.x, outer_node(n), p0, p1);
    double v20 = p0 - xc0, v21 = p1 - xc1;
    double abs_v2 = sqrt(v20 * v20 + v21 * v21);
    for (int i = 0; i < nf; ++i)
    {
      const int32_t node_ip1 = outer_node(i);
      xy(a.x, node_ip1, p0, p1);
      const double v30 = p0 - xc0, v31 = p1 - xc1;
      const double abs_v3 = sqrt(v30 * v30 + v31 * v31);
      for (int j = 0; j < 3; ++j)
      {
        const double v10 = cn0[j] - xc0, v11 = cn1[j] - xc1;
        const double abs_v1 = sqrt(v10 * v10 + v11 * v11);
        phi_min[j] = fmin(phi_min[j], acos((v10 * v20 + v11 * v21) / (abs_v1 * abs_v2)));
        phi_min[j] = fmin(phi_min[j], acos((v10 * v30 + v11 * v31) / (abs_v1 * abs_v3)));
      }
      node_i = node_ip1;
      xc0 = p0;
      xc1 = p1;
      v20 = -v30;
      v21 = -v31;
      abs_v2 = abs_v3;
    }
    (void)node_i;
    theta_min = fmax(fmax(phi_min[0], phi_min[1]), phi_min[2]);
  }
  const double sn = sin(theta_min / 2);
  a.cks[node] = 2.0 / (sn * sn);
}

// korn[cell] += (gdim + 1) * (cks[v0] + cks[v1] + cks[v2])   (se/reconstruction.hpp:295-303)
__global__ void __launch_bounds__(256)
k_korn_cells(int32_t ncells, const int32_t* cell_nodes, const double* cks, double* korn)
{
  const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells)
    return;
  const int32_t* cn = cell_nodes + 3 * (int64_t)c;
  korn[c] += 3.0 * cks[cn[0]] + 3.0 * cks[cn[1]] + 3.0 * cks[cn[2]];
}

void launch_korn(const DeviceMesh& m, const int64_t* node_slot, const int64_t* node_patch,
                 const int32_t* slot_cell, const uint32_t* slot_info, const uint8_t* pn,
                 const uint8_t* pflag, double* cks, double* korn, hipStream_t stream)
{
  KornArgs a{m.nnodes, m.ncells, m.x, m.cell_nodes, node_slot, node_patch, slot_cell, slot_info, pn, pflag, cks};
  hipLaunchKernelGGL(k_korn_patch, dim3((m.nnodes + 255) / 256), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_korn_cells, dim3((m.ncells + 255) / 256), dim3(256), 0, stream, m.ncells,
                     m.cell_nodes, cks, korn);
}

} // namespace eqlb
