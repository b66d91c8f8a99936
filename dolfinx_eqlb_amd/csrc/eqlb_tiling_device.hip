// Tiles of the tiled launches by recursive coordinate bisection ON THE DEVICE (the set-up of every handle:
// OrientedPatch / PatchData construction of the reference happens inside its timed call,
// cpp/dolfinx_eqlb/se/reconstruction.hpp:275-313, python/test/performance/perftest.py:145-147 - this is the part of
// the "cold" path that took 16 of 28 ms on the host threads).
//
// Same tree as rcb_split (eqlb_api.hip): a segment of `ntile` tiles gives its first ntile / 2 tiles (x TC cells) to
// the left child, cut across the longer side of its bounding box.  The tree's shape - offsets and sizes of all
// segments - does not depend on the data, so the host lays it out once; per level the device
//   1. builds a 64-bit key per position: (segment number of the level << 32) | coordinate along the segment's axis
//      (centroid coordinates are stored relative to the bounding box: non-negative floats order like their bits),
//   2. sorts (key, cell) pairs with ONE device-wide radix sort - segments stay in place, each is ordered along its
//      axis, so its first nl positions are its left child (rocprim::radix_sort_pairs, stable, deterministic),
//   3. derives the children's bounding boxes from the parent's box and the coordinate at the cut.
// A last sort by (tile, cell id) orders the cells inside every tile by their ids (long runs in the flush).
// Not taken for meshes with stretched cells (aspect ratio above ~3 in more than 0.1 % of the cells): there the
// host bisection chooses the cut of the last levels by the nodes it separates (rcb_split), which needs the mesh
// connectivity per trial cut.
#include "eqlb_internal.h"

#include <cstring> // (rocprim's texture iterator calls memset)
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <vector>

namespace eqlb
{
namespace
{
struct SegTree
{
  // all segments of all levels in breadth-first order; a level's segments are ordered by position
  std::vector<int32_t> off, n, nl, left, right; // nl = 0: leaf (not split)
  std::vector<int32_t> level_begin;             // [nlevels + 1] into the arrays above
};

SegTree build_tree(int32_t nc, int32_t ntiles, int tc)
{
  SegTree t;
  struct Item
  {
    int32_t off, n, ntile;
  };
  std::vector<Item> cur{{0, nc, ntiles}};
  t.level_begin.push_back(0);
  while (!cur.empty())
  {
    std::vector<Item> next;
    bool any = false;
    const int32_t base = (int32_t)t.off.size();
    for (const Item& s : cur)
    {
      t.off.push_back(s.off);
      t.n.push_back(s.n);
      if (s.ntile <= 1 || s.n <= tc)
      {
        t.nl.push_back(0);
        t.left.push_back(-1);
        t.right.push_back(-1);
        // a leaf stays a segment of the following levels (the device-wide sort must leave it in place)
        next.push_back({s.off, s.n, 1});
        continue;
      }
      any = true;
      const int32_t tl = s.ntile / 2;
      const int32_t nl = (int32_t)std::min<int64_t>(s.n, (int64_t)tl * tc);
      t.nl.push_back(nl);
      t.left.push_back(0);  // filled below
      t.right.push_back(0);
      next.push_back({s.off, nl, tl});
      next.push_back({s.off + nl, s.n - nl, s.ntile - tl});
    }
    // children ids: position of the children in the next level
    int32_t child = base + (int32_t)cur.size();
    for (size_t i = 0; i < cur.size(); ++i)
    {
      if (t.nl[base + i] == 0)
      {
        ++child; // the leaf's copy
        continue;
      }
      t.left[base + i] = child;
      t.right[base + i] = child + 1;
      child += 2;
    }
    t.level_begin.push_back((int32_t)t.off.size());
    if (!any)
      break;
    cur.swap(next);
  }
  return t;
}

__global__ void __launch_bounds__(256)
k_centroids(int32_t nc, const double* __restrict__ x, const int32_t* __restrict__ cell_nodes, double blo0, double blo1,
            double inv, float* __restrict__ cx, float* __restrict__ cy, int32_t* __restrict__ ord,
            unsigned long long* __restrict__ nstretched)
{
  const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc)
    return;
  const int32_t* cn = cell_nodes + 3 * (int64_t)c;
  const double* p0 = x + 3 * (int64_t)cn[0];
  const double* p1 = x + 3 * (int64_t)cn[1];
  const double* p2 = x + 3 * (int64_t)cn[2];
  cx[c] = (float)((p0[0] + p1[0] + p2[0] - 3.0 * blo0) * inv);
  cy[c] = (float)((p0[1] + p1[1] + p2[1] - 3.0 * blo1) * inv);
  ord[c] = c;
  const double e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e2x = p2[0] - p0[0], e2y = p2[1] - p0[1];
  const double l2 = fmax(fmax(e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y),
                         (e2x - e1x) * (e2x - e1x) + (e2y - e1y) * (e2y - e1y));
  if (l2 > 6.0 * fabs(e1x * e2y - e1y * e2x))
    atomicAdd(nstretched, 1ull);
}

// key of position p at a level: segment number (position order) and the coordinate along its axis
__global__ void __launch_bounds__(256)
k_level_keys(int32_t nc, int32_t nseg, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_nl,
             const float* __restrict__ box, const float* __restrict__ cx, const float* __restrict__ cy,
             const int32_t* __restrict__ ord, unsigned long long* __restrict__ keys)
{
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nc)
    return;
  // last segment with off <= p
  int lo = 0, hi = nseg - 1;
  while (lo < hi)
  {
    const int mid = (lo + hi + 1) >> 1;
    if (seg_off[mid] <= p)
      lo = mid;
    else
      hi = mid - 1;
  }
  unsigned int coord = 0u;
  if (seg_nl[lo] > 0)
  {
    const float* b = box + 4 * (int64_t)lo; // xlo, xhi, ylo, yhi
    const int axis = (b[1] - b[0] >= b[3] - b[2]) ? 0 : 1;
    const int32_t c = ord[p];
    coord = __float_as_uint(axis ? cy[c] : cx[c]);
  }
  else
    coord = (unsigned int)(p - seg_off[lo]); // a leaf keeps its order
  keys[p] = ((unsigned long long)lo << 32) | coord;
}

// boxes of the next level from the sorted keys of this one
__global__ void __launch_bounds__(256)
k_level_children(int32_t nseg, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_n,
                 const int32_t* __restrict__ seg_nl, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                 int32_t level_first, int32_t next_first, const float* __restrict__ box, float* __restrict__ box_next,
                 const unsigned long long* __restrict__ keys_sorted)
{
  const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg)
    return;
  const float* b = box + 4 * (int64_t)s;
  if (seg_nl[s] == 0)
  {
    // the leaf's copy: the segments of a level are numbered in position order, so is the next level
    return;
  }
  const int axis = (b[1] - b[0] >= b[3] - b[2]) ? 0 : 1;
  const int32_t cut = seg_off[s] + seg_nl[s];
  // coordinate of the first element of the right part (== of the last one when nothing is right of the cut)
  const int32_t at = (seg_nl[s] < seg_n[s]) ? cut : cut - 1;
  const float m = __uint_as_float((unsigned int)(keys_sorted[at] & 0xffffffffull));
  float* bl = box_next + 4 * (int64_t)(left[s] - next_first);
  float* br = box_next + 4 * (int64_t)(right[s] - next_first);
  for (int i = 0; i < 4; ++i)
    bl[i] = br[i] = b[i];
  bl[2 * axis + 1] = m;
  br[2 * axis] = m;
  (void)level_first;
}

__global__ void __launch_bounds__(256)
k_tile_keys(int32_t nc, int tc, const int32_t* __restrict__ ord, unsigned long long* __restrict__ keys)
{
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < nc)
    keys[p] = ((unsigned long long)(p / tc) << 32) | (unsigned int)ord[p];
}

struct DevBuf
{
  void* p = nullptr;
  ~DevBuf()
  {
    if (p)
      (void)hipFree(p);
  }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
  template <typename T>
  T* as()
  {
    return static_cast<T*>(p);
  }
};
} // namespace

// order [nc]: cells in tile order (tile t = positions [t TC, (t + 1) TC)), ascending ids inside a tile.
// Returns 0 on success, 1 if the mesh has stretched cells (the caller takes the host bisection), < 0 on a device error.
int device_tile_order(const DeviceMesh& m, int tc, int32_t ntiles, const double blo[2], const double bhi[2], double inv,
                      std::vector<int32_t>& order)
{
  const int32_t nc = m.ncells;
  const SegTree t = build_tree(nc, ntiles, tc);
  const int nlevels = (int)t.level_begin.size() - 1;
  const int32_t nseg_all = (int32_t)t.off.size();
  int32_t maxseg = 1;
  for (int l = 0; l < nlevels; ++l)
    maxseg = std::max(maxseg, t.level_begin[l + 1] - t.level_begin[l]);

  DevBuf d_cx, d_cy, d_ord[2], d_keys[2], d_off, d_n, d_nl, d_left, d_right, d_box[2], d_cnt, d_tmp;
  if (!d_cx.alloc(sizeof(float) * nc) || !d_cy.alloc(sizeof(float) * nc) || !d_ord[0].alloc(sizeof(int32_t) * nc)
      || !d_ord[1].alloc(sizeof(int32_t) * nc) || !d_keys[0].alloc(8 * (size_t)nc) || !d_keys[1].alloc(8 * (size_t)nc)
      || !d_off.alloc(4 * (size_t)nseg_all) || !d_n.alloc(4 * (size_t)nseg_all) || !d_nl.alloc(4 * (size_t)nseg_all)
      || !d_left.alloc(4 * (size_t)nseg_all) || !d_right.alloc(4 * (size_t)nseg_all)
      || !d_box[0].alloc(16 * (size_t)maxseg) || !d_box[1].alloc(16 * (size_t)maxseg) || !d_cnt.alloc(8))
    return -1;
  if (hipMemcpy(d_off.p, t.off.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice) != hipSuccess
      || hipMemcpy(d_n.p, t.n.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice) != hipSuccess
      || hipMemcpy(d_nl.p, t.nl.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice) != hipSuccess
      || hipMemcpy(d_left.p, t.left.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice) != hipSuccess
      || hipMemcpy(d_right.p, t.right.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice) != hipSuccess
      || hipMemset(d_cnt.p, 0, 8) != hipSuccess)
    return -1;
  const unsigned grid = (unsigned)((nc + 255) / 256);
  hipLaunchKernelGGL(k_centroids, dim3(grid), dim3(256), 0, 0, nc, m.x, m.cell_nodes, blo[0], blo[1], inv,
                     d_cx.as<float>(), d_cy.as<float>(), d_ord[0].as<int32_t>(), d_cnt.as<unsigned long long>());
  unsigned long long nstretched = 0;
  if (hipMemcpy(&nstretched, d_cnt.p, 8, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  if (nstretched * 1000ull > (unsigned long long)nc)
    return 1;
  {
    // root box: the box of the NODES in the scale of the centroids (the centroids lie inside it; only the ratio
    // of its sides matters for the choice of the first cut)
    float b[4] = {0.0f, (float)((bhi[0] - blo[0]) * 3.0 * inv), 0.0f, (float)((bhi[1] - blo[1]) * 3.0 * inv)};
    if (hipMemcpy(d_box[0].p, b, sizeof(b), hipMemcpyHostToDevice) != hipSuccess)
      return -1;
  }
  size_t tmp_bytes = 0;
  unsigned end_bit = 32;
  for (int32_t v = maxseg; v > 0; v >>= 1)
    ++end_bit;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys[0].as<unsigned long long>(), d_keys[1].as<unsigned long long>(),
                                d_ord[0].as<int32_t>(), d_ord[1].as<int32_t>(), (size_t)nc, 0u, 64u, (hipStream_t)0)
      != hipSuccess)
    return -1;
  if (!d_tmp.alloc(tmp_bytes))
    return -1;
  int cur = 0, bx = 0;
  for (int l = 0; l < nlevels; ++l)
  {
    const int32_t first = t.level_begin[l], nseg = t.level_begin[l + 1] - first;
    bool any = false;
    for (int32_t s = 0; s < nseg && !any; ++s)
      any = t.nl[first + s] > 0;
    if (!any)
      break;
    hipLaunchKernelGGL(k_level_keys, dim3(grid), dim3(256), 0, 0, nc, nseg, d_off.as<int32_t>() + first,
                       d_nl.as<int32_t>() + first, d_box[bx].as<float>(), d_cx.as<float>(), d_cy.as<float>(),
                       d_ord[cur].as<int32_t>(), d_keys[0].as<unsigned long long>());
    size_t tb = tmp_bytes;
    if (rocprim::radix_sort_pairs(d_tmp.p, tb, d_keys[0].as<unsigned long long>(), d_keys[1].as<unsigned long long>(),
                                  d_ord[cur].as<int32_t>(), d_ord[1 - cur].as<int32_t>(), (size_t)nc, 0u, end_bit,
                                  (hipStream_t)0)
        != hipSuccess)
      return -1;
    cur = 1 - cur;
    if (l + 1 < nlevels)
    {
      const int32_t nfirst = t.level_begin[l + 1];
      hipLaunchKernelGGL(k_level_children, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, 0, nseg,
                         d_off.as<int32_t>() + first, d_n.as<int32_t>() + first, d_nl.as<int32_t>() + first,
                         d_left.as<int32_t>() + first, d_right.as<int32_t>() + first, first, nfirst,
                         d_box[bx].as<float>(), d_box[1 - bx].as<float>(), d_keys[1].as<unsigned long long>());
      bx = 1 - bx;
    }
  }
  // ascending cell ids inside a tile
  hipLaunchKernelGGL(k_tile_keys, dim3(grid), dim3(256), 0, 0, nc, tc, d_ord[cur].as<int32_t>(),
                     d_keys[0].as<unsigned long long>());
  {
    size_t tb = tmp_bytes;
    if (rocprim::radix_sort_pairs(d_tmp.p, tb, d_keys[0].as<unsigned long long>(), d_keys[1].as<unsigned long long>(),
                                  d_ord[cur].as<int32_t>(), d_ord[1 - cur].as<int32_t>(), (size_t)nc, 0u, 64u,
                                  (hipStream_t)0)
        != hipSuccess)
      return -1;
    cur = 1 - cur;
  }
  order.resize((size_t)nc);
  if (hipMemcpy(order.data(), d_ord[cur].p, sizeof(int32_t) * (size_t)nc, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return (hipGetLastError() == hipSuccess) ? 0 : -1;
}

} // namespace eqlb
