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
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

__global__ void k_tiling_touch(int* p)
{
  if (p && threadIdx.x == 1024)
    *p = 0;
}

// HIP loads the code object of a translation unit at the first launch of one of its kernels (3 - 4 ms for this
// one with its rocPRIM sort kernels): done when the mesh is created, not inside the first eqlb_se_set_boundary
void device_tiling_prepare()
{
  hipLaunchKernelGGL(k_tiling_touch, dim3(1), dim3(1), 0, 0, (int*)nullptr);
  (void)hipGetLastError();
}

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

  // EQLB_PROFILE_SETUP=1: wall time of the stages on stderr
  const bool prof = getenv("EQLB_PROFILE_SETUP") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!prof)
      return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[eqlb setup]   device tiling: %-20s %6.2f ms\n", what,
            std::chrono::duration<double, std::milli>(t1 - t_last).count());
    t_last = t1;
  };
  // one allocation for everything (a dozen hipMalloc / hipFree pairs cost more than the sorts)
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_f = up(sizeof(float) * (size_t)nc), b_i = up(sizeof(int32_t) * (size_t)nc), b_k = up(8 * (size_t)nc),
               b_s = up(4 * (size_t)nseg_all), b_b = up(16 * (size_t)maxseg);
  size_t tmp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                (int32_t*)nullptr, (int32_t*)nullptr, (size_t)nc, 0u, 64u, (hipStream_t)0)
      != hipSuccess)
    return -1;
  const size_t total = 2 * b_f + 2 * b_i + 2 * b_k + 5 * b_s + 2 * b_b + 256 + up(tmp_bytes);
  DevBuf pool;
  if (!pool.alloc(total))
    return -1;
  char* base = pool.as<char>();
  auto take = [&](size_t b) {
    char* q = base;
    base += b;
    return q;
  };
  float* cx = reinterpret_cast<float*>(take(b_f));
  float* cy = reinterpret_cast<float*>(take(b_f));
  int32_t* ord2[2] = {reinterpret_cast<int32_t*>(take(b_i)), reinterpret_cast<int32_t*>(take(b_i))};
  unsigned long long* keys2[2] = {reinterpret_cast<unsigned long long*>(take(b_k)),
                                  reinterpret_cast<unsigned long long*>(take(b_k))};
  int32_t* s_off = reinterpret_cast<int32_t*>(take(b_s));
  int32_t* s_n = reinterpret_cast<int32_t*>(take(b_s));
  int32_t* s_nl = reinterpret_cast<int32_t*>(take(b_s));
  int32_t* s_left = reinterpret_cast<int32_t*>(take(b_s));
  int32_t* s_right = reinterpret_cast<int32_t*>(take(b_s));
  float* box2[2] = {reinterpret_cast<float*>(take(b_b)), reinterpret_cast<float*>(take(b_b))};
  unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(take(256));
  void* d_tmp = take(up(tmp_bytes));
  if (hipMemcpyAsync(s_off, t.off.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice, 0) != hipSuccess
      || hipMemcpyAsync(s_n, t.n.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice, 0) != hipSuccess
      || hipMemcpyAsync(s_nl, t.nl.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice, 0) != hipSuccess
      || hipMemcpyAsync(s_left, t.left.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice, 0) != hipSuccess
      || hipMemcpyAsync(s_right, t.right.data(), 4 * (size_t)nseg_all, hipMemcpyHostToDevice, 0) != hipSuccess
      || hipMemsetAsync(d_cnt, 0, 8, 0) != hipSuccess)
    return -1;
  lap("alloc + tree");
  const unsigned grid = (unsigned)((nc + 255) / 256);
  hipLaunchKernelGGL(k_centroids, dim3(grid), dim3(256), 0, 0, nc, m.x, m.cell_nodes, blo[0], blo[1], inv, cx, cy,
                     ord2[0], d_cnt);
  unsigned long long nstretched = 0;
  if (hipMemcpy(&nstretched, d_cnt, 8, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  if (nstretched * 1000ull > (unsigned long long)nc)
    return 1;
  {
    // root box: the box of the NODES in the scale of the centroids (the centroids lie inside it; only the ratio
    // of its sides matters for the choice of the first cut)
    float b[4] = {0.0f, (float)((bhi[0] - blo[0]) * 3.0 * inv), 0.0f, (float)((bhi[1] - blo[1]) * 3.0 * inv)};
    if (hipMemcpy(box2[0], b, sizeof(b), hipMemcpyHostToDevice) != hipSuccess)
      return -1;
  }
  lap("centroids");
  unsigned end_bit = 32;
  for (int32_t v = maxseg; v > 0; v >>= 1)
    ++end_bit;
  int cur = 0, bx = 0;
  for (int l = 0; l < nlevels; ++l)
  {
    const int32_t first = t.level_begin[l], nseg = t.level_begin[l + 1] - first;
    bool any = false;
    for (int32_t s = 0; s < nseg && !any; ++s)
      any = t.nl[first + s] > 0;
    if (!any)
      break;
    hipLaunchKernelGGL(k_level_keys, dim3(grid), dim3(256), 0, 0, nc, nseg, s_off + first, s_nl + first, box2[bx], cx,
                       cy, ord2[cur], keys2[0]);
    size_t tb = tmp_bytes;
    if (rocprim::radix_sort_pairs(d_tmp, tb, keys2[0], keys2[1], ord2[cur], ord2[1 - cur], (size_t)nc, 0u, end_bit,
                                  (hipStream_t)0)
        != hipSuccess)
      return -1;
    cur = 1 - cur;
    if (l + 1 < nlevels)
    {
      const int32_t nfirst = t.level_begin[l + 1];
      hipLaunchKernelGGL(k_level_children, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, 0, nseg, s_off + first,
                         s_n + first, s_nl + first, s_left + first, s_right + first, first, nfirst, box2[bx],
                         box2[1 - bx], keys2[1]);
      bx = 1 - bx;
    }
  }
  lap("levels");
  // ascending cell ids inside a tile
  hipLaunchKernelGGL(k_tile_keys, dim3(grid), dim3(256), 0, 0, nc, tc, ord2[cur], keys2[0]);
  {
    size_t tb = tmp_bytes;
    unsigned tile_bits = 0;
    for (int32_t v = ntiles; v > 0; v >>= 1)
      ++tile_bits;
    if (rocprim::radix_sort_pairs(d_tmp, tb, keys2[0], keys2[1], ord2[cur], ord2[1 - cur], (size_t)nc, 0u,
                                  32u + tile_bits, (hipStream_t)0)
        != hipSuccess)
      return -1;
    cur = 1 - cur;
  }
  order.resize((size_t)nc);
  if (hipMemcpy(order.data(), ord2[cur], sizeof(int32_t) * (size_t)nc, hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  lap("tile sort + download");
  return (hipGetLastError() == hipSuccess) ? 0 : -1;
}

} // namespace eqlb
