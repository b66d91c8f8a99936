// Host side of the C ABI (include/eqlb.h): device residency of mesh / patch SoA, binning of
// patches by size, kernel launches.  Mirrors the driver se::reconstruction<T>
// (cpp/dolfinx_eqlb/se/reconstruction.hpp:337-407) with the per-call setup hoisted into the
// handle.  There is NO CPU fallback: without a HIP device every compute entry point fails.
#include "eqlb_internal.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

namespace
{
thread_local std::string g_error;

// EQLB_PROFILE_SETUP=1: wall time of the set-up phases on stderr
struct SetupTimer
{
  bool on;
  std::chrono::steady_clock::time_point t0;
  SetupTimer() : on(getenv("EQLB_PROFILE_SETUP") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* what)
  {
    if (!on)
      return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[eqlb setup] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

} // namespace
namespace eqlb
{
// error message + code for the other translation units of the C ABI (eqlb_halo_rccl.hip)
int set_error(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}
} // namespace eqlb
namespace
{

#define HIP_TRY(expr)                                                                             \
  do                                                                                              \
  {                                                                                               \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(EQLB_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));                \
  } while (0)

template <typename T>
int upload(T** dst, const T* src, size_t n)
{
  *dst = nullptr;
  if (n == 0)
    n = 1;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T)));
  if (src)
    HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

template <typename T>
void dfree(T*& p)
{
  if (p)
    (void)hipFree(p);
  p = nullptr;
}

void free_boundary(eqlb_se* h)
{
  dfree(h->facet_type);
  dfree(h->node_ws);
  dfree(h->node_group);
  dfree(h->node_wslevel);
  h->ws_levels = 1;
  dfree(h->rest_cells);
  h->nrest_cells = 0;
  dfree(h->bvals);
  dfree(h->node_slot);
  dfree(h->node_patch);
  dfree(h->slot_cell);
  dfree(h->slot_info);
  dfree(h->pn);
  dfree(h->pflag);
  dfree(h->slots); // re-zeroed on the next call (node_mask may have changed)
  dfree(h->t_tiles);
  dfree(h->t_tile_cells);
  dfree(h->t_facet_owner);
  dfree(h->t_slot_cell);
  dfree(h->t_slot_info);
  dfree(h->t_pn);
  dfree(h->t_pflag);
  h->ntiles = 0;
  h->t_mixed = false;
  h->boundary_set = false;
}
// Grouped boundary patches of the stress path (se/reconstruction.hpp:170-234, se/Patch.cpp:60-104,
// 762-784; RT_2 only): a node whose two boundary facets carry flux BCs on both stress rows
// (base/BoundaryData.cpp:611-631) and that has two cells is grouped with the adjacent internal patch.
// The reference treats the groups one after the other in node order and lets the weak-symmetry step of a
// group see what the EARLIER groups added to the global stress on the cells of its internal patch
// (se/solve_patch_weaksym.hpp:100-131 reads the global vector).  On the device all row-wise sweeps come first
// and every (cell, vertex) contribution keeps its own slot row, so "what has been added so far" is a sum of
// slot rows: own row + rows of the vertices that are two-cell members of the own group + rows of the vertices
// that belong to an EARLIER group (group ids are handed out in the reference's discovery order; the patch
// builder marks those vertices).  The symmetry step of an earlier group has modified the rows of its internal
// patch, so overlapping groups are ordered: level of a group = 1 + the highest level among the earlier groups
// that own a vertex of one of its internal patch's cells; the weak-symmetry kernel runs level by level.
// ws: 0 normal, 1 two-cell member, 2 internal patch; level [nnodes]: level of the node's group (0 elsewhere).
int find_stress_groups(const eqlb::DeviceMesh& m, const int8_t* facet_type, const uint8_t* node_mask,
                       std::vector<int8_t>& ws, std::vector<int32_t>& group, std::vector<int8_t>& level,
                       int& nlevels, bool& any)
{
  const int32_t nn = m.nnodes;
  ws.assign(nn, 0);
  group.assign(nn, -1);
  any = false;
  std::vector<int8_t> cnt(nn, 0);
  for (int r = 0; r < 2; ++r)
    for (int32_t f = 0; f < m.nfacets; ++f)
      if (facet_type[(size_t)r * m.nfacets + f] == EQLB_FACET_ESSNT_DUAL)
      {
        ++cnt[m.h_facet_nodes[2 * (size_t)f]];
        ++cnt[m.h_facet_nodes[2 * (size_t)f + 1]];
      }
  int32_t ngroups = 0;
  for (int32_t node = 0; node < nn; ++node)
  {
    if (node_mask && !node_mask[node])
      continue;
    if (cnt[node] != 4 || group[node] >= 0 || m.h_node_ncells[node] != 2)
      continue;
    int32_t inner = -1;
    for (int32_t q = m.h_node_facets_off[node]; q < m.h_node_facets_off[node + 1] && inner < 0; ++q)
    {
      const int32_t f = m.h_node_facets[q];
      if (facet_type[f] == EQLB_FACET_INTERNAL)
        inner = (m.h_facet_nodes[2 * (size_t)f] == node) ? m.h_facet_nodes[2 * (size_t)f + 1]
                                                         : m.h_facet_nodes[2 * (size_t)f];
    }
    if (inner < 0)
      continue;
    std::vector<int32_t> members{inner};
    for (int32_t q = m.h_node_cells_off[inner]; q < m.h_node_cells_off[inner + 1]; ++q)
      for (int v = 0; v < 3; ++v)
      {
        const int32_t pnt = m.h_cell_nodes[3 * (size_t)m.h_node_cells[q] + v];
        if (cnt[pnt] == 4 && m.h_node_ncells[pnt] == 2
            && std::find(members.begin(), members.end(), pnt) == members.end())
          members.push_back(pnt);
      }
    if (members.size() < 2)
      continue;
    for (int32_t nd : members)
    {
      if (group[nd] >= 0 || (node_mask && !node_mask[nd]))
        return fail(EQLB_ERR_UNSUPPORTED,
                    "Incompatible mesh! To many patches with 2 cells on neumann boundary.");
      group[nd] = ngroups;
      ws[nd] = (nd == inner) ? 2 : 1;
    }
    ++ngroups;
    any = true;
  }
  // levels of overlapping groups (ascending group id = the reference's order)
  level.assign(nn, 0);
  nlevels = 1;
  if (any)
  {
    std::vector<int32_t> inner_of(ngroups, -1);
    for (int32_t node = 0; node < nn; ++node)
      if (ws[node] == 2)
        inner_of[group[node]] = node;
    std::vector<int> glevel(ngroups, 0);
    for (int32_t g = 0; g < ngroups; ++g)
    {
      const int32_t node = inner_of[g];
      int lv = 0;
      for (int32_t q = m.h_node_cells_off[node]; q < m.h_node_cells_off[node + 1]; ++q)
        for (int v = 0; v < 3; ++v)
        {
          const int32_t nd = m.h_cell_nodes[3 * (size_t)m.h_node_cells[q] + v];
          if (group[nd] >= 0 && group[nd] < g)
            lv = std::max(lv, glevel[group[nd]] + 1);
        }
      glevel[g] = lv;
      nlevels = std::max(nlevels, lv + 1);
    }
    if (nlevels > eqlb::WS_MAX_LEVELS)
      return fail(EQLB_ERR_UNSUPPORTED, "more than %d levels of overlapping groups of boundary patches",
                  eqlb::WS_MAX_LEVELS);
    for (int32_t node = 0; node < nn; ++node)
      if (group[node] >= 0)
        level[node] = (int8_t)glevel[group[node]];
  }
  return EQLB_OK;
}

// Recursive coordinate bisection of the cell centroids into chunks of exactly `tc` cells (the last
// one may be short): compact tiles keep the share of rim patches, which are solved by every tile
// they touch, small.
// (centroids relative to the bounding box of the mesh, in single precision: the bisection only compares them,
// ties go by the cell id, and a 12-byte item moves through the selection passes twice as fast as a 24-byte one)
struct TileItem
{
  float x, y;
  int32_t cell;
};

// Context of the bisection: cell -> nodes and a per-node stamp to count the nodes a cut separates
struct RcbCtx
{
  const int32_t* cell_nodes;
  std::vector<int64_t> stamp; // [nnodes] 2 * epoch + side of the last cell that touched the node
  std::vector<int64_t> cut;   // [nnodes] epoch in which the node was counted as cut
  int64_t epoch = 0;
};

static inline bool rcb_less(const TileItem& p, const TileItem& q, int axis)
{
  const float u = axis ? p.y : p.x, v = axis ? q.y : q.x;
  return u < v || (u == v && p.cell < q.cell);
}

// std::vector without value initialisation: the big scratch arrays of the tile builder are written completely by the
// worker threads - a zero fill by the calling thread would touch (page-fault) tens of MB serially first
template <typename T>
struct default_init_alloc : std::allocator<T>
{
  template <typename U>
  struct rebind
  {
    using other = default_init_alloc<U>;
  };
  template <typename U, typename... A>
  void construct(U* p, A&&... a)
  {
    if constexpr (sizeof...(A) == 0)
      ::new (static_cast<void*>(p)) U;
    else
      ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
};
template <typename T>
using uvec = std::vector<T, default_init_alloc<T>>;

// Host worker threads of the set-up: capped (the tile builder keeps an O(nnodes) stamp per worker: 16 MB each at
// 4M nodes, on every rank of a node) and exception safe - an exception inside a std::thread would call
// std::terminate; the first one is kept and re-thrown by join() in the calling thread, where the C entry points
// turn it into an error code (EQLB_GUARD).
static int host_workers(int64_t wanted)
{
  const int64_t hw = std::max<int64_t>(1, std::min<int64_t>(std::thread::hardware_concurrency(), 32));
  return (int)std::max<int64_t>(1, std::min<int64_t>(hw, wanted));
}
struct Workers
{
  std::vector<std::thread> th;
  std::exception_ptr err;
  std::mutex mu;
  template <typename F>
  void spawn(F f)
  {
    th.emplace_back([this, f]() {
      try
      {
        f();
      }
      catch (...)
      {
        std::lock_guard<std::mutex> g(mu);
        if (!err)
          err = std::current_exception();
      }
    });
  }
  void join()
  {
    for (auto& x : th)
      x.join();
    th.clear();
    if (err)
    {
      std::exception_ptr e = err;
      err = nullptr;
      std::rethrow_exception(e);
    }
  }
  ~Workers()
  {
    for (auto& x : th)
      if (x.joinable())
        x.join();
  }
};

// The same partition as rcb_partition (the key (coordinate, cell id) is a total order, so the two halves are
// determined as SETS) on the host threads, for the few large segments at the top of the recursion where the
// subtrees do not yet occupy the cores: histogram of the coordinate -> bucket of the splitting element ->
// exact splitter inside that bucket -> counting partition through a scratch array.
static void rcb_partition_parallel(TileItem* a, int64_t n, int64_t nl, int axis, float lo, float hi,
                                   std::vector<TileItem>& tmp)
{
  const int nt = host_workers(n / (1 << 15));
  constexpr int NBK = 4096;
  const float scale = (hi > lo) ? (float)NBK / (hi - lo) : 0.0f;
  auto bucket = [&](const TileItem& t) {
    const int b = (int)(((axis ? t.y : t.x) - lo) * scale);
    return b < 0 ? 0 : (b >= NBK ? NBK - 1 : b);
  };
  auto run = [&](auto f) {
    Workers w;
    for (int t = 1; t < nt; ++t)
      w.spawn([f, t]() { f(t); });
    f(0);
    w.join();
  };
  std::vector<int64_t> hist((size_t)nt * NBK, 0);
  run([&](int t) {
    int64_t* hh = &hist[(size_t)t * NBK];
    for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i)
      ++hh[bucket(a[i])];
  });
  int bs = 0;
  int64_t before = 0;
  for (; bs < NBK; ++bs)
  {
    int64_t c = 0;
    for (int t = 0; t < nt; ++t)
      c += hist[(size_t)t * NBK + bs];
    if (before + c > nl)
      break;
    before += c;
  }
  if (bs == NBK) // nl == n: nothing to split
    return;
  // the nl-th smallest element lives in bucket bs (buckets are ordered by the coordinate)
  std::vector<TileItem> cand;
  for (int64_t i = 0; i < n; ++i)
    if (bucket(a[i]) == bs)
      cand.push_back(a[i]);
  std::nth_element(cand.begin(), cand.begin() + (nl - before), cand.end(),
                   [axis](const TileItem& p, const TileItem& q) { return rcb_less(p, q, axis); });
  const TileItem piv = cand[(size_t)(nl - before)];
  // counting partition: [elements below the splitter | the rest]
  std::vector<int64_t> cnt((size_t)nt + 1, 0);
  run([&](int t) {
    int64_t c = 0;
    for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i)
      c += rcb_less(a[i], piv, axis) ? 1 : 0;
    cnt[(size_t)t + 1] = c;
  });
  for (int t = 0; t < nt; ++t)
    cnt[(size_t)t + 1] += cnt[(size_t)t];
  if ((int64_t)tmp.size() < n)
    tmp.resize((size_t)n);
  run([&](int t) {
    const int64_t b = n * t / nt, e = n * (t + 1) / nt;
    int64_t l = cnt[(size_t)t], r = nl + (b - cnt[(size_t)t]);
    for (int64_t i = b; i < e; ++i)
    {
      if (rcb_less(a[i], piv, axis))
        tmp[(size_t)l++] = a[i];
      else
        tmp[(size_t)r++] = a[i];
    }
  });
  run([&](int t) {
    const int64_t b = n * t / nt, e = n * (t + 1) / nt;
    std::copy(tmp.begin() + b, tmp.begin() + e, a + b);
  });
}

static void rcb_partition(TileItem* a, int64_t n, int64_t nl, int axis)
{
  if (axis == 0)
    std::nth_element(a, a + nl, a + n, [](const TileItem& p, const TileItem& q) {
      return p.x < q.x || (p.x == q.x && p.cell < q.cell);
    });
  else
    std::nth_element(a, a + nl, a + n, [](const TileItem& p, const TileItem& q) {
      return p.y < q.y || (p.y == q.y && p.cell < q.cell);
    });
}

// nodes with cells on both sides of the partition [0, nl) | [nl, n): their patches are solved twice
static int64_t rcb_cut_nodes(const TileItem* a, int64_t n, int64_t nl, RcbCtx& c)
{
  const int64_t ep = ++c.epoch;
  int64_t ncut = 0;
  for (int64_t i = 0; i < n; ++i)
  {
    const int64_t tag = 2 * ep + (i < nl ? 0 : 1);
    const int32_t* cn = c.cell_nodes + 3 * (size_t)a[i].cell;
    for (int j = 0; j < 3; ++j)
    {
      int64_t& st = c.stamp[cn[j]];
      if (st / 2 == ep && st != tag && c.cut[cn[j]] != ep)
      {
        c.cut[cn[j]] = ep;
        ++ncut;
      }
      st = tag;
    }
  }
  return ncut;
}

// pool of bisection contexts for the worker threads (a context is [nnodes]-sized)
struct RcbPool
{
  const int32_t* cell_nodes;
  int32_t nnodes;
  const uint8_t* stretched = nullptr; // [ncells] 1: longest edge^2 > 6 |det J| (aspect ratio above ~3)
  std::mutex mtx;
  std::vector<std::unique_ptr<RcbCtx>> free_list;
  std::unique_ptr<RcbCtx> acquire()
  {
    {
      std::lock_guard<std::mutex> g(mtx);
      if (!free_list.empty())
      {
        auto c = std::move(free_list.back());
        free_list.pop_back();
        return c;
      }
    }
    return std::unique_ptr<RcbCtx>(new RcbCtx{cell_nodes, std::vector<int64_t>(nnodes, -1), std::vector<int64_t>(nnodes, -1), 0});
  }
  void release(std::unique_ptr<RcbCtx> c)
  {
    std::lock_guard<std::mutex> g(mtx);
    free_list.push_back(std::move(c));
  }
};

// The subtrees are independent of one another (a context only remembers the nodes of ITS current cut), so
// the upper levels hand their halves to other host threads: same tiles as the serial recursion.
void rcb_split(TileItem* a, int64_t n, int64_t ntile, int tc, RcbPool& pool, RcbCtx* c, int depth)
{
  if (ntile <= 1 || n <= tc)
    return;
  float lo[2] = {3e38f, 3e38f}, hi[2] = {-3e38f, -3e38f};
  int64_t nstretched = 0;
  const bool last_levels = ntile <= 64;
  for (int64_t i = 0; i < n; ++i)
  {
    lo[0] = std::min(lo[0], a[i].x);
    hi[0] = std::max(hi[0], a[i].x);
    lo[1] = std::min(lo[1], a[i].y);
    hi[1] = std::max(hi[1], a[i].y);
    if (last_levels && pool.stretched)
      nstretched += pool.stretched[a[i].cell];
  }
  const int64_t tl = ntile / 2;
  const int64_t nl = std::min<int64_t>(n, tl * tc);
  int axis = (hi[0] - lo[0] >= hi[1] - lo[1]) ? 0 : 1;
  // the last levels decide the shape of the tiles: there the cut is chosen by what it costs - the
  // nodes it separates - not by the extent of the bounding box (which misleads on stretched cells:
  // boundary layers, polar meshes)
  // (where the cells of the segment are not stretched - fewer than 2 % with an aspect ratio above ~3 - the longer
  //  side of the bounding box IS the cheaper cut, and the two trial partitions with their node counts, which
  //  dominated the set-up time of isotropic meshes, are skipped)
  std::unique_ptr<RcbCtx> own;
  if (last_levels && nstretched * 50 > n)
  {
    if (!c)
    {
      own = pool.acquire();
      c = own.get();
    }
    rcb_partition(a, n, nl, axis);
    const int64_t c0 = rcb_cut_nodes(a, n, nl, *c);
    std::vector<TileItem> first(a, a + n); // the partition along `axis`, in case it wins
    rcb_partition(a, n, nl, 1 - axis);
    const int64_t c1 = rcb_cut_nodes(a, n, nl, *c);
    if (c1 < c0)
      axis = 1 - axis; // already partitioned along it
    else
      std::copy(first.begin(), first.end(), a);
  }
  else if (n >= (1 << 18) && depth <= 2) // the top of the tree: few segments, many idle cores
  {
    std::vector<TileItem> tmp;
    rcb_partition_parallel(a, n, nl, axis, lo[axis], hi[axis], tmp);
  }
  else
    rcb_partition(a, n, nl, axis);
  constexpr int PAR_DEPTH = 5; // up to 32 concurrent subtrees
  if (depth < PAR_DEPTH && n > 16 * (int64_t)tc)
  {
    // (a context taken above stays with this thread's half)
    auto left = std::async(std::launch::async, [&]() { rcb_split(a, nl, tl, tc, pool, nullptr, depth + 1); });
    rcb_split(a + nl, n - nl, ntile - tl, tc, pool, c, depth + 1);
    left.get();
  }
  else
  {
    rcb_split(a, nl, tl, tc, pool, c, depth + 1);
    rcb_split(a + nl, n - nl, ntile - tl, tc, pool, c, depth + 1);
  }
  if (own)
    pool.release(std::move(own));
}

// f(i) for i in [0, n) on the host threads (contiguous chunks)
template <typename F>
void parallel_for(int64_t n, int64_t min_chunk, F f)
{
  const int64_t nt = host_workers(n / std::max<int64_t>(min_chunk, 1));
  if (nt <= 1)
  {
    for (int64_t i = 0; i < n; ++i)
      f(i);
    return;
  }
  Workers w;
  for (int64_t t = 0; t < nt; ++t)
    w.spawn([=]() {
      for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i)
        f(i);
    });
  w.join();
}

// Tiled SoA of the plain flux equilibration (EQLB_SCATTER_TILED): cells bisected recursively by
// their centroids into tiles of TC cells; a tile lists every (masked-in) node of its cells.
int build_tiles(eqlb_se* h, const std::vector<int8_t>& node_bin_all, eqlb::BuildArgs a, int tc_fixed = 0, int max_bin = eqlb::MAX_BINS,
                bool full_only = false)
{
  // nodes of bins >= max_bin are left out (like masked-out nodes): another path equilibrates them.
  // full_only (fused stress launch on the crossed benchmark meshes): so are all patches that are not FULL (interior,
  // as many cells as lanes); the lists of a tile are padded to whole wave-blocks with copies of a full patch that
  // own no cell
  const eqlb::DeviceMesh& m = h->mesh->m;
  const int32_t nc = m.ncells;
  std::vector<int8_t> node_bin(node_bin_all);
  std::vector<uint8_t> is_rest(max_bin < eqlb::MAX_BINS ? m.nnodes : 0, 0);
  h->t_rest = 0;
  for (int32_t i = 0; i < m.nnodes; ++i)
  {
    int8_t& b = node_bin[i];
    if (b < 0)
      continue;
    const bool full = m.h_node_ncells[i] == m.h_node_nfcts[i] && m.h_node_ncells[i] == eqlb::BIN_P[b];
    if (b >= max_bin || (full_only && !full))
    {
      b = -1;
      ++h->t_rest;
      if (!is_rest.empty())
        is_rest[i] = 1;
    }
  }
  dfree(h->rest_cells);
  h->nrest_cells = 0;
  if (!is_rest.empty() && h->t_rest > 0)
  {
    // cells with a vertex whose patch the generic kernels take: the compact reduction of their slot rows
    std::vector<int32_t> rc;
    for (int32_t c = 0; c < nc; ++c)
      for (int j = 0; j < 3; ++j)
      {
        const int32_t nd = m.h_cell_nodes[3 * (size_t)c + j];
        if (is_rest[nd])
        {
          rc.push_back(c);
          break;
        }
      }
    h->nrest_cells = (int64_t)rc.size();
    if (upload(&h->rest_cells, rc.data(), std::max<size_t>(rc.size(), 1)))
      return EQLB_ERR_DEVICE;
  }
  // Tile size: the default, or - on meshes that fill the chip several times over - the size that
  // makes the tiles fill whole rounds of the 512 workgroup slots (2 per CU): 1M triangles in 2 045
  // tiles of 489 cells run in 4 rounds, 2 084 tiles of 480 cells leave 36 tiles for a fifth
  int TC = tc_fixed > 0 ? tc_fixed : eqlb::tile_cells_of(h->k);
  if (tc_fixed > 0)
  {
    // fused stress launch (tc_fixed = the largest tile its LDS holds): ONE workgroup per CU, so a partial last
    // round of the 256 slots costs a full round - fit the tile size to whole rounds as below
    const int64_t slots = 256, tcmax = tc_fixed;
    TC = (int)std::min<int64_t>(tcmax, 448);
    if ((int64_t)nc >= slots * 256)
    {
      const int64_t rounds = ((int64_t)nc + slots * tcmax - 1) / (slots * tcmax);
      TC = (int)(((int64_t)nc + rounds * slots - 1) / (rounds * slots));
    }
    if (h->tile_cells_user > 0)
      TC = (int)std::min<int64_t>(h->tile_cells_user, tcmax);
  }
  if (tc_fixed <= 0)
  {
    // resident workgroup slots of the chip: two per CU for k <= 2, one for k = 3
    const bool ev3 = h->mode == 1 && h->k >= 3; // EV mode of RT_3 stages 7 KB more tensors: smaller tiles
    const int64_t slots = (h->k <= 2) ? 512 : 256, tcmax = ev3 ? eqlb::tile_cells_ev_of(h->k) : eqlb::tile_cells_max_of(h->k);
    if (ev3)
      TC = eqlb::tile_cells_ev_of(h->k);
    if ((int64_t)nc >= slots * 256)
    {
      const int64_t rounds = ((int64_t)nc + slots * tcmax - 1) / (slots * tcmax);
      TC = (int)(((int64_t)nc + rounds * slots - 1) / (rounds * slots));
    }
    if (h->tile_cells_user > 0) // tuning knob (option "tile_cells"), capped by what the LDS of a workgroup holds
      TC = (int)std::min<int64_t>(h->tile_cells_user, tcmax);
  }
  SetupTimer tm;
  uvec<TileItem> items(nc);
  const int32_t ntiles = (nc + TC - 1) / TC;
  bool cached = false;
  {
    std::lock_guard<std::mutex> g(h->mesh->tiling_mutex);
    auto it = h->mesh->tiling_order.find(TC);
    if (it != h->mesh->tiling_order.end() && (int32_t)it->second.size() == nc)
    {
      for (int32_t p = 0; p < nc; ++p)
        items[p] = {0.0f, 0.0f, it->second[p]};
      cached = true;
    }
  }
  if (!cached)
  {
  std::vector<uint8_t> stretched(nc);
  // bounding box of the nodes: the centroids are stored relative to it (one scale for both directions)
  double blo[2] = {1e300, 1e300}, bhi[2] = {-1e300, -1e300};
  for (int32_t i = 0; i < m.nnodes; ++i)
    for (int d = 0; d < 2; ++d)
    {
      blo[d] = std::min(blo[d], m.h_x[3 * (size_t)i + d]);
      bhi[d] = std::max(bhi[d], m.h_x[3 * (size_t)i + d]);
    }
  const double ext = std::max(bhi[0] - blo[0], bhi[1] - blo[1]);
  const double inv = (ext > 0.0) ? 1.0 / (3.0 * ext) : 0.0;
  // the bisection on the device (one radix sort per level of the tree; eqlb_tiling_device.hip) unless the mesh
  // has stretched cells, where the host bisection below picks the cuts of the last levels by their cost
  bool on_device = false;
  {
    const char* env = getenv("EQLB_TILING");
    if (!(env && !strcmp(env, "host")) && nc >= 4096)
    {
      std::vector<int32_t> dord;
      const int r = eqlb::device_tile_order(m, TC, ntiles, blo, bhi, inv, dord);
      if (r < 0)
        return fail(EQLB_ERR_DEVICE, "tiling on the device failed");
      if (r == 0)
      {
        for (int32_t p = 0; p < nc; ++p)
          items[p] = {0.0f, 0.0f, dord[p]};
        on_device = true;
        tm.lap("tiles: bisection (device)");
      }
    }
  }
  if (!on_device)
  {
  parallel_for(nc, 1 << 16, [&](int64_t c) {
    const int32_t* cn = &m.h_cell_nodes[3 * (size_t)c];
    double cx = 0.0, cy = 0.0;
    for (int j = 0; j < 3; ++j)
    {
      cx += m.h_x[3 * (size_t)cn[j]] - blo[0];
      cy += m.h_x[3 * (size_t)cn[j] + 1] - blo[1];
    }
    items[c] = {(float)(cx * inv), (float)(cy * inv), (int32_t)c};
    const double* p0 = &m.h_x[3 * (size_t)cn[0]];
    const double* p1 = &m.h_x[3 * (size_t)cn[1]];
    const double* p2 = &m.h_x[3 * (size_t)cn[2]];
    const double e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e2x = p2[0] - p0[0], e2y = p2[1] - p0[1];
    const double l2 = std::max(std::max(e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y),
                               (e2x - e1x) * (e2x - e1x) + (e2y - e1y) * (e2y - e1y));
    stretched[c] = l2 > 6.0 * std::fabs(e1x * e2y - e1y * e2x) ? 1 : 0;
  });
  RcbPool pool{m.h_cell_nodes.data(), m.nnodes, stretched.data(), {}, {}};
  tm.lap("tiles: centroids");
  rcb_split(items.data(), nc, ntiles, TC, pool, nullptr, 0);
  tm.lap("tiles: bisection");
  // ascending cell ids inside a tile: the flush of a tile then touches flux_hdiv in long runs
  parallel_for(ntiles, 16, [&](int64_t t) {
    std::sort(items.begin() + (size_t)t * TC, items.begin() + std::min<size_t>((size_t)(t + 1) * TC, nc),
              [](const TileItem& p, const TileItem& q) { return p.cell < q.cell; });
  });
  }
  std::vector<int32_t> ord(nc);
  for (int32_t p = 0; p < nc; ++p)
    ord[p] = items[p].cell;
  std::lock_guard<std::mutex> g(h->mesh->tiling_mutex);
  h->mesh->tiling_order[TC] = std::move(ord);
  }
  // tiles that own a priority cell (ghost rows a neighbour rank waits for) are numbered first: a
  // first launch over them, the halo exchange, and the launch over the rest then overlap
  std::vector<int32_t> order(ntiles);
  {
    std::vector<uint8_t> tile_prio(ntiles, 0);
    if (!h->prio_cells.empty())
    {
      std::vector<uint8_t> is_prio(nc, 0);
      for (int32_t c : h->prio_cells)
        if (c >= 0 && c < nc)
          is_prio[c] = 1;
      for (int32_t p = 0; p < nc; ++p)
        if (is_prio[items[p].cell])
          tile_prio[p / TC] = 1;
    }
    int32_t np = 0;
    for (int32_t t = 0; t < ntiles; ++t)
      if (tile_prio[t])
        order[np++] = t;
    h->t_nprio = np;
    for (int32_t t = 0; t < ntiles; ++t)
      if (!tile_prio[t])
        order[np++] = t;
  }
  tm.lap("tiles: sort + priority");
  uvec<int32_t> tile_cells((size_t)ntiles * TC), cell_tile(nc), cell_pos(nc);
  parallel_for(ntiles, 16, [&](int64_t t) {
    const int64_t src = (int64_t)order[t] * TC, len = std::min<int64_t>(TC, nc - src);
    for (int64_t q = 0; q < len; ++q)
    {
      const int32_t c = items[src + q].cell;
      tile_cells[(size_t)t * TC + q] = c;
      cell_tile[c] = (int32_t)t;
      cell_pos[c] = (int32_t)((int64_t)t * TC + q);
    }
    for (int64_t q = len; q < TC; ++q)
      tile_cells[(size_t)t * TC + q] = -1;
  });
  std::vector<eqlb::TileDesc> tiles(ntiles);
  // pass 1 (host threads, a chunk of tiles each): the nodes of every tile by bin - full interior patches
  // (as many cells as lanes, no boundary facet: their wave-blocks run the specialised body of the kernel)
  // first -, in order of first appearance; flat storage, 3 TC entries per tile
  constexpr int NB = eqlb::MAX_BINS;
  uvec<int32_t> tnodes((size_t)ntiles * 3 * TC);
  constexpr int NCL = 6; // classes of a bin: full | interior with P - 1, P - 2, P - 3 cells | other interior | boundary
  std::vector<int32_t> tcount((size_t)ntiles * NCL * NB, 0); // [tile][bin][class]
  auto tile_chunks = [&](auto work) {
    const int64_t nt = host_workers(ntiles / 32);
    if (nt <= 1)
    {
      work(0, ntiles);
      return;
    }
    Workers wk;
    for (int64_t w = 0; w < nt; ++w)
      wk.spawn([&work, ntiles, w, nt]() { work((int64_t)ntiles * w / nt, (int64_t)ntiles * (w + 1) / nt); });
    wk.join();
  };
  // sort key of a node: NCL * bin + class (full interior patch 0 | interior patch with P - 1, P - 2, P - 3 cells 1, 2, 3 |
  // other interior patch 4 | boundary patch 5); -1: not listed
  // (one byte per node, cache resident, instead of three scattered reads per visit of a node)
  std::vector<int8_t> nkey(m.nnodes);
  parallel_for(m.nnodes, 1 << 16, [&](int64_t nd) {
    const int b_ = node_bin[nd];
    if (b_ < 0)
    {
      nkey[nd] = -1;
      return;
    }
    const bool interior = m.h_node_ncells[nd] == m.h_node_nfcts[nd]; // no boundary facet at the node
    const int missing = eqlb::BIN_P[b_] - m.h_node_ncells[nd];         // idle lanes of the patch group
    nkey[nd] = (int8_t)(NCL * b_ + (interior ? ((missing >= 0 && missing <= 3) ? missing : 4) : 5));
  });
  tile_chunks([&](int64_t t0, int64_t t1) {
    std::vector<int32_t> stamp(m.nnodes, -1), seen(3 * (size_t)TC);
    for (int64_t t = t0; t < t1; ++t)
    {
      int nseen = 0;
      int32_t* cnt = &tcount[(size_t)t * NCL * NB];
      auto key = [&](int32_t nd) { return (int)nkey[nd]; };
      for (int q = 0; q < TC; ++q)
      {
        const int32_t c = tile_cells[(size_t)t * TC + q];
        if (c < 0)
          continue;
        for (int j = 0; j < 3; ++j)
        {
          const int32_t nd = m.h_cell_nodes[3 * (size_t)c + j];
          if (nkey[nd] < 0)
            tiles[t].zero = 1; // masked-out vertex: the (cell, vertex) row of this tile stays unwritten
          if (nkey[nd] < 0 || stamp[nd] == (int32_t)t)
            continue;
          stamp[nd] = (int32_t)t;
          seen[nseen++] = nd;
          ++cnt[key(nd)];
        }
      }
      int32_t pos[NCL * NB], acc = 0; // stable counting sort by (bin, class)
      for (int q = 0; q < NCL * NB; ++q)
      {
        pos[q] = acc;
        acc += cnt[q];
      }
      int32_t* out = &tnodes[(size_t)t * 3 * TC];
      for (int i = 0; i < nseen; ++i)
        out[pos[key(seen[i])]++] = seen[i];
      for (int b_ = 0; b_ < NB; ++b_)
      {
        const int32_t* cb = cnt + NCL * b_;
        tiles[t].nfull[b_] = cb[0];
        tiles[t].nint[b_] = cb[0] + cb[1] + cb[2] + cb[3] + cb[4];
        tiles[t].npatch[b_] = tiles[t].nint[b_] + cb[5];
        if (b_ < 2)
        {
          tiles[t].nval[b_][0] = cb[0] + cb[1];
          tiles[t].nval[b_][1] = cb[0] + cb[1] + cb[2];
          tiles[t].nval[b_][2] = cb[0] + cb[1] + cb[2] + cb[3];
        }
        if (full_only)
        {
          // whole wave-blocks: nint keeps the number of real patches, the others are copies (pass 2)
          const int per = 64 / eqlb::BIN_P[b_];
          const int padded = (per > 0) ? (cb[0] + per - 1) / per * per : cb[0];
          tiles[t].nfull[b_] = padded;
          tiles[t].npatch[b_] = padded;
          if (b_ < 2)
            tiles[t].nval[b_][0] = tiles[t].nval[b_][1] = tiles[t].nval[b_][2] = padded;
        }
      }
    }
  });
  // lane slots and patch instances in tile order (serial prefix), then filled by the host threads
  int64_t slotctr = 0, ninst = 0;
  for (int32_t t = 0; t < ntiles; ++t)
    for (int b_ = 0; b_ < NB; ++b_)
    {
      tiles[t].slot_start[b_] = (int32_t)slotctr;
      tiles[t].patch_start[b_] = (int32_t)ninst;
      slotctr += (int64_t)tiles[t].npatch[b_] * eqlb::BIN_P[b_];
      ninst += tiles[t].npatch[b_];
      slotctr = (slotctr + 63) & ~(int64_t)63;
      if (slotctr > 0x7fffff00)
        return fail(EQLB_ERR_UNSUPPORTED, "tiled patch SoA exceeds 2^31 lane slots");
    }
  uvec<int32_t> inst_node((size_t)ninst), inst_slot((size_t)ninst), inst_tile((size_t)ninst);
  tile_chunks([&](int64_t t0, int64_t t1) {
    for (int64_t t = t0; t < t1; ++t)
    {
      const int32_t* src = &tnodes[(size_t)t * 3 * TC];
      for (int b_ = 0; b_ < NB; ++b_)
      {
        int32_t slot = tiles[t].slot_start[b_];
        const int32_t nreal = full_only ? tiles[t].nint[b_] : tiles[t].npatch[b_];
        for (int32_t i = 0, p_ = tiles[t].patch_start[b_]; i < tiles[t].npatch[b_]; ++i, ++p_, slot += eqlb::BIN_P[b_])
        {
          // (padding copy: the last real patch once more, tile -1 = it owns no cell and stores nothing)
          inst_node[p_] = (i < nreal) ? *src++ : src[-1];
          inst_slot[p_] = slot;
          inst_tile[p_] = (i < nreal) ? (int32_t)t : -1;
        }
      }
    }
  });
  tm.lap("tiles: patch lists");
  h->ntiles = ntiles;
  h->tile_tc = TC;
  h->t_nslots = slotctr;
  h->t_npatch = (int64_t)inst_node.size();
  int32_t *d_inode = nullptr, *d_islot = nullptr, *d_itile = nullptr, *d_ctile = nullptr, *d_cpos = nullptr;
  int st = 0;
  st |= upload(&h->t_tiles, tiles.data(), tiles.size());
  st |= upload(&h->t_tile_cells, tile_cells.data(), tile_cells.size());
  st |= upload<int32_t>(&h->t_slot_cell, nullptr, (size_t)std::max<int64_t>(slotctr, 1));
  st |= upload<uint32_t>(&h->t_slot_info, nullptr, (size_t)std::max<int64_t>(slotctr, 1));
  st |= upload<uint8_t>(&h->t_pn, nullptr, (size_t)std::max<int64_t>(h->t_npatch, 1));
  st |= upload<uint8_t>(&h->t_pflag, nullptr, (size_t)std::max<int64_t>(h->t_npatch, 1) * h->nrhs);
  st |= upload(&d_inode, inst_node.data(), std::max<size_t>(inst_node.size(), 1));
  st |= upload(&d_islot, inst_slot.data(), std::max<size_t>(inst_slot.size(), 1));
  st |= upload(&d_itile, inst_tile.data(), std::max<size_t>(inst_tile.size(), 1));
  st |= upload(&d_ctile, cell_tile.data(), cell_tile.size());
  st |= upload(&d_cpos, cell_pos.data(), cell_pos.size());
  hipError_t e = hipSuccess;
  if (!st)
  {
    e = hipMemset(h->t_slot_cell, 0xff, sizeof(int32_t) * std::max<int64_t>(slotctr, 1));
    if (e == hipSuccess)
      e = hipMemset(h->t_slot_info, 0, sizeof(uint32_t) * std::max<int64_t>(slotctr, 1));
    a.ninst = h->t_npatch;
    a.inst_node = d_inode;
    a.inst_slot = d_islot;
    a.inst_tile = d_itile;
    a.cell_tile = d_ctile;
    a.cell_pos = d_cpos;
    a.tile_cells = TC;
    a.npatch_total = h->t_npatch;
    a.slot_cell = h->t_slot_cell;
    a.slot_info = h->t_slot_info;
    a.pn = h->t_pn;
    a.pflag = h->t_pflag;
    a.stride = 0;
    a.ex_ncells = nullptr;
    if (e == hipSuccess && a.ninst > 0)
    {
      eqlb::launch_build_patches(a, nullptr);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
  }
  tm.lap("tiles: upload + builder kernel");
  dfree(d_inode);
  dfree(d_islot);
  dfree(d_itile);
  dfree(d_ctile);
  dfree(d_cpos);
  if (st)
    return EQLB_ERR_DEVICE;
  if (e != hipSuccess)
    return fail(EQLB_ERR_DEVICE, "tiled patch builder: %s", hipGetErrorString(e));
  return EQLB_OK;
}
} // namespace

extern "C" {

// Nothing may leave an extern "C" entry point as an exception (a ctypes / cgo / JNI caller would be terminated):
// function-try-blocks around the entries that allocate on the host or start worker threads.
#define EQLB_CATCH_ALL                                                                                       \
  catch (const std::bad_alloc&) { return fail(EQLB_ERR_NO_MEMORY, "host memory exhausted"); }                \
  catch (const std::exception& e) { return fail(EQLB_ERR_DEVICE, "internal error: %s", e.what()); }         \
  catch (...) { return fail(EQLB_ERR_DEVICE, "internal error"); }

const char* eqlb_last_error(void) { return g_error.c_str(); }

int eqlb_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int eqlb_mesh_create(int32_t nnodes, int32_t ncells, int32_t nfacets, const double* x,
                     const int32_t* cell_nodes, const int32_t* cell_facets,
                     const int32_t* facet_nodes, const int32_t* facet_cells_offsets,
                     const int32_t* facet_cells, const int32_t* node_cells_offsets,
                     const int32_t* node_cells, const int32_t* node_facets_offsets,
                     const int32_t* node_facets, const uint8_t* facet_perm, eqlb_mesh_t** mesh)
try
{
  if (!mesh || nnodes <= 0 || ncells <= 0 || nfacets <= 0 || !x || !cell_nodes || !cell_facets
      || !facet_nodes || !facet_cells_offsets || !facet_cells || !node_cells_offsets || !node_cells
      || !node_facets_offsets || !node_facets || !facet_perm)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_mesh_create: null or empty input");
  if (eqlb_device_count() < 1)
    return fail(EQLB_ERR_DEVICE, "eqlb_mesh_create: no HIP device available");
  // The kernels index with these tables unchecked: a bad entry would fault on the device, so the
  // connectivities are validated here (O(size) on the host, once per mesh).
  {
    auto csr_ok = [](const int32_t* off, int32_t n, const int32_t* val, int32_t bound) {
      if (off[0] != 0)
        return false;
      for (int32_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i])
          return false;
      for (int32_t q = 0; q < off[n]; ++q)
        if (val[q] < 0 || val[q] >= bound)
          return false;
      return true;
    };
    bool ok = true;
    for (size_t i = 0; i < (size_t)ncells * 3 && ok; ++i)
      ok = cell_nodes[i] >= 0 && cell_nodes[i] < nnodes && cell_facets[i] >= 0 && cell_facets[i] < nfacets
           && facet_perm[i] <= 1;
    for (size_t i = 0; i < (size_t)nfacets * 2 && ok; ++i)
      ok = facet_nodes[i] >= 0 && facet_nodes[i] < nnodes;
    ok = ok && csr_ok(facet_cells_offsets, nfacets, facet_cells, ncells)
         && csr_ok(node_cells_offsets, nnodes, node_cells, ncells)
         && csr_ok(node_facets_offsets, nnodes, node_facets, nfacets);
    for (int32_t f = 0; f < nfacets && ok; ++f)
    {
      const int32_t nc = facet_cells_offsets[f + 1] - facet_cells_offsets[f];
      ok = (nc == 1 || nc == 2);
    }
    if (!ok)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_mesh_create: inconsistent connectivity tables");
  }
  eqlb_mesh* m = new eqlb_mesh();
  eqlb::DeviceMesh& d = m->m;
  d.nnodes = nnodes;
  d.ncells = ncells;
  d.nfacets = nfacets;
  d.h_node_ncells.resize(nnodes);
  d.h_node_nfcts.resize(nnodes);
  for (int32_t i = 0; i < nnodes; ++i)
  {
    d.h_node_ncells[i] = node_cells_offsets[i + 1] - node_cells_offsets[i];
    d.h_node_nfcts[i] = node_facets_offsets[i + 1] - node_facets_offsets[i];
    d.ncells_max = std::max(d.ncells_max, d.h_node_ncells[i]);
  }
  d.h_x.assign(x, x + (size_t)nnodes * 3);
  d.h_cell_nodes.assign(cell_nodes, cell_nodes + (size_t)ncells * 3);
  d.h_facet_nodes.assign(facet_nodes, facet_nodes + (size_t)nfacets * 2);
  d.h_node_facets_off.assign(node_facets_offsets, node_facets_offsets + (size_t)nnodes + 1);
  d.h_node_facets.assign(node_facets, node_facets + (size_t)node_facets_offsets[nnodes]);
  d.h_node_cells_off.assign(node_cells_offsets, node_cells_offsets + (size_t)nnodes + 1);
  d.h_node_cells.assign(node_cells, node_cells + (size_t)node_cells_offsets[nnodes]);
  int st = 0;
  st |= upload(&d.x, x, (size_t)nnodes * 3);
  st |= upload(&d.cell_nodes, cell_nodes, (size_t)ncells * 3);
  st |= upload(&d.cell_facets, cell_facets, (size_t)ncells * 3);
  st |= upload(&d.facet_nodes, facet_nodes, (size_t)nfacets * 2);
  st |= upload(&d.facet_cells_off, facet_cells_offsets, (size_t)nfacets + 1);
  st |= upload(&d.facet_cells, facet_cells, (size_t)facet_cells_offsets[nfacets]);
  st |= upload(&d.node_cells_off, node_cells_offsets, (size_t)nnodes + 1);
  st |= upload(&d.node_facets_off, node_facets_offsets, (size_t)nnodes + 1);
  st |= upload(&d.node_facets, node_facets, (size_t)node_facets_offsets[nnodes]);
  st |= upload(&d.facet_perm, facet_perm, (size_t)ncells * 3);
  st |= upload<double>(&d.cellJ, nullptr, (size_t)ncells * 4);
  if (st)
  {
    eqlb_mesh_destroy(m);
    return EQLB_ERR_DEVICE;
  }
  eqlb::launch_cell_geometry(ncells, d.x, d.cell_nodes, d.cellJ, nullptr);
  eqlb::device_tiling_prepare(); // code object of the tile builder loaded here, not inside the first set_boundary
  if (hipDeviceSynchronize() != hipSuccess)
  {
    eqlb_mesh_destroy(m);
    return fail(EQLB_ERR_DEVICE, "eqlb_mesh_create: geometry kernel failed");
  }
  *mesh = m;
  return EQLB_OK;
}
EQLB_CATCH_ALL

void eqlb_mesh_destroy(eqlb_mesh_t* m)
{
  if (!m)
    return;
  eqlb::DeviceMesh& d = m->m;
  dfree(d.x);
  dfree(d.cellJ);
  dfree(d.cell_nodes);
  dfree(d.cell_facets);
  dfree(d.facet_nodes);
  dfree(d.facet_cells_off);
  dfree(d.facet_cells);
  dfree(d.node_cells_off);
  dfree(d.node_facets_off);
  dfree(d.node_facets);
  dfree(d.node_cells);
  dfree(d.facet_perm);
  delete m;
}

int32_t eqlb_mesh_max_patch_cells(const eqlb_mesh_t* mesh) { return mesh ? mesh->m.ncells_max : 0; }

int eqlb_se_create(eqlb_mesh_t* mesh, int32_t k, int32_t degree_dg, int32_t nrhs,
                   int32_t reconstruct_stress, int32_t estimate_korn, eqlb_se_t** handle)
try
{
  if (!mesh || !handle || nrhs < 1)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  if (k < 1)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Degree must be at least 1");
  // se/reconstruction.hpp:363-373
  if (degree_dg > k - 1 || degree_dg < 0)
    return fail(EQLB_ERR_INVALID_ARGUMENT,
                "Equilibration: Wrong polynomial degree of the projected RHS");
  if (reconstruct_stress)
  {
    // se/reconstruction.hpp:376-388
    if (nrhs < 2)
      return fail(EQLB_ERR_INVALID_ARGUMENT,
                  "Stress equilibration: Specify all rows of stress tensor");
    if (k < 2)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "Stress equilibration: RT_k with k>1 required!");
  }
  (void)estimate_korn;
  std::vector<double> tab;
  if (eqlb::fill_tables_host(k, degree_dg, tab) != 0 || degree_dg != k - 1 || k > 4)
    return fail(EQLB_ERR_UNSUPPORTED, "RT_%d with DG_%d data%s is not in this build", k, degree_dg,
                reconstruct_stress ? " (stress)" : "");
  eqlb_se* h = new eqlb_se();
  h->mesh = mesh;
  h->k = k;
  h->deg = degree_dg;
  h->nrhs = nrhs;
  h->stress = reconstruct_stress ? 1 : 0;
  // default result path: tiled launch where it is the fastest (measured)
  h->scatter = EQLB_SCATTER_AUTO;
  // (k = 4, three interior unknowns per cell: register solver as well since round 3 - 0.38 ms against 14.9 ms of
  // the dense LDS Cholesky at 250 000 triangles; the EV patch problems at k = 4 stay on the dense solver)
  h->nrt = k * (k + 2);
  h->nd = (degree_dg + 1) * (degree_dg + 2) / 2;
  int st = upload(&h->tables, tab.data(), tab.size());
  st |= upload<int32_t>(&h->status, nullptr, 1);
  if (st)
  {
    eqlb_se_destroy(h);
    return EQLB_ERR_DEVICE;
  }
  (void)hipMemset(h->status, 0, sizeof(int32_t));
  *handle = h;
  return EQLB_OK;
}
EQLB_CATCH_ALL

void eqlb_se_destroy(eqlb_se_t* h)
{
  if (!h)
    return;
  free_boundary(h);
  dfree(h->tables);
  dfree(h->slots);
  dfree(h->status);
  dfree(h->d_flux_dg);
  dfree(h->d_rhs_dg);
  dfree(h->d_flux_hdiv);
  dfree(h->d_cks);
  dfree(h->d_korn);
  if (h->ev)
  {
    for (int i = 0; i < eqlb_se::EV_RING * eqlb_se::EV_PER_SET; ++i)
      if (h->ev[i])
        (void)hipEventDestroy(h->ev[i]);
    delete[] h->ev;
  }
  if (h->ev_fork)
    (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join)
    (void)hipEventDestroy(h->ev_join);
  if (h->side_stream)
    (void)hipStreamDestroy(h->side_stream);
  delete h;
}

int eqlb_se_set_option(eqlb_se_t* h, const char* key, int32_t value)
{
  if (!h || !key)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_set_option: null argument");
  if (!strcmp(key, "solver"))
  {
#ifdef EQLB_EXP_SOLVER9 // timing-only variant without the solve (wrong results): experiment builds only
    const bool exp9 = value == 9;
#else
    const bool exp9 = false;
#endif
    if (value != EQLB_SOLVER_LDS_CHOLESKY && value != EQLB_SOLVER_SHUFFLE && !exp9)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "unknown solver %d", value);
    h->solver = value;
  }
  else if (!strcmp(key, "scatter"))
  {
    if (value != EQLB_SCATTER_SLOTS && value != EQLB_SCATTER_ATOMIC && value != EQLB_SCATTER_TILED
        && value != EQLB_SCATTER_AUTO)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "unknown scatter mode %d", value);
    h->scatter = value;
  }
  else if (!strcmp(key, "fused"))
    h->fused = value;
  else if (!strcmp(key, "timing"))
  {
    h->timing = value;
    h->ev_calls = 0;
  }
  else if (!strcmp(key, "accumulate"))
  {
    if (value != 0 && value != 1)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "accumulate must be 0 or 1");
    h->accumulate = value;
  }
  else if (!strcmp(key, "multi_rhs"))
  {
    if (value != 0 && value != 1)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "multi_rhs must be 0 or 1");
    h->multi_rhs = value;
  }
  else if (!strcmp(key, "tile_first"))
  {
    if (value < 0)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "tile_first must not be negative");
    h->tile_first = value;
  }
  else if (!strcmp(key, "tile_count"))
    h->tile_count = value;
  else if (!strcmp(key, "tile_cells"))
  {
    if (value < 0)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "tile_cells must not be negative");
    h->tile_cells_user = value;
  }
  else
    return fail(EQLB_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
  return EQLB_OK;
}

int eqlb_se_set_boundary(eqlb_se_t* h, const int8_t* facet_type, const double* boundary_values,
                         const uint8_t* node_mask)
try
{
  if (!h || !facet_type)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_set_boundary: null argument");
  SetupTimer tm;
  const eqlb::DeviceMesh& m = h->mesh->m;
  for (size_t i = 0; i < (size_t)h->nrhs * m.nfacets; ++i)
    if (facet_type[i] < EQLB_FACET_INTERNAL || facet_type[i] > EQLB_FACET_ESSNT_DUAL)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_set_boundary: facet type %d out of range", (int)facet_type[i]);
  bool inhomogeneous = false;
  if (boundary_values)
  {
    const size_t nb = (size_t)h->nrhs * m.ncells * h->nrt;
    for (size_t i = 0; i < nb && !inhomogeneous; ++i)
      inhomogeneous = (boundary_values[i] != 0.0);
  }
  h->stress_flux_bcs = false;
  if (h->stress)
    for (size_t i = 0; i < (size_t)2 * m.nfacets && !h->stress_flux_bcs; ++i)
      h->stress_flux_bcs = (facet_type[i] == EQLB_FACET_ESSNT_DUAL);
  // OrientedPatch::set_max_patch_size (se/Patch.cpp:337-404): every local node is checked
  for (int32_t i = 0; i < m.nnodes; ++i)
  {
    if (node_mask && !node_mask[i])
      continue; // the reference loops the owned nodes only (size_local)
    if (m.h_node_ncells[i] == 1)
      return fail(EQLB_ERR_PATCH_TOO_SMALL, "Patch around node %d has only 1 cells.", i);
    if (m.h_node_ncells[i] < 1)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "node %d belongs to no cell", i);
  }
  tm.lap("checks");
  free_boundary(h);
  tm.lap("free old tables");

  // bins by lanes per patch: P = smallest of {4,8,16,32,64} >= number of patch facets
  std::vector<int64_t> node_slot(m.nnodes, -1), node_patch(m.nnodes, -1);
  int64_t count[eqlb::MAX_BINS] = {0, 0, 0, 0, 0};
  std::vector<int8_t> node_bin(m.nnodes, -1);
  for (int32_t i = 0; i < m.nnodes; ++i)
  {
    if (node_mask && !node_mask[i])
      continue;
    const int nf = m.h_node_nfcts[i];
    int b = 0;
    while (b < eqlb::MAX_BINS && eqlb::BIN_P[b] < nf)
      ++b;
    if (b == eqlb::MAX_BINS || m.h_node_ncells[i] > 63)
      return fail(EQLB_ERR_PATCH_TOO_LARGE, "Patch around node %d has %d cells (limit 63)", i,
                  m.h_node_ncells[i]);
    node_bin[i] = (int8_t)b;
    ++count[b];
  }
  int64_t slot_off = 0, patch_off = 0;
  for (int b = 0; b < eqlb::MAX_BINS; ++b)
  {
    h->bins[b].P = eqlb::BIN_P[b];
    h->bins[b].npatch = count[b];
    h->bins[b].slot_offset = slot_off;
    h->bins[b].patch_offset = patch_off;
    slot_off += count[b] * eqlb::BIN_P[b];
    patch_off += count[b];
    count[b] = 0;
  }
  h->nslots = slot_off;
  h->npatch_total = patch_off;
  // fused stress launch (RT_2, no flux BCs on the stress rows): it takes the FULL patches of the bins 0, 1 -
  // interior, as many cells as lanes -, listed first in their bin; the generic kernels take the patches behind them
  const bool full_first = h->stress && h->k == 2 && !h->stress_flux_bcs && h->mode == 0;
  auto is_full = [&](int32_t i) {
    const int b = node_bin[i];
    return full_first && b >= 0 && b < 2 && m.h_node_ncells[i] == m.h_node_nfcts[i]
           && m.h_node_ncells[i] == eqlb::BIN_P[b];
  };
  for (int pass = 0; pass < 2; ++pass)
  {
    for (int32_t i = 0; i < m.nnodes; ++i)
    {
      const int b = node_bin[i];
      if (b < 0 || is_full(i) != (pass == 0))
        continue;
      node_patch[i] = h->bins[b].patch_offset + count[b];
      node_slot[i] = h->bins[b].slot_offset + count[b] * eqlb::BIN_P[b];
      ++count[b];
    }
    if (pass == 0)
      for (int b = 0; b < eqlb::MAX_BINS; ++b)
        h->bins[b].nfull = count[b];
  }

  tm.lap("binning");
  int st = 0;
  st |= upload(&h->facet_type, facet_type, (size_t)h->nrhs * m.nfacets);
  if (inhomogeneous)
    st |= upload(&h->bvals, boundary_values, (size_t)h->nrhs * m.ncells * h->nrt);
  st |= upload(&h->node_slot, node_slot.data(), (size_t)m.nnodes);
  st |= upload(&h->node_patch, node_patch.data(), (size_t)m.nnodes);
  st |= upload<int32_t>(&h->slot_cell, nullptr, (size_t)h->nslots);
  st |= upload<uint32_t>(&h->slot_info, nullptr, (size_t)h->nslots);
  st |= upload<uint8_t>(&h->pn, nullptr, (size_t)h->npatch_total);
  st |= upload<uint8_t>(&h->pflag, nullptr, (size_t)h->npatch_total * h->nrhs);
  if (st)
    return EQLB_ERR_DEVICE;
  HIP_TRY(hipMemset(h->slot_cell, 0xff, sizeof(int32_t) * std::max<int64_t>(h->nslots, 1)));
  HIP_TRY(hipMemset(h->slot_info, 0, sizeof(uint32_t) * std::max<int64_t>(h->nslots, 1)));

  eqlb::BuildArgs a{};
  a.nnodes = m.nnodes;
  a.nfacets = m.nfacets;
  a.nrhs = h->nrhs;
  a.cell_nodes = m.cell_nodes;
  a.cell_facets = m.cell_facets;
  a.facet_nodes = m.facet_nodes;
  a.facet_cells_off = m.facet_cells_off;
  a.facet_cells = m.facet_cells;
  a.node_cells_off = m.node_cells_off;
  a.node_facets_off = m.node_facets_off;
  a.node_facets = m.node_facets;
  a.facet_perm = m.facet_perm;
  a.facet_type = h->facet_type;
  if (h->stress && h->k == 2 && h->stress_flux_bcs)
  {
    std::vector<int8_t> ws, lvl;
    std::vector<int32_t> grp;
    bool any = false;
    h->ws_levels = 1;
    const int stg = find_stress_groups(m, facet_type, node_mask, ws, grp, lvl, h->ws_levels, any);
    if (stg)
      return stg;
    if (any)
    {
      if (upload(&h->node_ws, ws.data(), ws.size()) || upload(&h->node_group, grp.data(), grp.size())
          || upload(&h->node_wslevel, lvl.data(), lvl.size()))
        return EQLB_ERR_DEVICE;
      a.node_ws = h->node_ws;
      a.node_group = h->node_group;
      a.node_wslevel = h->node_wslevel;
    }
  }
  a.node_slot = h->node_slot;
  a.node_patch = h->node_patch;
  a.npatch_total = h->npatch_total;
  a.slot_cell = h->slot_cell;
  a.slot_info = h->slot_info;
  a.pn = h->pn;
  a.pflag = h->pflag;
  a.stride = 0;
  eqlb::launch_build_patches(a, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  tm.lap("plain SoA: upload + builder");
  h->t_stress = h->stress && h->k == 2 && !h->stress_flux_bcs && h->mode == 0;
  if (h->t_stress || (!h->stress && h->k <= 3))
  {
    // fused stress launch: its own tile size, patches of up to 8 facets (bins 0, 1)
    h->t_mixed = false;
    if (h->t_stress)
    {
      // Patches of the bins 0, 1 that are not full (interior with fewer cells than lanes, boundary): on the crossed
      // benchmark meshes the boundary patches only (0.4 %) - the tiles list the full patches and the others go with
      // the rest (generic kernels on a side stream next to the fused kernel); on unstructured meshes most patches -
      // the tiles list every patch of the two bins and the kernel carries both instances of the body.
      // EQLB_STRESS_MIXED_TILES=0/1 forces the choice.
      int64_t nlisted = 0, nnotfull = 0;
      for (int32_t i = 0; i < m.nnodes; ++i)
      {
        const int8_t b = node_bin[i];
        if (b < 0 || b >= 2)
          continue;
        ++nlisted;
        if (!(m.h_node_ncells[i] == m.h_node_nfcts[i] && m.h_node_ncells[i] == eqlb::BIN_P[b]))
          ++nnotfull;
      }
      h->t_mixed = nnotfull * 20 > nlisted;
      if (const char* env = getenv("EQLB_STRESS_MIXED_TILES"))
        h->t_mixed = env[0] != '0';
    }
    const int stt = h->t_stress ? build_tiles(h, node_bin, a, eqlb::stress_tile_cells(), 2, !h->t_mixed)
                                : build_tiles(h, node_bin, a);
    if (stt)
      return stt;
    if (h->mode == 1)
    {
      const int64_t ne = (int64_t)h->ntiles * h->tile_tc * 3;
      if (upload<int32_t>(&h->t_facet_owner, nullptr, (size_t)std::max<int64_t>(ne, 1)))
        return EQLB_ERR_DEVICE;
      eqlb::launch_tile_facet_owner(m, ne, h->t_tile_cells, h->t_facet_owner, nullptr);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
    }
  }
  tm.lap("tiles (total)");
  h->boundary_set = true;
  return EQLB_OK;
}
EQLB_CATCH_ALL

int eqlb_se_kornconst(eqlb_se_t* h, double* cells_kornconst, int32_t memspace, void* stream_)
try
{
  if (!h || !cells_kornconst)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  if (!h->boundary_set)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_kornconst: boundary data not set");
  const eqlb::DeviceMesh& m = h->mesh->m;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!h->d_cks && upload<double>(&h->d_cks, nullptr, (size_t)m.nnodes))
    return EQLB_ERR_DEVICE;
  double* d_korn = cells_kornconst;
  if (memspace == EQLB_MEM_HOST)
  {
    if (!h->d_korn && upload<double>(&h->d_korn, nullptr, (size_t)m.ncells))
      return EQLB_ERR_DEVICE;
    HIP_TRY(hipMemcpyAsync(h->d_korn, cells_kornconst, sizeof(double) * m.ncells, hipMemcpyHostToDevice, stream));
    d_korn = h->d_korn;
  }
  eqlb::launch_korn(m, h->node_slot, h->node_patch, h->slot_cell, h->slot_info, h->pn, h->pflag,
                    h->d_cks, d_korn, stream);
  HIP_TRY(hipGetLastError());
  if (memspace == EQLB_MEM_HOST)
  {
    HIP_TRY(hipMemcpyAsync(cells_kornconst, d_korn, sizeof(double) * m.ncells, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
  }
  return EQLB_OK;
}
EQLB_CATCH_ALL

int eqlb_se_equilibrate_with_kornconst(eqlb_se_t* h, const double* flux_dg, const double* rhs_dg,
                                       double* flux_hdiv, double* cells_kornconst,
                                       int32_t memspace, void* stream_)
try
{
  if (!cells_kornconst)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  const int st = eqlb_se_equilibrate(h, flux_dg, rhs_dg, flux_hdiv, memspace, stream_);
  if (st)
    return st;
  return eqlb_se_kornconst(h, cells_kornconst, memspace, stream_);
}
EQLB_CATCH_ALL

int64_t eqlb_se_num_patches(const eqlb_se_t* h) { return h ? h->npatch_total : 0; }

int eqlb_se_tiling_info(const eqlb_se_t* h, int64_t* ntiles, int64_t* cells_per_tile,
                        int64_t* npatch_instances, int64_t* nlane_slots)
{
  if (!h || !h->boundary_set)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_tiling_info: set the boundary first");
  if (ntiles)
    *ntiles = h->ntiles;
  if (cells_per_tile)
    *cells_per_tile = h->tile_tc;
  if (npatch_instances)
    *npatch_instances = h->t_npatch;
  if (nlane_slots)
    *nlane_slots = h->t_nslots;
  return EQLB_OK;
}

int eqlb_se_set_priority_cells(eqlb_se_t* h, const int32_t* cells, int32_t n)
try
{
  if (!h || n < 0 || (n > 0 && !cells))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_set_priority_cells: invalid argument");
  h->prio_cells.assign(cells, cells + n);
  return EQLB_OK;
}
EQLB_CATCH_ALL

int32_t eqlb_se_num_priority_tiles(const eqlb_se_t* h) { return (h && h->boundary_set) ? h->t_nprio : 0; }

int eqlb_se_equilibrate_tiles(eqlb_se_t* h, const double* flux_dg, const double* rhs_dg, double* flux_hdiv,
                              int32_t tile_first, int32_t tile_count, void* stream)
try
{
  if (!h || tile_first < 0)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_equilibrate_tiles: invalid argument");
  const int32_t f0 = h->tile_first, c0 = h->tile_count;
  h->tile_first = tile_first;
  h->tile_count = tile_count;
  const int st = eqlb_se_equilibrate(h, flux_dg, rhs_dg, flux_hdiv, EQLB_MEM_DEVICE, stream);
  h->tile_first = f0;
  h->tile_count = c0;
  return st;
}
EQLB_CATCH_ALL

int eqlb_se_export_patches(eqlb_se_t* h, int32_t stride, int32_t* ncells, int32_t* cells,
                           int32_t* fcts, int8_t* fcts_local, int8_t* inodes_local,
                           int8_t* reversed)
try
{
  if (!h || !ncells || !cells || !fcts || !fcts_local || !inodes_local || !reversed)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_export_patches: null argument");
  if (!h->boundary_set)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_export_patches: set the boundary first");
  const eqlb::DeviceMesh& m = h->mesh->m;
  if (stride < m.ncells_max + 2)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_export_patches: stride too small");
  const size_t nn = (size_t)m.nnodes;
  int32_t *d_n = nullptr, *d_c = nullptr, *d_f = nullptr;
  int8_t *d_fl = nullptr, *d_il = nullptr, *d_rv = nullptr;
  int st = 0;
  st |= upload<int32_t>(&d_n, nullptr, nn);
  st |= upload<int32_t>(&d_c, nullptr, nn * stride);
  st |= upload<int32_t>(&d_f, nullptr, nn * stride);
  st |= upload<int8_t>(&d_fl, nullptr, nn * stride * 2);
  st |= upload<int8_t>(&d_il, nullptr, nn * stride);
  st |= upload<int8_t>(&d_rv, nullptr, nn * stride * 2);
  if (!st)
  {
    eqlb::BuildArgs a{};
    a.nnodes = m.nnodes;
    a.nfacets = m.nfacets;
    a.nrhs = h->nrhs;
    a.cell_nodes = m.cell_nodes;
    a.cell_facets = m.cell_facets;
    a.facet_nodes = m.facet_nodes;
    a.facet_cells_off = m.facet_cells_off;
    a.facet_cells = m.facet_cells;
    a.node_cells_off = m.node_cells_off;
    a.node_facets_off = m.node_facets_off;
    a.node_facets = m.node_facets;
    a.facet_perm = m.facet_perm;
    a.facet_type = h->facet_type;
    a.node_slot = nullptr;
    a.node_patch = nullptr;
    a.npatch_total = 0;
    a.stride = stride;
    a.ex_ncells = d_n;
    a.ex_cells = d_c;
    a.ex_fcts = d_f;
    a.ex_fl = d_fl;
    a.ex_il = d_il;
    a.ex_rev = d_rv;
    eqlb::launch_build_patches(a, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess)
      e = hipMemcpy(ncells, d_n, nn * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      e = hipMemcpy(cells, d_c, nn * stride * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      e = hipMemcpy(fcts, d_f, nn * stride * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      e = hipMemcpy(fcts_local, d_fl, nn * stride * 2, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      e = hipMemcpy(inodes_local, d_il, nn * stride, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      e = hipMemcpy(reversed, d_rv, nn * stride * 2, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
      st = fail(EQLB_ERR_DEVICE, "eqlb_se_export_patches: %s", hipGetErrorString(e));
  }
  dfree(d_n);
  dfree(d_c);
  dfree(d_f);
  dfree(d_fl);
  dfree(d_il);
  dfree(d_rv);
  return st ? EQLB_ERR_DEVICE : EQLB_OK;
}
EQLB_CATCH_ALL

// The sweep on per-right-hand-side arrays: g[r], f[r], x[r] are the blocks of RHS r (host or device).
static int equilibrate_lists(eqlb_se_t* h, const double* const* g_in, const double* const* f_in,
                             double* const* x_io, int32_t memspace, void* stream_)
{
  if (!h || !g_in || !f_in || !x_io)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  for (int r = 0; r < h->nrhs; ++r)
    if (!g_in[r] || !f_in[r] || !x_io[r])
      return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  if (!h->boundary_set)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_equilibrate: boundary data not set");
  const eqlb::DeviceMesh& m = h->mesh->m;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const size_t s_g = (size_t)m.ncells * h->nd * 2, s_f = (size_t)m.ncells * h->nd; // block sizes
  const size_t n_g = (size_t)h->nrhs * s_g, n_f = (size_t)h->nrhs * s_f;
  // EQLB_SCATTER_AUTO: the tiled launch where it applies and is the fastest (k <= 3, plain flux
  // equilibration, shuffle solver), else slots + reduction
  int scatter_eff = h->scatter;
  // stress of RT_2 without flux BCs on the stress rows: rows 0, 1 and their weak symmetry in one tiled launch
  const bool stress_fused = h->stress && h->t_stress && h->ntiles > 0 && h->solver == EQLB_SOLVER_SHUFFLE
                            && (scatter_eff == EQLB_SCATTER_AUTO || scatter_eff == EQLB_SCATTER_TILED);
  if (scatter_eff == EQLB_SCATTER_AUTO)
    scatter_eff = (stress_fused || (!h->stress && h->k <= 3 && h->solver == EQLB_SOLVER_SHUFFLE && h->ntiles > 0))
                      ? EQLB_SCATTER_TILED
                      : EQLB_SCATTER_SLOTS;
  h->scatter_last = scatter_eff;
  const size_t s_slot = (size_t)m.ncells * h->nrt, n_slot = (size_t)h->nrhs * s_slot;
  // EV mode writes conforming DOFs unless the broken layout is requested
  const bool ev_conf = h->mode == 1 && h->ev_output == 0;
  const size_t s_x = ev_conf ? (size_t)h->ev_ndofs : s_slot, n_x = (size_t)h->nrhs * s_x;
  if (!h->accumulate && scatter_eff == EQLB_SCATTER_ATOMIC)
    return fail(EQLB_ERR_UNSUPPORTED, "\"accumulate\" = 0 is not available with the atomic scatter");
  if (h->mode == 1 && (scatter_eff == EQLB_SCATTER_ATOMIC || (h->solver != EQLB_SOLVER_SHUFFLE && h->k != 4)))
    return fail(EQLB_ERR_UNSUPPORTED, "EV equilibration runs with the shuffle solver (tiled or slot scatter)");

  std::vector<const double*> d_g(g_in, g_in + h->nrhs), d_f(f_in, f_in + h->nrhs);
  std::vector<double*> d_x(x_io, x_io + h->nrhs);
  if (memspace == EQLB_MEM_HOST)
  {
    if (!h->d_flux_dg)
    {
      if (upload<double>(&h->d_flux_dg, nullptr, n_g) || upload<double>(&h->d_rhs_dg, nullptr, n_f)
          || upload<double>(&h->d_flux_hdiv, nullptr, n_x))
        return EQLB_ERR_DEVICE;
    }
    for (int r = 0; r < h->nrhs; ++r)
    {
      HIP_TRY(hipMemcpyAsync(h->d_flux_dg + r * s_g, g_in[r], s_g * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemcpyAsync(h->d_rhs_dg + r * s_f, f_in[r], s_f * sizeof(double), hipMemcpyHostToDevice, stream));
      if (h->accumulate)
        HIP_TRY(hipMemcpyAsync(h->d_flux_hdiv + r * s_x, x_io[r], s_x * sizeof(double), hipMemcpyHostToDevice, stream));
      d_g[r] = h->d_flux_dg + r * s_g;
      d_f[r] = h->d_rhs_dg + r * s_f;
      d_x[r] = h->d_flux_hdiv + r * s_x;
    }
  }
  else if (memspace != EQLB_MEM_DEVICE)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_equilibrate: unknown memory space");

  hipEvent_t* evs = nullptr;
  if (h->timing)
  {
    if (h->ev && !h->ev[eqlb_se::EV_RING * eqlb_se::EV_PER_SET - 1])
    {
      // an earlier hipEventCreate failed half way: start over
      for (int i = 0; i < eqlb_se::EV_RING * eqlb_se::EV_PER_SET; ++i)
        if (h->ev[i])
          (void)hipEventDestroy(h->ev[i]);
      delete[] h->ev;
      h->ev = nullptr;
    }
    if (!h->ev)
    {
      h->ev = new hipEvent_t[eqlb_se::EV_RING * eqlb_se::EV_PER_SET](); // null until created
      for (int i = 0; i < eqlb_se::EV_RING * eqlb_se::EV_PER_SET; ++i)
        HIP_TRY(hipEventCreate(&h->ev[i]));
    }
    evs = h->ev + (h->ev_calls % eqlb_se::EV_RING) * eqlb_se::EV_PER_SET;
  }

  eqlb::SeArgs a{};
  a.cellJ = m.cellJ;
  a.slot_cell = h->slot_cell;
  a.slot_info = h->slot_info;
  a.pn = h->pn;
  a.pflag = h->pflag;
  a.tables = h->tables;
  a.bvals = h->bvals;
  a.status = h->status;
  a.npatch_total = h->npatch_total;
  a.ncells = m.ncells;
  a.nrhs = h->nrhs;
  // data of right-hand side r: the kernels address block rhs_in of flux_dg / rhs_dg and block rhs_out
  // of out; the caller's arrays arrive block by block, the slot buffer is one array
  auto select_rhs = [&](eqlb::SeArgs& aa, int r, bool to_slots) {
    aa.rhs = r;
    aa.flux_dg = d_g[r];
    aa.rhs_dg = d_f[r];
    aa.rhs_in = 0;
    if (to_slots)
    {
      aa.out = h->slots;
      aa.rhs_out = r;
    }
    else
    {
      aa.out = d_x[r];
      aa.rhs_out = 0;
    }
  };

  // ---- slot path: (cell, vertex) rows into the slot buffer, weak symmetry on the slot rows, reduction.
  // first_bin > 0: only the patches of the bins >= first_bin (the rest of a fused stress launch);
  // their sums are ADDED to what the tiled launch wrote ----
  // first_bin = -1: the REST of a fused stress launch - in the bins 0, 1 the patches behind the full ones
  // (Bin::nfull), the higher bins entirely; sums added by the compact reduction over the cells they touch
  // sp_stream / sp_phase: stream of the launches; phase 0 everything, 1 the patch kernels only, 2 the reduction only
  // (the rest of a fused stress launch runs its patch kernels on a side stream next to the fused kernel)
  hipStream_t sp_stream = stream;
  int sp_phase = 0;
  auto run_slot_path = [&](int first_bin, int accumulate) -> int {
    const bool rest = first_bin < 0;
    auto bin_np = [&](int b) -> int64_t {
      if (rest) // (tiles with every patch of the bins 0, 1: the higher bins only)
        return (b < 2) ? (h->t_mixed ? 0 : h->bins[b].npatch - h->bins[b].nfull) : h->bins[b].npatch;
      return (b >= first_bin) ? h->bins[b].npatch : 0;
    };
    auto bin_po = [&](int b) -> int64_t { return h->bins[b].patch_offset + ((rest && b < 2) ? h->bins[b].nfull : 0); };
    auto bin_so = [&](int b) -> int64_t {
      return h->bins[b].slot_offset + ((rest && b < 2) ? h->bins[b].nfull * h->bins[b].P : 0);
    };
    const int cover = rest ? 1 : first_bin; // 0: every patch writes its slot rows
    if (sp_phase == 2)
    {
      for (int r = 0; r < h->nrhs; ++r)
        if (eqlb::launch_reduce_slots_cells(h->nrt, m.ncells, h->nrest_cells, h->rest_cells,
                                            h->slots + (size_t)r * s_slot * 3, d_x[r], sp_stream))
          return fail(EQLB_ERR_UNSUPPORTED, "compact slot reduction for %d DOFs per cell is not in this build", h->nrt);
      return EQLB_OK;
    }
    if (!h->slots)
    {
      if (upload<double>(&h->slots, nullptr, n_slot * 3))
        return EQLB_ERR_DEVICE;
      // slots of (cell, vertex) pairs whose node is not equilibrated here (node_mask, other path) stay zero
      // (on the stream of the patch kernels: a fill on the null stream is not ordered against the non-blocking side
      // stream of a fused stress launch and could wipe rows its kernels have already written)
      HIP_TRY(hipMemsetAsync(h->slots, 0, n_slot * 3 * sizeof(double), sp_stream));
      h->slots_first_bin = eqlb::MAX_BINS;
    }
    // The reduction adds ALL slot rows of a cell.  A run over the bins >= first_bin rewrites only their rows: rows
    // of the lower bins left by an earlier run over more bins (option "scatter" / "solver" changed on this handle)
    // would be added again on top of what the tiled launch wrote
    if (h->slots_first_bin < cover)
      HIP_TRY(hipMemsetAsync(h->slots, 0, n_slot * 3 * sizeof(double), sp_stream));
    h->slots_first_bin = cover;
    eqlb::SeArgs as = a;
    if ((h->mode == 1 && h->k <= 3) || (h->fused && h->solver == EQLB_SOLVER_SHUFFLE && h->k <= 3))
    {
      // all bins in one launch; timing slot 0 holds the fused kernel
      eqlb::FusedBins fb{};
      int64_t nb = 0;
      for (int b = 0; b < eqlb::MAX_BINS; ++b)
      {
        fb.block_start[b] = nb;
        fb.npatch[b] = bin_np(b);
        fb.slot_offset[b] = bin_so(b);
        fb.patch_offset[b] = bin_po(b);
        nb += (fb.npatch[b] * h->bins[b].P + 255) / 256;
      }
      fb.block_start[eqlb::MAX_BINS] = nb;
      for (int r = 0; r < h->nrhs; ++r)
      {
        select_rhs(as, r, true);
        if (evs && r == 0 && first_bin == 0)
          HIP_TRY(hipEventRecord(evs[0], sp_stream));
        const int st = (h->mode == 1) ? eqlb::launch_ev_patch_fused(h->k, as, fb, sp_stream)
                                      : eqlb::launch_se_patch_fused(h->k, h->deg, EQLB_SCATTER_SLOTS, as, fb, sp_stream);
        if (st)
          return fail(st, "fused patch kernel launch failed (k=%d)", h->k);
      }
      if (evs && first_bin == 0)
        HIP_TRY(hipEventRecord(evs[1], sp_stream));
    }
    else
      for (int b = 0; b < eqlb::MAX_BINS; ++b)
      {
        if (bin_np(b) == 0)
          continue;
        as.npatch = bin_np(b);
        as.slot_offset = bin_so(b);
        as.patch_offset = bin_po(b);
        if (evs && first_bin == 0)
          HIP_TRY(hipEventRecord(evs[2 * b], sp_stream));
        for (int r = 0; r < h->nrhs; ++r)
        {
          select_rhs(as, r, true);
          const int st = eqlb::launch_se_patch(h->k, h->deg, h->bins[b].P, h->solver, EQLB_SCATTER_SLOTS, as, sp_stream, h->mode);
          if (st)
            return fail(st, "patch kernel launch failed (k=%d, P=%d)", h->k, h->bins[b].P);
        }
        if (evs && first_bin == 0)
          HIP_TRY(hipEventRecord(evs[2 * b + 1], sp_stream));
      }
    if (h->stress)
    {
      // weak symmetry of rows 0, 1 on the patch-local stresses held in the slots
      // (se/reconstruction.hpp:237-270; the grouped boundary patches of :170-234 are flagged by the
      // patch builder: PFLAG_WS_SKIP / PFLAG_WS_GROUP)
      if (evs && first_bin == 0)
        HIP_TRY(hipEventRecord(evs[2 * eqlb::MAX_BINS + 2], sp_stream));
      select_rhs(as, 0, true); // the kernel works on the slot rows of RHS 0 and 1
      // (overlapping groups of boundary patches: one pass per level, a pass skips the patches of other levels)
      for (int lv = 0; lv < h->ws_levels; ++lv)
        for (int b = 0; b < eqlb::MAX_BINS; ++b)
        {
          if (bin_np(b) == 0)
            continue;
          as.npatch = bin_np(b);
          as.slot_offset = bin_so(b);
          as.patch_offset = bin_po(b);
          as.ws_level = lv;
          const int st = eqlb::launch_se_weaksym(h->k, h->bins[b].P, !h->stress_flux_bcs, as, sp_stream);
          if (st)
            return fail(st, "weak-symmetry kernel launch failed (k=%d, P=%d)", h->k, h->bins[b].P);
        }
      if (evs && first_bin == 0)
        HIP_TRY(hipEventRecord(evs[2 * eqlb::MAX_BINS + 3], sp_stream));
    }
    if (evs && first_bin == 0)
      HIP_TRY(hipEventRecord(evs[2 * eqlb::MAX_BINS], sp_stream));
    // blocks that lie behind one another (one array, the usual case) are reduced by one launch
    bool contiguous = true;
    for (int r = 1; r < h->nrhs; ++r)
      contiguous = contiguous && d_x[r] == d_x[0] + r * s_x;
    const int nlaunch = contiguous ? 1 : h->nrhs, per = contiguous ? h->nrhs : 1;
    if (rest && sp_phase == 1)
      return EQLB_OK;
    if (rest)
    {
      // only the cells that a patch of the generic kernels touches (the slot rows of their other vertices are zero)
      for (int r = 0; r < h->nrhs; ++r)
        if (eqlb::launch_reduce_slots_cells(h->nrt, m.ncells, h->nrest_cells, h->rest_cells,
                                            h->slots + (size_t)r * s_slot * 3, d_x[r], sp_stream))
          return fail(EQLB_ERR_UNSUPPORTED, "compact slot reduction for %d DOFs per cell is not in this build", h->nrt);
      return EQLB_OK;
    }
    for (int l = 0; l < nlaunch; ++l)
    {
      const double* sl = h->slots + (size_t)l * s_slot * 3;
      if (ev_conf)
        eqlb::launch_ev_reduce(m, h->k, per, h->ev_cell_dofs, h->ev_ndofs, sl, d_x[l], accumulate, h->ev_basis,
                               (h->ev_basis && h->ev_basis_has_R) ? h->ev_basis + h->nrt * h->nrt : nullptr, sp_stream);
      else if (eqlb::launch_reduce_slots(h->nrt, m.ncells, per, sl, d_x[l], accumulate, sp_stream))
        return fail(EQLB_ERR_UNSUPPORTED, "slot reduction for %d DOFs per cell is not in this build", h->nrt);
    }
    if (evs && first_bin == 0)
      HIP_TRY(hipEventRecord(evs[2 * eqlb::MAX_BINS + 1], sp_stream));
    return EQLB_OK;
  };

  if (scatter_eff == EQLB_SCATTER_TILED)
  {
    if ((h->stress && !stress_fused) || h->solver != EQLB_SOLVER_SHUFFLE || h->ntiles == 0)
      return fail(EQLB_ERR_UNSUPPORTED,
                  "the tiled scatter is available for k <= 3 with the shuffle solver (stress: RT_2 without "
                  "flux boundary conditions on the stress rows)");
    if (h->tile_first > h->ntiles)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "tile_first %d beyond the %d tiles", h->tile_first, h->ntiles);
    const int32_t tcount = (h->tile_count < 0) ? h->ntiles - h->tile_first
                                               : std::min(h->tile_count, h->ntiles - h->tile_first);
    eqlb::TileArgs ta{h->t_tiles, h->t_tile_cells, tcount, h->tile_tc,
                      ev_conf ? h->t_facet_owner : nullptr, h->ev_cell_dofs, h->ev_ndofs, m.nfacets,
                      h->tile_first, h->accumulate, ev_conf ? h->ev_basis : nullptr,
                      (ev_conf && h->ev_basis && h->ev_basis_has_R) ? h->ev_basis + h->nrt * h->nrt : nullptr};
    eqlb::SeArgs at = a;
    at.slot_cell = h->t_slot_cell;
    at.slot_info = h->t_slot_info;
    at.pn = h->t_pn;
    at.pflag = h->t_pflag;
    at.npatch_total = h->t_npatch;
    if (evs)
      HIP_TRY(hipEventRecord(evs[0], stream));
    int r0 = 0;
    // the rest of a fused stress launch (boundary patches, interior patches that are not full, bins of more than 8
    // lanes): its patch kernels - a handful of small launches, 50 us back to back at 1M triangles - run on a side
    // stream NEXT TO the fused kernel, their sums are added behind it.  With the FIRST range of tiles of a two-phase
    // sweep: its patches touch ghost cells like any other, and the caller packs the ghost rows behind that range
    // (option accumulate = 0: the tiled launches STORE, the rest can only be added behind the last of them; an empty
    // range - a rank without priority tiles, or with priority tiles only - takes nothing along)
    const bool with_first_range = tcount > 0 && (h->accumulate ? h->tile_first == 0 : h->tile_first + tcount == h->ntiles);
    const bool rest_now = stress_fused && h->t_rest > 0
#ifdef EQLB_EXP_REST_LAST // (the order before the fix, to show that tests/test_gpu_halo.py sees it)
                          && (h->tile_first + tcount == h->ntiles);
#else
                          && with_first_range;
#endif
    if (rest_now)
    {
      if (!h->side_stream)
      {
        HIP_TRY(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(h->ev_fork, stream));
      HIP_TRY(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
      hipEvent_t* keep = evs;
      evs = nullptr;
      sp_stream = h->side_stream;
      sp_phase = 1;
      const int st = run_slot_path(-1, 1);
      sp_stream = stream;
      sp_phase = 0;
      evs = keep;
      if (st)
        return st;
      HIP_TRY(hipEventRecord(h->ev_join, h->side_stream));
    }
    if (stress_fused)
    {
      // rows 0, 1 of the stress and their weak symmetry in one launch
      select_rhs(at, 0, false);
      const int st = eqlb::launch_se_stress_tiled(at, ta, d_g.data(), d_f.data(), d_x.data(), stream, h->t_mixed);
      if (st)
        return fail(st, "fused stress kernel launch failed");
      r0 = 2;
    }
    if (h->multi_rhs && h->nrhs - r0 > 1)
    {
      // all (remaining) right-hand sides in one launch per chunk of MULTI_RHS_MAX
      for (int rb = r0; rb < h->nrhs; rb += eqlb::MULTI_RHS_MAX)
      {
        eqlb::MultiRhs mr{};
        mr.n = std::min(eqlb::MULTI_RHS_MAX, h->nrhs - rb);
        mr.rhs0 = rb;
        for (int i = 0; i < mr.n; ++i)
        {
          mr.g[i] = d_g[rb + i];
          mr.f[i] = d_f[rb + i];
          mr.x[i] = d_x[rb + i];
        }
        select_rhs(at, rb, false);
        const int st = eqlb::launch_se_patch_tiled_multi(h->k, h->deg, h->mode, at, ta, mr, stream);
        if (st)
          return fail(st, "tiled multi-RHS patch kernel launch failed (k=%d)", h->k);
      }
    }
    else
      for (int r = r0; r < h->nrhs; ++r)
      {
        select_rhs(at, r, false);
        const int st = eqlb::launch_se_patch_tiled(h->k, h->deg, h->mode, at, ta, stream);
        if (st)
          return fail(st, "tiled patch kernel launch failed (k=%d)", h->k);
      }
    if (evs)
      HIP_TRY(hipEventRecord(evs[1], stream));
    if (rest_now)
    {
      HIP_TRY(hipStreamWaitEvent(stream, h->ev_join, 0));
      hipEvent_t* keep = evs;
      evs = nullptr;
      sp_phase = 2;
      const int st = run_slot_path(-1, 1);
      sp_phase = 0;
      evs = keep;
      if (st)
        return st;
    }
  }
  else if (scatter_eff == EQLB_SCATTER_SLOTS)
  {
    const int st = run_slot_path(0, h->accumulate);
    if (st)
      return st;
  }
  else
  {
    // fp64 global atomics straight into flux_hdiv
    if (h->stress)
      return fail(EQLB_ERR_UNSUPPORTED, "stress equilibration needs the slot or the tiled scatter");
    eqlb::SeArgs aa = a;
    if (h->fused && h->solver == EQLB_SOLVER_SHUFFLE && h->k <= 3)
    {
      eqlb::FusedBins fb{};
      int64_t nb = 0;
      for (int b = 0; b < eqlb::MAX_BINS; ++b)
      {
        fb.block_start[b] = nb;
        fb.npatch[b] = h->bins[b].npatch;
        fb.slot_offset[b] = h->bins[b].slot_offset;
        fb.patch_offset[b] = h->bins[b].patch_offset;
        nb += (h->bins[b].npatch * h->bins[b].P + 255) / 256;
      }
      fb.block_start[eqlb::MAX_BINS] = nb;
      for (int r = 0; r < h->nrhs; ++r)
      {
        select_rhs(aa, r, false);
        if (evs && r == 0)
          HIP_TRY(hipEventRecord(evs[0], stream));
        const int st = eqlb::launch_se_patch_fused(h->k, h->deg, scatter_eff, aa, fb, stream);
        if (st)
          return fail(st, "fused patch kernel launch failed (k=%d)", h->k);
      }
      if (evs)
        HIP_TRY(hipEventRecord(evs[1], stream));
    }
    else
      for (int b = 0; b < eqlb::MAX_BINS; ++b)
      {
        if (h->bins[b].npatch == 0)
          continue;
        aa.npatch = h->bins[b].npatch;
        aa.slot_offset = h->bins[b].slot_offset;
        aa.patch_offset = h->bins[b].patch_offset;
        if (evs)
          HIP_TRY(hipEventRecord(evs[2 * b], stream));
        for (int r = 0; r < h->nrhs; ++r)
        {
          select_rhs(aa, r, false);
          const int st = eqlb::launch_se_patch(h->k, h->deg, h->bins[b].P, h->solver, scatter_eff, aa, stream);
          if (st)
            return fail(st, "patch kernel launch failed (k=%d, P=%d)", h->k, h->bins[b].P);
        }
        if (evs)
          HIP_TRY(hipEventRecord(evs[2 * b + 1], stream));
      }
  }
  if (evs)
    ++h->ev_calls;
  HIP_TRY(hipGetLastError());

  if (memspace == EQLB_MEM_HOST)
  {
    for (int r = 0; r < h->nrhs; ++r)
      HIP_TRY(hipMemcpyAsync(x_io[r], d_x[r], s_x * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    int32_t status = 0;
    HIP_TRY(hipMemcpy(&status, h->status, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (status)
    {
      (void)hipMemset(h->status, 0, sizeof(int32_t));
      return fail(EQLB_ERR_SINGULAR, "patch system not positive definite");
    }
  }
  return EQLB_OK;
}

int eqlb_se_equilibrate_lists(eqlb_se_t* h, const double* const* flux_dg, const double* const* rhs_dg,
                              double* const* flux_hdiv, int32_t memspace, void* stream)
try
{
  return equilibrate_lists(h, flux_dg, rhs_dg, flux_hdiv, memspace, stream);
}
EQLB_CATCH_ALL

int eqlb_se_equilibrate(eqlb_se_t* h, const double* flux_dg, const double* rhs_dg,
                        double* flux_hdiv, int32_t memspace, void* stream_)
try
{
  if (!h || !flux_dg || !rhs_dg || !flux_hdiv)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  const eqlb::DeviceMesh& m = h->mesh->m;
  const size_t s_g = (size_t)m.ncells * h->nd * 2, s_f = (size_t)m.ncells * h->nd;
  const size_t s_x = (h->mode == 1 && h->ev_output == 0) ? (size_t)h->ev_ndofs : (size_t)m.ncells * h->nrt;
  std::vector<const double*> g(h->nrhs), f(h->nrhs);
  std::vector<double*> x(h->nrhs);
  for (int r = 0; r < h->nrhs; ++r)
  {
    g[r] = flux_dg + r * s_g;
    f[r] = rhs_dg + r * s_f;
    x[r] = flux_hdiv + r * s_x;
  }
  return equilibrate_lists(h, g.data(), f.data(), x.data(), memspace, stream_);
}
EQLB_CATCH_ALL

int eqlb_se_check_status(eqlb_se_t* h, void* stream_)
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_check_status: null handle");
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int32_t status = 0;
  HIP_TRY(hipMemcpyAsync(&status, h->status, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (status)
  {
    HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(int32_t), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return fail(EQLB_ERR_SINGULAR, "patch system not positive definite");
  }
  return EQLB_OK;
}

double eqlb_se_last_kernel_ms(const eqlb_se_t* h, int32_t which)
{
  // which = b (0..4): patch kernel of bin b (P = 4 << b); 5: slot reduction; 6: weak-symmetry kernels.
  // Average device time per launch over the calls recorded since timing was enabled
  // (at most the last EV_RING calls).  Synchronises with the recorded events.
  if (!h || !h->ev || h->ev_calls == 0 || which < 0 || which > eqlb::MAX_BINS + 1)
    return 0.0;
  if (which == eqlb::MAX_BINS + 1 && !h->stress)
    return 0.0;
  const bool fused_run = (h->mode == 1 && h->k <= 3) || h->scatter_last == EQLB_SCATTER_TILED
                         || (h->fused && h->solver == EQLB_SOLVER_SHUFFLE && h->k <= 3);
  if (which < eqlb::MAX_BINS && ((fused_run && which != 0) || (!fused_run && h->bins[which].npatch == 0)))
    return 0.0;
  if (which == eqlb::MAX_BINS && h->scatter_last != EQLB_SCATTER_SLOTS)
    return 0.0;
  const int64_t nset = std::min<int64_t>(h->ev_calls, eqlb_se::EV_RING);
  double sum = 0.0;
  for (int64_t s = 0; s < nset; ++s)
  {
    hipEvent_t* evs = h->ev + s * eqlb_se::EV_PER_SET;
    float ms = 0.f;
    if (hipEventSynchronize(evs[2 * which + 1]) != hipSuccess
        || hipEventElapsedTime(&ms, evs[2 * which], evs[2 * which + 1]) != hipSuccess)
      return 0.0;
    sum += ms;
  }
  return sum / (double)nset;
}

int eqlb_project_dg(eqlb_mesh_t* mesh, int32_t degree, int32_t bs, int32_t nrhs, int32_t nq,
                    const double* qpoints, const double* qweights, const double* qvalues,
                    double* out, int32_t memspace, void* stream_)
try
{
  if (!mesh || !qpoints || !qweights || !qvalues || !out || bs < 1 || nrhs < 1)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Local solver: Input sizes does not match");
  std::vector<double> Pm;
  const int st = eqlb::projection_matrix_host(degree, nq, qpoints, qweights, Pm);
  if (st)
    return fail(st, "eqlb_project_dg: unsupported degree %d or number of points %d", degree, nq);
  const int nd = (degree + 1) * (degree + 2) / 2;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t ncells = (int64_t)mesh->m.ncells * nrhs; // the RHS are stacked cell blocks
  const size_t n_in = (size_t)ncells * nq * bs, n_out = (size_t)ncells * nd * bs;
  double *d_P = nullptr, *d_in = nullptr, *d_out = nullptr;
  if (upload(&d_P, Pm.data(), Pm.size()))
    return EQLB_ERR_DEVICE;
  int rc = EQLB_OK;
  if (memspace == EQLB_MEM_HOST)
  {
    if (upload(&d_in, qvalues, n_in) || upload<double>(&d_out, nullptr, n_out))
      rc = EQLB_ERR_DEVICE;
  }
  else
  {
    d_in = const_cast<double*>(qvalues);
    d_out = out;
  }
  if (!rc)
  {
    eqlb::launch_project_dg(ncells, nd, nq, bs, d_P, d_in, d_out, stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && memspace == EQLB_MEM_HOST)
      e = hipMemcpy(out, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream); // d_P is freed below
    if (e != hipSuccess)
      rc = fail(EQLB_ERR_DEVICE, "eqlb_project_dg: %s", hipGetErrorString(e));
  }
  dfree(d_P);
  if (memspace == EQLB_MEM_HOST)
  {
    dfree(d_in);
    dfree(d_out);
  }
  return rc;
}
EQLB_CATCH_ALL

int eqlb_get_reference_table(int32_t k, int32_t degree_dg, const char* name, double* out,
                             int32_t capacity)
{
  std::vector<double> tab;
  if (!name || !out || eqlb::fill_tables_host(k, degree_dg, tab) != 0)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_get_reference_table: unknown table");
  const int nrt = k * (k + 2), nd = (degree_dg + 1) * (degree_dg + 2) / 2, nq = k * (k + 1) / 2;
  // layout of the table buffer (fill_tables_host): S | F | H | D | ...; the three rows of H are padded
  // to an even number of doubles there (Sizes::HROW) and returned without the padding
  const size_t hrow = (size_t)nd * nq, hrow_pad = hrow + (hrow & 1);
  const size_t nS = (size_t)3 * nrt * nrt, nF = (size_t)9 * nd * k, nH = 3 * hrow, nHp = 3 * hrow_pad,
               nD = (size_t)6 * nd * nq;
  size_t off = 0, len = 0;
  if (!strcmp(name, "S"))
  {
    off = 0;
    len = nS;
  }
  else if (!strcmp(name, "F"))
  {
    off = nS;
    len = nF;
  }
  else if (!strcmp(name, "H"))
  {
    if ((size_t)capacity < nH)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_get_reference_table: capacity too small");
    for (int n = 0; n < 3; ++n)
      std::copy(tab.begin() + nS + nF + n * hrow_pad, tab.begin() + nS + nF + n * hrow_pad + hrow,
                out + n * hrow);
    return (int)nH;
  }
  else if (!strcmp(name, "D"))
  {
    off = nS + nF + nHp;
    len = nD;
  }
  else
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_get_reference_table: unknown table '%s'", name);
  if ((size_t)capacity < len)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_get_reference_table: capacity too small");
  std::copy(tab.begin() + off, tab.begin() + off + len, out);
  return (int)len;
}

static int estimate_impl(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux_hdiv,
                         const double* flux_dg, const double* rhs_dg, double* cell_div2,
                         double* cell_sig2, double* facet_jump, int32_t memspace, void* stream_,
                         double alpha, double beta)
{
  if (!mesh || !flux_hdiv || !flux_dg || !rhs_dg || nrhs < 1 || k < 1 || k > 4)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_estimate: invalid argument");
  const eqlb::DeviceMesh& m = mesh->m;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int nrt = k * (k + 2), nd = k * (k + 1) / 2;
  const size_t n_x = (size_t)nrhs * m.ncells * nrt, n_g = (size_t)nrhs * m.ncells * nd * 2,
               n_f = (size_t)nrhs * m.ncells * nd;
  const size_t n_c = (size_t)nrhs * m.ncells, n_e = (size_t)nrhs * m.nfacets;
  if (memspace == EQLB_MEM_DEVICE)
  {
    const int st = eqlb::launch_estimate(m, k, nrhs, flux_hdiv, flux_dg, rhs_dg, cell_div2, cell_sig2,
                                         facet_jump, alpha, beta, stream);
    return st ? fail(st, "eqlb_se_estimate: kernel launch failed") : EQLB_OK;
  }
  if (memspace != EQLB_MEM_HOST)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_estimate: unknown memory space");
  double *d_x = nullptr, *d_g = nullptr, *d_f = nullptr, *d_d = nullptr, *d_s = nullptr, *d_j = nullptr;
  int st = upload(&d_x, flux_hdiv, n_x) | upload(&d_g, flux_dg, n_g) | upload(&d_f, rhs_dg, n_f);
  if (cell_div2)
    st |= upload<double>(&d_d, nullptr, n_c);
  if (cell_sig2)
    st |= upload<double>(&d_s, nullptr, n_c);
  if (facet_jump)
    st |= upload<double>(&d_j, nullptr, n_e);
  int rc = st ? EQLB_ERR_DEVICE : eqlb::launch_estimate(m, k, nrhs, d_x, d_g, d_f, d_d, d_s, d_j, alpha, beta, stream);
  hipError_t e = hipSuccess;
  if (!rc && cell_div2)
    e = hipMemcpy(cell_div2, d_d, n_c * sizeof(double), hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && cell_sig2)
    e = hipMemcpy(cell_sig2, d_s, n_c * sizeof(double), hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && facet_jump)
    e = hipMemcpy(facet_jump, d_j, n_e * sizeof(double), hipMemcpyDeviceToHost);
  dfree(d_x);
  dfree(d_g);
  dfree(d_f);
  dfree(d_d);
  dfree(d_s);
  dfree(d_j);
  if (rc || e != hipSuccess)
    return fail(EQLB_ERR_DEVICE, "eqlb_se_estimate: device error");
  return EQLB_OK;
}

int eqlb_se_estimate(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux_hdiv,
                     const double* flux_dg, const double* rhs_dg, double* cell_div2,
                     double* cell_sig2, double* facet_jump, int32_t memspace, void* stream)
{
  return estimate_impl(mesh, k, nrhs, flux_hdiv, flux_dg, rhs_dg, cell_div2, cell_sig2, facet_jump,
                       memspace, stream, 0.0, 1.0);
}

int eqlb_ev_estimate(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux_broken,
                     const double* flux_dg, const double* rhs_dg, double* cell_div2,
                     double* cell_sig2, double* facet_jump, int32_t memspace, void* stream)
{
  return estimate_impl(mesh, k, nrhs, flux_broken, flux_dg, rhs_dg, cell_div2, cell_sig2, facet_jump,
                       memspace, stream, -1.0, 0.0);
}

// Host arrays of a call staged on the device for its duration (inputs copied in, outputs copied back).
namespace
{
struct Staging
{
  std::vector<double*> bufs;
  struct Out
  {
    double *host, *dev;
    size_t n;
  };
  std::vector<Out> outs;
  bool bad = false;
  const double* in(const double* host, size_t n)
  {
    if (!host)
      return nullptr;
    double* d = nullptr;
    if (upload(&d, host, n))
      bad = true;
    bufs.push_back(d);
    return d;
  }
  double* out(double* host, size_t n)
  {
    if (!host)
      return nullptr;
    double* d = nullptr;
    if (upload<double>(&d, nullptr, n))
      bad = true;
    bufs.push_back(d);
    outs.push_back({host, d, n});
    return d;
  }
  bool fetch()
  {
    for (const Out& o : outs)
      if (hipMemcpy(o.host, o.dev, o.n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return false;
    return true;
  }
  ~Staging()
  {
    for (double* b : bufs)
      if (b)
        (void)hipFree(b);
  }
};
} // namespace

int eqlb_se_estimate_stress(eqlb_mesh_t* mesh, int32_t k, const double* flux_hdiv, const double* korn,
                            double pi_1, double* cell_energy, double* cell_wsym, double* node_asym,
                            int32_t memspace, void* stream_)
{
  if (!mesh || !flux_hdiv || k < 1 || k > 4 || !(pi_1 > -1.0))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_estimate_stress: invalid argument");
  if (memspace != EQLB_MEM_DEVICE && memspace != EQLB_MEM_HOST)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_se_estimate_stress: unknown memory space");
  eqlb::DeviceMesh& m = mesh->m;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (node_asym)
  {
    std::lock_guard<std::mutex> g(mesh->tiling_mutex); // (the mesh-level lock: first use from several threads)
    if (!m.node_cells && upload(&m.node_cells, m.h_node_cells.data(), m.h_node_cells.size()))
      return fail(EQLB_ERR_DEVICE, "eqlb_se_estimate_stress: device allocation failed");
  }
  const size_t nx = (size_t)m.ncells * k * (k + 2);
  int rc;
  if (memspace == EQLB_MEM_DEVICE)
    rc = eqlb::launch_estimate_stress(m, m.node_cells, k, flux_hdiv, flux_hdiv + nx, korn, pi_1, cell_energy,
                                      cell_wsym, node_asym, stream);
  else
  {
    Staging s;
    const double* d_x = s.in(flux_hdiv, 2 * nx);
    const double* d_k = s.in(korn, m.ncells);
    double* d_e = s.out(cell_energy, m.ncells);
    double* d_w = s.out(cell_wsym, m.ncells);
    double* d_a = s.out(node_asym, m.nnodes);
    if (s.bad)
      return fail(EQLB_ERR_DEVICE, "eqlb_se_estimate_stress: device allocation failed");
    rc = eqlb::launch_estimate_stress(m, m.node_cells, k, d_x, d_x + nx, d_k, pi_1, d_e, d_w, d_a, stream);
    if (!rc && !s.fetch())
      rc = EQLB_ERR_DEVICE;
  }
  return rc ? fail(rc, "eqlb_se_estimate_stress: device error") : EQLB_OK;
}

int eqlb_oscillation(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, const double* flux, const double* flux_dg,
                     int32_t nq, const double* qpoints, const double* qweights, const double* fvalues,
                     const double* korn, double* out, int32_t memspace, void* stream_)
{
  if (!mesh || !flux || !qpoints || !qweights || !fvalues || !out || nrhs < 1 || k < 1 || k > 4 || nq < 1
      || nq > 128)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_oscillation: invalid argument");
  if (memspace != EQLB_MEM_DEVICE && memspace != EQLB_MEM_HOST)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_oscillation: unknown memory space");
  const eqlb::DeviceMesh& m = mesh->m;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const size_t nx = (size_t)nrhs * m.ncells * k * (k + 2), ng = (size_t)nrhs * m.ncells * k * (k + 1);
  int rc;
  if (memspace == EQLB_MEM_DEVICE)
    rc = eqlb::launch_oscillation(m, k, nrhs, flux, flux_dg, nq, qpoints, qweights, fvalues, korn, out, stream);
  else
  {
    Staging s;
    const double* d_x = s.in(flux, nx);
    const double* d_g = s.in(flux_dg, ng);
    const double* d_f = s.in(fvalues, (size_t)nrhs * m.ncells * nq);
    const double* d_k = s.in(korn, m.ncells);
    double* d_o = s.out(out, (size_t)nrhs * m.ncells);
    if (s.bad)
      return fail(EQLB_ERR_DEVICE, "eqlb_oscillation: device allocation failed");
    rc = eqlb::launch_oscillation(m, k, nrhs, d_x, d_g, nq, qpoints, qweights, d_f, d_k, d_o, stream);
    if (!rc && !s.fetch())
      rc = EQLB_ERR_DEVICE;
  }
  return rc ? fail(rc, "eqlb_oscillation: device error") : EQLB_OK;
}

int eqlb_halo_pack(int32_t nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells,
                   double* x, double* buf, int32_t clear, void* stream)
{
  if (nrhs < 0 || nlist < 0 || nrt < 1 || (nlist > 0 && (!cells || !x || !buf)))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_pack: invalid argument");
  eqlb::launch_halo_pack(nrhs, nlist, nrt, ncells, cells, x, buf, clear, reinterpret_cast<hipStream_t>(stream));
  return (hipGetLastError() == hipSuccess) ? EQLB_OK : fail(EQLB_ERR_DEVICE, "eqlb_halo_pack: launch failed");
}

int eqlb_halo_unpack_add(int32_t nrhs, int32_t nlist, int32_t nrt, int64_t ncells, const int64_t* cells,
                         double* x, const double* buf, void* stream)
{
  if (nrhs < 0 || nlist < 0 || nrt < 1 || (nlist > 0 && (!cells || !x || !buf)))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_unpack_add: invalid argument");
  eqlb::launch_halo_unpack_add(nrhs, nlist, nrt, ncells, cells, x, buf, reinterpret_cast<hipStream_t>(stream));
  return (hipGetLastError() == hipSuccess) ? EQLB_OK
                                           : fail(EQLB_ERR_DEVICE, "eqlb_halo_unpack_add: launch failed");
}

// ---- constrained-minimisation (EV) equilibrator ---------------------------------------------------
int eqlb_ev_create(eqlb_mesh_t* mesh, int32_t k, int32_t nrhs, eqlb_ev_t** handle)
try
{
  if (!handle)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_create: null argument");
  eqlb_se* se = nullptr;
  const int st = eqlb_se_create(mesh, k, k - 1, nrhs, 0, 0, &se);
  if (st)
    return st;
  se->mode = 1;
  se->ev_ndofs = (int64_t)mesh->m.nfacets * k + (int64_t)mesh->m.ncells * (k * k - k);
  eqlb_ev* h = new eqlb_ev();
  h->se = se;
  *handle = h;
  return EQLB_OK;
}
EQLB_CATCH_ALL

void eqlb_ev_destroy(eqlb_ev_t* h)
{
  if (!h)
    return;
  if (h->se)
  {
    dfree(h->se->ev_cell_dofs);
    dfree(h->se->ev_basis);
    eqlb_se_destroy(h->se);
  }
  delete h;
}

int eqlb_ev_set_option(eqlb_ev_t* h, const char* key, int32_t value)
{
  if (!h || !key)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_option: null argument");
  if (!strcmp(key, "output"))
  {
    if (value != 0 && value != 1)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "unknown output layout %d", value);
    h->se->ev_output = value;
    dfree(h->se->d_flux_hdiv); // staging size depends on the layout
    dfree(h->se->d_flux_dg);
    dfree(h->se->d_rhs_dg);
    return EQLB_OK;
  }
  if (!strcmp(key, "boundary_basis"))
  {
    if (value != 0 && value != 1)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "unknown boundary basis %d", value);
    h->se->ev_bv_hier = value;
    return EQLB_OK;
  }
  if (!strcmp(key, "timing") || !strcmp(key, "scatter") || !strcmp(key, "accumulate") || !strcmp(key, "tile_cells")
      || !strcmp(key, "multi_rhs"))
    return eqlb_se_set_option(h->se, key, value);
  return fail(EQLB_ERR_INVALID_ARGUMENT, "unknown option '%s'", key);
}

int eqlb_ev_set_dofmap(eqlb_ev_t* h, const int32_t* cell_dofs, int64_t ndofs)
try
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_dofmap: null argument");
  eqlb_se* se = h->se;
  const eqlb::DeviceMesh& m = se->mesh->m;
  dfree(se->ev_cell_dofs);
  dfree(se->d_flux_hdiv);
  dfree(se->d_flux_dg);
  dfree(se->d_rhs_dg);
  if (!cell_dofs)
  {
    se->ev_ndofs = (int64_t)m.nfacets * se->k + (int64_t)m.ncells * (se->k * se->k - se->k);
    return EQLB_OK;
  }
  const size_t n = (size_t)m.ncells * se->nrt;
  for (size_t i = 0; i < n; ++i)
    if (cell_dofs[i] < 0 || cell_dofs[i] >= ndofs)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_dofmap: DOF %d out of range", cell_dofs[i]);
  if (upload(&se->ev_cell_dofs, cell_dofs, n))
    return EQLB_ERR_DEVICE;
  se->ev_ndofs = ndofs;
  return EQLB_OK;
}
EQLB_CATCH_ALL

int eqlb_ev_set_basis_transform(eqlb_ev_t* h, const double* C, const double* R)
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_basis_transform: null argument");
  eqlb_se* se = h->se;
  dfree(se->ev_basis);
  se->ev_basis_has_R = false;
  if (!C)
    return EQLB_OK;
  const int nrt = se->nrt, k = se->k;
  // facet rows of C must not see anything but their own facet block (the other functions of the
  // hierarchic element have no normal trace there): the two cells of a facet would disagree otherwise
  for (int f = 0; f < 3; ++f)
    for (int j = 0; j < k; ++j)
      for (int c = 0; c < nrt; ++c)
        if ((c < f * k || c >= (f + 1) * k) && C[(f * k + j) * nrt + c] != 0.0)
          return fail(EQLB_ERR_INVALID_ARGUMENT,
                      "eqlb_ev_set_basis_transform: facet DOF %d of the target element depends on DOF %d outside its "
                      "facet", f * k + j, c);
  // [C | R | facet maps]: broken facet DOFs = (facet block of C)^-1 [R^-1] x target facet DOFs, for the boundary values
  std::vector<double> buf((size_t)nrt * nrt + k * k + 6 * k * k, 0.0);
  std::copy(C, C + (size_t)nrt * nrt, buf.begin());
  double* Rd = buf.data() + (size_t)nrt * nrt;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j)
      Rd[i * k + j] = R ? R[i * k + j] : (i == j ? 1.0 : 0.0);
  auto invert = [k](const double* A, double* Ai) -> bool { // Gauss-Jordan with partial pivoting, k <= 4
    double w[4][8];
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
      {
        w[i][j] = A[i * k + j];
        w[i][k + j] = (i == j) ? 1.0 : 0.0;
      }
    for (int c = 0; c < k; ++c)
    {
      int p = c;
      for (int r = c + 1; r < k; ++r)
        if (std::fabs(w[r][c]) > std::fabs(w[p][c]))
          p = r;
      if (w[p][c] == 0.0)
        return false;
      for (int j = 0; j < 2 * k; ++j)
        std::swap(w[c][j], w[p][j]);
      const double ip = 1.0 / w[c][c];
      for (int j = 0; j < 2 * k; ++j)
        w[c][j] *= ip;
      for (int r = 0; r < k; ++r)
        if (r != c)
        {
          const double f_ = w[r][c];
          for (int j = 0; j < 2 * k; ++j)
            w[r][j] -= f_ * w[c][j];
        }
    }
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
        Ai[i * k + j] = w[i][k + j];
    return true;
  };
  double Ri[16], Cf[16], Cfi[16];
  if (!invert(Rd, Ri))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_basis_transform: R is singular");
  double* maps = Rd + k * k;
  for (int f = 0; f < 3; ++f)
  {
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
        Cf[i * k + j] = C[(f * k + i) * nrt + f * k + j];
    if (!invert(Cf, Cfi))
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_basis_transform: facet block %d of C is singular", f);
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
      {
        maps[((f * 2 + 0) * k + i) * k + j] = Cfi[i * k + j];
        double a_ = 0.0;
        for (int q = 0; q < k; ++q)
          a_ += Cfi[i * k + q] * Ri[q * k + j];
        maps[((f * 2 + 1) * k + i) * k + j] = a_;
      }
  }
  if (upload(&se->ev_basis, buf.data(), buf.size()))
    return EQLB_ERR_DEVICE;
  se->ev_basis_has_R = R != nullptr;
  return EQLB_OK;
}

int64_t eqlb_ev_num_dofs(const eqlb_ev_t* h) { return h ? h->se->ev_ndofs : 0; }

int eqlb_ev_set_boundary(eqlb_ev_t* h, const int8_t* facet_type, const double* boundary_values,
                         const uint8_t* node_mask)
try
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_set_boundary: null argument");
  eqlb_se* se = h->se;
  const int st = eqlb_se_set_boundary(se, facet_type, nullptr, node_mask);
  if (st)
    return st;
  const eqlb::DeviceMesh& m = se->mesh->m;
  bool inhomogeneous = false;
  const size_t nb = (size_t)se->nrhs * se->ev_ndofs;
  if (boundary_values)
    for (size_t i = 0; i < nb && !inhomogeneous; ++i)
      inhomogeneous = (boundary_values[i] != 0.0);
  if (inhomogeneous)
  {
    // conforming boundary DOFs -> the broken per-cell layout the patch kernel reads
    double* d_conf = nullptr;
    if (upload(&d_conf, boundary_values, nb)
        || upload<double>(&se->bvals, nullptr, (size_t)se->nrhs * m.ncells * se->nrt))
    {
      dfree(d_conf);
      return EQLB_ERR_DEVICE;
    }
    hipError_t e = hipMemset(se->bvals, 0, sizeof(double) * (size_t)se->nrhs * m.ncells * se->nrt);
    if (e == hipSuccess)
    {
      eqlb::launch_ev_boundary_to_broken(m, se->k, se->nrhs, se->ev_cell_dofs, se->ev_ndofs, d_conf, se->bvals,
                                         (se->ev_basis && !se->ev_bv_hier) ? se->ev_basis + se->nrt * se->nrt + se->k * se->k : nullptr, nullptr);
      e = hipDeviceSynchronize();
    }
    dfree(d_conf);
    if (e != hipSuccess)
      return fail(EQLB_ERR_DEVICE, "eqlb_ev_set_boundary: %s", hipGetErrorString(e));
  }
  return EQLB_OK;
}
EQLB_CATCH_ALL

int eqlb_ev_equilibrate(eqlb_ev_t* h, const double* flux_dg, const double* rhs_dg,
                        double* flux_hdiv, int32_t memspace, void* stream)
try
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  return eqlb_se_equilibrate(h->se, flux_dg, rhs_dg, flux_hdiv, memspace, stream);
}
EQLB_CATCH_ALL

int eqlb_ev_equilibrate_lists(eqlb_ev_t* h, const double* const* flux_dg, const double* const* rhs_dg,
                              double* const* flux_hdiv, int32_t memspace, void* stream)
try
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "Equilibration: Input sizes does not match");
  return equilibrate_lists(h->se, flux_dg, rhs_dg, flux_hdiv, memspace, stream);
}
EQLB_CATCH_ALL

int64_t eqlb_ev_num_patches(const eqlb_ev_t* h) { return h ? h->se->npatch_total : 0; }

int eqlb_ev_check_status(eqlb_ev_t* h, void* stream)
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_ev_check_status: null handle");
  return eqlb_se_check_status(h->se, stream);
}

double eqlb_ev_last_kernel_ms(const eqlb_ev_t* h, int32_t which)
{
  return h ? eqlb_se_last_kernel_ms(h->se, which) : 0.0;
}

} // extern "C"
