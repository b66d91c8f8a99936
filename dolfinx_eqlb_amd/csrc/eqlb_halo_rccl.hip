// Reverse halo of the node-ownership decomposition over RCCL (the reference has no
// distributed equilibration, cpp/dolfinx_eqlb/se/reconstruction.hpp:90 loops the owned nodes and the flux of
// ghost cells is never reduced, python/dolfinx_eqlb/eqlb/FluxEqlbSE.py:164 "TODO"): the C++ host's own
// transport - grouped ncclSend / ncclRecv per neighbour on the caller's communicator and stream, between the
// two halo kernels.  xGMI is point to point: a neighbour exchange uses the direct links of the (at most a few)
// neighbour pairs, there is no ring and no collective on the data path.
//
// RCCL is NOT linked: the five entry points are resolved at run time, first among the libraries already
// loaded into the process (a caller that made its communicator with the RCCL bundled with PyTorch must get
// THAT library's ncclSend for it), then from librccl.so(.1) of the ROCm installation.
#include "eqlb_internal.h"

#include <dlfcn.h>

#include <climits>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#define fail eqlb::set_error

namespace
{
// ABI of rccl.h (ROCm 7: ncclDataType_t ncclFloat64 = 8, ncclResult_t ncclSuccess = 0, 128-byte unique id)
constexpr int NCCL_FLOAT64 = 8;
struct UniqueId
{
  char internal[128];
};
using comm_t = void*;
struct Rccl
{
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    void* h = nullptr; // RTLD_DEFAULT first: the RCCL the caller already uses
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(RTLD_DEFAULT, name);
      if (!p)
      {
        if (!h)
          h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
          h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h)
          p = dlsym(h, name);
      }
      return p;
    };
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetUniqueId && r.CommInitRank && r.CommDestroy;
  });
  return r;
}

int need_rccl(const char* who)
{
  if (!rccl().ok)
    return fail(EQLB_ERR_UNSUPPORTED, "%s: RCCL (librccl.so) is not available in this process", who);
  return EQLB_OK;
}

int nccl_fail(const char* who, const char* what, int rc)
{
  const char* msg = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
  return fail(EQLB_ERR_DEVICE, "%s: %s failed: %s (ncclResult %d)", who, what, msg, rc);
}
} // namespace

extern "C" {

int eqlb_rccl_get_unique_id(void* id128)
{
  if (!id128)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_rccl_get_unique_id: null argument");
  if (int st = need_rccl("eqlb_rccl_get_unique_id"))
    return st;
  const int rc = rccl().GetUniqueId(reinterpret_cast<UniqueId*>(id128));
  return rc ? nccl_fail("eqlb_rccl_get_unique_id", "ncclGetUniqueId", rc) : EQLB_OK;
}

int eqlb_rccl_comm_create(const void* id128, int32_t nranks, int32_t rank, void** comm)
{
  if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_rccl_comm_create: invalid argument");
  if (int st = need_rccl("eqlb_rccl_comm_create"))
    return st;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  comm_t c = nullptr;
  const int rc = rccl().CommInitRank(&c, nranks, id, rank);
  if (rc)
    return nccl_fail("eqlb_rccl_comm_create", "ncclCommInitRank", rc);
  *comm = c;
  return EQLB_OK;
}

void eqlb_rccl_comm_destroy(void* comm)
{
  if (comm && rccl().ok)
    (void)rccl().CommDestroy(comm);
}

int eqlb_halo_exchange(void* comm, int32_t npeers, const int32_t* peers, const double* const* send_buf,
                       const int64_t* send_count, double* const* recv_buf, const int64_t* recv_count,
                       void* stream_)
{
  if (!comm || npeers < 0 || (npeers > 0 && (!peers || !send_buf || !send_count || !recv_buf || !recv_count)))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_exchange: invalid argument");
  for (int32_t i = 0; i < npeers; ++i)
    if (send_count[i] < 0 || recv_count[i] < 0 || (send_count[i] > 0 && !send_buf[i])
        || (recv_count[i] > 0 && !recv_buf[i]) || peers[i] < 0)
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_exchange: invalid buffer / count of peer entry %d", i);
  if (int st = need_rccl("eqlb_halo_exchange"))
    return st;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  Rccl& r = rccl();
  // one group: all sends and receives of the step are posted together, so no pair of ranks can wait for
  // each other's matching call (a rank's receive from q and q's send to it are both inside their groups)
  int rc = r.GroupStart();
  if (rc)
    return nccl_fail("eqlb_halo_exchange", "ncclGroupStart", rc);
  int first = 0;
  const char* what = "";
  for (int32_t i = 0; i < npeers; ++i)
  {
    if (send_count[i] > 0 && !first)
    {
      first = r.Send(send_buf[i], (size_t)send_count[i], NCCL_FLOAT64, peers[i], comm, stream);
      what = "ncclSend";
    }
    if (recv_count[i] > 0 && !first)
    {
      first = r.Recv(recv_buf[i], (size_t)recv_count[i], NCCL_FLOAT64, peers[i], comm, stream);
      what = "ncclRecv";
    }
  }
  rc = r.GroupEnd(); // always closed, also after a failed post
  if (first)
    return nccl_fail("eqlb_halo_exchange", what, first);
  if (rc)
    return nccl_fail("eqlb_halo_exchange", "ncclGroupEnd", rc);
  return EQLB_OK;
}

int eqlb_halo_reduce(void* comm, int32_t nrhs, int32_t nrt, int64_t nentries, double* x, int32_t npeers,
                     const int32_t* peers, const int64_t* const* send_idx, const int64_t* nsend,
                     double* const* send_buf, const int64_t* const* recv_idx, const int64_t* nrecv,
                     double* const* recv_buf, void* stream_)
{
  if (nrhs < 1 || nrt < 1 || !x || npeers < 0
      || (npeers > 0 && (!peers || !send_idx || !nsend || !send_buf || !recv_idx || !nrecv || !recv_buf)))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_reduce: invalid argument");
  for (int32_t i = 0; i < npeers; ++i)
    if (nsend[i] < 0 || nrecv[i] < 0 || nsend[i] > INT32_MAX || nrecv[i] > INT32_MAX
        || (nsend[i] > 0 && (!send_idx[i] || !send_buf[i])) || (nrecv[i] > 0 && (!recv_idx[i] || !recv_buf[i])))
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_reduce: invalid list of peer entry %d", i);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // gather + clear of the ghost rows, one launch per peer
  for (int32_t i = 0; i < npeers; ++i)
    if (nsend[i] > 0)
      eqlb::launch_halo_pack(nrhs, (int32_t)nsend[i], nrt, nentries, send_idx[i], x, send_buf[i], 1, stream);
  if (hipGetLastError() != hipSuccess)
    return fail(EQLB_ERR_DEVICE, "eqlb_halo_reduce: pack kernel launch failed");
  std::vector<int64_t> sc((size_t)npeers), rc_((size_t)npeers);
  for (int32_t i = 0; i < npeers; ++i)
  {
    sc[i] = nsend[i] * nrhs * nrt;
    rc_[i] = nrecv[i] * nrhs * nrt;
  }
  if (int st = eqlb_halo_exchange(comm, npeers, peers, send_buf, sc.data(), recv_buf, rc_.data(), stream_))
    return st;
  for (int32_t i = 0; i < npeers; ++i)
    if (nrecv[i] > 0)
      eqlb::launch_halo_unpack_add(nrhs, (int32_t)nrecv[i], nrt, nentries, recv_idx[i], x, recv_buf[i], stream);
  if (hipGetLastError() != hipSuccess)
    return fail(EQLB_ERR_DEVICE, "eqlb_halo_reduce: unpack kernel launch failed");
  return EQLB_OK;
}

// ---- halo plan: the lists on the device and the staging buffers, owned by the library -----------------
struct eqlb_halo
{
  int32_t nrhs = 0, nrt = 0, npeers = 0;
  int64_t nentries = 0;
  std::vector<int32_t> peers;
  std::vector<int64_t> nsend, nrecv;
  std::vector<int64_t*> send_idx, recv_idx; // device
  std::vector<double*> send_buf, recv_buf;  // device
};

int eqlb_halo_create(int32_t nrhs, int32_t nrt, int64_t nentries, int32_t npeers, const int32_t* peers,
                     const int64_t* const* send_idx, const int64_t* nsend, const int64_t* const* recv_idx,
                     const int64_t* nrecv, eqlb_halo_t** handle)
try
{
  if (!handle || nrhs < 1 || nrt < 1 || nentries < 0 || npeers < 0
      || (npeers > 0 && (!peers || !send_idx || !nsend || !recv_idx || !nrecv)))
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_create: invalid argument");
  for (int32_t i = 0; i < npeers; ++i)
  {
    if (nsend[i] < 0 || nrecv[i] < 0 || nsend[i] > INT32_MAX || nrecv[i] > INT32_MAX || (nsend[i] > 0 && !send_idx[i])
        || (nrecv[i] > 0 && !recv_idx[i]))
      return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_create: invalid list of peer entry %d", i);
    for (int64_t j = 0; j < nsend[i]; ++j)
      if (send_idx[i][j] < 0 || send_idx[i][j] >= nentries)
        return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_create: send index %lld of peer entry %d out of range",
                    (long long)send_idx[i][j], i);
    for (int64_t j = 0; j < nrecv[i]; ++j)
      if (recv_idx[i][j] < 0 || recv_idx[i][j] >= nentries)
        return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_create: receive index %lld of peer entry %d out of range",
                    (long long)recv_idx[i][j], i);
  }
  std::unique_ptr<eqlb_halo, void (*)(eqlb_halo*)> h(new eqlb_halo(), [](eqlb_halo* p) { eqlb_halo_destroy(p); });
  h->nrhs = nrhs;
  h->nrt = nrt;
  h->nentries = nentries;
  h->npeers = npeers;
  h->peers.assign(peers, peers + npeers);
  h->nsend.assign(nsend, nsend + npeers);
  h->nrecv.assign(nrecv, nrecv + npeers);
  h->send_idx.assign(npeers, nullptr);
  h->recv_idx.assign(npeers, nullptr);
  h->send_buf.assign(npeers, nullptr);
  h->recv_buf.assign(npeers, nullptr);
  auto dev = [](void** p, const void* src, size_t bytes) {
    if (bytes == 0)
      return true;
    if (hipMalloc(p, bytes) != hipSuccess)
      return false;
    return !src || hipMemcpy(*p, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  for (int32_t i = 0; i < npeers; ++i)
  {
    const size_t bs = (size_t)nsend[i] * nrhs * nrt * sizeof(double), br = (size_t)nrecv[i] * nrhs * nrt * sizeof(double);
    if (!dev(reinterpret_cast<void**>(&h->send_idx[i]), send_idx[i], (size_t)nsend[i] * sizeof(int64_t))
        || !dev(reinterpret_cast<void**>(&h->recv_idx[i]), recv_idx[i], (size_t)nrecv[i] * sizeof(int64_t))
        || !dev(reinterpret_cast<void**>(&h->send_buf[i]), nullptr, bs)
        || !dev(reinterpret_cast<void**>(&h->recv_buf[i]), nullptr, br))
      return fail(EQLB_ERR_DEVICE, "eqlb_halo_create: device allocation failed");
  }
  *handle = h.release();
  return EQLB_OK;
}
catch (const std::bad_alloc&)
{
  return fail(EQLB_ERR_NO_MEMORY, "host memory exhausted");
}

void eqlb_halo_destroy(eqlb_halo_t* h)
{
  if (!h)
    return;
  for (auto* p : h->send_idx)
    if (p)
      (void)hipFree(p);
  for (auto* p : h->recv_idx)
    if (p)
      (void)hipFree(p);
  for (auto* p : h->send_buf)
    if (p)
      (void)hipFree(p);
  for (auto* p : h->recv_buf)
    if (p)
      (void)hipFree(p);
  delete h;
}

int eqlb_halo_bytes(const eqlb_halo_t* h, int64_t* bytes_sent, int64_t* bytes_received)
{
  if (!h)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_bytes: null handle");
  int64_t s = 0, r = 0;
  for (int32_t i = 0; i < h->npeers; ++i)
  {
    s += h->nsend[i];
    r += h->nrecv[i];
  }
  if (bytes_sent)
    *bytes_sent = s * h->nrhs * h->nrt * (int64_t)sizeof(double);
  if (bytes_received)
    *bytes_received = r * h->nrhs * h->nrt * (int64_t)sizeof(double);
  return EQLB_OK;
}

int eqlb_halo_reduce_plan(eqlb_halo_t* h, void* comm, double* x, void* stream)
{
  if (!h || !x)
    return fail(EQLB_ERR_INVALID_ARGUMENT, "eqlb_halo_reduce_plan: null argument");
  return eqlb_halo_reduce(comm, h->nrhs, h->nrt, h->nentries, x, h->npeers, h->peers.data(), h->send_idx.data(),
                          h->nsend.data(), h->send_buf.data(), h->recv_idx.data(), h->nrecv.data(),
                          h->recv_buf.data(), stream);
}

} // extern "C"
