// cutfemx_amd: multi-GPU exchange steps of the hot path behind the C ABI.
//
// The reference's collectives on this path are neighbour exchanges of DOLFINx index maps:
//   phi.x.scatter_forward()                      python/demo/demo_poisson.py:157      owner -> ghost copy
//   A.scatter_reverse(); b.scatter_reverse(add)  python/demo/demo_poisson.py:51-54    ghost -> owner add
//   indicator.scatter_rev(plus) + scatter_fwd    cpp/cutfemx/fem/deactivate.h:180-181 ghost -> owner OR, then copy back
// One rank per GPU; a rank talks to the few ranks it shares dofs with (two for z-slabs).  Every exchange is
// one grouped RCCL send/recv per peer over xGMI (point-to-point links: no ring, no all-reduce), with the
// contiguous case (vertex planes of a slab) sent straight from / received next to the array it belongs to,
// and the general case (index lists of an index map) packed / unpacked by a kernel.
//
// Transport: RCCL (librccl.so.1, resolved at run time with dlopen: the engine itself has no link-time
// dependency on it and single-GPU use never loads it), or a caller-supplied host callback (MPI, gloo): the
// library stages the slices through pinned host memory and the callback moves the bytes.  The host mode is
// what the two-process tests drive on a one-GPU box, where RCCL cannot place two ranks on one device.
#include <dlfcn.h>

#include <cstring>
#include <vector>

#include "cfx_device.h"

using namespace cfx;

namespace
{

// ---- RCCL entry points (rccl.h), resolved by name
typedef void* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0 };

struct Rccl
{
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl& rccl()
{
  static Rccl r;
  if (r.handle) return r;
  // resolved into a local table and published (handle included) only when every symbol was found: a failed attempt
  // leaves `r` empty, so the next call tries again and reports the same error instead of handing out null pointers
  Rccl t;
  // a process that already holds RCCL (PyTorch-ROCm bundles one under the same soname) keeps its copy
  for (const char* name : {"librccl.so.1", "librccl.so"})
  {
    t.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (!t.handle) t.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (t.handle) break;
  }
  if (!t.handle) throw Error(CFX_ERR_RUNTIME, std::string("cfx_dist: cannot load librccl.so.1: ") + dlerror());
  auto sym = [&](const char* n) {
    void* p = dlsym(t.handle, n);
    if (!p) throw Error(CFX_ERR_RUNTIME, std::string("cfx_dist: RCCL symbol missing: ") + n);
    return p;
  };
  t.GetUniqueId = reinterpret_cast<decltype(t.GetUniqueId)>(sym("ncclGetUniqueId"));
  t.CommInitRank = reinterpret_cast<decltype(t.CommInitRank)>(sym("ncclCommInitRank"));
  t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(sym("ncclCommDestroy"));
  t.Send = reinterpret_cast<decltype(t.Send)>(sym("ncclSend"));
  t.Recv = reinterpret_cast<decltype(t.Recv)>(sym("ncclRecv"));
  t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(sym("ncclGroupStart"));
  t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(sym("ncclGroupEnd"));
  t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(sym("ncclGetErrorString"));
  r = t;
  return r;
}

void nccl_check(int rc, const char* what)
{
  if (rc != ncclSuccess)
    throw Error(CFX_ERR_RUNTIME, std::string("cfx_dist: ") + what + ": " + rccl().GetErrorString(rc));
}

} // namespace

// page-locked staging buffer of the host-staged transport: grows on demand, lives with its communicator
struct PinnedBuf
{
  void* p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  PinnedBuf(PinnedBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  void* reserve(size_t bytes)
  {
    if (bytes <= cap) return p;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    CFX_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return p;
  }
};

struct cfx_comm_s
{
  int world = 1, rank = 0;
  bool is_rccl = false;
  ncclComm_t comm = nullptr;
  cfx_host_exchange_fn host_fn = nullptr;
  void* host_user = nullptr;
  cfx_device_exchange_fn device_fn = nullptr; // the caller's own GPU-to-GPU transport (cfx_dist_comm_create_device)
  void* device_user = nullptr;
  std::vector<PinnedBuf> stage_send, stage_recv; // per segment slot of an exchange, reused by every later exchange
};

namespace
{

// ---- pack / unpack / combine kernels (elements of ES bytes moved as such; ADD on f64, OR on i8)
template <typename T>
__global__ void __launch_bounds__(kBlock) dist_pack_kernel(int64_t n, const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                           T* __restrict__ dst)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// op: 0 copy, 1 add, 2 or.  idx == nullptr: dst is the contiguous destination range itself
template <typename T>
__global__ void __launch_bounds__(kBlock) dist_combine_kernel(int64_t n, const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                              T* __restrict__ dst, int op)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  T* d = dst + (idx ? (int64_t)idx[i] : i);
  if (op == 0) *d = src[i];
  else if (op == 1) *d = *d + src[i]; // (index lists of one exchange are duplicate-free: one writer per entry)
  else *d = (T)((*d != 0 || src[i] != 0) ? 1 : 0);
}

struct Segment
{
  int peer;
  const void* send; // device
  int64_t send_bytes;
  void* recv;       // device
  int64_t recv_bytes;
};

// all sends and receives of one exchange step, posted together
void transport(cfx_comm_s* c, std::vector<Segment>& seg)
{
  Context& cx = ctx();
  if (c->is_rccl)
  {
    Rccl& r = rccl();
    nccl_check(r.GroupStart(), "ncclGroupStart");
    for (const Segment& s : seg)
    {
      if (s.send_bytes > 0) nccl_check(r.Send(s.send, (size_t)s.send_bytes, ncclInt8, s.peer, c->comm, cx.stream), "ncclSend");
      if (s.recv_bytes > 0) nccl_check(r.Recv(s.recv, (size_t)s.recv_bytes, ncclInt8, s.peer, c->comm, cx.stream), "ncclRecv");
    }
    nccl_check(r.GroupEnd(), "ncclGroupEnd");
    return;
  }
  if (c->device_fn)
  {
    // the caller's GPU-to-GPU transport (a GPU-aware MPI, the launcher's own RCCL process group): it is handed the device
    // segments once everything that fills them has run; whatever it enqueues must have completed when it returns
    const int n = (int)seg.size();
    std::vector<int32_t> peers(n);
    std::vector<const void*> sp(n, nullptr);
    std::vector<void*> rp(n, nullptr);
    std::vector<int64_t> sb(n), rb(n);
    for (int i = 0; i < n; ++i)
    {
      peers[i] = seg[i].peer; sb[i] = seg[i].send_bytes; rb[i] = seg[i].recv_bytes;
      sp[i] = sb[i] > 0 ? seg[i].send : nullptr; rp[i] = rb[i] > 0 ? seg[i].recv : nullptr;
    }
    CFX_HIP(hipStreamSynchronize(cx.stream));
    const int rc = c->device_fn(c->device_user, n, peers.data(), sp.data(), sb.data(), rp.data(), rb.data());
    if (rc != 0) throw Error(CFX_ERR_RUNTIME, "cfx_dist: the device exchange callback failed");
    return;
  }
  // host-staged: device -> pinned host, the caller's callback (MPI / gloo), pinned host -> device.  The staging
  // buffers belong to the communicator (RAII: an exception below leaks nothing, and a step allocates nothing new)
  const int n = (int)seg.size();
  if ((int)c->stage_send.size() < n) { c->stage_send.resize(n); c->stage_recv.resize(n); }
  std::vector<void*> hr(n, nullptr);
  std::vector<const void*> sp(n, nullptr);
  std::vector<int32_t> peers(n);
  std::vector<int64_t> sb(n), rb(n);
  for (int i = 0; i < n; ++i)
  {
    peers[i] = seg[i].peer; sb[i] = seg[i].send_bytes; rb[i] = seg[i].recv_bytes;
    if (sb[i] > 0)
    {
      void* h = c->stage_send[i].reserve((size_t)sb[i]);
      CFX_HIP(hipMemcpyAsync(h, seg[i].send, (size_t)sb[i], hipMemcpyDeviceToHost, cx.stream));
      sp[i] = h;
    }
    if (rb[i] > 0) hr[i] = c->stage_recv[i].reserve((size_t)rb[i]);
  }
  CFX_HIP(hipStreamSynchronize(cx.stream));
  const int rc = c->host_fn(c->host_user, n, peers.data(), sp.data(), sb.data(), hr.data(), rb.data());
  if (rc != 0) throw Error(CFX_ERR_RUNTIME, "cfx_dist: the host exchange callback failed");
  for (int i = 0; i < n; ++i)
    if (rb[i] > 0) CFX_HIP(hipMemcpyAsync(seg[i].recv, hr[i], (size_t)rb[i], hipMemcpyHostToDevice, cx.stream));
  CFX_HIP(hipStreamSynchronize(cx.stream)); // the buffers are reused by the next exchange
}

// One exchange step on an array of T: send my [send] parts, combine what arrives into my [recv] parts.
template <typename T>
void exchange_apply(cfx_comm_s* c, T* data, int n, const cfx_dist_exchange* ex, int op)
{
  require(c != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_dist: null communicator");
  require(n == 0 || (ex != nullptr && data != nullptr), CFX_ERR_INVALID_ARGUMENT, "cfx_dist: null argument");
  require(is_device_pointer(data), CFX_ERR_INVALID_ARGUMENT, "cfx_dist: the array must live in HBM");
  std::vector<Segment> seg;
  std::vector<DevArray<T>> sendbuf(n), recvbuf(n);
  for (int i = 0; i < n; ++i)
  {
    const cfx_dist_exchange& e = ex[i];
    require(e.peer >= 0 && e.peer < c->world && e.peer != c->rank, CFX_ERR_INVALID_ARGUMENT, "cfx_dist: bad peer rank");
    require(e.send_count >= 0 && e.recv_count >= 0, CFX_ERR_INVALID_ARGUMENT, "cfx_dist: negative count");
    Segment s{};
    s.peer = e.peer;
    s.send_bytes = e.send_count * (int64_t)sizeof(T);
    s.recv_bytes = e.recv_count * (int64_t)sizeof(T);
    if (e.send_index && e.send_count > 0)
    {
      require(is_device_pointer(e.send_index), CFX_ERR_INVALID_ARGUMENT, "cfx_dist: index lists must live in HBM");
      sendbuf[i].alloc(e.send_count);
      launch("dist_pack", dist_pack_kernel<T>, grid_for(e.send_count), dim3(kBlock), 0, e.send_count, (const T*)data,
             e.send_index, sendbuf[i].p);
      s.send = sendbuf[i].p;
    }
    else
      s.send = data + e.send_offset;
    // a contiguous copy lands in place; everything else (add / or / index list) goes through a receive buffer
    if (op == 0 && !e.recv_index) s.recv = data + e.recv_offset;
    else if (e.recv_count > 0)
    {
      if (e.recv_index) require(is_device_pointer(e.recv_index), CFX_ERR_INVALID_ARGUMENT, "cfx_dist: index lists must live in HBM");
      recvbuf[i].alloc(e.recv_count);
      s.recv = recvbuf[i].p;
    }
    seg.push_back(s);
  }
  transport(c, seg);
  for (int i = 0; i < n; ++i)
  {
    const cfx_dist_exchange& e = ex[i];
    if (e.recv_count == 0 || (op == 0 && !e.recv_index)) continue;
    launch("dist_combine", dist_combine_kernel<T>, grid_for(e.recv_count), dim3(kBlock), 0, e.recv_count,
           (const T*)recvbuf[i].p, e.recv_index, e.recv_index ? data : data + e.recv_offset, op);
  }
}

} // namespace

extern "C" {

int cfx_dist_unique_id(char id[CFX_DIST_ID_BYTES])
{
  CFX_API_BEGIN
  require(id != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_dist_unique_id: null argument");
  static_assert(CFX_DIST_ID_BYTES == sizeof(ncclUniqueId), "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  nccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
  std::memcpy(id, u.internal, sizeof(u.internal));
  CFX_API_END
}

int cfx_dist_comm_create(int world, int rank, const char id[CFX_DIST_ID_BYTES], cfx_comm_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out && id && world >= 1 && rank >= 0 && rank < world, CFX_ERR_INVALID_ARGUMENT, "cfx_dist_comm_create: bad argument");
  auto c = std::make_unique<cfx_comm_s>();
  c->world = world; c->rank = rank; c->is_rccl = true;
  ncclUniqueId u;
  std::memcpy(u.internal, id, sizeof(u.internal));
  nccl_check(rccl().CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");
  *out = c.release();
  CFX_API_END
}

int cfx_dist_comm_create_host(int world, int rank, cfx_host_exchange_fn fn, void* user, cfx_comm_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out && fn && world >= 1 && rank >= 0 && rank < world, CFX_ERR_INVALID_ARGUMENT, "cfx_dist_comm_create_host: bad argument");
  auto c = std::make_unique<cfx_comm_s>();
  c->world = world; c->rank = rank; c->is_rccl = false; c->host_fn = fn; c->host_user = user;
  *out = c.release();
  CFX_API_END
}

int cfx_dist_comm_create_device(int world, int rank, cfx_device_exchange_fn fn, void* user, cfx_comm_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out && fn && world >= 1 && rank >= 0 && rank < world, CFX_ERR_INVALID_ARGUMENT, "cfx_dist_comm_create_device: bad argument");
  auto c = std::make_unique<cfx_comm_s>();
  c->world = world; c->rank = rank; c->is_rccl = false; c->device_fn = fn; c->device_user = user;
  *out = c.release();
  CFX_API_END
}

int cfx_dist_comm_info(cfx_comm_t c, int* world, int* rank, int* is_rccl)
{
  CFX_API_BEGIN
  require(c != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_dist_comm_info: null communicator");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (is_rccl) *is_rccl = c->is_rccl ? 1 : 0;
  CFX_API_END
}

int cfx_dist_comm_destroy(cfx_comm_t c)
{
  CFX_API_BEGIN
  if (c)
  {
    if (c->is_rccl && c->comm)
    {
      (void)hipStreamSynchronize(ctx().stream);
      (void)rccl().CommDestroy(c->comm);
    }
    delete c;
  }
  CFX_API_END
}

int cfx_dist_scatter_forward(cfx_comm_t c, double* x, int n, const cfx_dist_exchange* ex)
{
  CFX_API_BEGIN
  exchange_apply<double>(c, x, n, ex, 0);
  CFX_API_END
}

int cfx_dist_scatter_reverse_add(cfx_comm_t c, double* x, int n, const cfx_dist_exchange* ex)
{
  CFX_API_BEGIN
  exchange_apply<double>(c, x, n, ex, 1);
  CFX_API_END
}

int cfx_dist_scatter_reverse_matrix(cfx_comm_t c, cfx_pattern_t P, double* values, int n, const cfx_dist_row_exchange* ex)
{
  CFX_API_BEGIN
  require(c && P && (n == 0 || (ex && values)), CFX_ERR_INVALID_ARGUMENT, "cfx_dist_scatter_reverse_matrix: null argument");
  // rows -> ranges of the CSR value array: 4 row pointers per exchange, read back in one copy
  std::vector<int64_t> rows;
  for (int i = 0; i < n; ++i)
  {
    require(ex[i].send_row_lo >= 0 && ex[i].send_row_lo <= ex[i].send_row_hi && ex[i].send_row_hi <= P->nrows
                && ex[i].recv_row_lo >= 0 && ex[i].recv_row_lo <= ex[i].recv_row_hi && ex[i].recv_row_hi <= P->nrows,
            CFX_ERR_OUT_OF_RANGE, "cfx_dist_scatter_reverse_matrix: row range outside the matrix");
    rows.insert(rows.end(), {ex[i].send_row_lo, ex[i].send_row_hi, ex[i].recv_row_lo, ex[i].recv_row_hi});
  }
  std::vector<int64_t> ptr(rows.size());
  for (size_t k = 0; k < rows.size(); ++k)
    CFX_HIP(hipMemcpyAsync(&ptr[k], P->indptr.p + rows[k], sizeof(int64_t), hipMemcpyDeviceToHost, ctx().stream));
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  std::vector<cfx_dist_exchange> e(n);
  for (int i = 0; i < n; ++i)
  {
    e[i] = cfx_dist_exchange{};
    e[i].peer = ex[i].peer;
    e[i].send_offset = ptr[4 * i]; e[i].send_count = ptr[4 * i + 1] - ptr[4 * i];
    e[i].recv_offset = ptr[4 * i + 2]; e[i].recv_count = ptr[4 * i + 3] - ptr[4 * i + 2];
  }
  // both sides built the rows of a shared plane from the same entities (DOLFINx keeps the ghost rows' sparsity
  // on both ranks), so the slices must have equal lengths.  Each side sends BOTH of its counts and checks both
  // directions against the peer's pair: a mismatch in either direction is seen by both ranks, which then raise
  // together -- a one-sided check would let the other rank go on into a value exchange nobody answers
  {
    DevArray<double> cnt(4 * (int64_t)std::max(n, 1));
    std::vector<double> h(4 * (size_t)std::max(n, 1), 0.0);
    for (int i = 0; i < n; ++i) { h[4 * i] = (double)e[i].send_count; h[4 * i + 1] = (double)e[i].recv_count; }
    CFX_HIP(hipMemcpyAsync(cnt.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx().stream));
    std::vector<cfx_dist_exchange> ce(n);
    for (int i = 0; i < n; ++i)
    {
      ce[i] = cfx_dist_exchange{};
      ce[i].peer = ex[i].peer; ce[i].send_offset = 4 * i; ce[i].send_count = 2; ce[i].recv_offset = 4 * i + 2; ce[i].recv_count = 2;
    }
    exchange_apply<double>(c, cnt.p, n, ce.data(), 0);
    CFX_HIP(hipMemcpyAsync(h.data(), cnt.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx().stream));
    CFX_HIP(hipStreamSynchronize(ctx().stream));
    for (int i = 0; i < n; ++i)
      require((int64_t)h[4 * i + 2] == e[i].recv_count && (int64_t)h[4 * i + 3] == e[i].send_count, CFX_ERR_RUNTIME,
              "cfx_dist_scatter_reverse_matrix: the peer's rows hold a different number of entries (sparsity of the "
              "shared rows must be built from the same entities on both ranks)");
  }
  exchange_apply<double>(c, values, n, e.data(), 1);
  CFX_API_END
}

int cfx_dist_indicator_or(cfx_comm_t c, int8_t* indicator, int n, const cfx_dist_exchange* ex)
{
  CFX_API_BEGIN
  exchange_apply<int8_t>(c, indicator, n, ex, 2);
  CFX_API_END
}

int cfx_dist_indicator_forward(cfx_comm_t c, int8_t* indicator, int n, const cfx_dist_exchange* ex)
{
  CFX_API_BEGIN
  exchange_apply<int8_t>(c, indicator, n, ex, 0);
  CFX_API_END
}

} // extern "C"
