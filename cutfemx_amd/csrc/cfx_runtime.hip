// cutfemx_amd: runtime (device/stream/profile), scans, incidence inversion.
#include "cfx_device.h"

namespace cfx
{

thread_local std::string g_last_error;

Context& ctx()
{
  static Context c;
  return c;
}

void Context::ensure()
{
  if (initialised) return;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error(CFX_ERR_HIP, "cutfemx_amd: no HIP device available (this engine has no CPU fallback)");
  if (device < 0) device = 0;
  if (device >= n) throw Error(CFX_ERR_HIP, "cutfemx_amd: device index out of range");
  CFX_HIP(hipSetDevice(device));
  initialised = true;
}

hipEvent_t Context::get_event()
{
  if (!event_pool.empty())
  {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  CFX_HIP(hipEventCreate(&e));
  return e;
}

void publish_across_lanes()
{
  Context& c = ctx();
  if (!c.overlap || !c.side_stream) return;
  hipStream_t other = c.stream == c.main_stream ? c.side_stream : c.main_stream;
  hipEvent_t e = c.get_event();
  CFX_HIP(hipEventRecord(e, c.stream));
  CFX_HIP(hipStreamWaitEvent(other, e, 0));
  c.event_pool.push_back(e);
}

int Context::entry(const char* name)
{
  auto it = entry_index.find(name);
  if (it != entry_index.end()) return it->second;
  entries.push_back({name, 0.0, 0});
  entry_index[name] = (int)entries.size() - 1;
  return (int)entries.size() - 1;
}

void Context::flush_profile()
{
  if (pending.empty()) return;
  CFX_HIP(hipStreamSynchronize(main_stream));
  if (side_stream) CFX_HIP(hipStreamSynchronize(side_stream));
  for (auto& p : pending)
  {
    float ms = 0.f;
    CFX_HIP(hipEventElapsedTime(&ms, p.a, p.b));
    entries[p.entry].total_ms += ms;
    entries[p.entry].launches += 1;
    event_pool.push_back(p.a);
    event_pool.push_back(p.b);
  }
  pending.clear();
}

// ---------------------------------------------------------------------------
// HBM block cache (see cfx_common.h)
// ---------------------------------------------------------------------------
namespace
{
struct LiveBlock { size_t size; uint64_t serial; };
struct BlockCache
{
  std::multimap<size_t, void*> free_blocks; // size -> block
  std::map<void*, LiveBlock> live;          // block -> size, serial of this hand-out
  size_t cached = 0, in_use = 0, peak = 0;  // bytes: cached, handed out, high-water of (in_use + cached)
  void hand_out(void* p, size_t size)
  {
    live[p] = LiveBlock{size, next_serial()};
    in_use += size;
    peak = std::max(peak, in_use + cached);
  }
};
BlockCache& cache()
{
  static BlockCache c;
  return c;
}
size_t round_size(size_t bytes)
{
  const size_t g = bytes < (1u << 20) ? 4096 : (size_t)2 << 20; // 4 KiB / 2 MiB granules
  return (bytes + g - 1) / g * g;
}
} // namespace

void* pinned_scratch()
{
  static void* p = []() {
    void* q = nullptr;
    CFX_HIP(hipHostMalloc(&q, 256, hipHostMallocDefault));
    return q;
  }();
  return p;
}

int64_t& sync_counter()
{
  static int64_t n = 0;
  static const bool report = []() {
    const char* e = getenv("CFX_COUNT_SYNC");
    if (e && e[0] == '2') ctx().trace_sync = true;
    if (e && (e[0] == '1' || e[0] == '2')) atexit([]() { fprintf(stderr, "cutfemx_amd: %lld size read-backs\n", (long long)sync_counter()); });
    return true;
  }();
  (void)report;
  return n;
}

void* dev_alloc(size_t bytes)
{
  BlockCache& c = cache();
  const size_t want = round_size(bytes > 0 ? bytes : 1);
  // best fit that does not waste more than half of the block
  auto it = c.free_blocks.lower_bound(want);
  if (it != c.free_blocks.end() && it->first <= want + want / 2 + ((size_t)4 << 20))
  {
    void* p = it->second;
    const size_t size = it->first;
    c.cached -= size;
    c.free_blocks.erase(it);
    c.hand_out(p, size);
    return p;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess)
  {
    (void)hipGetLastError();
    dev_cache_release(); // out of memory: give the cached blocks back and retry once
    e = hipMalloc(&p, want);
    if (e != hipSuccess) throw Error(CFX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  c.hand_out(p, want);
  return p;
}

void dev_free(void* p)
{
  if (!p) return;
  // two streams in flight: a block handed back now could reach a launch on the other stream while queued work of
  // this one still uses it -- it stays out of the cache until the section is joined
  if (ctx().overlap) { ctx().deferred_free.push_back(p); return; }
  BlockCache& c = cache();
  auto it = c.live.find(p);
  if (it == c.live.end()) return; // not ours
  c.free_blocks.emplace(it->second.size, p);
  c.cached += it->second.size;
  c.in_use -= it->second.size;
  c.live.erase(it);
}

uint64_t next_serial()
{
  static uint64_t n = 0;
  return ++n;
}

uint64_t dev_block_serial(const void* p)
{
  BlockCache& c = cache();
  auto it = c.live.find(const_cast<void*>(p));
  return it == c.live.end() ? 0 : it->second.serial;
}

namespace
{
std::map<uint64_t, bool>& live_rules()
{
  static std::map<uint64_t, bool> m;
  return m;
}
} // namespace
void rules_serial_live(uint64_t s, bool live)
{
  if (live) live_rules()[s] = true; else live_rules().erase(s);
}
bool rules_serial_is_live(uint64_t s) { return live_rules().count(s) != 0; }

void device_memory_stats(size_t& live, size_t& cached, size_t& peak)
{
  BlockCache& c = cache();
  live = c.in_use; cached = c.cached; peak = c.peak;
}
void device_memory_reset_peak()
{
  BlockCache& c = cache();
  c.peak = c.in_use + c.cached;
}

void dev_cache_release()
{
  BlockCache& c = cache();
  if (c.free_blocks.empty()) return;
  (void)hipStreamSynchronize(ctx().main_stream); // a cached block may still be in use by queued work
  if (ctx().side_stream) (void)hipStreamSynchronize(ctx().side_stream);
  for (auto& kv : c.free_blocks) (void)hipFree(kv.second);
  c.free_blocks.clear();
  c.cached = 0;
}

bool is_device_pointer(const void* p)
{
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess)
  {
    (void)hipGetLastError(); // plain host memory: clear the sticky error
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// ---------------------------------------------------------------------------
// exclusive scan: tile reduce -> (recursive) scan of tile sums -> tile scan
// ---------------------------------------------------------------------------
template <typename Tin, typename Tout>
__global__ void __launch_bounds__(kBlock) scan_reduce_kernel(const Tin* __restrict__ in, int64_t n, Tout* tile_sums)
{
  const int64_t tile = (int64_t)blockIdx.x * kTile; // the order of a sum is free: coalesced, block-strided loads
  Tout s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = tile + k * kBlock + threadIdx.x;
    if (i < n) s += (Tout)in[i];
  }
  Tout total;
  (void)block_exclusive_scan<Tout>(s, total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

template <typename Tin, typename Tout>
__global__ void __launch_bounds__(kBlock) scan_write_kernel(const Tin* __restrict__ in, int64_t n,
                                                            const Tout* __restrict__ tile_offsets, Tout* out)
{
  // a thread scans kScanItems consecutive items; the tile goes through LDS both ways so that the global loads
  // and stores are coalesced (lane i touches element i of a 256-element row, not its own 64 B run)
  __shared__ Tout s_v[kTile];
  const int64_t tile = (int64_t)blockIdx.x * kTile;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int i = k * kBlock + threadIdx.x;
    s_v[i] = (tile + i < n) ? (Tout)in[tile + i] : (Tout)0;
  }
  __syncthreads();
  Tout v[kScanItems];
  Tout s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    v[k] = s_v[threadIdx.x * kScanItems + k];
    s += v[k];
  }
  Tout total;
  Tout off = block_exclusive_scan<Tout>(s, total) + tile_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    s_v[threadIdx.x * kScanItems + k] = off;
    off += v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int i = k * kBlock + threadIdx.x;
    if (tile + i < n) out[tile + i] = s_v[i];
  }
  // the element one past the end receives the grand total
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) out[n] = tile_offsets[blockIdx.x] + total;
}

template <typename Tin, typename Tout>
__global__ void scan_small_kernel(const Tin* in, int64_t n, Tout* out)
{
  // single thread block, n <= kTile: used for the top of the recursion
  const int64_t base = (int64_t)threadIdx.x * kScanItems;
  Tout v[kScanItems];
  Tout s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    v[k] = (base + k < n) ? (Tout)in[base + k] : (Tout)0;
    s += v[k];
  }
  Tout total;
  Tout off = block_exclusive_scan<Tout>(s, total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    if (base + k < n) out[base + k] = off;
    off += v[k];
  }
  if (threadIdx.x == 0) out[n] = total;
}

// Single-pass scan (chained tiles with wave-wide look-back): one launch, the input is read
// once.  A tile takes its number from an atomic ticket when its block starts, so every
// predecessor it waits for is already resident and makes progress.  Tile state = one 64-bit
// word (2-bit flag | 62-bit value) exchanged with agent-scope atomics: the L2s of the eight
// XCDs are not coherent with each other for plain loads.
constexpr unsigned long long kScanAggregate = 1ull << 62, kScanPrefix = 2ull << 62, kScanValueMask = (1ull << 62) - 1ull;

template <typename Tin, typename Tout>
__global__ void __launch_bounds__(kBlock) scan_chained_kernel(const Tin* __restrict__ in, int64_t n,
                                                              unsigned long long* __restrict__ state,
                                                              unsigned int* __restrict__ ticket, Tout* __restrict__ out)
{
  __shared__ unsigned int s_tile;
  __shared__ unsigned long long s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const unsigned int tile = s_tile;
  // the tile goes through LDS both ways: coalesced global loads and stores (see scan_write_kernel)
  __shared__ Tout s_v[kTile];
  const int64_t tbase = (int64_t)tile * kTile;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int i = k * kBlock + threadIdx.x;
    s_v[i] = (tbase + i < n) ? (Tout)in[tbase + i] : (Tout)0;
  }
  __syncthreads();
  Tout v[kScanItems];
  Tout s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    v[k] = s_v[threadIdx.x * kScanItems + k];
    s += v[k];
  }
  Tout total;
  const Tout local = block_exclusive_scan<Tout>(s, total);
  if (threadIdx.x < 64)
  {
    const int lane = threadIdx.x;
    unsigned long long prefix = 0;
    if (tile == 0)
    {
      if (lane == 0)
        __hip_atomic_store(&state[0], kScanPrefix | ((unsigned long long)total & kScanValueMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
      if (lane == 0)
        __hip_atomic_store(&state[tile], kScanAggregate | ((unsigned long long)total & kScanValueMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      // look back 64 tiles at a time until a tile with a complete prefix is found
      int64_t hi = (int64_t)tile - 1;
      while (true)
      {
        const int64_t p = hi - lane;
        unsigned long long w = kScanPrefix; // tiles before the first: prefix 0
        if (p >= 0)
        {
          do
          {
            w = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((w >> 62) == 0ull);
        }
        const unsigned long long done = __ballot((w >> 62) == 2ull);
        const int first = done ? __ffsll((long long)done) - 1 : 64; // nearest tile whose prefix is complete
        unsigned long long part = (lane <= first) ? (w & kScanValueMask) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        prefix += part;
        if (done) break;
        hi -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&state[tile], kScanPrefix | ((prefix + (unsigned long long)total) & kScanValueMask),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  Tout off = local + (Tout)s_prefix;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    s_v[threadIdx.x * kScanItems + k] = off;
    off += v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int i = k * kBlock + threadIdx.x;
    if (tbase + i < n) out[tbase + i] = s_v[i];
  }
  // the element one past the end receives the grand total
  if ((int64_t)(tile + 1) * kTile >= n && threadIdx.x == kBlock - 1) out[n] = (Tout)s_prefix + total;
}

template <typename Tin, typename Tout>
static void scan_impl(const Tin* in, Tout* out, int64_t n)
{
  if (n == 0)
  {
    cfx::dev_fill(out, 0, sizeof(Tout));
    return;
  }
  const int64_t ntiles = (n + kTile - 1) / kTile;
  // chained up to 16 M elements (one launch instead of three to five: what matters for the many
  // short scans of a step and for the slabs of a multi-GPU run); longer arrays keep the three-kernel
  // form, whose tiles never wait on each other (measured at 135 M elements: 0.60 vs 0.75 ms)
  static const int64_t chained_max = getenv("CFX_SCAN_CHAINED_TILES") ? atoll(getenv("CFX_SCAN_CHAINED_TILES")) : 8192;
  if (ntiles > 1 && ntiles <= chained_max)
  {
    // tile states + ticket: a zeroed slice of a pool that is cleared with one fill when it wraps (a step runs
    // a dozen short scans: one memset each otherwise)
    constexpr int64_t kPoolWords = 1 << 20;
    // (one pool per stream: the refill at wrap-around is ordered against earlier users by the stream it runs on)
    struct ScanPool { unsigned long long* pool = nullptr; int64_t next = kPoolWords; };
    static std::map<hipStream_t, ScanPool> pools;
    ScanPool& sp = pools[ctx().stream];
    unsigned long long*& pool = sp.pool;
    int64_t& next = sp.next;
    unsigned long long* state;
    DevArray<unsigned long long> own;
    if (ntiles + 1 > kPoolWords / 4)
    {
      own.alloc(ntiles + 1);
      own.zero();
      state = own.p;
    }
    else
    {
      if (!pool) pool = static_cast<unsigned long long*>(dev_alloc(sizeof(unsigned long long) * kPoolWords));
      if (next + ntiles + 1 > kPoolWords)
      {
        cfx::dev_fill(pool, 0, sizeof(unsigned long long) * kPoolWords); // stream order keeps earlier users ahead of it
        next = 0;
      }
      state = pool + next;
      next += ntiles + 1;
    }
    launch("scan_chained", scan_chained_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, in, n, state,
           reinterpret_cast<unsigned int*>(state + ntiles), out);
    return;
  }
  if (ntiles == 1)
  {
    launch("scan_top", scan_small_kernel<Tin, Tout>, dim3(1), dim3(kBlock), 0, in, n, out);
    return;
  }
  DevArray<Tout> sums(ntiles), offs(ntiles + 1);
  launch("scan_reduce", scan_reduce_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, in, n, sums.p);
  if (ntiles <= kTile)
    launch("scan_top", scan_small_kernel<Tout, Tout>, dim3(1), dim3(kBlock), 0, sums.p, ntiles, offs.p);
  else
    scan_impl<Tout, Tout>(sums.p, offs.p, ntiles);
  launch("scan_write", scan_write_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, in, n, offs.p, out);
}

void exclusive_scan(const int32_t* in, int64_t* out, int64_t n) { scan_impl<int32_t, int64_t>(in, out, n); }
void exclusive_scan(const int32_t* in, int32_t* out, int64_t n) { scan_impl<int32_t, int32_t>(in, out, n); }
void exclusive_scan(const int64_t* in, int64_t* out, int64_t n) { scan_impl<int64_t, int64_t>(in, out, n); }

// ---------------------------------------------------------------------------
// incidence inversion: map[ncells][width] (item ids) -> item -> cells (CSR).
// Counting sort with integer atomics, then each segment is sorted so the
// result does not depend on the atomic arrival order.
// ---------------------------------------------------------------------------
__global__ void adj_count_kernel(const int32_t* __restrict__ map, int64_t nentries, int32_t* counts)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nentries) atomicAdd(&counts[map[i]], 1);
}

__global__ void adj_fill_kernel(const int32_t* __restrict__ map, int64_t nentries, int width,
                                const int64_t* __restrict__ offsets, int32_t* cursor, int32_t* cells)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nentries) return;
  const int32_t item = map[i];
  const int32_t pos = atomicAdd(&cursor[item], 1);
  cells[offsets[item] + pos] = (int32_t)(i / width);
}

// Every item's list in ascending order (the fill above went through an atomic cursor).  One wavefront per 64
// consecutive items: their lists are one contiguous run of `cells`, staged in LDS with coalesced loads, sorted there
// (insertion sort: the lists are short and nearly sorted, cells were issued in ascending order) and written back
// coalesced; a thread sorting its own list in global memory touched 64 different lines per load (80 ms at 512^3).
constexpr int kAdjSortCap = 4096;
__global__ void __launch_bounds__(64) adj_sort_kernel(int64_t nitems, const int64_t* __restrict__ offsets, int32_t* cells)
{
  __shared__ int32_t s_c[kAdjSortCap];
  const int lane = threadIdx.x;
  const int64_t it0 = (int64_t)blockIdx.x * 64;
  const int64_t it = it0 + lane;
  const int64_t last = it0 + 64 < nitems ? it0 + 64 : nitems;
  const int64_t base = offsets[it0], end = offsets[last];
  const int span = (int)(end - base);
  const int64_t b = it < nitems ? offsets[it] : end, e = it < nitems ? offsets[it + 1] : end;
  if (span <= kAdjSortCap)
  {
    for (int k = lane; k < span; k += 64) s_c[k] = cells[base + k];
    __syncthreads();
    const int lb = (int)(b - base), le = (int)(e - base);
    for (int i = lb + 1; i < le; ++i)
    {
      const int32_t v = s_c[i];
      int j = i - 1;
      while (j >= lb && s_c[j] > v) { s_c[j + 1] = s_c[j]; --j; }
      s_c[j + 1] = v;
    }
    __syncthreads();
    for (int k = lane; k < span; k += 64) cells[base + k] = s_c[k];
    return;
  }
  for (int64_t i = b + 1; i < e; ++i) // (very long lists: in place)
  {
    const int32_t v = cells[i];
    int64_t j = i - 1;
    while (j >= b && cells[j] > v) { cells[j + 1] = cells[j]; --j; }
    cells[j + 1] = v;
  }
}

void build_adjacency(const int32_t* map, int64_t ncells, int width, int64_t nitems, Adjacency& adj)
{
  const int64_t nentries = ncells * width;
  DevArray<int32_t> counts(nitems);
  counts.zero();
  launch("adj_count", adj_count_kernel, grid_for(nentries), dim3(kBlock), 0, map, nentries, counts.p);
  adj.offsets.alloc(nitems + 1);
  exclusive_scan(counts.p, adj.offsets.p, nitems);
  adj.cells.alloc(nentries);
  counts.zero();
  launch("adj_fill", adj_fill_kernel, grid_for(nentries), dim3(kBlock), 0, map, nentries, width,
         adj.offsets.p, counts.p, adj.cells.p);
  launch("adj_sort", adj_sort_kernel, dim3((unsigned)((nitems + 63) / 64)), dim3(64), 0, nitems, adj.offsets.p, adj.cells.p);
  adj.built = true;
  publish_across_lanes();
}

} // namespace cfx

using namespace cfx;

// ---------------------------------------------------------------------------
// C ABI: runtime
// ---------------------------------------------------------------------------
namespace
{
// streaming fill, 16 B per lane per store, 4 stores per thread (zeroing the 3 GB of CSR values and the 1 GB
// right-hand side is part of every step); the bytes past the last 16 B chunk go out bytewise from block 0
__global__ void __launch_bounds__(kBlock) fill16_kernel(unsigned char* __restrict__ p, size_t bytes, unsigned word)
{
  const int64_t n16 = (int64_t)(bytes / 16);
  const uint4 v = make_uint4(word, word, word, word);
  const int64_t base = (int64_t)blockIdx.x * (kBlock * 4) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const int64_t i = base + (int64_t)k * kBlock;
    if (i < n16) reinterpret_cast<uint4*>(p)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) p[(size_t)n16 * 16 + threadIdx.x] = (unsigned char)word;
}
} // namespace

namespace cfx
{
int* zero_flag()
{
  // two alternating pools: only the pool being switched TO is refilled, so a flag a caller still holds
  // (written by kernels, not yet read back) survives until 4096 further flags have been handed out
  constexpr int kFlags = 4096;
  struct FlagPool { int* pool[2] = {nullptr, nullptr}; int cur = 1, next = kFlags; };
  static std::map<hipStream_t, FlagPool> pools; // one per stream, as for the scan states
  FlagPool& fp = pools[ctx().stream];
  int* (&pool)[2] = fp.pool;
  int& cur = fp.cur;
  int& next = fp.next;
  if (next == kFlags)
  {
    cur ^= 1;
    if (!pool[cur]) pool[cur] = static_cast<int*>(dev_alloc(sizeof(int) * kFlags));
    // stream order: every earlier user of this pool is ahead of the fill, every later one behind it
    dev_fill(pool[cur], 0, sizeof(int) * kFlags);
    next = 0;
  }
  return pool[cur] + next++;
}

void dev_fill(void* p, int byte, size_t bytes)
{
  if (bytes == 0) return;
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0)
  {
    CFX_HIP(hipMemsetAsync(p, byte, bytes, ctx().stream));
    return;
  }
  const unsigned b = (unsigned)byte & 0xffu, word = b | (b << 8) | (b << 16) | (b << 24);
  const int64_t n16 = (int64_t)(bytes / 16);
  launch("fill", fill16_kernel, grid_for(std::max<int64_t>(n16, 1), kBlock * 4), dim3(kBlock), 0,
         static_cast<unsigned char*>(p), bytes, word);
}
} // namespace cfx

extern "C" {

const char* cfx_last_error(void) { return g_last_error.c_str(); }

int cfx_init(int device)
{
  CFX_API_BEGIN
  Context& c = ctx();
  // one process drives one GPU (one rank per device): the block cache, the flag / scan-state pools and every
  // live handle hold pointers of the first device, so a later call must name the same device
  if (c.initialised && c.device != device)
    throw Error(CFX_ERR_INVALID_ARGUMENT, "cfx_init: the library is already bound to device " + std::to_string(c.device)
                                              + "; use one process per GPU");
  c.device = device;
  c.ensure();
  CFX_API_END
}

int cfx_set_stream(void* s)
{
  CFX_API_BEGIN
  ctx().ensure();
  ctx().flush_profile();
  // cached blocks and pool slices handed out earlier may still be in use by work queued on the old stream
  require(!ctx().overlap, CFX_ERR_RUNTIME, "cfx_set_stream inside an overlap section");
  if (ctx().stream != (hipStream_t)s) CFX_HIP(hipStreamSynchronize(ctx().stream));
  ctx().stream = ctx().main_stream = (hipStream_t)s;
  CFX_API_END
}

int cfx_synchronize(void)
{
  CFX_API_BEGIN
  ctx().ensure();
  CFX_HIP(hipStreamSynchronize(ctx().main_stream));
  if (ctx().side_stream) CFX_HIP(hipStreamSynchronize(ctx().side_stream));
  CFX_API_END
}

int cfx_overlap_begin(void)
{
  CFX_API_BEGIN
  Context& c = ctx();
  c.ensure();
  require(!c.overlap, CFX_ERR_RUNTIME, "cfx_overlap_begin: already inside an overlap section");
  if (!c.side_stream) CFX_HIP(hipStreamCreateWithFlags(&c.side_stream, hipStreamNonBlocking));
  hipEvent_t e = c.get_event();
  CFX_HIP(hipEventRecord(e, c.main_stream));          // the side lane starts behind everything queued so far
  CFX_HIP(hipStreamWaitEvent(c.side_stream, e, 0));
  c.event_pool.push_back(e);
  c.overlap = true;
  CFX_API_END
}

int cfx_overlap_side(int side)
{
  CFX_API_BEGIN
  Context& c = ctx();
  require(c.overlap, CFX_ERR_RUNTIME, "cfx_overlap_side: not inside an overlap section");
  c.stream = side ? c.side_stream : c.main_stream;
  CFX_API_END
}

int cfx_overlap_end(void)
{
  CFX_API_BEGIN
  Context& c = ctx();
  require(c.overlap, CFX_ERR_RUNTIME, "cfx_overlap_end: not inside an overlap section");
  hipEvent_t e = c.get_event();
  CFX_HIP(hipEventRecord(e, c.side_stream));          // join: the main lane continues behind the side lane's work
  CFX_HIP(hipStreamWaitEvent(c.main_stream, e, 0));
  c.event_pool.push_back(e);
  c.stream = c.main_stream;
  c.overlap = false;
  for (void* p : c.deferred_free) dev_free(p);        // stream order on the main lane now covers both lanes' users
  c.deferred_free.clear();
  CFX_API_END
}

int cfx_copy(void* dst, const void* src, size_t bytes)
{
  CFX_API_BEGIN
  ctx().ensure();
  if (bytes > 0)
  {
    require(dst && src, CFX_ERR_INVALID_ARGUMENT, "cfx_copy: null pointer");
    CFX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx().stream));
    CFX_HIP(hipStreamSynchronize(ctx().stream));
  }
  CFX_API_END
}

int cfx_device_alloc(void** ptr, size_t bytes)
{
  CFX_API_BEGIN
  ctx().ensure();
  *ptr = dev_alloc(bytes); // cached blocks: a real hipMalloc/hipFree pair per step costs milliseconds
  CFX_API_END
}

int cfx_device_free(void* ptr)
{
  CFX_API_BEGIN
  ctx().ensure();
  dev_free(ptr);
  CFX_API_END
}

int cfx_device_cache_release(void)
{
  CFX_API_BEGIN
  ctx().ensure();
  dev_cache_release();
  CFX_API_END
}

int cfx_device_memory_stats(size_t* in_use, size_t* cached, size_t* peak, int reset_peak)
{
  CFX_API_BEGIN
  size_t a = 0, b = 0, c = 0;
  device_memory_stats(a, b, c);
  if (in_use) *in_use = a;
  if (cached) *cached = b;
  if (peak) *peak = c;
  if (reset_peak) device_memory_reset_peak();
  CFX_API_END
}

int cfx_device_memset(void* ptr, int byte, size_t bytes)
{
  CFX_API_BEGIN
  ctx().ensure();
  if (bytes && !ptr) throw Error(CFX_ERR_INVALID_ARGUMENT, "cfx_device_memset: null pointer");
  if (bytes) cfx::dev_fill(ptr, byte, bytes);
  CFX_API_END
}

int cfx_profile_enable(int on)
{
  CFX_API_BEGIN
  ctx().ensure();
  ctx().flush_profile();
  ctx().profile = on != 0;
  CFX_API_END
}

int cfx_profile_reset(void)
{
  CFX_API_BEGIN
  ctx().flush_profile();
  for (auto& e : ctx().entries) { e.total_ms = 0.0; e.launches = 0; }
  CFX_API_END
}

int cfx_profile_count(void)
{
  try { ctx().flush_profile(); } catch (...) { return 0; }
  return (int)ctx().entries.size();
}

int cfx_profile_get(int i, const char** name, double* total_ms, int64_t* launches)
{
  CFX_API_BEGIN
  ctx().flush_profile();
  require(i >= 0 && i < (int)ctx().entries.size(), CFX_ERR_OUT_OF_RANGE, "profile index out of range");
  const ProfileEntry& e = ctx().entries[i];
  if (name) *name = e.name.c_str();
  if (total_ms) *total_ms = e.total_ms;
  if (launches) *launches = e.launches;
  CFX_API_END
}

int cfx_event_create(void** ev)
{
  CFX_API_BEGIN
  ctx().ensure();
  hipEvent_t e;
  CFX_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  CFX_API_END
}

int cfx_event_record(void* ev)
{
  CFX_API_BEGIN
  CFX_HIP(hipEventRecord((hipEvent_t)ev, ctx().stream));
  CFX_API_END
}

int cfx_event_elapsed_ms(void* a, void* b, double* ms)
{
  CFX_API_BEGIN
  CFX_HIP(hipEventSynchronize((hipEvent_t)b));
  float f = 0.f;
  CFX_HIP(hipEventElapsedTime(&f, (hipEvent_t)a, (hipEvent_t)b));
  *ms = f;
  CFX_API_END
}

int cfx_event_destroy(void* ev)
{
  CFX_API_BEGIN
  CFX_HIP(hipEventDestroy((hipEvent_t)ev));
  CFX_API_END
}

} // extern "C"
