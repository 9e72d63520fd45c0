// cutfemx_amd: device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once

#include "cfx_common.h"

namespace cfx
{

constexpr int kBlock = 256;      // 4 wavefronts
constexpr int kScanItems = 8;    // items per thread in scan/compaction tiles
constexpr int kTile = kBlock * kScanItems;

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Row-ordered kernels remap the block id so that every XCD walks one
// contiguous chunk of the rows and neighbouring rows meet in the same L2.
// Launch with xcd_grid(); placement only affects speed, never results.
__device__ __forceinline__ int64_t xcd_block_id()
{
  const int64_t per = gridDim.x >> 3; // gridDim.x is a multiple of 8
  return (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

inline dim3 xcd_grid(int64_t nblocks)
{
  const int64_t g = (nblocks + 7) / 8 * 8;
  if (g > 2147483647LL) throw Error(CFX_ERR_RUNTIME, "grid too large");
  return dim3((unsigned)g);
}

// the block's threads fill the contiguous run p[0..n) with v: 16 B per lane (8 B stores reach about half the write
// rate), one value ahead of the pairs when the run starts on an odd index
__device__ __forceinline__ void block_fill_run(double* __restrict__ p, int n, double v)
{
  const int head = (int)((reinterpret_cast<uintptr_t>(p) >> 3) & 1) & (n > 0 ? 1 : 0);
  if (threadIdx.x == 0 && head) p[0] = v;
  const int npair = (n - head) >> 1;
  double2* q = reinterpret_cast<double2*>(p + head);
  const double2 vv = make_double2(v, v);
  for (int k = threadIdx.x; k < npair; k += blockDim.x) q[k] = vv;
  if (threadIdx.x == 0 && ((n - head) & 1)) p[n - 1] = v;
}

// 32-bit finaliser (murmur3): cell ids of a structured mesh are arithmetic progressions, which a
// bare multiplicative hash maps onto a few residues of a power-of-two table (measured: 2x longer
// probe chains on one of eight slabs)
__host__ __device__ __forceinline__ uint32_t cfx_hash32(uint32_t h)
{
  h ^= h >> 16; h *= 0x85ebca6bu;
  h ^= h >> 13; h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

// Integer atomics of a workgroup combined in LDS before they go to memory (incidence inversions: adj_*_lds_kernel,
// facet_dof_*_kernel): kAdjRun consecutive entries per workgroup, <= kAdjRun distinct keys in 2 kAdjRun slots; the
// return value is the slot, `rank` the arrival number of this entry among the workgroup's entries with the same key.
constexpr int kAdjPer = 4, kAdjRun = kBlock * kAdjPer, kAdjSlots = 2 * kAdjRun;
__device__ __forceinline__ int adj_lds_insert(int32_t* s_key, int32_t* s_cnt, int32_t item, int& rank)
{
  unsigned h = cfx_hash32((uint32_t)item) & (kAdjSlots - 1);
  for (;;)
  {
    const int32_t prev = atomicCAS(&s_key[h], -1, item);
    if (prev == -1 || prev == item) break;
    h = (h + 1) & (kAdjSlots - 1);
  }
  rank = atomicAdd(&s_cnt[h], 1);
  return (int)h;
}

// inclusive scan over the 64 lanes of a wavefront
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    T up = __shfl_up(v, d, 64);
    if (lane >= d) v += up;
  }
  return v;
}

// exclusive scan across a 256-thread block; returns the exclusive prefix of
// `v`, writes the block total to `total` (all threads)
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T& total)
{
  __shared__ T wave_sums[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T incl = wave_inclusive_scan(v);
  __syncthreads(); // protect wave_sums reuse across calls
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w)
  {
    T s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + incl - v;
}

// ---------------------------------------------------------------------------
// Chained tiles: the exclusive prefix of a per-tile total over the tiles of ONE launch (single-pass scans, and
// count + offset + write kernels that would otherwise be three launches around a scan of the tile totals).
// A tile takes its number from an atomic ticket when its block starts, so every predecessor it waits for is already
// resident and makes progress.  Tile state = one 64-bit word (2-bit flag | 62-bit value) exchanged with agent-scope
// atomics: the L2s of the eight XCDs are not coherent with each other for plain loads.  `state` holds one zeroed word
// per tile, `ticket` one zeroed counter (cfx::chain_state hands out both).
// ---------------------------------------------------------------------------
constexpr int kChainLook = 4;
constexpr unsigned long long kScanAggregate = 1ull << 62, kScanPrefix = 2ull << 62, kScanValueMask = (1ull << 62) - 1ull;

struct ChainState
{
  unsigned long long* state = nullptr;
  unsigned int* ticket = nullptr;
};
// zeroed state words for `ntiles` tiles + the ticket, from a pool that is cleared with one fill when it wraps; state ==
// nullptr when ntiles is beyond what the pool serves (the caller keeps its multi-launch form); cfx_runtime.hip
ChainState chain_state(int64_t ntiles);
// count + offsets + write kernels of a sync-free step as ONE chained launch each, up to CFX_FUSED_TILES tiles (0: never;
// the three launches around a scan of the tile totals, as outside a step).  Longer lists keep the passes that never
// wait on each other.  Measured on MI355X (ticket on one address + look-back): ~13 ns per tile on top of the work, so
// the two launch boundaries saved (~10 us each) are spent at ~1500 tiles; break-even was seen at 128^3 (500 - 1000
// tiles per site), 32^3 gains 0.03 of 0.48 ms.  Groups of 8 tiles per ticket were tried: slower at every size (few
// workgroups walking their tiles one after the other).
inline ChainState fused_chain(bool publish, int64_t ntiles)
{
  static const int64_t max_tiles = []() { const char* e = getenv("CFX_FUSED_TILES"); return e ? atoll(e) : (int64_t)512; }();
  return publish && ntiles <= max_tiles ? chain_state(ntiles) : ChainState{};
}

// all threads of the block: the number of this block's tile
__device__ __forceinline__ unsigned int chain_take_tile(unsigned int* __restrict__ ticket)
{
  __shared__ unsigned int s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  return s_tile;
}

// all threads of the block (>= 64 threads), `total` block-uniform: sum of the totals of the tiles before `tile`
__device__ __forceinline__ unsigned long long chain_exclusive_prefix(unsigned long long* __restrict__ state,
                                                                     const unsigned int tile, const unsigned long long total)
{
  __shared__ unsigned long long s_prefix;
  if (threadIdx.x < 64)
  {
    const int lane = threadIdx.x;
    unsigned long long prefix = 0;
    if (tile == 0)
    {
      if (lane == 0)
        __hip_atomic_store(&state[0], kScanPrefix | (total & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
      if (lane == 0)
        __hip_atomic_store(&state[tile], kScanAggregate | (total & kScanValueMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      // look back kChainLook x 64 tiles per round trip (the loads of a round are in flight together: the front of
      // complete prefixes moves that many tiles per memory latency) until a tile with a complete prefix is found
      int64_t hi = (int64_t)tile - 1;
      unsigned long long acc = 0; // per lane; summed over the wave at the end
      bool found = false;
      while (!found)
      {
        unsigned long long w[kChainLook];
#pragma unroll
        for (int q = 0; q < kChainLook; ++q)
        {
          const int64_t p = hi - lane - 64 * q;
          w[q] = kScanPrefix; // tiles before the first: prefix 0
          if (p >= 0) w[q] = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < kChainLook; ++q)
        {
          const int64_t p = hi - lane - 64 * q;
          while ((w[q] >> 62) == 0ull) w[q] = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < kChainLook; ++q)
        {
          if (found) break;
          const unsigned long long done = __ballot((w[q] >> 62) == 2ull);
          const int first = done ? __ffsll((long long)done) - 1 : 64; // nearest tile whose prefix is complete
          acc += (lane <= first) ? (w[q] & kScanValueMask) : 0ull;
          found = done != 0ull;
        }
        hi -= 64 * kChainLook;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      prefix = acc;
      if (lane == 0)
        __hip_atomic_store(&state[tile], kScanPrefix | ((prefix + total) & kScanValueMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  return s_prefix;
}

// ---------------------------------------------------------------------------
// stream compaction: indices i in [0,n) with pred(i), ascending.
// Three launches: per-tile counts, scan, per-tile write.
// ---------------------------------------------------------------------------
template <typename Pred>
__global__ void __launch_bounds__(kBlock) compact_count_kernel(int64_t n, Pred pred, int32_t* tile_counts)
{
  const int64_t base = (int64_t)blockIdx.x * kTile;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < n && pred(i)) ++c;
  }
  int total;
  (void)block_exclusive_scan<int>(c, total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

template <typename Pred>
__global__ void __launch_bounds__(kBlock) compact_write_kernel(int64_t n, Pred pred, const int64_t* tile_offsets,
                                                               int32_t* out)
{
  // thread-major item order (thread t owns items t*kScanItems..) keeps output ascending
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kScanItems;
  bool f[kScanItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = base + k;
    f[k] = (i < n) && pred(i);
    c += f[k] ? 1 : 0;
  }
  int total;
  int off = block_exclusive_scan<int>(c, total);
  int64_t o = tile_offsets[blockIdx.x] + off;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (f[k]) out[o++] = (int32_t)(base + k);
}

// returns count; `out` is allocated to exactly that size
template <typename Pred>
inline int64_t compact(const char* name, int64_t n, Pred pred, DevArray<int32_t>& out)
{
  const int64_t ntiles = (n + kTile - 1) / kTile;
  if (ntiles == 0) { out.alloc(0); return 0; }
  DevArray<int32_t> counts(ntiles);
  DevArray<int64_t> offsets(ntiles + 1);
  launch(name, compact_count_kernel<Pred>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, pred, counts.p);
  exclusive_scan(counts.p, offsets.p, ntiles);
  const int64_t total = read_scalar(offsets.p + ntiles);
  out.alloc(total);
  launch(name, compact_write_kernel<Pred>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, pred, offsets.p, out.p);
  return total;
}

// the same compaction over a list whose own length may still be in HBM (`n`), the total a count site (cfx_common.h)
template <typename Pred>
__global__ void __launch_bounds__(kBlock) compact_count_n_kernel(DevN n_d, Pred pred, int32_t* tile_counts)
{
  const int64_t n = dev_n(n_d);
  const int64_t base = (int64_t)blockIdx.x * kTile;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < n && pred(i)) ++c;
  }
  int total;
  (void)block_exclusive_scan<int>(c, total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

template <typename Pred>
__global__ void __launch_bounds__(kBlock) compact_write_n_kernel(DevN n_d, Pred pred, const int64_t* tile_offsets,
                                                                 int32_t* out, DevN out_n)
{
  const int64_t n = dev_n(n_d);
  const int64_t cap = out_n.dev ? dev_n(out_n) : INT64_MAX;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kScanItems;
  bool f[kScanItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = base + k;
    f[k] = (i < n) && pred(i);
    c += f[k] ? 1 : 0;
  }
  int total;
  int off = block_exclusive_scan<int>(c, total);
  int64_t o = tile_offsets[blockIdx.x] + off;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (f[k]) { if (o < cap) out[o] = (int32_t)(base + k); ++o; }
}

// count + offsets + write in ONE launch (tiles chained by look-back): inside a sync-free step, where `out` is sized by the
// previous step's total before anything is counted.  emit(o, i): what else the caller derives from hit i at position o.
struct NoEmit
{
  __device__ void operator()(int64_t, int64_t) const {}
};
template <typename Pred, typename Emit>
__global__ void __launch_bounds__(kBlock) compact_chained_kernel(DevN n_d, Pred pred, Emit emit, int32_t* __restrict__ out,
                                                                 int64_t cap, ChainState chain,
                                                                 int64_t* __restrict__ total_out, CountJobs after)
{
  const int64_t n = dev_n(n_d);
  const unsigned int tile = chain_take_tile(chain.ticket);
  const int64_t base = (int64_t)tile * kTile + (int64_t)threadIdx.x * kScanItems;
  bool f[kScanItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = base + k;
    f[k] = (i < n) && pred(i);
    c += f[k] ? 1 : 0;
  }
  int total;
  const int off = block_exclusive_scan<int>(c, total);
  const int64_t prefix = (int64_t)chain_exclusive_prefix(chain.state, tile, (unsigned long long)total);
  int64_t o = prefix + off;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (f[k]) { if (o < cap) { out[o] = (int32_t)(base + k); emit(o, base + k); } ++o; }
  if (tile == gridDim.x - 1 && threadIdx.x == kBlock - 1)
  {
    *total_out = prefix + total;
    if (after.n > 0) count_publish(after);
  }
}

// `pre(capacity)`: called once the capacity of `out` is known and before the write pass (allocates what emit writes to)
template <typename Pred, typename Emit = NoEmit>
inline Count compact_count(const char* name, const char* site, DevN n, Pred pred, DevArray<int32_t>& out,
                           Emit* emit = nullptr, const std::function<void(int64_t)>& pre = nullptr)
{
  const int64_t ntiles = (n.cap + kTile - 1) / kTile;
  if (ntiles == 0)
  {
    out.alloc(0);
    step_record(site, 0);
    if (pre) pre(0);
    return Count(0);
  }
  DevArray<int32_t> counts;
  DevArray<int64_t> offsets(ntiles + 1);
  Count total;
  const CountSource src{offsets.p + ntiles, kCountI64, kCountUpTo};
  CountPlan cp(1, &site, &src);
  const ChainState chain = fused_chain(cp.publish, ntiles);
  if (chain.state)
  {
    const CountJobs after = cp.take_jobs();
    cp.finish(&total);
    out.alloc(total.cap());
    out.count = total;
    if (pre) pre(total.cap());
    launch(name, compact_chained_kernel<Pred, Emit>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, pred, emit ? *emit : Emit{},
           out.p, total.cap(), chain, offsets.p + ntiles, after);
    return total;
  }
  counts.alloc(ntiles);
  launch(name, compact_count_n_kernel<Pred>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, pred, counts.p);
  exclusive_scan(counts.p, offsets.p, ntiles, &cp);
  cp.finish(&total);
  out.alloc(total.cap());
  if (total.cell) out.count = total;
  launch(name, compact_write_n_kernel<Pred>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, pred, offsets.p, out.p, total.devn());
  return total;
}

// ---------------------------------------------------------------------------
// compaction of a byte array (classification codes, flags): 16 B per lane per
// load, so a wave streams 1 KiB per instruction instead of 64 B.
// ---------------------------------------------------------------------------
constexpr int kByteItems = 16;
constexpr int kByteTile = kBlock * kByteItems;

template <typename ByteTest>
__device__ __forceinline__ unsigned byte_flags(const uint8_t* __restrict__ bytes, int64_t base, int64_t n, ByteTest test)
{
  unsigned f = 0;
  if (base + kByteItems <= n)
  {
    const uint4 v = *reinterpret_cast<const uint4*>(bytes + base);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < kByteItems; ++k) f |= test((uint8_t)(w[k >> 2] >> (8 * (k & 3)))) ? (1u << k) : 0u;
  }
  else
  {
    for (int k = 0; k < kByteItems; ++k)
      if (base + k < n && test(bytes[base + k])) f |= 1u << k;
  }
  return f;
}

template <typename ByteTest>
__global__ void __launch_bounds__(kBlock) compact_bytes_count_kernel(int64_t n, const uint8_t* __restrict__ bytes,
                                                                     ByteTest test, int32_t* tile_counts)
{
  const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kByteItems;
  const int c = base < n ? __popc(byte_flags(bytes, base, n, test)) : 0;
  int total;
  (void)block_exclusive_scan<int>(c, total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

template <typename ByteTest>
__global__ void __launch_bounds__(kBlock) compact_bytes_write_kernel(int64_t n, const uint8_t* __restrict__ bytes,
                                                                     ByteTest test,
                                                                     const int64_t* __restrict__ tile_offsets,
                                                                     int32_t* __restrict__ out, DevN out_n)
{
  // (a list sized by the previous step: nothing beyond its published length -- 0 in a void step -- is written)
  const int64_t cap = out_n.dev ? dev_n(out_n) : INT64_MAX;
  // the tile's hits are packed in LDS first: the global stores are then one contiguous, coalesced run per tile
  // (direct stores leave every lane with its own short run: 64 partial sectors per store instruction)
  __shared__ int32_t s_out[kByteTile];
  const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kByteItems;
  const unsigned f = base < n ? byte_flags(bytes, base, n, test) : 0u;
  int total;
  int off = block_exclusive_scan<int>(__popc(f), total);
#pragma unroll
  for (int k = 0; k < kByteItems; ++k)
    if (f & (1u << k)) s_out[off++] = (int32_t)(base + k);
  __syncthreads();
  const int64_t o = tile_offsets[blockIdx.x];
  for (int i = threadIdx.x; i < total; i += kBlock)
    if (o + i < cap) out[o + i] = s_out[i];
}

static __global__ void ride_along_kernel(const int64_t* a, const int* b, int64_t* out)
{
  out[0] = *a; out[1] = *b;
}

// `bytes` must be 16-byte aligned
// `tile_counts` (optional): matches per tile of kByteTile bytes already known to the caller
template <typename ByteTest>
// `known_total` (optional): the number of matches when the caller already has it on the host -- no size read-back
// `ride_flag` / `ride_out` (optional): a device int the caller wants on the host as well -- it rides on the size read-back
inline int64_t compact_bytes(const char* name, int64_t n, const uint8_t* bytes, ByteTest test, DevArray<int32_t>& out,
                             const int32_t* tile_counts = nullptr, int64_t known_total = -1, const int* ride_flag = nullptr,
                             int* ride_out = nullptr)
{
  const int64_t ntiles = (n + kByteTile - 1) / kByteTile;
  if (ntiles == 0) { out.alloc(0); return 0; }
  DevArray<int32_t> counts;
  DevArray<int64_t> offsets(ntiles + 1);
  if (!tile_counts)
  {
    counts.alloc(ntiles);
    launch(name, compact_bytes_count_kernel<ByteTest>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, bytes, test,
           counts.p);
  }
  exclusive_scan(tile_counts ? tile_counts : counts.p, offsets.p, ntiles);
  int64_t total;
  if (known_total >= 0) total = known_total;
  else if (ride_flag)
  {
    DevArray<int64_t> two(2);
    launch(name, ride_along_kernel, dim3(1), dim3(1), 0, offsets.p + ntiles, ride_flag, two.p);
    struct Two { int64_t v[2]; };
    const Two t = read_scalar(reinterpret_cast<const Two*>(two.p));
    total = t.v[0];
    *ride_out = (int)t.v[1];
  }
  else total = read_scalar(offsets.p + ntiles);
  out.alloc(total);
  launch(name, compact_bytes_write_kernel<ByteTest>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, bytes, test,
         offsets.p, out.p, DevN());
  return total;
}

// the same compaction for lists that the kernels of a sync-free step consume: the total is a count site (`site`: its
// name in the step's history) -- read back at once outside a step, left in HBM inside one, where `out` is sized by the
// previous step's total and carries the published length (out.count)
template <typename ByteTest>
inline Count compact_bytes_count(const char* name, const char* site, int64_t n, const uint8_t* bytes, ByteTest test,
                                 DevArray<int32_t>& out, const int32_t* tile_counts = nullptr)
{
  const int64_t ntiles = (n + kByteTile - 1) / kByteTile;
  if (ntiles == 0)
  {
    out.alloc(0);
    step_record(site, 0);
    return Count(0);
  }
  DevArray<int32_t> counts;
  DevArray<int64_t> offsets(ntiles + 1);
  if (!tile_counts)
  {
    counts.alloc(ntiles);
    launch(name, compact_bytes_count_kernel<ByteTest>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, bytes, test,
           counts.p);
  }
  Count total;
  {
    const CountSource src{offsets.p + ntiles, kCountI64, kCountUpTo};
    CountPlan cp(1, &site, &src);
    exclusive_scan(tile_counts ? tile_counts : counts.p, offsets.p, ntiles, &cp);
    cp.finish(&total);
  }
  out.alloc(total.cap());
  if (total.cell) out.count = total;
  launch(name, compact_bytes_write_kernel<ByteTest>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, bytes, test,
         offsets.p, out.p, total.devn());
  return total;
}

struct ByteNonZero
{
  __device__ bool operator()(uint8_t v) const { return v != 0; }
};
struct ByteZero
{
  __device__ bool operator()(uint8_t v) const { return v == 0; }
};
// classification code (-1, 0, 1) against a selector mask: bit0 inside, bit1 cut, bit2 outside
struct DomainMask
{
  int mask;
  __device__ bool operator()(uint8_t v) const { return (mask >> ((int)(int8_t)v + 1)) & 1; }
};

// ---------------------------------------------------------------------------
// small geometry helpers (affine simplices, gdim == tdim)
// ---------------------------------------------------------------------------
template <int TDIM>
struct Geo
{
  double x[TDIM + 1][TDIM]; // vertex coordinates
  double K[TDIM][TDIM];     // K[t][d] = d xi_t / d x_d
  double detJ;
};

template <int TDIM>
__device__ __forceinline__ void load_cell(const double* __restrict__ x, const int32_t* __restrict__ conn,
                                          int64_t cell, Geo<TDIM>& g)
{
#pragma unroll
  for (int i = 0; i <= TDIM; ++i)
  {
    const int64_t v = conn[cell * (TDIM + 1) + i];
#pragma unroll
    for (int d = 0; d < TDIM; ++d) g.x[i][d] = x[3 * v + d];
  }
}

template <int TDIM>
__device__ __forceinline__ void jacobian(Geo<TDIM>& g)
{
  double J[TDIM][TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
#pragma unroll
    for (int t = 0; t < TDIM; ++t) J[d][t] = g.x[t + 1][d] - g.x[0][d];
  if constexpr (TDIM == 2)
  {
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double inv = 1.0 / det; // one FP64 division per cell instead of four
    g.K[0][0] = J[1][1] * inv;  g.K[0][1] = -J[0][1] * inv;
    g.K[1][0] = -J[1][0] * inv; g.K[1][1] = J[0][0] * inv;
    g.detJ = det;
  }
  else
  {
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double inv = 1.0 / det; // one FP64 division per cell instead of nine
    g.K[0][0] = c00 * inv;
    g.K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * inv;
    g.K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * inv;
    g.K[1][0] = c01 * inv;
    g.K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * inv;
    g.K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * inv;
    g.K[2][0] = c02 * inv;
    g.K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * inv;
    g.K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * inv;
    g.detJ = det;
  }
}

// UFL CellDiameter: largest vertex-to-vertex distance
template <int TDIM>
__device__ __forceinline__ double cell_diameter(const Geo<TDIM>& g)
{
  double h2 = 0.0;
#pragma unroll
  for (int i = 0; i <= TDIM; ++i)
#pragma unroll
    for (int j = i + 1; j <= TDIM; ++j)
    {
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < TDIM; ++d) d2 += (g.x[i][d] - g.x[j][d]) * (g.x[i][d] - g.x[j][d]);
      h2 = d2 > h2 ? d2 : h2;
    }
  return sqrt(h2);
}

} // namespace cfx
