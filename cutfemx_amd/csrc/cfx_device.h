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

template <typename Pred>
inline Count compact_count(const char* name, const char* site, DevN n, Pred pred, DevArray<int32_t>& out)
{
  const int64_t ntiles = (n.cap + kTile - 1) / kTile;
  if (ntiles == 0)
  {
    out.alloc(0);
    step_record(site, 0);
    return Count(0);
  }
  DevArray<int32_t> counts(ntiles);
  DevArray<int64_t> offsets(ntiles + 1);
  launch(name, compact_count_n_kernel<Pred>, dim3((unsigned)ntiles), dim3(kBlock), 0, n, pred, counts.p);
  Count total;
  {
    const CountSource src{offsets.p + ntiles, kCountI64, kCountUpTo};
    CountPlan cp(1, &site, &src);
    exclusive_scan(counts.p, offsets.p, ntiles, &cp);
    cp.finish(&total);
  }
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

__global__ inline void ride_along_kernel(const int64_t* a, const int* b, int64_t* out)
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
