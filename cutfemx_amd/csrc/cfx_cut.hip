// cutfemx_amd: level-set classification, selector scan, cut-cell
// sub-triangulation + runtime quadrature, per-point level-set evaluators and
// ghost-penalty facet selection -- HIP kernels for gfx950 and their C ABI.
//
// Replaces (paths relative to the CutFEMx tree):
//   cpp/cutfemx/cut/cut.cpp:845-868   update()  -> cutcells::cut        (a1, a2)
//   cpp/cutfemx/cut/cut.cpp:877-924   locate_entities()                  (a4)
//   cpp/cutfemx/cut/cut.cpp:1311-1335 runtime_quadrature()               (a3)
//   cpp/cutfemx/level_set/normal.h:39-187, value.h:34-119                (a12)
//   python/cutfemx/cut.py:340-380     ghost_penalty_facets()             (a7)
#include <cctype>
#include <cstdlib>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "cfx_device.h"

#define CFX_QUAD_TABLE_QUALIFIER static __device__ const
#include "cfx_quadrature_tables.h"
#undef CFX_QUAD_TABLE_QUALIFIER

using namespace cfx;

namespace cfx
{
// number of points of the reference rule of `degree` on a `dim`-simplex (cfx_quadhost.cpp)
int quad_npoints(int dim, int degree);
} // namespace cfx

namespace
{

// ---------------------------------------------------------------------------
// selector: DNF over (level set, relation) clauses.  relation -> set of
// domains, cut.cpp:323-342.  bit0 inside, bit1 intersected, bit2 outside.
// ---------------------------------------------------------------------------
constexpr int kMaxClauses = 16;

struct Selector
{
  int n = 0;
  int ls[kMaxClauses];
  int mask[kMaxClauses];
  int term[kMaxClauses];
};

Selector parse_selector(const char* s, int nls)
{
  require(s != nullptr, CFX_ERR_INVALID_ARGUMENT, "selector is null");
  Selector sel;
  int term = 0;
  const char* p = s;
  auto fail = [&](const char* why)
  { throw Error(CFX_ERR_INVALID_ARGUMENT, std::string("invalid selector '") + s + "': " + why); };
  for (;;)
  {
    while (*p && isspace((unsigned char)*p)) ++p;
    if (!*p) fail("expected a level-set name");
    const char* b = p;
    while (*p && (isalnum((unsigned char)*p) || *p == '_')) ++p;
    if (p == b) fail("expected a level-set name");
    const size_t len = (size_t)(p - b);
    int ls = -1;
    if (len >= 3 && strncmp(b, "phi", 3) == 0)
    {
      ls = 0;
      for (size_t i = 3; i < len; ++i)
      {
        if (!isdigit((unsigned char)b[i])) { ls = -1; break; }
        ls = 10 * ls + (b[i] - '0');
      }
    }
    if (ls < 0 || ls >= nls) fail("unknown level-set name");
    while (*p && isspace((unsigned char)*p)) ++p;
    int mask = 0;
    if (p[0] == '<' && p[1] == '=') { mask = 1 | 2; p += 2; }
    else if (p[0] == '>' && p[1] == '=') { mask = 4 | 2; p += 2; }
    else if (p[0] == '=' && p[1] == '=') { mask = 2; p += 2; }
    else if (p[0] == '<') { mask = 1; p += 1; }
    else if (p[0] == '>') { mask = 4; p += 1; }
    else if (p[0] == '=') { mask = 2; p += 1; }
    else fail("expected a relation (<, <=, =, >=, >)");
    while (*p && isspace((unsigned char)*p)) ++p;
    char* e = nullptr;
    const double rhs = strtod(p, &e);
    if (e == p || rhs != 0.0) fail("right-hand side must be 0");
    p = e;
    if (sel.n >= kMaxClauses) fail("too many clauses");
    sel.ls[sel.n] = ls; sel.mask[sel.n] = mask; sel.term[sel.n] = term; ++sel.n;
    while (*p && isspace((unsigned char)*p)) ++p;
    if (!*p) break;
    if (strncmp(p, "and", 3) == 0) p += 3;
    else if (strncmp(p, "&&", 2) == 0) p += 2;
    else if (*p == '&') p += 1;
    else if (strncmp(p, "or", 2) == 0) { p += 2; ++term; }
    else if (strncmp(p, "||", 2) == 0) { p += 2; ++term; }
    else if (*p == '|') { p += 1; ++term; }
    else fail("expected 'and' / 'or'");
  }
  return sel;
}

struct SelectorPred
{
  const int8_t* domain;
  int64_t ncells;
  Selector sel;
  __device__ bool operator()(int64_t c) const
  {
    const int nterm = sel.term[sel.n - 1] + 1;
    for (int t = 0; t < nterm; ++t)
    {
      bool ok = true, any = false;
      for (int k = 0; k < sel.n; ++k)
      {
        if (sel.term[k] != t) continue;
        any = true;
        const int d = domain[(int64_t)sel.ls[k] * ncells + c];
        if (!((sel.mask[k] >> (d + 1)) & 1)) { ok = false; break; }
      }
      if (any && ok) return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------------------
// a1 classification: one thread per cell, coalesced dofmap rows (16 B/lane for
// tets), gathered level-set values (L2/MALL resident), 1 B/cell out.
// ---------------------------------------------------------------------------
// sign code of every level-set dof: 1 negative, 2 positive, 0 zero.  The bitwise AND of a cell's
// codes is 1 iff all its values are negative, 2 iff all are positive, 0 otherwise -- and the byte
// table (135 MB at 512^3) stays in the Infinity Cache where the 1.1 GB of doubles does not.
// (zero_n > 0: the first zero_n threads also clear the per-tile counters of the classification that follows -- a fill
// launch less per step)
// (touch, optional: one byte per level-set dof, cleared here and set by the classification on the dofs of every cut cell)
__global__ void __launch_bounds__(kBlock) sign_codes_kernel(int64_t n, const double* __restrict__ phi, uint8_t* __restrict__ code,
                                                            int32_t* __restrict__ zero_this, int64_t zero_n,
                                                            uint8_t* __restrict__ touch)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < zero_n) zero_this[i] = 0;
  if (i >= n) return;
  const double v = phi[i];
  code[i] = v < 0.0 ? (uint8_t)1 : (v > 0.0 ? (uint8_t)2 : (uint8_t)0);
  if (touch) touch[i] = 0;
}

// the same, four values per thread: two 16 B loads and one 4 B store per lane (a byte per lane and store reached
// 4.1 TB/s of the level set at 512^3).  phi and code start on 16 B / 4 B boundaries (checked at the launch); the last
// n % 4 values go to the last thread one by one.
__global__ void __launch_bounds__(kBlock) sign_codes4_kernel(int64_t n, const double* __restrict__ phi, uint8_t* __restrict__ code,
                                                             int32_t* __restrict__ zero_this, int64_t zero_n,
                                                             uint8_t* __restrict__ touch)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < zero_n) zero_this[i] = 0;
  const int64_t b = 4 * i;
  if (b >= n) return;
  auto sign = [](double v) { return v < 0.0 ? 1u : (v > 0.0 ? 2u : 0u); };
  if (b + 4 <= n)
  {
    const double2 p = *reinterpret_cast<const double2*>(phi + b), q = *reinterpret_cast<const double2*>(phi + b + 2);
    *reinterpret_cast<uint32_t*>(code + b) = sign(p.x) | (sign(p.y) << 8) | (sign(q.x) << 16) | (sign(q.y) << 24);
    if (touch) *reinterpret_cast<uint32_t*>(touch + b) = 0u; // (as aligned as the codes: checked at the launch)
    return;
  }
  for (int64_t k = b; k < n; ++k)
  {
    code[k] = (uint8_t)sign(phi[k]);
    if (touch) touch[k] = 0;
  }
}

#ifndef CFX_CLASSIFY_UNROLL
#define CFX_CLASSIFY_UNROLL 4
#endif
template <int ND>
__global__ void __launch_bounds__(kBlock) classify_kernel(int64_t ncells, const int32_t* __restrict__ dofmap,
                                                          const uint8_t* __restrict__ code, int8_t* __restrict__ domain,
                                                          int32_t* tiles_inside, int32_t* tiles_cut,
                                                          uint8_t* __restrict__ touch = nullptr)
{
  // U cells per thread, a block-wide stride apart: U independent 16 B/lane streaming loads in
  // flight per lane before the first dependent level-set gather
  constexpr int U = CFX_CLASSIFY_UNROLL;
  const int64_t c0 = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  int32_t d[U][ND];
  int n_in = 0, n_cut = 0;
#pragma unroll
  for (int u = 0; u < U; ++u)
  {
    const int64_t c = c0 + (int64_t)u * kBlock;
    if (c >= ncells) continue;
    if constexpr (ND == 4)
    {
      typedef int cfx_i4 __attribute__((ext_vector_type(4)));
      const cfx_i4 v = __builtin_nontemporal_load(reinterpret_cast<const cfx_i4*>(dofmap + c * 4)); // streamed once
      d[u][0] = v.x; d[u][1] = v.y; d[u][2] = v.z; d[u][3] = v.w;
    }
    else
    {
#pragma unroll
      for (int i = 0; i < ND; ++i) d[u][i] = dofmap[c * ND + i];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
  {
    const int64_t c = c0 + (int64_t)u * kBlock;
    if (c >= ncells) continue;
    unsigned all = 3u;
#pragma unroll
    for (int i = 0; i < ND; ++i) all &= code[d[u][i]];
    domain[c] = all == 1u ? (int8_t)CFX_INSIDE : (all == 2u ? (int8_t)CFX_OUTSIDE : (int8_t)CFX_INTERSECTED);
    n_in += all == 1u ? 1 : 0;
    n_cut += (all != 1u && all != 2u) ? 1 : 0;
    if (touch && all != 1u && all != 2u) // (a cut cell: its dofs are next to the interface; every writer stores 1)
    {
#pragma unroll
      for (int i = 0; i < ND; ++i) touch[d[u][i]] = 1;
    }
  }
  // the two selector scans every solve starts with ("phi<0", "phi=0") get their tile counts here
  if (tiles_inside)
  {
    static_assert(kByteTile % (kBlock * U) == 0, "classification blocks must nest in compaction tiles");
    int tot_in, tot_cut;
    (void)block_exclusive_scan<int>(n_in, tot_in);
    (void)block_exclusive_scan<int>(n_cut, tot_cut);
    if (threadIdx.x == 0)
    {
      const int64_t tile = ((int64_t)blockIdx.x * (kBlock * U)) / kByteTile;
      if (tot_in) atomicAdd(&tiles_inside[tile], tot_in);
      if (tot_cut) atomicAdd(&tiles_cut[tile], tot_cut);
    }
  }
}

// ---------------------------------------------------------------------------
// Classification with block culling.  The reference classifies cell by cell (cut.cpp:743-786); so did rounds 1-3,
// streaming the 16 B connectivity row of every one of the 805 M cells of the 512^3 mesh -- 12.9 GB, 2.8 ms, for a
// domain whose interface touches 0.35 % of the cells.  A block of kClassBlock consecutive cells whose vertices ALL have
// negative (all positive) level-set values consists of inside (outside) cells: that is decided from the block's
// mesh-static vertex summary (cfx_mesh_s::class_runs: the distinct vertices as runs of consecutive ids -- 4 to 12 runs on
// meshes whose numbering has any locality, ~0.7 KB of sign codes per block read coalesced) and written as one 1 KB
// run of the domain array; only blocks with vertices on both sides (or on the interface) read their connectivity and
// go cell by cell as before.  Same domain array bit for bit, any mesh; a block without a summary (more than kClassRuns
// runs: no locality in the numbering) always takes the cell-by-cell path.
// ---------------------------------------------------------------------------
template <int ND, int CELLS, int RUNS, int N, int H>
__global__ void __launch_bounds__(kBlock) class_summary_kernel(int64_t ncells, const int32_t* __restrict__ conn,
                                                               int32_t* __restrict__ nruns, int2* __restrict__ runs,
                                                               int2* __restrict__ sub_runs)
{
  constexpr int kClassBlock = CELLS, kClassRuns = RUNS; // (this kernel's block: the block proper, or a quarter of it)
  // the block's ND * kClassBlock vertex ids -> distinct ids (LDS hash set: a block of a mesh with any locality has a
  // few hundred) -> sorted (bitonic over N slots) -> runs of consecutive ids.  More than N distinct ids: no summary.
  constexpr int Q = N / kBlock;
  static_assert(N % kBlock == 0 && (H & (H - 1)) == 0 && H >= ND * CELLS, "summary table sizes");
  __shared__ int32_t s_h[H];
  __shared__ int32_t s_v[N];
  __shared__ int32_t s_start[kClassRuns + 1], s_d0[kClassRuns + 1];
  __shared__ int s_cnt;
  const int64_t c0 = (int64_t)blockIdx.x * kClassBlock;
  const int nloc = (int)(ncells - c0 < kClassBlock ? ncells - c0 : kClassBlock) * ND;
  for (int i = threadIdx.x; i < H; i += kBlock) s_h[i] = -1;
  for (int i = threadIdx.x; i < N; i += kBlock) s_v[i] = 0x7fffffff;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < nloc; i += kBlock)
  {
    const int32_t v = conn[c0 * ND + i];
    unsigned h = cfx_hash32((uint32_t)v) & (H - 1);
    for (int probe = 0; probe < H; ++probe)
    {
      const int32_t old = s_h[h];
      if (old == v) break;
      if (old == -1)
      {
        const int32_t prev = atomicCAS(&s_h[h], -1, v);
        if (prev == -1)
        {
          const int k = atomicAdd(&s_cnt, 1);
          if (k < N) s_v[k] = v;
          break;
        }
        if (prev == v) break;
      }
      h = (h + 1) & (H - 1);
    }
  }
  __syncthreads();
  const int ndist = s_cnt;
  if (ndist > N)
  {
    if (threadIdx.x == 0 && nruns) nruns[blockIdx.x] = -1;
    if (threadIdx.x < kClassRuns) runs[(int64_t)blockIdx.x * kClassRuns + threadIdx.x] = make_int2(0, threadIdx.x == 0 ? -1 : 0);
    if (sub_runs && threadIdx.x < kClassSub * kClassSubRuns)
      sub_runs[(int64_t)blockIdx.x * kClassSub * kClassSubRuns + threadIdx.x] = make_int2(0, threadIdx.x % kClassSubRuns == 0 ? -1 : 0);
    return;
  }
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1)
    {
      for (int i = threadIdx.x; i < N; i += kBlock)
      {
        const int p = i ^ j;
        if (p > i)
        {
          const int32_t a = s_v[i], b = s_v[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { s_v[i] = b; s_v[p] = a; }
        }
      }
      __syncthreads();
    }
  // the run starts among the (distinct, ascending) ids
  int isstart[Q], cs = 0;
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    const int i = threadIdx.x * Q + q;
    isstart[q] = (i < ndist && (i == 0 || s_v[i - 1] != s_v[i] - 1)) ? 1 : 0;
    cs += isstart[q];
  }
  int tot_s;
  int os = block_exclusive_scan<int>(cs, tot_s);
  if (tot_s <= kClassRuns)
  {
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      const int i = threadIdx.x * Q + q;
      if (isstart[q]) { s_start[os] = s_v[i]; s_d0[os] = i; ++os; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && nruns) nruns[blockIdx.x] = tot_s <= kClassRuns ? tot_s : -1;
  // every slot is written: unused ones hold length 0, a block without a summary length -1 in slot 0 (the classification
  // reads the slots alone -- one dependent load less per wavefront)
  if (threadIdx.x < kClassRuns)
  {
    const int j = threadIdx.x;
    int2 r = make_int2(0, 0);
    if (tot_s > kClassRuns) r.y = j == 0 ? -1 : 0;
    else if (j < tot_s) r = make_int2(s_start[j], (j + 1 < tot_s ? s_d0[j + 1] : ndist) - s_d0[j]);
    runs[(int64_t)blockIdx.x * kClassRuns + j] = r;
  }
  if (sub_runs == nullptr) return;
  // The quarter blocks: per run of the block the span [min, max] of the ids the quarter's cells refer to -- a superset of
  // the quarter's vertices when they do not fill the span, which keeps the culling exact (one sign on a superset is one
  // sign on the set) and costs a search and two LDS atomics per reference where a summary of its own cost a hash set
  // and a sort per quarter (36 ms at 512^3).  Quarters that touch more than kClassSubRuns runs get no summary.
  __shared__ int32_t s_lo[kClassSub][RUNS], s_hi[kClassSub][RUNS];
  const bool have = tot_s <= kClassRuns;
  for (int i = threadIdx.x; i < kClassSub * RUNS; i += kBlock) { (&s_lo[0][0])[i] = 0x7fffffff; (&s_hi[0][0])[i] = -1; }
  __syncthreads();
  if (have)
    for (int i = threadIdx.x; i < nloc; i += kBlock)
    {
      const int32_t v = conn[c0 * ND + i];
      const int sq = (i / ND) / (CELLS / kClassSub);
      int lo = 0, hi = tot_s; // last run whose start is <= v
      while (hi - lo > 1)
      {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= v) lo = mid; else hi = mid;
      }
      atomicMin(&s_lo[sq][lo], v);
      atomicMax(&s_hi[sq][lo], v);
    }
  __syncthreads();
  if (threadIdx.x < kClassSub)
  {
    const int sq = threadIdx.x;
    int2* out = sub_runs + ((int64_t)blockIdx.x * kClassSub + sq) * kClassSubRuns;
    int n = 0;
    bool ok = have;
    for (int j = 0; ok && j < tot_s; ++j)
      if (s_hi[sq][j] >= 0)
      {
        if (n == kClassSubRuns) { ok = false; break; }
        out[n++] = make_int2(s_lo[sq][j], s_hi[sq][j] - s_lo[sq][j] + 1);
      }
    if (!ok) { n = 0; out[n++] = make_int2(0, -1); }
    for (; n < kClassSubRuns; ++n) out[n] = make_int2(0, 0);
  }
}

// AND of the sign codes of the vertices a run table lists (RUNS slots of (start, length), one per lane; length 0: unused,
// -1 in slot 0: no summary -> 0).  The elements are numbered through (prefix sums of the lengths) and dealt to the lanes E
// at a time, so that a lane's loads are independent and in flight together: walking the runs one after the other left
// every wavefront ~12 dependent load latencies long.
template <int RUNS, int E>
__device__ __forceinline__ unsigned class_runs_and(const int2* __restrict__ table, const uint8_t* __restrict__ code, int lane)
{
  const int2 mine = lane < RUNS ? table[lane] : make_int2(0, 0);
  const bool summary = __shfl(mine.y, 0, 64) >= 0;
  const int nr = summary ? __popcll(__ballot(mine.y > 0)) : 0;
  unsigned all = summary ? 3u : 0u;
  const int incl = wave_inclusive_scan<int>(summary ? mine.y : 0);
  const int total = __shfl(incl, 63, 64);
  for (int t0 = 0; t0 < total; t0 += 64 * E)
  {
    int addr[E];
#pragma unroll
    for (int e = 0; e < E; ++e) addr[e] = -1;
    for (int j = 0; j < nr; ++j)
    {
      const int start = __shfl(mine.x, j, 64), hi = __shfl(incl, j, 64), lo = hi - __shfl(mine.y, j, 64);
#pragma unroll
      for (int e = 0; e < E; ++e)
      {
        const int t = t0 + lane + 64 * e;
        addr[e] = (t >= lo && t < hi) ? start + (t - lo) : addr[e];
      }
    }
    unsigned v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = addr[e] >= 0 ? (unsigned)code[addr[e]] : 3u;
#pragma unroll
    for (int e = 0; e < E; ++e) all &= v[e];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) all &= __shfl_xor(all, o, 64);
  return all;
}

// CELLS equal domain bytes from cbase on, written by the wavefront
template <int CELLS>
__device__ __forceinline__ void class_fill(int8_t* __restrict__ domain, int64_t cbase, int64_t ncells, int8_t d, int lane)
{
  const uint32_t w4 = 0x01010101u * (uint32_t)(uint8_t)d;
  // 16 B per lane when the run is whole and aligned: one store instruction per 1024 cells.  (The kernel is not bound by
  // its stores -- 0.78 -> 0.75 ms with every block uniform -- nor helped by a persistent grid with the next block's
  // runs prefetched: 1.5 ms; 786 k short wavefronts live on how many of them are in flight.)
  if (cbase + CELLS <= ncells && (reinterpret_cast<uintptr_t>(domain + cbase) & 15) == 0)
  {
    if (lane < CELLS / 16) reinterpret_cast<uint4*>(domain + cbase)[lane] = make_uint4(w4, w4, w4, w4);
    return;
  }
#pragma unroll
  for (int q4 = 0; q4 < CELLS / 256; ++q4)
  {
    const int64_t c = cbase + 4 * (lane + 64 * q4);
    if (c + 4 <= ncells && (reinterpret_cast<uintptr_t>(domain + c) & 3) == 0) *reinterpret_cast<uint32_t*>(domain + c) = w4;
    else
      for (int64_t q = c; q < c + 4 && q < ncells; ++q) domain[q] = d;
  }
}

// The same for TWO tables at once (lanes 0..31 hold the runs of table A, lanes 32..63 those of table B; RUNS <= 32): both
// tables arrive with one load latency and both sets of sign codes with a second one.  The culled classification is
// bound by how many of its short wavefronts are in flight (786 k of them at 512^3, two dependent loads each): a
// wavefront that decides two blocks halves their number.
template <int RUNS, int E>
__device__ __forceinline__ void class_runs_and2(const int2* __restrict__ tableA, const int2* __restrict__ tableB, bool haveB,
                                                const uint8_t* __restrict__ code, int lane, unsigned& allA, unsigned& allB)
{
  static_assert(RUNS <= 32, "two tables in one wavefront");
  const int half = lane >> 5, hl = lane & 31;
  int2 mine = make_int2(0, 0);
  if (hl < RUNS && (half == 0 || haveB)) mine = (half == 0 ? tableA : tableB)[hl];
  const bool sumA = __shfl(mine.y, 0, 64) >= 0, sumB = haveB && __shfl(mine.y, 32, 64) >= 0;
  const unsigned long long pos = __ballot(mine.y > 0);
  const int nrA = sumA ? __popcll(pos & 0xffffffffull) : 0, nrB = sumB ? __popcll(pos >> 32) : 0;
  const bool mysum = half == 0 ? sumA : sumB;
  int incl = wave_inclusive_scan<int>(mysum ? mine.y : 0);
  const int totA = __shfl(incl, 31, 64);
  const int totB = __shfl(incl, 63, 64) - totA;
  if (half) incl -= totA; // prefix inside the own table
  allA = sumA ? 3u : 0u; allB = sumB ? 3u : 0u;
  const int total = max(totA, totB);
  for (int t0 = 0; t0 < total; t0 += 64 * E)
  {
    int addrA[E], addrB[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { addrA[e] = -1; addrB[e] = -1; }
    for (int j = 0; j < max(nrA, nrB); ++j)
    {
      const int sa = __shfl(mine.x, j, 64), ha = __shfl(incl, j, 64), la = ha - __shfl(mine.y, j, 64);
      const int sb = __shfl(mine.x, 32 + j, 64), hb = __shfl(incl, 32 + j, 64), lb = hb - __shfl(mine.y, 32 + j, 64);
#pragma unroll
      for (int e = 0; e < E; ++e)
      {
        const int t = t0 + lane + 64 * e;
        addrA[e] = (j < nrA && t >= la && t < ha) ? sa + (t - la) : addrA[e];
        addrB[e] = (j < nrB && t >= lb && t < hb) ? sb + (t - lb) : addrB[e];
      }
    }
    unsigned va[E], vb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) va[e] = addrA[e] >= 0 ? (unsigned)code[addrA[e]] : 3u;
#pragma unroll
    for (int e = 0; e < E; ++e) vb[e] = addrB[e] >= 0 ? (unsigned)code[addrB[e]] : 3u;
#pragma unroll
    for (int e = 0; e < E; ++e) { allA &= va[e]; allB &= vb[e]; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { allA &= __shfl_xor(allA, o, 64); allB &= __shfl_xor(allB, o, 64); }
}

// a block with vertices on both sides (or on the interface): quarter by quarter, and only the quarters that are mixed
// themselves cell by cell (classify_kernel).  The whole wavefront works on the one block.
template <int ND>
__device__ __forceinline__ void classify_block_mixed(int64_t ncells, int64_t b, const int32_t* __restrict__ dofmap,
                                                     const int2* __restrict__ sub_runs, const uint8_t* __restrict__ code,
                                                     int8_t* __restrict__ domain, int32_t* tiles_inside, int32_t* tiles_cut,
                                                     uint8_t* __restrict__ touch, int lane)
{
  constexpr int SUB = kClassBlock / kClassSub, U = SUB / 64;
  const int64_t cbase = b * kClassBlock;
  const int64_t tile = cbase / kByteTile;
  int n_in = 0, n_cut = 0;
  // (the four quarter tables decided in one go -- 16 lanes per table, one round of loads -- measured in round 5: no change,
  // 1.056 against 1.044 ms at 512^3: the mixed blocks live on their cell loops)
  for (int sq = 0; sq < kClassSub; ++sq)
  {
    const int64_t sbase = cbase + (int64_t)sq * SUB;
    if (sbase >= ncells) break;
    const unsigned sall = class_runs_and<kClassSubRuns, 4>(sub_runs + (b * kClassSub + sq) * kClassSubRuns, code, lane);
    if (sall == 1u || sall == 2u)
    {
      class_fill<SUB>(domain, sbase, ncells, sall == 1u ? (int8_t)CFX_INSIDE : (int8_t)CFX_OUTSIDE, lane);
      if (sall == 1u && lane == 0) n_in += (int)(ncells - sbase < SUB ? ncells - sbase : SUB);
      continue;
    }
    int32_t d[U][ND];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t c = sbase + lane + (int64_t)u * 64;
      if (c >= ncells) continue;
      if constexpr (ND == 4)
      {
        const int4 v = *reinterpret_cast<const int4*>(dofmap + c * 4);
        d[u][0] = v.x; d[u][1] = v.y; d[u][2] = v.z; d[u][3] = v.w;
      }
      else
      {
#pragma unroll
        for (int i = 0; i < ND; ++i) d[u][i] = dofmap[c * ND + i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t c = sbase + lane + (int64_t)u * 64;
      if (c >= ncells) continue;
      unsigned a = 3u;
#pragma unroll
      for (int i = 0; i < ND; ++i) a &= code[d[u][i]];
      domain[c] = a == 1u ? (int8_t)CFX_INSIDE : (a == 2u ? (int8_t)CFX_OUTSIDE : (int8_t)CFX_INTERSECTED);
      n_in += a == 1u ? 1 : 0;
      n_cut += (a != 1u && a != 2u) ? 1 : 0;
      if (touch && a != 1u && a != 2u) // (a cut cell: its dofs are next to the interface; every writer stores 1)
      {
#pragma unroll
        for (int i = 0; i < ND; ++i) touch[d[u][i]] = 1;
      }
    }
  }
  if (tiles_inside)
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n_in += __shfl_xor(n_in, o, 64); n_cut += __shfl_xor(n_cut, o, 64); }
    if (lane == 0)
    {
      if (n_in) atomicAdd(&tiles_inside[tile], n_in);
      if (n_cut) atomicAdd(&tiles_cut[tile], n_cut);
    }
  }
}

// one wavefront per PAIR of blocks of kClassBlock cells
#ifndef CFX_CLASSIFY_WAVES
#define CFX_CLASSIFY_WAVES 5 // (round 5, two blocks per wavefront, 512^3 sphere: 4 -> 1.21 ms, 5 -> 1.04, 6 -> 1.09)
#endif
#ifndef CFX_CLASSIFY_E
#define CFX_CLASSIFY_E 12 // sign codes per lane, table and round of loads (a block has ~690: one round)
#endif
template <int ND>
__global__ void __launch_bounds__(kBlock, CFX_CLASSIFY_WAVES) classify_culled_kernel(int64_t ncells, int64_t nblocks, const int32_t* __restrict__ dofmap,
                                                                 const int2* __restrict__ runs, const int2* __restrict__ sub_runs,
                                                                 const uint8_t* __restrict__ code, int8_t* __restrict__ domain,
                                                                 int32_t* tiles_inside, int32_t* tiles_cut,
                                                                 uint8_t* __restrict__ block_class, uint8_t* __restrict__ touch)
{
  static_assert(kByteTile % kClassBlock == 0, "classification blocks must nest in compaction tiles");
  const int lane = threadIdx.x & 63;
  const int64_t b0 = 2 * ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
  if (b0 >= nblocks) return;
  const bool have1 = b0 + 1 < nblocks;
  unsigned all[2];
  class_runs_and2<kClassRuns, CFX_CLASSIFY_E>(runs + b0 * kClassRuns, runs + (b0 + 1) * kClassRuns, have1, code, lane, all[0], all[1]);
  for (int k = 0; k < (have1 ? 2 : 1); ++k)
  {
    const int64_t b = b0 + k;
    const int64_t cbase = b * kClassBlock;
    const unsigned a = all[k];
    if (a == 1u || a == 2u)
    {
      // every vertex of the block on one side
      class_fill<kClassBlock>(domain, cbase, ncells, a == 1u ? (int8_t)CFX_INSIDE : (int8_t)CFX_OUTSIDE, lane);
      if (tiles_inside && a == 1u && lane == 0)
        atomicAdd(&tiles_inside[cbase / kByteTile], (int32_t)(ncells - cbase < kClassBlock ? ncells - cbase : kClassBlock));
      if (block_class && lane == 0) block_class[b] = (uint8_t)a;
      continue;
    }
    if (block_class && lane == 0) block_class[b] = 0;
    classify_block_mixed<ND>(ncells, b, dofmap, sub_runs, code, domain, tiles_inside, tiles_cut, touch, lane);
  }
}

// Implicit-structured variant of a1 for the generated box / slab meshes with the level set on the geometry dofmap
// (CFX_IMPLICIT_BOX=1): the connectivity of a Kuhn-split cube is a function of the cube index, so one thread
// classifies the cells of a cube from its 2^tdim corner codes and the 12.9 GB connectivity stream of
// classify_kernel (16 of its 18.3 algorithmic bytes per cell) is not read at all.  Same domain array.
template <int TDIM>
__global__ void __launch_bounds__(kBlock) classify_box_kernel(int64_t ncubes, int n, const uint8_t* __restrict__ code,
                                                              int8_t* __restrict__ domain, int32_t* tiles_inside,
                                                              int32_t* tiles_cut, uint8_t* __restrict__ touch)
{
  constexpr int NC = TDIM == 3 ? 6 : 2; // cells per cube
  constexpr int NV = TDIM + 1;
  // corners of the Kuhn simplices, local corner i = bx + 2 by + 4 bz (the generator's tables, cfx_mesh.hip)
  constexpr int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};
  constexpr int tri[2][3] = {{0, 1, 3}, {0, 3, 2}};
  __shared__ int8_t s_dom[kBlock * NC];
  __shared__ int s_cnt[4]; // inside / cut cells of the block in the two compaction tiles it can overlap
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t h = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t tile0 = ((int64_t)blockIdx.x * kBlock * NC) / kByteTile;
  if (h < ncubes)
  {
    const int64_t n1 = n + 1;
    const int64_t ix = h % n, iy = (h / n) % n, iz = TDIM == 3 ? h / ((int64_t)n * n) : 0;
    const int64_t v0 = ix + n1 * (iy + n1 * iz);
    unsigned c[1 << TDIM];
#pragma unroll
    for (int i = 0; i < (1 << TDIM); ++i)
      c[i] = code[v0 + (i & 1) + n1 * (((i >> 1) & 1) + n1 * ((i >> 2) & 1))];
#pragma unroll
    for (int k = 0; k < NC; ++k)
    {
      unsigned all = 3u;
#pragma unroll
      for (int j = 0; j < NV; ++j) all &= c[TDIM == 3 ? tet[k][j] : tri[k][j]];
      s_dom[threadIdx.x * NC + k] = all == 1u ? (int8_t)CFX_INSIDE : (all == 2u ? (int8_t)CFX_OUTSIDE : (int8_t)CFX_INTERSECTED);
      if (touch && all != 1u && all != 2u) // (a cut cell: its vertices are next to the interface; every writer stores 1)
      {
#pragma unroll
        for (int j = 0; j < NV; ++j)
        {
          const int i = TDIM == 3 ? tet[k][j] : tri[k][j];
          touch[v0 + (i & 1) + n1 * (((i >> 1) & 1) + n1 * ((i >> 2) & 1))] = 1;
        }
      }
      if (tiles_inside && all != 2u)
        atomicAdd(&s_cnt[2 * (int)((h * NC + k) / kByteTile - tile0) + (all == 1u ? 0 : 1)], 1);
    }
  }
  __syncthreads();
  if (tiles_inside && threadIdx.x < 4 && s_cnt[threadIdx.x])
    atomicAdd((threadIdx.x & 1 ? tiles_cut : tiles_inside) + tile0 + (threadIdx.x >> 1), s_cnt[threadIdx.x]);
  // the block's cells are contiguous: coalesced byte stores out of LDS
  const int64_t first = (int64_t)blockIdx.x * kBlock * NC;
  const int64_t count = min((int64_t)kBlock, ncubes - (int64_t)blockIdx.x * kBlock) * NC;
  for (int i = threadIdx.x; i < count; i += kBlock) domain[first + i] = s_dom[i];
}

// ---------------------------------------------------------------------------
// a2 sub-triangulation tables.  A vertex is "negative" iff phi < 0 (zeros side
// with the positive part).  Local point ids: 0..tdim parent vertices, then the
// cut points on edges (a negative, b non-negative).  Built once on the host.
// ---------------------------------------------------------------------------
struct CutCase
{
  int8_t n_in, n_out, n_if, npts;
  int8_t edge[4][2];
  int8_t in[3][4];
  int8_t out[3][4];
  int8_t iface[2][3];
};

__constant__ CutCase c_cases[2][16];

void prism(int8_t dst[3][4], int a0, int a1, int a2, int b0, int b1, int b2)
{
  const int t[3][4] = {{a0, a1, a2, b2}, {a0, a1, b1, b2}, {a0, b0, b1, b2}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) dst[i][j] = (int8_t)t[i][j];
}

CutCase make_case(int tdim, int mask)
{
  CutCase s;
  memset(&s, 0, sizeof(s));
  const int nv = tdim + 1;
  int neg[4], pos[4], nn = 0, np = 0;
  for (int v = 0; v < nv; ++v)
    if ((mask >> v) & 1) neg[nn++] = v; else pos[np++] = v;
  int npts = nv;
  auto cutp = [&](int a, int b)
  {
    s.edge[npts - nv][0] = (int8_t)a; s.edge[npts - nv][1] = (int8_t)b;
    return npts++;
  };
  auto set = [](int8_t* d, int a, int b, int c, int e = 0) { d[0] = (int8_t)a; d[1] = (int8_t)b; d[2] = (int8_t)c; d[3] = (int8_t)e; };
  auto set3 = [](int8_t* d, int a, int b, int c) { d[0] = (int8_t)a; d[1] = (int8_t)b; d[2] = (int8_t)c; };
  if (tdim == 2)
  {
    if (nn == 0) { s.n_out = 1; set(s.out[0], 0, 1, 2); }
    else if (nn == 3) { s.n_in = 1; set(s.in[0], 0, 1, 2); }
    else if (nn == 1)
    {
      const int a = neg[0], b0 = pos[0], b1 = pos[1];
      const int q0 = cutp(a, b0), q1 = cutp(a, b1);
      s.n_in = 1; set(s.in[0], a, q0, q1);
      s.n_out = 2; set(s.out[0], q0, b0, b1); set(s.out[1], q0, b1, q1);
      s.n_if = 1; set3(s.iface[0], q0, q1, 0);
    }
    else
    {
      const int a0 = neg[0], a1 = neg[1], b = pos[0];
      const int q0 = cutp(a0, b), q1 = cutp(a1, b);
      s.n_in = 2; set(s.in[0], a0, a1, q1); set(s.in[1], a0, q1, q0);
      s.n_out = 1; set(s.out[0], b, q0, q1);
      s.n_if = 1; set3(s.iface[0], q0, q1, 0);
    }
  }
  else
  {
    if (nn == 0) { s.n_out = 1; set(s.out[0], 0, 1, 2, 3); }
    else if (nn == 4) { s.n_in = 1; set(s.in[0], 0, 1, 2, 3); }
    else if (nn == 1)
    {
      const int a = neg[0];
      const int q0 = cutp(a, pos[0]), q1 = cutp(a, pos[1]), q2 = cutp(a, pos[2]);
      s.n_in = 1; set(s.in[0], a, q0, q1, q2);
      s.n_out = 3; prism(s.out, q0, q1, q2, pos[0], pos[1], pos[2]);
      s.n_if = 1; set3(s.iface[0], q0, q1, q2);
    }
    else if (nn == 3)
    {
      const int b = pos[0];
      const int q0 = cutp(neg[0], b), q1 = cutp(neg[1], b), q2 = cutp(neg[2], b);
      s.n_in = 3; prism(s.in, q0, q1, q2, neg[0], neg[1], neg[2]);
      s.n_out = 1; set(s.out[0], b, q0, q1, q2);
      s.n_if = 1; set3(s.iface[0], q0, q1, q2);
    }
    else
    {
      const int a0 = neg[0], a1 = neg[1], b0 = pos[0], b1 = pos[1];
      const int q00 = cutp(a0, b0), q01 = cutp(a0, b1), q10 = cutp(a1, b0), q11 = cutp(a1, b1);
      s.n_in = 3; prism(s.in, a0, q00, q01, a1, q10, q11);
      s.n_out = 3; prism(s.out, b0, q00, q10, b1, q01, q11);
      s.n_if = 2; set3(s.iface[0], q00, q01, q11); set3(s.iface[1], q00, q11, q10);
    }
  }
  s.npts = (int8_t)npts;
  return s;
}

CutCase h_cases[2][16];
bool g_cases_ready = false;

void ensure_cases()
{
  if (g_cases_ready) return;
  for (int t = 2; t <= 3; ++t)
    for (int m = 0; m < 16; ++m) h_cases[t - 2][m] = make_case(t, m & ((1 << (t + 1)) - 1));
  CFX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_cases), h_cases, sizeof(h_cases)));
  g_cases_ready = true;
}

enum Part { PART_IN = 0, PART_OUT = 1, PART_IF = 2 };

template <int TDIM>
__device__ __forceinline__ int sign_mask(const double* phi)
{
  int m = 0;
#pragma unroll
  for (int v = 0; v <= TDIM; ++v) m |= (phi[v] < 0.0) ? (1 << v) : 0;
  return m;
}

constexpr int kPackShift = cfx::kCountPackShift; // (points, rules) of a cut cell packed into one int64 (see cfx_runtime_quadrature)
constexpr int64_t kPackMask = (1ll << kPackShift) - 1;

// per cut cell: number of rules and points it will emit for each of the (one or two) parts asked for
struct CountParts
{
  int n;
  int part[2], nref[2];
  int64_t* packed[2];
};
template <int TDIM>
__global__ void __launch_bounds__(kBlock) cut_count_kernel(DevN ncut_d, const int32_t* __restrict__ cut_cells,
                                                           const int32_t* __restrict__ ls_dofmap,
                                                           const double* __restrict__ phi_v, CountParts P)
{
  const int64_t ncut = dev_n(ncut_d);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncut)
  {
    if (i < ncut_d.cap) // list shorter than its capacity: the scans run over the capacity
      for (int k = 0; k < P.n; ++k) P.packed[k][i] = 0;
    return;
  }
  const int64_t c = cut_cells[i];
  double phi[TDIM + 1];
#pragma unroll
  for (int v = 0; v <= TDIM; ++v) phi[v] = phi_v[ls_dofmap[c * (TDIM + 1) + v]];
  const CutCase& cs = c_cases[TDIM - 2][sign_mask<TDIM>(phi)];
  for (int k = 0; k < P.n; ++k)
  {
    const int part = P.part[k];
    const int ns = part == PART_IN ? cs.n_in : (part == PART_OUT ? cs.n_out : cs.n_if);
    // points in the low kPackShift bits, rules above: one scan gives both offsets
    P.packed[k][i] = (int64_t)(ns * P.nref[k]) | ((int64_t)(part == PART_IF ? ns : (ns > 0 ? 1 : 0)) << kPackShift);
  }
}

// parent-reference coordinates of local point p of the cut case
template <int TDIM>
__device__ __forceinline__ void local_point(const CutCase& cs, int p, const double* phi, double* X)
{
#pragma unroll
  for (int d = 0; d < TDIM; ++d) X[d] = 0.0;
  if (p <= TDIM)
  {
#pragma unroll
    for (int d = 0; d < TDIM; ++d) X[d] = (p == d + 1) ? 1.0 : 0.0;
    return;
  }
  const int a = cs.edge[p - (TDIM + 1)][0], b = cs.edge[p - (TDIM + 1)][1];
  double pa = 0.0, pb = 0.0;
#pragma unroll
  for (int v = 0; v <= TDIM; ++v)
  {
    pa = (v == a) ? phi[v] : pa;
    pb = (v == b) ? phi[v] : pb;
  }
  const double t = pa / (pa - pb); // a negative, b non-negative: t in [0,1]
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    const double xa = (a == d + 1) ? 1.0 : 0.0, xb = (b == d + 1) ? 1.0 : 0.0;
    X[d] = xa + t * (xb - xa);
  }
}

template <int TDIM>
__device__ __forceinline__ const double* ref_points(int dim, int degree, int& n, const double*& w)
{
  if (dim == 1)
  {
    n = cfx_quad_offset_1d[degree + 1] - cfx_quad_offset_1d[degree];
    w = cfx_quad_weights_1d + cfx_quad_offset_1d[degree];
    return cfx_quad_points_1d + cfx_quad_offset_1d[degree];
  }
  if (dim == 2)
  {
    n = cfx_quad_offset_2d[degree + 1] - cfx_quad_offset_2d[degree];
    w = cfx_quad_weights_2d + cfx_quad_offset_2d[degree];
    return cfx_quad_points_2d + 2 * cfx_quad_offset_2d[degree];
  }
  n = cfx_quad_offset_3d[degree + 1] - cfx_quad_offset_3d[degree];
  w = cfx_quad_weights_3d + cfx_quad_offset_3d[degree];
  return cfx_quad_points_3d + 3 * cfx_quad_offset_3d[degree];
}

// ---------------------------------------------------------------------------
// a3 with several level sets: runtime_quadrature(cut([phi, phi1, ...]), "phi<0 and phi1>0", k)
// (cpp/cutfemx/cut/cut.h:122-181, docs/user-guide/element-classification.md:145-160).  One conjunction of
// clauses; each P1 level set is planar inside a cell, so the cell's part of the region is the parent simplex
// clipped by one half-space per clause, in the order the clauses are written: every simplex of the current list
// is sub-triangulated with the single-level-set case tables (phi_k interpolated at its vertices) and its negative
// ("<") or positive (">") part kept.  With an "=0" clause the list starts from that level set's
// interface sub-facets and the other clauses clip them.  One thread per rule cell, the simplex lists in
// private memory: a few thousand cells along the curves / surfaces where two level sets meet, not a hot path.
// ---------------------------------------------------------------------------
constexpr int kMultiMaxClauses = 4; // clauses of a conjunction (at most 3 of them clip)
constexpr int kMultiMaxSimp = 27;   // 3 clips of 3 sub-simplices each

struct MultiArgs
{
  int64_t n;                  // rule cells
  const int32_t* cells;
  const double* x;
  const int32_t* conn;
  const int32_t* ls_dofmap;
  const double* phi[kMultiMaxClauses]; // dof values of the clause's level set
  int ncl, eq;                // eq: index of the "=0" clause or -1
  int mask[kMultiMaxClauses];
  int order;
  int64_t* packed;            // count pass: (points | rules << kPackShift) per cell
  const int64_t* packed_off;  // emit pass
  double* points;
  double* weights;
  int32_t* offsets;
  int32_t* parent_map;
};

struct MultiRuleCell
{
  const int8_t* domain;
  int64_t ncells;
  int ncl;
  int ls[kMultiMaxClauses], mask[kMultiMaxClauses];
  __device__ bool operator()(int64_t c) const
  {
    bool ok = true, any_cut = false;
    for (int k = 0; k < ncl; ++k)
    {
      const int d = domain[(int64_t)ls[k] * ncells + c];
      if (d == CFX_INTERSECTED) any_cut = true;
      else if (d == CFX_NOT_CANDIDATE || mask[k] == 2 || !((mask[k] >> (d + 1)) & 1)) ok = false;
    }
    return ok && any_cut;
  }
};

template <int TDIM>
struct MSimp
{
  double V[TDIM + 1][TDIM]; // vertices in parent reference coordinates (DIM + 1 of them used)
};

// the psi < 0 part (keep_out false) or the psi > 0 part (keep_out: the "out" simplices of the same case, so that a
// ">" clause triangulates exactly like the single-level-set "phi>0" rules) of simplex `in` (dimension DIM),
// appended to out[m...]; returns the new m
template <int TDIM, int DIM>
__device__ int multi_clip_one(const MSimp<TDIM>& in, const double* psi, bool keep_out, MSimp<TDIM>* out, int m)
{
  if constexpr (DIM == 1)
  {
    const bool n0 = psi[0] < 0.0, n1 = psi[1] < 0.0;
    if (n0 == n1)
    {
      if (n0 != keep_out) { out[m] = in; return m + 1; } // wholly on the kept side
      return m;
    }
    // a = the negative end, b = the other; cut point q; negative part (a, q), positive part (q, b)
    const int a = n0 ? 0 : 1, b = 1 - a;
    const double t = psi[a] / (psi[a] - psi[b]);
    const double La = (a == 1) ? 1.0 : 0.0, Lb = (b == 1) ? 1.0 : 0.0; // local coordinate of the two ends
    const double Lq = La + t * (Lb - La);
    const double L0 = keep_out ? Lq : La, L1 = keep_out ? Lb : Lq;
#pragma unroll
    for (int d = 0; d < TDIM; ++d)
    {
      out[m].V[0][d] = in.V[0][d] + L0 * (in.V[1][d] - in.V[0][d]);
      out[m].V[1][d] = in.V[0][d] + L1 * (in.V[1][d] - in.V[0][d]);
    }
    return m + 1;
  }
  else
  {
    int sm = 0;
#pragma unroll
    for (int v = 0; v <= DIM; ++v) sm |= (psi[v] < 0.0) ? (1 << v) : 0;
    if (sm == (keep_out ? 0 : (1 << (DIM + 1)) - 1)) { out[m] = in; return m + 1; } // wholly on the kept side: unchanged
    const CutCase& cs = c_cases[DIM - 2][sm];
    const int ns = keep_out ? cs.n_out : cs.n_in;
    for (int k = 0; k < ns; ++k)
    {
      for (int v = 0; v <= DIM; ++v)
      {
        double L[DIM];
        local_point<DIM>(cs, keep_out ? cs.out[k][v] : cs.in[k][v], psi, L); // the sub-simplex vertex in the coordinates of `in`
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          double xx = in.V[0][d];
#pragma unroll
          for (int t = 0; t < DIM; ++t) xx += L[t] * (in.V[t + 1][d] - in.V[0][d]);
          out[m].V[v][d] = xx;
        }
      }
      ++m;
    }
    return m;
  }
}

template <int TDIM>
__device__ __forceinline__ double multi_phi_at(const double* phi, const double* X)
{
  double l0 = 1.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t) l0 -= X[t];
  double v = l0 * phi[0];
#pragma unroll
  for (int t = 0; t < TDIM; ++t) v += X[t] * phi[t + 1];
  return v;
}

// DIM = TDIM: volume rules; DIM = TDIM - 1: rules on the interface of clause A.eq
template <int TDIM, int DIM, bool EMIT>
__global__ void __launch_bounds__(kBlock) multi_rules_kernel(MultiArgs A)
{
  constexpr int NV = TDIM + 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= A.n) return;
  const int64_t c = A.cells[i];
  double phi[kMultiMaxClauses][NV];
  for (int k = 0; k < A.ncl; ++k)
#pragma unroll
    for (int v = 0; v < NV; ++v) phi[k][v] = A.phi[k][A.ls_dofmap[c * NV + v]];
  // base list: the parent simplex, or the interface sub-facets of the "=0" level set
  MSimp<TDIM> base[2];
  int nbase = 1;
  if constexpr (DIM == TDIM)
  {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int d = 0; d < TDIM; ++d) base[0].V[v][d] = (v == d + 1) ? 1.0 : 0.0;
  }
  else
  {
    const CutCase& cs = c_cases[TDIM - 2][sign_mask<TDIM>(phi[A.eq])];
    nbase = cs.n_if;
    for (int f = 0; f < nbase; ++f)
      for (int v = 0; v < TDIM; ++v) local_point<TDIM>(cs, cs.iface[f][v], phi[A.eq], base[f].V[v]);
  }
  int nref;
  const double* wref;
  const double* pref = ref_points<TDIM>(DIM, A.order, nref, wref);
  Geo<TDIM> g;
  int64_t pq = 0, pr = 0;
  if constexpr (EMIT)
  {
    load_cell<TDIM>(A.x, A.conn, c, g);
    jacobian<TDIM>(g);
    const int64_t po = A.packed_off[i];
    pq = po & kPackMask; pr = po >> kPackShift;
  }
  int total_pts = 0, total_rules = 0;
  const int ngroups = DIM == TDIM ? 1 : nbase;
  MSimp<TDIM> la[kMultiMaxSimp], lb[kMultiMaxSimp];
  for (int gi = 0; gi < ngroups; ++gi)
  {
    MSimp<TDIM>* cur = la;
    MSimp<TDIM>* nxt = lb;
    int n = 1;
    cur[0] = base[gi];
    for (int k = 0; k < A.ncl && n > 0; ++k)
    {
      if (k == A.eq) continue;
      const bool keep_out = !(A.mask[k] & 1);
      int m = 0;
      for (int q = 0; q < n; ++q)
      {
        double psi[DIM + 1];
#pragma unroll
        for (int v = 0; v <= DIM; ++v) psi[v] = multi_phi_at<TDIM>(phi[k], cur[q].V[v]);
        m = multi_clip_one<TDIM, DIM>(cur[q], psi, keep_out, nxt, m);
      }
      n = m;
      MSimp<TDIM>* t = cur; cur = nxt; nxt = t;
    }
    if (n == 0) continue;
    total_rules += 1;
    total_pts += n * nref;
    if constexpr (EMIT)
    {
      for (int q = 0; q < n; ++q)
      {
        double scale;
        if constexpr (DIM == TDIM)
        {
          double e[TDIM][TDIM];
#pragma unroll
          for (int t = 0; t < TDIM; ++t)
#pragma unroll
            for (int d = 0; d < TDIM; ++d) e[t][d] = cur[q].V[t + 1][d] - cur[q].V[0][d];
          double det;
          if constexpr (TDIM == 2) det = e[0][0] * e[1][1] - e[0][1] * e[1][0];
          else
            det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0])
                  + e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
          scale = fabs(det) * fabs(g.detJ);
        }
        else
        {
          // physical vertices of the piece: x_0 + J X, J[d][t] = x_{t+1,d} - x_{0,d}
          double xp[TDIM][TDIM];
#pragma unroll
          for (int v = 0; v < TDIM; ++v)
#pragma unroll
            for (int d = 0; d < TDIM; ++d)
            {
              double xx = g.x[0][d];
#pragma unroll
              for (int t = 0; t < TDIM; ++t) xx += (g.x[t + 1][d] - g.x[0][d]) * cur[q].V[v][t];
              xp[v][d] = xx;
            }
          if constexpr (TDIM == 2)
          {
            const double dx = xp[1][0] - xp[0][0], dy = xp[1][1] - xp[0][1];
            scale = sqrt(dx * dx + dy * dy);
          }
          else
          {
            double a[3], b[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { a[d] = xp[1][d] - xp[0][d]; b[d] = xp[2][d] - xp[0][d]; }
            const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
            scale = sqrt(cx * cx + cy * cy + cz * cz);
          }
        }
        for (int r = 0; r < nref; ++r)
        {
          double l0 = 1.0;
#pragma unroll
          for (int t = 0; t < DIM; ++t) l0 -= pref[r * DIM + t];
#pragma unroll
          for (int d = 0; d < TDIM; ++d)
          {
            double v = l0 * cur[q].V[0][d];
#pragma unroll
            for (int t = 0; t < DIM; ++t) v += pref[r * DIM + t] * cur[q].V[t + 1][d];
            A.points[(pq + r) * TDIM + d] = v;
          }
          A.weights[pq + r] = wref[r] * scale;
        }
        pq += nref;
      }
      A.parent_map[pr] = (int32_t)c;
      A.offsets[pr + 1] = (int32_t)pq;
      ++pr;
    }
  }
  if constexpr (!EMIT) A.packed[i] = (int64_t)total_pts | ((int64_t)total_rules << kPackShift);
}

#ifndef CFX_EMIT_LANES
#define CFX_EMIT_LANES 4
#endif
constexpr int kEmitLanes = CFX_EMIT_LANES; // lanes per cut cell (64 = one wavefront per cell)
static_assert(kEmitLanes >= 4 && 64 % kEmitLanes == 0, "lanes 0..tdim of a group stage the cell's vertices");

// ---------------------------------------------------------------------------
// a2+a3 emit: a lane group per cut cell.  Lanes 0..tdim stage the cell's
// vertex coordinates and level-set values in LDS; every lane then owns one
// (sub-simplex, reference point) pair, so the wave writes one contiguous run
// of points/weights.  points = parent-reference coords; weights = physical
// measure (w_ref |det sub->parent| |det parent->phys|, surface measure for the
// interface).  Volume parts: one rule per cut cell; interface: one rule per
// sub-facet (cut.cpp:1286-1294).
// ---------------------------------------------------------------------------
struct EmitJobs
{
  int n;
  int part[2];
  const int64_t* packed_off[2];
  double* points[2];
  double* weights[2];
  int32_t* offsets[2];
  int32_t* parent_map[2];
};

#define CFX_EMIT_STORE(dst, val) (dst) = (val)
#ifndef CFX_EMIT_WAVES
#define CFX_EMIT_WAVES 4 // waves per SIMD the emit kernel is compiled for (112 registers in 3-D without a bound: 4)
#endif
template <int TDIM>
__global__ void __launch_bounds__(kBlock, CFX_EMIT_WAVES) cut_emit_kernel(
    DevN ncut_d, const int32_t* __restrict__ cut_cells, const double* __restrict__ x,
    const int32_t* __restrict__ conn, const int32_t* __restrict__ ls_dofmap, const double* __restrict__ phi_v,
    int degree, EmitJobs jobs)
{
  const int64_t ncut = dev_n(ncut_d);
  constexpr int NV = TDIM + 1;
  // kEmitLanes lanes share one cut cell (16 cells per wavefront): a cell emits
  // 6-42 points; measured at 256^3: 64 lanes 705 us, 32 -> 445, 16 -> 293, 8 -> 247: more cells in
  // flight wins for this latency-bound kernel; with the local points staged in LDS, at 512^3:
  // 16 lanes 928 us, 8 -> 763, 4 -> 648
  __shared__ double s_phi[kBlock / kEmitLanes][NV];
  __shared__ double s_x[kBlock / kEmitLanes][NV][TDIM];
  const int wave = threadIdx.x / kEmitLanes, lane = threadIdx.x % kEmitLanes; // group, lane in group
  const int64_t i = (int64_t)blockIdx.x * (kBlock / kEmitLanes) + wave;
  const bool live = i < ncut;
  const int64_t c = live ? cut_cells[i] : 0;
  if (live && lane < NV)
  {
    const int64_t v = conn[c * NV + lane];
    s_phi[wave][lane] = phi_v[ls_dofmap[c * NV + lane]];
#pragma unroll
    for (int d = 0; d < TDIM; ++d) s_x[wave][lane][d] = x[3 * v + d];
  }
  __syncthreads();

  double phi[NV];
  Geo<TDIM> g;
#pragma unroll
  for (int v = 0; v < NV; ++v)
  {
    phi[v] = s_phi[wave][v];
#pragma unroll
    for (int d = 0; d < TDIM; ++d) g.x[v][d] = s_x[wave][v][d];
  }
  const CutCase& cs = c_cases[TDIM - 2][sign_mask<TDIM>(phi)];
  // the cell's local points (vertices + edge cut points, at most 2 NV) once per cell: one lane each, so the
  // division of a cut point is done once instead of once per (quadrature point, sub-simplex vertex)
  __shared__ double s_P[kBlock / kEmitLanes][2 * NV][TDIM];
  if (live)
    for (int p = lane; p < cs.npts; p += kEmitLanes)
    {
      double X[TDIM];
      local_point<TDIM>(cs, p, phi, X);
#pragma unroll
      for (int d = 0; d < TDIM; ++d) s_P[wave][p][d] = X[d];
    }
  __syncthreads();
  if (!live) return;
  // one or two rule sets of the same cut (runtime_quadratures: e.g. "phi<0" and "phi=0"): the staging above is shared
  // (the offsets of both sets are requested before the first store: a load issued behind the first set's stores waits
  // for them -- vmcnt counts loads and stores in order on this architecture)
  int64_t po_job[2];
#pragma unroll
  for (int job = 0; job < 2; ++job) po_job[job] = job < jobs.n ? jobs.packed_off[job][i] : 0;
  for (int job = 0; job < jobs.n; ++job)
  {
  const int part = jobs.part[job];
  double* __restrict__ points = jobs.points[job];
  double* __restrict__ weights = jobs.weights[job];
  int32_t* __restrict__ offsets = jobs.offsets[job];
  int32_t* __restrict__ parent_map = jobs.parent_map[job];
  const int ns = part == PART_IN ? cs.n_in : (part == PART_OUT ? cs.n_out : cs.n_if);
  // the last cut cell closes the parent list with a sentinel (the list is allocated one entry longer): kernels that
  // walk "the rules of cell c" stop there without knowing the number of rules, which may still be in HBM
  const int64_t po = job == 0 ? po_job[0] : po_job[1];
  if (lane == 0 && i == ncut - 1)
    parent_map[(po >> kPackShift) + (part == PART_IF ? ns : (ns > 0 ? 1 : 0))] = -1;
  if (ns == 0) continue;
  int nref;
  const double* wref;
  const double* pref = ref_points<TDIM>(part == PART_IF ? TDIM - 1 : TDIM, degree, nref, wref);
  const int npts = ns * nref;
  const int32_t pbase = (int32_t)(po & kPackMask), rbase = (int32_t)(po >> kPackShift);

  // J[d][t] = x_{t+1}[d] - x_0[d]
  double J[TDIM][TDIM];
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
#pragma unroll
    for (int t = 0; t < TDIM; ++t) J[d][t] = g.x[t + 1][d] - g.x[0][d];
  double detJ;
  if constexpr (TDIM == 2) detJ = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  else
    detJ = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2])
           + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  detJ = fabs(detJ);

  for (int pt = lane; pt < npts; pt += kEmitLanes)
  {
    const int k = pt / nref, q = pt - k * nref;
    double V[NV][TDIM];
    if (part == PART_IF)
    {
#pragma unroll
      for (int j = 0; j < TDIM; ++j)
      {
        const int p = cs.iface[k][j];
#pragma unroll
        for (int d = 0; d < TDIM; ++d) V[j][d] = s_P[wave][p][d];
      }
      // physical sub-facet vertices -> surface measure
      double xp[TDIM][TDIM];
#pragma unroll
      for (int j = 0; j < TDIM; ++j)
#pragma unroll
        for (int d = 0; d < TDIM; ++d)
        {
          double v = g.x[0][d];
#pragma unroll
          for (int t = 0; t < TDIM; ++t) v += J[d][t] * V[j][t];
          xp[j][d] = v;
        }
      double scale;
      if constexpr (TDIM == 2)
      {
        const double dx = xp[1][0] - xp[0][0], dy = xp[1][1] - xp[0][1];
        scale = sqrt(dx * dx + dy * dy);
      }
      else
      {
        double a[3], b[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { a[d] = xp[1][d] - xp[0][d]; b[d] = xp[2][d] - xp[0][d]; }
        const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
        scale = sqrt(cx * cx + cy * cy + cz * cz);
      }
      const double* xi = pref + (TDIM - 1) * q;
      double l0 = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM - 1; ++t) l0 -= xi[t];
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = l0 * V[0][d];
#pragma unroll
        for (int t = 0; t < TDIM - 1; ++t) v += xi[t] * V[t + 1][d];
        CFX_EMIT_STORE(points[(int64_t)(pbase + pt) * TDIM + d], v);
      }
      CFX_EMIT_STORE(weights[pbase + pt], wref[q] * scale);
    }
    else
    {
      const int8_t* sx = part == PART_IN ? cs.in[k] : cs.out[k];
#pragma unroll
      for (int j = 0; j < NV; ++j)
      {
        const int p = sx[j];
#pragma unroll
        for (int d = 0; d < TDIM; ++d) V[j][d] = s_P[wave][p][d];
      }
      double dsub;
      if constexpr (TDIM == 2)
        dsub = (V[1][0] - V[0][0]) * (V[2][1] - V[0][1]) - (V[1][1] - V[0][1]) * (V[2][0] - V[0][0]);
      else
      {
        double a[3], b[3], e[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { a[d] = V[1][d] - V[0][d]; b[d] = V[2][d] - V[0][d]; e[d] = V[3][d] - V[0][d]; }
        dsub = a[0] * (b[1] * e[2] - b[2] * e[1]) - a[1] * (b[0] * e[2] - b[2] * e[0]) + a[2] * (b[0] * e[1] - b[1] * e[0]);
      }
      const double scale = fabs(dsub) * detJ;
      const double* xi = pref + TDIM * q;
      double l0 = 1.0;
#pragma unroll
      for (int t = 0; t < TDIM; ++t) l0 -= xi[t];
#pragma unroll
      for (int d = 0; d < TDIM; ++d)
      {
        double v = l0 * V[0][d];
#pragma unroll
        for (int t = 0; t < TDIM; ++t) v += xi[t] * V[t + 1][d];
        CFX_EMIT_STORE(points[(int64_t)(pbase + pt) * TDIM + d], v);
      }
      CFX_EMIT_STORE(weights[pbase + pt], wref[q] * scale);
    }
  }
  if (lane == 0)
  {
    if (rbase == 0) offsets[0] = 0; // (the first rule's first point: this cell opens the rule set)
    if (part == PART_IF)
    {
      for (int f = 0; f < ns; ++f)
      {
        parent_map[rbase + f] = (int32_t)c;
        offsets[rbase + f + 1] = pbase + (f + 1) * nref;
      }
    }
    else
    {
      parent_map[rbase] = (int32_t)c;
      offsets[rbase + 1] = pbase + npts;
    }
  }
  } // jobs
}

// whole-cell rules: reference points, weights * |detJ|
// (python/tests/quadrature_utils.py:40-61)
template <int TDIM>
__global__ void __launch_bounds__(kBlock) full_rules_kernel(int64_t n, const int32_t* __restrict__ cells,
                                                            const double* __restrict__ x,
                                                            const int32_t* __restrict__ conn, int degree,
                                                            double* __restrict__ points, double* __restrict__ weights,
                                                            int32_t* __restrict__ offsets, int32_t* __restrict__ parent_map)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  Geo<TDIM> g;
  load_cell<TDIM>(x, conn, cells[i], g);
  jacobian<TDIM>(g);
  int nref;
  const double* wref;
  const double* pref = ref_points<TDIM>(TDIM, degree, nref, wref);
  const double detJ = fabs(g.detJ);
  for (int q = 0; q < nref; ++q)
  {
#pragma unroll
    for (int d = 0; d < TDIM; ++d) points[(i * nref + q) * TDIM + d] = pref[q * TDIM + d];
    weights[i * nref + q] = wref[q] * detJ;
  }
  parent_map[i] = cells[i];
  offsets[i + 1] = (int32_t)((i + 1) * nref);
  if (i == 0) offsets[0] = 0;
}

// rule index of point q: largest r with offsets[r] <= q
__device__ __forceinline__ int64_t rule_of_point(const int32_t* __restrict__ offsets, int64_t nr, int64_t q)
{
  int64_t lo = 0, hi = nr; // offsets[lo] <= q < offsets[hi]
  while (hi - lo > 1)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= q) lo = mid; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// a12 per-point evaluators (P1 level set over P1 geometry): one thread per
// point.  normal = sign * K^T grad_ref(phi) / max(|.|, 1e-14)
// (cpp/cutfemx/level_set/normal.h:150-186)
// ---------------------------------------------------------------------------
// one thread per rule: the normal of a P1 level set over P1 geometry is constant on the parent cell,
// so it is formed once per rule and written to the rule's points (no search of the rule, one cell load)
template <int TDIM>
__global__ void __launch_bounds__(kBlock) normals_rule_kernel(DevN nr_d, const int32_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ parent_map,
                                                              const double* __restrict__ x, const int32_t* __restrict__ conn,
                                                              const int32_t* __restrict__ ls_dofmap,
                                                              const double* __restrict__ phi_v, double sign,
                                                              double* __restrict__ out)
{
  const int64_t nr = dev_n(nr_d);
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nr) return;
  const int64_t c = parent_map[r];
  Geo<TDIM> g;
  load_cell<TDIM>(x, conn, c, g);
  jacobian<TDIM>(g);
  double phi[TDIM + 1];
#pragma unroll
  for (int v = 0; v <= TDIM; ++v) phi[v] = phi_v[ls_dofmap[c * (TDIM + 1) + v]];
  double gref[TDIM], gp[TDIM];
#pragma unroll
  for (int t = 0; t < TDIM; ++t) gref[t] = phi[t + 1] - phi[0];
  double norm = 0.0;
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double v = 0.0;
#pragma unroll
    for (int t = 0; t < TDIM; ++t) v += g.K[t][d] * gref[t];
    gp[d] = v;
    norm += v * v;
  }
  norm = sqrt(norm);
  if (norm < 1.0e-14) norm = 1.0e-14;
#pragma unroll
  for (int d = 0; d < TDIM; ++d) gp[d] = sign * gp[d] / norm;
  for (int64_t q = offsets[r]; q < offsets[r + 1]; ++q)
  {
#pragma unroll
    for (int d = 0; d < TDIM; ++d) out[q * TDIM + d] = gp[d];
  }
}

template <int TDIM>
__global__ void __launch_bounds__(kBlock) values_kernel(int64_t nq, int64_t nr, const int32_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ parent_map,
                                                        const double* __restrict__ points,
                                                        const int32_t* __restrict__ ls_dofmap,
                                                        const double* __restrict__ phi_v, double* __restrict__ out)
{
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nq) return;
  const int64_t c = parent_map[rule_of_point(offsets, nr, q)];
  double l0 = 1.0, v = 0.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t)
  {
    const double X = points[q * TDIM + t];
    l0 -= X;
    v += X * phi_v[ls_dofmap[c * (TDIM + 1) + t + 1]];
  }
  out[q] = v + l0 * phi_v[ls_dofmap[c * (TDIM + 1)]];
}

template <int TDIM>
__global__ void __launch_bounds__(kBlock) physical_points_kernel(int64_t nq, int64_t nr,
                                                                 const int32_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ parent_map,
                                                                 const double* __restrict__ points,
                                                                 const double* __restrict__ x,
                                                                 const int32_t* __restrict__ conn,
                                                                 double* __restrict__ out)
{
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nq) return;
  const int64_t c = parent_map[rule_of_point(offsets, nr, q)];
  Geo<TDIM> g;
  load_cell<TDIM>(x, conn, c, g);
  double X[TDIM], l0 = 1.0;
#pragma unroll
  for (int t = 0; t < TDIM; ++t) { X[t] = points[q * TDIM + t]; l0 -= X[t]; }
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double v = l0 * g.x[0][d];
#pragma unroll
    for (int t = 0; t < TDIM; ++t) v += X[t] * g.x[t + 1][d];
    out[q * TDIM + d] = v;
  }
}

// ---------------------------------------------------------------------------
// a7 ghost-penalty band.  For every cut cell c (ascending) and local facet lf:
// the neighbour n across the facet is found through the vertex->cells
// incidence; the facet is kept when n exists, n is in (cut U selected) and --
// if n is itself cut -- c < n, so each facet is emitted once, by its smallest
// cut cell.  Row = (c0, lf0, c1, lf1) with c0 < c1.
// (python/cutfemx/cut.py:340-380, python/cutfemx/wrappers/cut.cpp:84-114)
// ---------------------------------------------------------------------------
template <int TDIM>
__device__ __forceinline__ bool facet_neighbour(const int32_t* __restrict__ conn, const int64_t* __restrict__ v2c_off,
                                                const int32_t* __restrict__ v2c, int64_t c, int lf, int32_t& nb,
                                                int& nb_lf)
{
  constexpr int NV = TDIM + 1;
  int32_t cv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) cv[i] = conn[c * NV + i];
  // pivot = first facet vertex
  int32_t pivot = -1;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i != lf && pivot < 0) pivot = cv[i];
  for (int64_t k = v2c_off[pivot]; k < v2c_off[pivot + 1]; ++k)
  {
    const int32_t o = v2c[k];
    if (o == c) continue;
    int32_t ov[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) ov[i] = conn[(int64_t)o * NV + i];
    // o shares the facet iff all facet vertices of c are vertices of o
    bool all = true;
#pragma unroll
    for (int i = 0; i < NV; ++i)
    {
      if (i == lf) continue;
      bool found = false;
#pragma unroll
      for (int j = 0; j < NV; ++j) found = found || (ov[j] == cv[i]);
      all = all && found;
    }
    if (!all) continue;
    // local facet of o = its vertex that is not on the facet
    int olf = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j)
    {
      bool onf = false;
#pragma unroll
      for (int i = 0; i < NV; ++i) onf = onf || (i != lf && cv[i] == ov[j]);
      if (!onf) olf = j;
    }
    nb = o; nb_lf = olf;
    return true;
  }
  return false;
}

// One thread per (cut cell, local facet): the neighbour search (a scan of the vertex->cells list of a
// facet vertex) runs once, its result is parked in `cand` and packed in cell / facet order afterwards.
// neighbour of cell c across local facet lf from the mesh's cell->cell table; nb_lf = the vertex of the
// neighbour that is not on the facet
template <int TDIM>
__device__ __forceinline__ bool facet_neighbour_tab(const int32_t* __restrict__ conn, const int32_t* __restrict__ c2c,
                                                    int64_t c, int lf, int32_t& nb, int& nb_lf)
{
  constexpr int NV = TDIM + 1;
  nb = c2c[c * NV + lf];
  if (nb < 0) return false;
  int32_t cv[NV], ov[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { cv[i] = conn[c * NV + i]; ov[i] = conn[(int64_t)nb * NV + i]; }
  nb_lf = 0;
#pragma unroll
  for (int j = 0; j < NV; ++j)
  {
    bool onf = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) onf = onf || (i != lf && cv[i] == ov[j]);
    if (!onf) nb_lf = j;
  }
  return true;
}

template <int TDIM, bool XCD>
__global__ void __launch_bounds__(kBlock) cell_neighbours_kernel(int64_t ncells, const int32_t* __restrict__ conn,
                                                                 const int64_t* __restrict__ v2c_off,
                                                                 const int32_t* __restrict__ v2c, int32_t* __restrict__ c2c)
{
  constexpr int NV = TDIM + 1;
  // (XCD: every XCD -- every L2 -- walks one contiguous chunk of the cells; the candidates of neighbouring cells are
  // the same rows of the connectivity: 95 -> 86 ms at 512^3)
  const int64_t c = (XCD ? xcd_block_id() : (int64_t)blockIdx.x) * kBlock + threadIdx.x;
  if (c >= ncells) return;
  // All facets of the cell from TWO incidence lists instead of one list per facet: a cell that shares NV - 1
  // vertices with c is its neighbour across the facet opposite the vertex it lacks; every facet but the one opposite
  // vertex 0 contains vertex 0 (candidates: the cells around vertex 0), that one contains vertex 1.
  // (Measured and dropped in round 4: deciding from the incidence lists alone -- a candidate is the neighbour across
  // facet i when it is in the lists of all vertices but vertex i, one binary search per (candidate, vertex) instead of
  // the candidate's connectivity row -- ran 5.6 x slower: 530 ms, ~600 dependent 4 B loads per cell.)
  int32_t cv[NV], out[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { cv[i] = conn[c * NV + i]; out[i] = -1; }
  constexpr int KB = 4; // candidates in flight: ids first, then their connectivity rows, then the compares
#pragma unroll
  for (int pass = 0; pass < 2; ++pass)
  {
    const int32_t pivot = cv[pass];
    const int64_t kb = v2c_off[pivot], ke = v2c_off[pivot + 1];
    for (int64_t k0 = kb; k0 < ke; k0 += KB)
    {
      int32_t o[KB], ov[KB][NV];
#pragma unroll
      for (int u = 0; u < KB; ++u) o[u] = v2c[k0 + u < ke ? k0 + u : ke - 1];
#pragma unroll
      for (int u = 0; u < KB; ++u)
      {
        if constexpr (NV == 4)
        {
          const int4 q = *reinterpret_cast<const int4*>(conn + (int64_t)o[u] * 4);
          ov[u][0] = q.x; ov[u][1] = q.y; ov[u][2] = q.z; ov[u][3] = q.w;
        }
        else
        {
#pragma unroll
          for (int i = 0; i < NV; ++i) ov[u][i] = conn[(int64_t)o[u] * NV + i];
        }
      }
#pragma unroll
      for (int u = 0; u < KB; ++u)
      {
        if (k0 + u >= ke || o[u] == c) continue;
        int shared = 0, missing = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i)
        {
          bool found = false;
#pragma unroll
          for (int j = 0; j < NV; ++j) found = found || (ov[u][j] == cv[i]);
          shared += found ? 1 : 0;
          missing = found ? missing : i;
        }
        // pass 0 settles the facets that contain vertex 0, pass 1 the one opposite to it
        if (shared == NV - 1 && (pass == 0 ? missing != 0 : missing == 0))
        {
#pragma unroll
          for (int i = 0; i < NV; ++i) out[i] = (i == missing && out[i] < 0) ? o[u] : out[i];
        }
      }
    }
  }
  if constexpr (NV == 4) *reinterpret_cast<int4*>(c2c + c * 4) = make_int4(out[0], out[1], out[2], out[3]);
  else
  {
#pragma unroll
    for (int i = 0; i < NV; ++i) c2c[c * NV + i] = out[i];
  }
}

template <int TDIM>
__global__ void __launch_bounds__(kBlock) ghost_facets_find_kernel(DevN ncut_d, const int32_t* __restrict__ cut_cells,
                                                                   const int32_t* __restrict__ conn,
                                                                   const int32_t* __restrict__ c2c,
                                                                   const int8_t* __restrict__ domain, SelectorPred sel,
                                                                   int32_t* __restrict__ counts, int32_t* __restrict__ cand)
{
  const int64_t ncut = dev_n(ncut_d);
  constexpr int NV = TDIM + 1;
  // one thread per cut cell: its NV facets one after the other, the count written once (no counter to zero first: a
  // launch less per step -- a kernel boundary costs ~10 us here)
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncut)
  {
    if (i < ncut_d.cap) counts[i] = 0; // (list shorter than its capacity: the scan runs over the capacity)
    return;
  }
  const int64_t c = cut_cells[i];
  int n = 0;
#pragma unroll
  for (int lf = 0; lf < NV; ++lf)
  {
    int4 r = make_int4(-1, -1, -1, -1);
    int32_t nb;
    int nlf;
    if (facet_neighbour_tab<TDIM>(conn, c2c, c, lf, nb, nlf))
    {
      const bool nb_cut = domain[nb] == CFX_INTERSECTED;
      if ((nb_cut || sel(nb)) && !(nb_cut && nb < c))
      {
        r = (c < nb) ? make_int4((int)c, lf, nb, nlf) : make_int4(nb, nlf, (int)c, lf);
        ++n;
      }
    }
    *reinterpret_cast<int4*>(cand + 4 * (i * NV + lf)) = r;
  }
  counts[i] = n;
}

template <int TDIM>
__global__ void __launch_bounds__(kBlock) ghost_facets_pack_kernel(DevN ncut_d, const int32_t* __restrict__ cand,
                                                                   const int64_t* __restrict__ offs,
                                                                   int32_t* __restrict__ rows)
{
  const int64_t ncut = dev_n(ncut_d);
  constexpr int NV = TDIM + 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncut) return;
  int64_t o = offs[i];
#pragma unroll
  for (int lf = 0; lf < NV; ++lf)
  {
    const int4 r = *reinterpret_cast<const int4*>(cand + 4 * (i * NV + lf));
    if (r.x >= 0) { *reinterpret_cast<int4*>(rows + 4 * o) = r; ++o; }
  }
}

// ---------------------------------------------------------------------------
// 8f-3 cell aggregation (cell_aggregation.cpp:143-270) without its sequential sweeps.
// Sweep k of the reference visits the unrooted cut cells in ascending order; cell c takes the
// first (ascending) active neighbour that is rooted AT THAT MOMENT: a root, a cell rooted in an
// earlier sweep, or a lower-numbered cell rooted earlier in this sweep.  Hence the sweep t(c) in
// which c is rooted is the least fixed point of
//     t(c) = max(0, min over active neighbours o of t(o) + [o > c]),   t(root) = -1,
// a shortest-path problem with 0/1 edge weights solved by parallel relaxation; the neighbour
// chosen in sweep t(c) is the first o with t(o) < t(c) or (t(o) == t(c) and o < c); roots,
// aggregate ids and depths then follow the parent pointers (pointer jumping).
// ---------------------------------------------------------------------------
constexpr int32_t kAggUnset = 0x7fffffff;

template <int TDIM>
__device__ __forceinline__ int sorted_neighbours(const int32_t* __restrict__ conn, const int64_t* __restrict__ v2c_off,
                                                 const int32_t* __restrict__ v2c, int64_t c, int32_t* nbs)
{
  int n = 0;
  for (int lf = 0; lf <= TDIM; ++lf)
  {
    int32_t nb;
    int nlf;
    if (facet_neighbour<TDIM>(conn, v2c_off, v2c, c, lf, nb, nlf)) nbs[n++] = nb;
  }
  for (int a = 1; a < n; ++a) // insertion sort, n <= 4
  {
    const int32_t v = nbs[a];
    int b = a - 1;
    while (b >= 0 && nbs[b] > v) { nbs[b + 1] = nbs[b]; --b; }
    nbs[b + 1] = v;
  }
  return n;
}

// selected volume fraction of the cut cells from order-1 volume rules
template <int TDIM>
__global__ void agg_fraction_kernel(int64_t nr, const int32_t* __restrict__ offsets, const int32_t* __restrict__ parent,
                                    const double* __restrict__ weights, const double* __restrict__ x,
                                    const int32_t* __restrict__ conn, double* fraction)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nr) return;
  double s = 0.0;
  for (int32_t q = offsets[e]; q < offsets[e + 1]; ++q) s += weights[q];
  Geo<TDIM> g;
  load_cell<TDIM>(x, conn, parent[e], g);
  jacobian<TDIM>(g);
  atomicAdd(&fraction[parent[e]], s / (fabs(g.detJ) / (TDIM == 2 ? 2.0 : 6.0)));
}

// cls bits: 1 interior, 2 cut, 4 well posed (root), 8 ill posed
__global__ void agg_init_kernel(int64_t nc, const int8_t* __restrict__ domain, int8_t selcode, int policy,
                                const double* __restrict__ fraction, double threshold, int32_t* t, uint8_t* cls)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const bool interior = domain[c] == selcode, cutc = domain[c] == CFX_INTERSECTED;
  const bool well = interior || (cutc && policy == 1 && fraction[c] >= threshold);
  t[c] = well ? -1 : kAggUnset;
  cls[c] = (uint8_t)((interior ? 1 : 0) | (cutc ? 2 : 0) | (well ? 4 : 0) | ((cutc && !well) ? 8 : 0));
}

template <int TDIM>
__global__ void agg_relax_kernel(int64_t n_ill, const int32_t* __restrict__ ill, const int32_t* __restrict__ conn,
                                 const int64_t* __restrict__ v2c_off, const int32_t* __restrict__ v2c,
                                 const uint8_t* __restrict__ cls, int32_t* t, int* changed)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ill) return;
  const int32_t c = ill[i];
  int32_t nbs[TDIM + 1];
  const int n = sorted_neighbours<TDIM>(conn, v2c_off, v2c, c, nbs);
  int32_t best = kAggUnset;
  for (int k = 0; k < n; ++k)
  {
    const int32_t o = nbs[k];
    if (!(cls[o] & 3)) continue; // active cells only
    const int32_t to = *reinterpret_cast<volatile int32_t*>(&t[o]);
    if (to == kAggUnset) continue;
    const int32_t cand = to + (o > c ? 1 : 0);
    best = cand < best ? cand : best;
  }
  if (best != kAggUnset && best < 0) best = 0;
  if (best < t[c]) { t[c] = best; *changed = 1; }
}

template <int TDIM>
__global__ void agg_parent_kernel(int64_t n_ill, const int32_t* __restrict__ ill, const int32_t* __restrict__ conn,
                                  const int64_t* __restrict__ v2c_off, const int32_t* __restrict__ v2c,
                                  const uint8_t* __restrict__ cls, const int32_t* __restrict__ t, int64_t limit,
                                  int32_t* parent)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ill) return;
  const int32_t c = ill[i];
  int32_t p = -1;
  const int32_t tc = t[c];
  if (tc != kAggUnset && (int64_t)tc < limit)
  {
    int32_t nbs[TDIM + 1];
    const int n = sorted_neighbours<TDIM>(conn, v2c_off, v2c, c, nbs);
    for (int k = 0; k < n && p < 0; ++k)
    {
      const int32_t o = nbs[k];
      if (!(cls[o] & 3) || t[o] == kAggUnset) continue;
      if (t[o] < tc || (t[o] == tc && o < c)) p = o;
    }
  }
  parent[i] = p;
}

__global__ void agg_roots_kernel(int64_t n_well, const int32_t* __restrict__ well, int32_t* root, int32_t* agg,
                                 int32_t* depth)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_well) return;
  const int32_t c = well[i];
  root[c] = c; agg[c] = (int32_t)i; depth[c] = 0; // aggregate ids follow the ascending root order (:211-217)
}

__global__ void agg_resolve_kernel(int64_t n_ill, const int32_t* __restrict__ ill, const int32_t* __restrict__ parent,
                                   int32_t* root, int32_t* agg, int32_t* depth, int* changed)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ill) return;
  const int32_t c = ill[i], p = parent[i];
  if (p < 0 || root[c] >= 0) return;
  const int32_t rp = *reinterpret_cast<volatile int32_t*>(&root[p]);
  if (rp < 0) return;
  // the parent's three fields were written before its root became visible only within one
  // thread; read depth/aggregate after the root and re-run until nothing changes
  depth[c] = depth[p] + 1; agg[c] = agg[p];
  __threadfence();
  root[c] = rp;
  *changed = 1;
}

__global__ void agg_flag_rootless_kernel(int64_t n_ill, const int32_t* __restrict__ ill, const int32_t* __restrict__ root,
                                         uint8_t* cls)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_ill && root[ill[i]] < 0) cls[ill[i]] |= 16;
}

__global__ void agg_pairs_kernel(int64_t n, const int32_t* __restrict__ bad, const int32_t* __restrict__ root,
                                 int32_t* rows)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *reinterpret_cast<int4*>(rows + 4 * i) = make_int4(bad[i], 0, root[bad[i]], 0);
}

struct ClsMask
{
  const uint8_t* cls;
  uint8_t any, none;
  __device__ bool operator()(int64_t c) const { return (cls[c] & any) != 0 && (cls[c] & none) == 0; }
};

// cells outside a candidate subset: a classification code no selector matches
__global__ void restrict_domain_kernel(int64_t n, const uint8_t* __restrict__ keep, int8_t* __restrict__ domain)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n && !keep[c]) domain[c] = (int8_t)CFX_NOT_CANDIDATE;
}

__global__ void mark_subset_kernel(int64_t n, const int32_t* __restrict__ cells, int64_t ncells, uint8_t* keep, int* bad)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t c = cells[i];
  if (c < 0 || c >= ncells) { *bad = 1; return; }
  keep[c] = 1;
}

// interior facets inside a cell set: thread per (listed cell, facet), emitted from the lower cell
template <int TDIM>
__global__ void __launch_bounds__(kBlock) interior_facets_find_kernel(int64_t n, const int32_t* __restrict__ cells,
                                                                      const int32_t* __restrict__ conn,
                                                                      const int64_t* __restrict__ v2c_off,
                                                                      const int32_t* __restrict__ v2c,
                                                                      const uint8_t* __restrict__ inset,
                                                                      int32_t* __restrict__ counts, int32_t* __restrict__ cand)
{
  constexpr int NV = TDIM + 1;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n * NV) return;
  const int64_t i = t / NV;
  const int lf = (int)(t - i * NV);
  const int64_t c = cells[i];
  int4 r = make_int4(-1, -1, -1, -1);
  int32_t nb;
  int nlf;
  if (facet_neighbour<TDIM>(conn, v2c_off, v2c, c, lf, nb, nlf) && inset[nb] && c < nb)
  {
    r = make_int4((int)c, lf, nb, nlf);
    atomicAdd(&counts[i], 1);
  }
  *reinterpret_cast<int4*>(cand + 4 * t) = r;
}

// ---------------------------------------------------------------------------
// 8f-4 facet hosts: cut(level_set, facets, tdim - 1) (cut.cpp:540-591, 788-830).  The hosts are facets of
// the mesh given as integration rows (cell, local facet[, cell1, local facet1]); host vertex j is either the
// j-th vertex of cell0 that is not opposite the facet (ascending local index) or the caller's
// entities_to_geometry row.  The sub-triangulation of a host is the (tdim-1)-dimensional marching case of its
// P1 level-set values; points live on the host's reference simplex, weights carry the physical measure.
// ---------------------------------------------------------------------------
template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_hosts_kernel(int64_t n, const int32_t* __restrict__ rows, int width,
                                                             const int32_t* __restrict__ geom,
                                                             const int32_t* __restrict__ conn, int64_t ncells,
                                                             const int32_t* __restrict__ ls_dofmap,
                                                             int32_t* __restrict__ verts, int32_t* __restrict__ ls, int* bad)
{
  constexpr int NV = TDIM + 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t c = rows[i * width];
  const int lf = rows[i * width + 1];
  if (c < 0 || c >= ncells || lf < 0 || lf > TDIM) { *bad = 1; return; }
  int32_t cv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) cv[k] = conn[c * NV + k];
  for (int j = 0; j < TDIM; ++j)
  {
    int loc = -1;
    if (geom)
    {
      const int32_t g = geom[i * TDIM + j];
      for (int k = 0; k < NV; ++k) loc = (k != lf && cv[k] == g) ? k : loc;
    }
    else
      loc = j < lf ? j : j + 1;
    if (loc < 0) { *bad = 2; return; }
    int32_t v = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) v = (k == loc) ? cv[k] : v;
    verts[i * TDIM + j] = v;
    ls[i * TDIM + j] = ls_dofmap[c * NV + loc];
  }
  if (width == 4)
  {
    const int64_t c1 = rows[i * 4 + 2];
    const int lf1 = rows[i * 4 + 3];
    if (c1 < 0 || c1 >= ncells || lf1 < 0 || lf1 > TDIM) { *bad = 1; return; }
    // the same facet seen from cell1: every facet vertex of cell0 is a vertex of cell1 other than lf1
    for (int k = 0; k < NV; ++k)
    {
      if (k == lf) continue;
      bool found = false;
      for (int m = 0; m < NV; ++m) found = found || (m != lf1 && conn[c1 * NV + m] == cv[k]);
      if (!found) { *bad = 3; return; }
    }
  }
}

// sub-simplices of a host of dimension HD with sign mask `mask` in `part`
template <int HD>
__device__ __forceinline__ int host_subcount(int mask, int part)
{
  if constexpr (HD == 1)
  {
    const int nn = __popc(mask & 3);
    if (part == PART_IF) return nn == 1 ? 1 : 0; // the cut point
    return part == PART_IN ? (nn >= 1 ? 1 : 0) : (nn <= 1 ? 1 : 0);
  }
  else
  {
    const CutCase& cs = c_cases[0][mask & 7];
    return part == PART_IF ? cs.n_if : (part == PART_IN ? cs.n_in : cs.n_out);
  }
}

template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_count_kernel(int64_t ncut, const int32_t* __restrict__ cut_hosts,
                                                             const int32_t* __restrict__ ls, const double* __restrict__ phi_v,
                                                             int part, int nref, int32_t* __restrict__ n_rules,
                                                             int32_t* __restrict__ n_points)
{
  constexpr int HD = TDIM - 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncut) return;
  const int64_t h = cut_hosts[i];
  double phi[HD + 1];
#pragma unroll
  for (int v = 0; v <= HD; ++v) phi[v] = phi_v[ls[h * TDIM + v]];
  const int ns = host_subcount<HD>(sign_mask<HD>(phi), part);
  n_rules[i] = ns > 0 ? 1 : 0;
  n_points[i] = ns * nref;
}

// physical measure factor of a host: |x1 - x0| (segment) or |(x1 - x0) x (x2 - x0)| (triangle)
template <int TDIM>
__device__ __forceinline__ double host_measure(const double (*xv)[TDIM])
{
  if constexpr (TDIM == 2)
  {
    const double dx = xv[1][0] - xv[0][0], dy = xv[1][1] - xv[0][1];
    return sqrt(dx * dx + dy * dy);
  }
  else
  {
    double a[3], b[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { a[d] = xv[1][d] - xv[0][d]; b[d] = xv[2][d] - xv[0][d]; }
    const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    return sqrt(cx * cx + cy * cy + cz * cz);
  }
}

// one thread per cut host (there are O(N^(tdim-2)) .. O(N^(tdim-1)) of them): one rule per host
template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_emit_kernel(
    int64_t ncut, const int32_t* __restrict__ cut_hosts, const double* __restrict__ x, const int32_t* __restrict__ verts,
    const int32_t* __restrict__ ls, const double* __restrict__ phi_v, const int32_t* __restrict__ host_ids, int part,
    int degree, const int32_t* __restrict__ rule_off, const int32_t* __restrict__ point_off, double* __restrict__ points,
    double* __restrict__ weights, int32_t* __restrict__ offsets, int32_t* __restrict__ parent_map,
    int32_t* __restrict__ rule_host)
{
  constexpr int HD = TDIM - 1, NV = HD + 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncut) return;
  const int64_t h = cut_hosts[i];
  double phi[NV], xv[NV][TDIM];
#pragma unroll
  for (int v = 0; v < NV; ++v)
  {
    phi[v] = phi_v[ls[h * TDIM + v]];
    const int64_t g = verts[h * TDIM + v];
#pragma unroll
    for (int d = 0; d < TDIM; ++d) xv[v][d] = x[3 * g + d];
  }
  const int mask = sign_mask<HD>(phi);
  const int ns = host_subcount<HD>(mask, part);
  if (ns == 0) return;
  const double measure = host_measure<TDIM>(xv);
  const int32_t pbase = point_off[i], rbase = rule_off[i];
  if (part == PART_IF)
  {
    // the set phi = 0 on the host (codimension 2 in the mesh): a point of a segment host (weight 1), a
    // straight segment across a triangle host (1-D rule, weights carry its physical length)
    if constexpr (HD == 1)
    {
      const int a = mask == 1 ? 0 : 1;
      const double pa = a == 0 ? phi[0] : phi[1], pb = a == 0 ? phi[1] : phi[0];
      const double t = pa / (pa - pb);
      const double xa = a == 0 ? 0.0 : 1.0, xb = a == 0 ? 1.0 : 0.0;
      points[pbase] = xa + t * (xb - xa);
      weights[pbase] = 1.0;
      offsets[rbase + 1] = pbase + 1;
    }
    else
    {
      const CutCase& cs = c_cases[0][mask & 7];
      double V[2][HD], xp[2][TDIM];
#pragma unroll
      for (int j = 0; j < 2; ++j)
      {
        local_point<2>(cs, cs.iface[0][j], phi, V[j]);
        const double l0 = 1.0 - V[j][0] - V[j][1];
#pragma unroll
        for (int d = 0; d < TDIM; ++d) xp[j][d] = l0 * xv[0][d] + V[j][0] * xv[1][d] + V[j][1] * xv[2][d];
      }
      double len = 0.0;
#pragma unroll
      for (int d = 0; d < TDIM; ++d) len += (xp[1][d] - xp[0][d]) * (xp[1][d] - xp[0][d]);
      len = sqrt(len);
      int n1;
      const double* w1;
      const double* p1 = ref_points<TDIM>(1, degree, n1, w1);
      for (int q = 0; q < n1; ++q)
      {
#pragma unroll
        for (int d = 0; d < HD; ++d) points[(int64_t)(pbase + q) * HD + d] = V[0][d] + p1[q] * (V[1][d] - V[0][d]);
        weights[pbase + q] = w1[q] * len;
      }
      offsets[rbase + 1] = pbase + n1;
    }
    parent_map[rbase] = host_ids[h];
    rule_host[rbase] = (int32_t)h;
    return;
  }
  int nref;
  const double* wref;
  const double* pref = ref_points<TDIM>(HD, degree, nref, wref);
  for (int k = 0; k < ns; ++k)
  {
    double V[NV][HD];
    double dsub;
    if constexpr (HD == 1)
    {
      // cut point from the negative vertex a towards the other vertex b
      double A = 0.0, B = 1.0;
      if (mask == 1 || mask == 2)
      {
        const int a = mask == 1 ? 0 : 1;
        const double pa = a == 0 ? phi[0] : phi[1], pb = a == 0 ? phi[1] : phi[0];
        const double t = pa / (pa - pb);
        const double xa = a == 0 ? 0.0 : 1.0, xb = a == 0 ? 1.0 : 0.0;
        const double q = xa + t * (xb - xa);
        if (part == PART_IN) { A = xa; B = q; } else { A = q; B = xb; }
      }
      V[0][0] = A; V[1][0] = B;
      dsub = B - A;
    }
    else
    {
      const CutCase& cs = c_cases[0][mask & 7];
      const int8_t* sx = part == PART_IN ? cs.in[k] : cs.out[k];
#pragma unroll
      for (int j = 0; j < NV; ++j) local_point<2>(cs, sx[j], phi, V[j]);
      dsub = (V[1][0] - V[0][0]) * (V[2][1] - V[0][1]) - (V[1][1] - V[0][1]) * (V[2][0] - V[0][0]);
    }
    const double scale = fabs(dsub) * measure;
    for (int q = 0; q < nref; ++q)
    {
      const double* xi = pref + HD * q;
      double l0 = 1.0;
#pragma unroll
      for (int t = 0; t < HD; ++t) l0 -= xi[t];
      const int64_t o = pbase + k * nref + q;
#pragma unroll
      for (int d = 0; d < HD; ++d)
      {
        double v = l0 * V[0][d];
#pragma unroll
        for (int t = 0; t < HD; ++t) v += xi[t] * V[t + 1][d];
        points[o * HD + d] = v;
      }
      weights[o] = wref[q] * scale;
    }
  }
  parent_map[rbase] = host_ids[h];
  rule_host[rbase] = (int32_t)h;
  offsets[rbase + 1] = pbase + ns * nref;
}

// whole-host rules (the standard facets of a mixed measure): reference points, weights * host measure
template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_full_rules_kernel(int64_t n, const int32_t* __restrict__ hosts,
                                                                  const double* __restrict__ x,
                                                                  const int32_t* __restrict__ verts,
                                                                  const int32_t* __restrict__ host_ids, int degree,
                                                                  double* __restrict__ points, double* __restrict__ weights,
                                                                  int32_t* __restrict__ offsets, int32_t* __restrict__ parent_map,
                                                                  int32_t* __restrict__ rule_host)
{
  constexpr int HD = TDIM - 1, NV = HD + 1;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t h = hosts ? hosts[i] : i;
  double xv[NV][TDIM];
#pragma unroll
  for (int v = 0; v < NV; ++v)
  {
    const int64_t g = verts[h * TDIM + v];
#pragma unroll
    for (int d = 0; d < TDIM; ++d) xv[v][d] = x[3 * g + d];
  }
  const double measure = host_measure<TDIM>(xv);
  int nref;
  const double* wref;
  const double* pref = ref_points<TDIM>(HD, degree, nref, wref);
  for (int q = 0; q < nref; ++q)
  {
#pragma unroll
    for (int d = 0; d < HD; ++d) points[(i * nref + q) * HD + d] = pref[q * HD + d];
    weights[i * nref + q] = wref[q] * measure;
  }
  parent_map[i] = host_ids[h];
  rule_host[i] = (int32_t)h;
  offsets[i + 1] = (int32_t)((i + 1) * nref);
  if (i == 0) offsets[0] = 0;
}

__global__ void __launch_bounds__(kBlock) gather_rows_kernel(int64_t n, const int32_t* __restrict__ idx, int width,
                                                             const int32_t* __restrict__ src, int32_t* __restrict__ dst)
{
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n * width) return;
  const int64_t i = t / width;
  const int k = (int)(t - i * width);
  dst[t] = src[(int64_t)idx[i] * width + k];
}

template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_physical_points_kernel(int64_t nq, int64_t nr,
                                                                       const int32_t* __restrict__ offsets,
                                                                       const int32_t* __restrict__ verts,
                                                                       const double* __restrict__ points,
                                                                       const double* __restrict__ x, double* __restrict__ out)
{
  constexpr int HD = TDIM - 1;
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nq) return;
  const int64_t r = rule_of_point(offsets, nr, q);
  double lam[HD + 1];
  lam[0] = 1.0;
#pragma unroll
  for (int t = 0; t < HD; ++t) { lam[t + 1] = points[q * HD + t]; lam[0] -= lam[t + 1]; }
#pragma unroll
  for (int d = 0; d < TDIM; ++d)
  {
    double v = 0.0;
#pragma unroll
    for (int j = 0; j <= HD; ++j) v += lam[j] * x[3 * (int64_t)verts[r * TDIM + j] + d];
    out[q * TDIM + d] = v;
  }
}

// facet-hosted rule r seen from cell `side` of its row: parent = that cell, points = the cell's reference
// coordinates of the same physical points (barycentric weights moved to the cell's local vertices) --
// what facet_runtime_quadrature_payload / interior_facet_runtime_quadrature_payload hand to the kernels
// (python/cutfemx/_runintgen_adapter.py:605-680).  Rule i of the output is rule perm[i] of the input.
template <int TDIM>
__global__ void __launch_bounds__(kBlock) facet_to_cell_kernel(int64_t nq, int64_t nr, const int32_t* __restrict__ new_off,
                                                               const int32_t* __restrict__ perm,
                                                               const int32_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ rows, int width, int side,
                                                               const int32_t* __restrict__ verts,
                                                               const double* __restrict__ points,
                                                               const double* __restrict__ weights,
                                                               const int32_t* __restrict__ conn,
                                                               double* __restrict__ out_points, double* __restrict__ out_weights,
                                                               int32_t* __restrict__ out_parent)
{
  constexpr int HD = TDIM - 1, NV = TDIM + 1;
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= nq) return;
  const int64_t i = rule_of_point(new_off, nr, p);
  const int64_t r = perm[i];
  const int64_t q = offsets[r] + (p - new_off[i]);
  const int64_t c = rows[r * width + 2 * side];
  double lam[HD + 1];
  lam[0] = 1.0;
#pragma unroll
  for (int t = 0; t < HD; ++t) { lam[t + 1] = points[q * HD + t]; lam[0] -= lam[t + 1]; }
  double X[TDIM];
#pragma unroll
  for (int t = 0; t < TDIM; ++t) X[t] = 0.0;
#pragma unroll
  for (int j = 0; j <= HD; ++j)
  {
    const int32_t g = verts[r * TDIM + j];
#pragma unroll
    for (int t = 0; t < TDIM; ++t) X[t] += (conn[c * NV + t + 1] == g) ? lam[j] : 0.0;
  }
#pragma unroll
  for (int t = 0; t < TDIM; ++t) out_points[p * TDIM + t] = X[t];
  out_weights[p] = weights[q];
  if (p == new_off[i]) out_parent[i] = (int32_t)c;
}

__global__ void __launch_bounds__(kBlock) rule_cell_keys_kernel(int64_t nr, const int32_t* __restrict__ rows, int width,
                                                                int side, int32_t* __restrict__ keys, int* unsorted)
{
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nr) return;
  const int32_t k = rows[r * width + 2 * side];
  keys[r] = k;
  if (r > 0 && rows[(r - 1) * width + 2 * side] > k) *unsorted = 1;
}

__global__ void __launch_bounds__(kBlock) permuted_counts_kernel(int64_t nr, const int32_t* __restrict__ perm,
                                                                 const int32_t* __restrict__ offsets, int32_t* __restrict__ counts)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nr) return;
  const int64_t r = perm ? perm[i] : i;
  counts[i] = offsets[r + 1] - offsets[r];
}

__global__ void __launch_bounds__(kBlock) iota_kernel(int64_t n, int32_t* __restrict__ a)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] = (int32_t)i;
}

// exterior facets: per cell the bit mask of its local facets without a neighbour
template <int TDIM>
__global__ void __launch_bounds__(kBlock) exterior_mask_kernel(int64_t ncells, const int32_t* __restrict__ conn,
                                                               const int64_t* __restrict__ v2c_off,
                                                               const int32_t* __restrict__ v2c, uint8_t* __restrict__ mask)
{
  constexpr int NV = TDIM + 1;
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= ncells) return;
  unsigned m = 0;
  for (int lf = 0; lf < NV; ++lf)
  {
    int32_t nb;
    int nlf;
    if (!facet_neighbour<TDIM>(conn, v2c_off, v2c, c, lf, nb, nlf)) m |= 1u << lf;
  }
  mask[c] = (uint8_t)m;
}

__global__ void __launch_bounds__(kBlock) exterior_count_kernel(int64_t n, const int32_t* __restrict__ cells,
                                                                const uint8_t* __restrict__ mask, int32_t* __restrict__ counts)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) counts[i] = __popc((unsigned)mask[cells[i]]);
}

__global__ void __launch_bounds__(kBlock) exterior_pack_kernel(int64_t n, const int32_t* __restrict__ cells,
                                                               const uint8_t* __restrict__ mask,
                                                               const int64_t* __restrict__ offs, int32_t* __restrict__ rows)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t c = cells[i];
  unsigned m = mask[c];
  int64_t o = offs[i];
  for (int lf = 0; lf < 4; ++lf)
    if ((m >> lf) & 1u) { rows[2 * o] = c; rows[2 * o + 1] = lf; ++o; }
}

struct ByteSet
{
  const uint8_t* b;
  __device__ bool operator()(int64_t i) const { return b[i] != 0; }
};

struct IsCut
{
  const int8_t* domain;
  __device__ bool operator()(int64_t c) const { return domain[c] == CFX_INTERSECTED; }
};

// the per-tile counters of the first level set, zeroed: two arrays in one block
static void tile_counters(cfx_cut_t cut, int64_t ntiles, bool zeroed_already)
{
  const int64_t stride = (ntiles + 3) & ~3LL;
  if (!zeroed_already)
  {
    cut->tile_block.alloc(2 * stride);
    cut->tile_block.zero();
  }
  cut->tiles_inside.release(); cut->tiles_cut.release();
  cut->tiles_inside.p = cut->tile_block.p; cut->tiles_inside.n = ntiles; cut->tiles_inside.owned = false;
  cut->tiles_cut.p = cut->tile_block.p + stride; cut->tiles_cut.n = ntiles; cut->tiles_cut.owned = false;
}

void classify(cfx_cut_t cut)
{
  const int64_t nc = cut->nhosts();
  cut->block_class.release(); // (set again by the culled classification of level set 0)
  ++cut->gen;                 // the located lists of the previous classification lose their provenance
  provenance_forget_cut(cut);
  for (int k = 0; k < cut->nls; ++k)
  {
    int8_t* dom = cut->domain.p + (int64_t)k * nc;
    // (the codes of level set 0 stay with the cut: a row plan classifies row tiles from them)
    // ... next to one byte per level-set dof that the classification sets on the dofs of every cut cell: a dof with
    // a negative (positive) value and no cut cell around it has only inside (outside) cells around it)
    DevArray<uint8_t> codes_k;
    if (k == 0 && cut->codes0.n != cut->ls_ndofs) { cut->codes0.alloc(cut->ls_ndofs); cut->touch0.alloc(cut->ls_ndofs); }
    if (k != 0) codes_k.alloc(cut->ls_ndofs);
    struct { uint8_t* p; } codes{k == 0 ? cut->codes0.p : codes_k.p};
    uint8_t* touch = k == 0 ? cut->touch0.p : nullptr;
    if (k == 0) cut->touch_valid = false;
    // (the per-tile counters of level set 0 are cleared by the same launch when they fit its grid)
    bool tiles_zeroed = false;
    int32_t* zero_this = nullptr;
    int64_t zero_n = 0;
    if (k == 0 && cut->host_mask.n == 0)
    {
      const int64_t stride = (((nc + kByteTile - 1) / kByteTile) + 3) & ~3LL;
      if (2 * stride <= cut->ls_ndofs)
      {
        cut->tile_block.alloc(2 * stride);
        zero_this = cut->tile_block.p; zero_n = 2 * stride;
        tiles_zeroed = true;
      }
    }
    // (four values per thread when the level set starts on a 16 B boundary and the grid still covers the counters)
    const int64_t quads = (cut->ls_ndofs + 3) / 4;
    if ((reinterpret_cast<uintptr_t>(cut->ls_values[k].p) & 15) == 0 && (reinterpret_cast<uintptr_t>(codes.p) & 3) == 0
        && (reinterpret_cast<uintptr_t>(touch) & 3) == 0 && zero_n <= quads)
      launch("sign_codes", sign_codes4_kernel, grid_for(quads), dim3(kBlock), 0, cut->ls_ndofs, cut->ls_values[k].p, codes.p,
             zero_this, zero_n, touch);
    else
      launch("sign_codes", sign_codes_kernel, grid_for(cut->ls_ndofs), dim3(kBlock), 0, cut->ls_ndofs, cut->ls_values[k].p,
             codes.p, zero_this, zero_n, touch);
    const uint8_t* phi = codes.p;
    {
      // implicit-structured variant (opt-in): generated box mesh, P1 level set on the geometry dofmap
      const char* ib = getenv("CFX_IMPLICIT_BOX");
      cfx_mesh_t mesh = cut->mesh;
      if (ib && ib[0] == '1' && cut->host_width == 0 && mesh->box_n > 0 && cut->ls_dofmap.p == mesh->conn.p
          && cut->ls_ndofs_cell == mesh->tdim + 1)
      {
        const int64_t ncubes = nc / (mesh->tdim == 3 ? 6 : 2);
        int32_t *b_in = nullptr, *b_cut = nullptr;
        if (k == 0 && cut->host_mask.n == 0)
        {
          const int64_t ntiles = (nc + kByteTile - 1) / kByteTile;
          tile_counters(cut, ntiles, tiles_zeroed);
          b_in = cut->tiles_inside.p; b_cut = cut->tiles_cut.p;
        }
        else if (k == 0) { cut->tiles_inside.release(); cut->tiles_cut.release(); }
        if (mesh->tdim == 3)
          launch("classify_box", classify_box_kernel<3>, grid_for(ncubes), dim3(kBlock), 0, ncubes, mesh->box_n, phi, dom,
                 b_in, b_cut, touch);
        else
          launch("classify_box", classify_box_kernel<2>, grid_for(ncubes), dim3(kBlock), 0, ncubes, mesh->box_n, phi, dom,
                 b_in, b_cut, touch);
        if (k == 0) cut->touch_valid = true;
        continue;
      }
    }
    int32_t *t_in = nullptr, *t_cut = nullptr;
    if (k == 0 && cut->host_mask.n == 0)
    {
      const int64_t ntiles = (nc + kByteTile - 1) / kByteTile;
      tile_counters(cut, ntiles, tiles_zeroed);
      t_in = cut->tiles_inside.p; t_cut = cut->tiles_cut.p;
    }
    else if (k == 0) { cut->tiles_inside.release(); cut->tiles_cut.release(); }
    {
      // block culling (classify_culled_kernel): cells as hosts, level set on the geometry dofmap (the summary is the
      // mesh's), P1; CFX_CLASSIFY_CULL=0: cell by cell
      cfx_mesh_t mesh = cut->mesh;
      const char* cc = getenv("CFX_CLASSIFY_CULL");
      const int nd = cut->ls_ndofs_cell;
      if (!(cc && cc[0] == '0') && cut->host_width == 0 && cut->ls_dofmap.p == mesh->conn.p && nd == mesh->tdim + 1
          && nc == mesh->ncells && (nd == 3 || nd == 4))
      {
        const int64_t nb = (nc + kClassBlock - 1) / kClassBlock;
        if (!mesh->class_built)
        {
          mesh->class_nruns.alloc(nb);
          mesh->class_runs.alloc(nb * kClassRuns);
          mesh->class_sub_runs.alloc(nb * kClassSub * kClassSubRuns);
          if (nd == 4)
            launch("classify_summary", class_summary_kernel<4, kClassBlock, kClassRuns, 1024, 4096>, dim3((unsigned)nb), dim3(kBlock), 0,
                   nc, mesh->conn.p, mesh->class_nruns.p, mesh->class_runs.p, mesh->class_sub_runs.p);
          else
            launch("classify_summary", class_summary_kernel<3, kClassBlock, kClassRuns, 1024, 4096>, dim3((unsigned)nb), dim3(kBlock), 0,
                   nc, mesh->conn.p, mesh->class_nruns.p, mesh->class_runs.p, mesh->class_sub_runs.p);
          mesh->class_built = true;
          publish_across_lanes();
        }
        const dim3 cgrid((unsigned)(((nb + 1) / 2 + kBlock / 64 - 1) / (kBlock / 64)));   // a wavefront per pair of blocks
        uint8_t* bclass = nullptr;
        if (k == 0 && t_in != nullptr) { cut->block_class.alloc(nb); bclass = cut->block_class.p; }
        if (nd == 4) launch("classify", classify_culled_kernel<4>, cgrid, dim3(kBlock), 0, nc, nb, cut->ls_dofmap.p, (const int2*)mesh->class_runs.p, (const int2*)mesh->class_sub_runs.p, phi, dom, t_in, t_cut, bclass, touch);
        else launch("classify", classify_culled_kernel<3>, cgrid, dim3(kBlock), 0, nc, nb, cut->ls_dofmap.p, (const int2*)mesh->class_runs.p, (const int2*)mesh->class_sub_runs.p, phi, dom, t_in, t_cut, bclass, touch);
        if (k == 0) cut->touch_valid = true;
        continue;
      }
    }
    switch (cut->ls_ndofs_cell)
    {
    case 2: launch("classify", classify_kernel<2>, grid_for(nc, kBlock * CFX_CLASSIFY_UNROLL), dim3(kBlock), 0, nc, cut->ls_dofmap.p, phi, dom, t_in, t_cut, touch); break;
    case 3: launch("classify", classify_kernel<3>, grid_for(nc, kBlock * CFX_CLASSIFY_UNROLL), dim3(kBlock), 0, nc, cut->ls_dofmap.p, phi, dom, t_in, t_cut, touch); break;
    case 4: launch("classify", classify_kernel<4>, grid_for(nc, kBlock * CFX_CLASSIFY_UNROLL), dim3(kBlock), 0, nc, cut->ls_dofmap.p, phi, dom, t_in, t_cut, touch); break;
    case 6: launch("classify", classify_kernel<6>, grid_for(nc, kBlock * CFX_CLASSIFY_UNROLL), dim3(kBlock), 0, nc, cut->ls_dofmap.p, phi, dom, t_in, t_cut, touch); break;
    case 10: launch("classify", classify_kernel<10>, grid_for(nc, kBlock * CFX_CLASSIFY_UNROLL), dim3(kBlock), 0, nc, cut->ls_dofmap.p, phi, dom, t_in, t_cut, touch); break;
    default: throw Error(CFX_ERR_INVALID_ARGUMENT, "unsupported level-set element (dofs per cell must be 3, 4, 6 or 10)");
    }
    if (k == 0) cut->touch_valid = true;
  }
  if (cut->host_mask.n > 0)
    for (int k = 0; k < cut->nls; ++k)
      launch("restrict_domain", restrict_domain_kernel, grid_for(nc), dim3(kBlock), 0, nc, cut->host_mask.p,
             cut->domain.p + (int64_t)k * nc);
  cut->located.clear();
  cut->located_ids.clear();
  cut->ghost_rows.clear();
}

// "phi<0" and "phi=0" of the first level set in one pass over the classification bytes: every solve asks for
// both (the uncut entities, and the cut cells behind runtime_quadrature), and the classification has counted both
__global__ void __launch_bounds__(kBlock) locate_inside_cut_kernel(int64_t n, const uint8_t* __restrict__ bytes,
                                                                   const int64_t* __restrict__ off_in,
                                                                   const int64_t* __restrict__ off_cut,
                                                                   int32_t* __restrict__ out_in, int32_t* __restrict__ out_cut,
                                                                   DevN n_in_d, DevN n_cut_d, const uint8_t* __restrict__ block_class)
{
  // (lists sized by the previous step: nothing is written when a total did not fit -- dev_n is 0 in a void step)
  const int64_t n_in = dev_n(n_in_d), n_cut = dev_n(n_cut_d);
  const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kByteItems;
  unsigned f_in = 0, f_cut = 0;
  // what the culled classification knows about whole blocks of kClassBlock cells saves the bytes of the uniform ones:
  // a tile of outside blocks (two thirds of the 512^3 mesh) is left at once, an inside block lists its cells unread
  unsigned cls = 0;
  if (block_class)
  {
    constexpr int BPT = kByteTile / kClassBlock;
    const int64_t nblocks = (n + kClassBlock - 1) / kClassBlock;
    unsigned tile_all = 3u;
#pragma unroll
    for (int j = 0; j < BPT; ++j)
    {
      const int64_t bj = (int64_t)blockIdx.x * BPT + j;
      tile_all &= bj < nblocks ? (unsigned)block_class[bj] : 2u;
    }
    if (tile_all == 2u) return;
    cls = base < n ? (unsigned)block_class[base / kClassBlock] : 2u;
  }
  if (base < n && cls == 0u)
  {
    f_in = byte_flags(bytes, base, n, DomainMask{1});
    f_cut = byte_flags(bytes, base, n, DomainMask{2});
  }
  else if (base < n && cls == 1u)
    f_in = base + kByteItems <= n ? 0xffffu : ((1u << (int)(n - base)) - 1u);
  // the inside cells (the dense list) are packed in LDS and stored as one coalesced run per tile; the few cut
  // cells go out directly
  __shared__ int32_t s_in[kByteTile];
  int total_in, total;
  int o_in = block_exclusive_scan<int>(__popc(f_in), total_in);
  const int o_cut = block_exclusive_scan<int>(__popc(f_cut), total);
  int64_t b = off_cut[blockIdx.x] + o_cut;
#pragma unroll
  for (int k = 0; k < kByteItems; ++k)
  {
    if (f_in & (1u << k)) s_in[o_in++] = (int32_t)(base + k);
    if (f_cut & (1u << k)) { if (b < n_cut) out_cut[b] = (int32_t)(base + k); ++b; }
  }
  __syncthreads();
  const int64_t a = off_in[blockIdx.x];
  for (int i = threadIdx.x; i < total_in; i += kBlock)
    if (a + i < n_in) out_in[a + i] = s_in[i];
}

const DevArray<int32_t>& locate(cfx_cut_t cut, const std::string& selector)
{
  auto it = cut->located.find(selector);
  if (it != cut->located.end()) return it->second;
  const int64_t nh = cut->nhosts();
  SelectorPred pred{cut->domain.p, nh, parse_selector(selector.c_str(), cut->nls)};
  DevArray<int32_t> out;
  const uint8_t* bytes = reinterpret_cast<const uint8_t*>(cut->domain.p) + (int64_t)pred.sel.ls[0] * nh;
  if (pred.sel.n == 1 && (reinterpret_cast<uintptr_t>(bytes) & 15) == 0)
  {
    // "phi<0" / "phi=0" of the first level set: the classification already counted the tiles
    if (pred.sel.ls[0] == 0 && cut->tiles_inside.n > 0 && (pred.sel.mask[0] == 1 || pred.sel.mask[0] == 2)
        && cut->located.find("phi<0") == cut->located.end() && cut->located.find("phi=0") == cut->located.end()
        && (selector == "phi<0" || selector == "phi=0"))
    {
      const int64_t ntiles = cut->tiles_inside.n;
      DevArray<int64_t> off_in(ntiles + 1), off_cut(ntiles + 1);
      // (both totals in one round trip -- or none: inside a step they stay in HBM, published by the second scan, and
      // the lists are sized by the last step)
      const char* names[2] = {"locate.inside", "locate.cut"};
      const CountSource src[2] = {{off_in.p + ntiles, kCountI64, kCountUpTo}, {off_cut.p + ntiles, kCountI64, kCountUpTo}};
      CountPlan cp(2, names, src);
      exclusive_scan_pair(cut->tiles_inside.p, off_in.p, cut->tiles_cut.p, off_cut.p, ntiles, &cp);
      Count cnt[2];
      cp.finish(cnt);
      DevArray<int32_t> l_in(cnt[0].cap()), l_cut(cnt[1].cap());
      l_in.count = cnt[0]; l_cut.count = cnt[1];
      launch("locate_entities", locate_inside_cut_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, nh, bytes, off_in.p,
             off_cut.p, l_in.p, l_cut.p, l_in.devn(), l_cut.devn(),
             cut->block_class.n == (nh + kClassBlock - 1) / kClassBlock ? (const uint8_t*)cut->block_class.p : (const uint8_t*)nullptr);
      list_register(l_in.p, l_in.count);
      list_register(l_cut.p, l_cut.count);
      if (cut->host_width == 0 && cut->host_mask.n == 0) provenance_register(l_in.p, cut, cut->gen, -1, l_in.n);
      cut->located.emplace("phi<0", std::move(l_in));
      cut->located.emplace("phi=0", std::move(l_cut));
      return cut->located.find(selector)->second;
    }
    const int32_t* known = nullptr;
    if (pred.sel.ls[0] == 0 && cut->tiles_inside.n > 0)
      known = pred.sel.mask[0] == 1 ? cut->tiles_inside.p : (pred.sel.mask[0] == 2 ? cut->tiles_cut.p : nullptr);
    compact_bytes("locate_entities", nh, bytes, DomainMask{pred.sel.mask[0]}, out, known); // 1 B/cell stream
    // ("phi<0" / "phi>0" of level set 0 over all cells: the list is a function of the classification)
    if (pred.sel.ls[0] == 0 && cut->host_width == 0 && cut->host_mask.n == 0 && (pred.sel.mask[0] == 1 || pred.sel.mask[0] == 4))
      provenance_register(out.p, cut, cut->gen, pred.sel.mask[0] == 1 ? -1 : 1, out.n);
  }
  else
    compact("locate_entities", nh, pred, out);
  auto res = cut->located.emplace(selector, std::move(out));
  return res.first->second;
}

// the list with its exact length on the host (callers that size host-side work by it)
const DevArray<int32_t>& locate_exact(cfx_cut_t cut, const std::string& selector)
{
  DevArray<int32_t>& a = const_cast<DevArray<int32_t>&>(locate(cut, selector));
  if (a.count.cell)
  {
    a.n = a.count.value();
    list_unregister(a.p);
    a.count = Count();
  }
  return a;
}

void facet_runtime_quadrature(cfx_cut_t cut, const char* selector, int order, bool whole_hosts, cfx_rules_t* out);

} // namespace

extern "C" {

int cfx_cut_options_default(cfx_cut_options* opt)
{
  CFX_API_BEGIN
  require(opt != nullptr, CFX_ERR_INVALID_ARGUMENT, "null options");
  opt->cut_approximation_order = 1;
  opt->max_refinement_iterations = 8;
  opt->edge_max_depth = 20;
  opt->reserved = 0;
  CFX_API_END
}

int cfx_cut_create(cfx_mesh_t mesh, int nls, const int32_t* ls_dofmap, int ls_ndofs_cell, int64_t ls_ndofs,
                   const double* const* ls_values, const cfx_cut_options* opt, cfx_cut_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(mesh && out, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create: null mesh/output");
  // cut.cpp:97-107: at least one level set
  require(nls >= 1 && ls_values, CFX_ERR_INVALID_ARGUMENT, "cutfemx.cut requires at least one level-set function");
  require(nls <= 8, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create: at most 8 level sets");
  require(ls_dofmap && ls_ndofs > 0, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create: empty level-set dofmap");
  ensure_cases();
  auto cut = std::make_unique<cfx_cut_s>();
  cut->mesh = mesh;
  cut->nls = nls;
  cut->ls_ndofs_cell = ls_ndofs_cell;
  cut->ls_ndofs = ls_ndofs;
  if (opt) cut->options = *opt; else cfx_cut_options_default(&cut->options);
  require(cut->options.cut_approximation_order == 1, CFX_ERR_INVALID_ARGUMENT,
          "cfx_cut_create: only straight (order-1) cut approximation is implemented");
  cut->ls_dofmap = to_device(ls_dofmap, mesh->ncells * (int64_t)ls_ndofs_cell);
  for (int k = 0; k < nls; ++k)
  {
    require(ls_values[k] != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create: null level-set values");
    cut->ls_values.push_back(to_device(ls_values[k], ls_ndofs));
  }
  cut->domain.alloc((int64_t)nls * mesh->ncells);
  classify(cut.get());
  *out = cut.release();
  CFX_API_END
}

int cfx_cut_restrict(cfx_cut_t cut, const int32_t* cells, int64_t n)
{
  CFX_API_BEGIN
  require(cut && (cells || n == 0) && n >= 0, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_restrict: null argument");
  require(cut->host_width == 0, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_restrict: the cut is hosted by facets (pass the subset to cfx_cut_create_facets)");
  const int64_t nc = cut->mesh->ncells;
  DevArray<int32_t> dcells = to_device(cells, n);
  cut->host_mask.alloc(nc);
  cut->host_mask.zero();
  ZeroFlag bad;
  launch("mark_subset", mark_subset_kernel, grid_for(n), dim3(kBlock), 0, n, dcells.p, nc, cut->host_mask.p, bad.p);
  require(!read_scalar(bad.p), CFX_ERR_OUT_OF_RANGE, "cfx_cut_restrict: cell index out of range");
  classify(cut);
  CFX_API_END
}

int cfx_cut_update(cfx_cut_t cut, const double* const* ls_values)
{
  CFX_API_BEGIN
  require(cut != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_update: null handle");
  if (ls_values)
    for (int k = 0; k < cut->nls; ++k)
      if (ls_values[k]) cut->ls_values[k] = to_device(ls_values[k], cut->ls_ndofs);
  classify(cut);
  CFX_API_END
}

int cfx_cut_info(cfx_cut_t cut, int* tdim, int* gdim, int64_t* ncells, int* nls)
{
  CFX_API_BEGIN
  require(cut != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_info: null handle");
  if (tdim) *tdim = cut->host_dim();
  if (gdim) *gdim = cut->mesh->gdim;
  if (ncells) *ncells = cut->nhosts();
  if (nls) *nls = cut->nls;
  CFX_API_END
}

int cfx_cut_domain(cfx_cut_t cut, int ls, const int8_t** domain)
{
  CFX_API_BEGIN
  require(cut && domain, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_domain: null argument");
  require(ls >= 0 && ls < cut->nls, CFX_ERR_OUT_OF_RANGE, "level-set index out of range");
  *domain = cut->domain.p + (int64_t)ls * cut->nhosts();
  CFX_API_END
}

int cfx_locate_entities(cfx_cut_t cut, const char* selector, const int32_t** entities, int64_t* n)
{
  CFX_API_BEGIN
  require(cut && selector && entities && n, CFX_ERR_INVALID_ARGUMENT, "cfx_locate_entities: null argument");
  if (cut->host_width != 0) (void)locate_exact(cut, selector);
  const DevArray<int32_t>& a = locate(cut, selector);
  if (cut->host_width != 0)
  {
    // host_parent_index (cut.cpp:352-359): facet hosts answer with the caller's facet ids
    auto it = cut->located_ids.find(selector);
    if (it == cut->located_ids.end())
    {
      DevArray<int32_t> ids(a.n);
      if (a.n > 0)
        launch("locate_entities", gather_rows_kernel, grid_for(a.n), dim3(kBlock), 0, a.n, a.p, 1, cut->host_ids.p, ids.p);
      it = cut->located_ids.emplace(selector, std::move(ids)).first;
    }
    *entities = it->second.p;
    *n = it->second.n;
    return CFX_OK;
  }
  *entities = a.p;
  *n = a.count.cell ? a.count.cap() : a.n; // (a capacity while the step that made the list is open, cutfemx_amd.h)
  CFX_API_END
}

} // extern "C"

namespace
{
void multi_runtime_quadrature(cfx_cut_t cut, const Selector& sel, int order, cfx_rules_t* out)
{
  cfx_mesh_t mesh = cut->mesh;
  const int tdim = mesh->tdim;
  require(sel.term[sel.n - 1] == 0, CFX_ERR_INVALID_ARGUMENT,
          "runtime quadrature over several level sets takes one conjunction (clauses joined by 'and')");
  require(sel.n <= kMultiMaxClauses, CFX_ERR_INVALID_ARGUMENT, "runtime quadrature: too many clauses in the selector");
  MultiArgs A{};
  MultiRuleCell pred{cut->domain.p, mesh->ncells, sel.n, {}, {}};
  A.eq = -1;
  int nclip = 0;
  for (int k = 0; k < sel.n; ++k)
  {
    require(sel.mask[k] != 7 && sel.mask[k] != 5, CFX_ERR_INVALID_ARGUMENT, "runtime quadrature: unsupported relation");
    if (sel.mask[k] == 2)
    {
      require(A.eq < 0, CFX_ERR_INVALID_ARGUMENT, "runtime quadrature: at most one '=0' clause (a codimension-1 interface)");
      A.eq = k;
    }
    else ++nclip;
    A.mask[k] = sel.mask[k]; A.phi[k] = cut->ls_values[sel.ls[k]].p;
    pred.ls[k] = sel.ls[k]; pred.mask[k] = sel.mask[k];
  }
  require(nclip <= 3, CFX_ERR_INVALID_ARGUMENT, "runtime quadrature: at most three clipping clauses per conjunction");
  A.ncl = sel.n; A.order = order;
  DevArray<int32_t> cells;
  A.n = compact("multi_rule_cells", mesh->ncells, pred, cells);
  A.cells = cells.p; A.x = mesh->x.p; A.conn = mesh->conn.p; A.ls_dofmap = cut->ls_dofmap.p;
  auto r = std::make_unique<cfx_rules_s>();
  r->mesh = mesh; r->tdim = tdim; r->gdim = mesh->gdim;
  DevArray<int64_t> packed(A.n), packed_off(A.n + 1);
  A.packed = packed.p;
  const bool iface = A.eq >= 0;
  auto run = [&](bool emit)
  {
    if (A.n == 0) return;
    const dim3 grid = grid_for(A.n);
#define CFX_MULTI(T, D)                                                                                   \
  do                                                                                                      \
  {                                                                                                       \
    if (emit) launch("multi_rules_emit", multi_rules_kernel<T, D, true>, grid, dim3(kBlock), 0, A);       \
    else launch("multi_rules_count", multi_rules_kernel<T, D, false>, grid, dim3(kBlock), 0, A);          \
  } while (0)
    if (tdim == 2) { if (iface) CFX_MULTI(2, 1); else CFX_MULTI(2, 2); }
    else { if (iface) CFX_MULTI(3, 2); else CFX_MULTI(3, 3); }
#undef CFX_MULTI
  };
  ensure_cases();
  run(false);
  require(A.n < (1ll << 26), CFX_ERR_RUNTIME, "runtime quadrature: more than 2^26 cut cells");
  exclusive_scan(packed.p, packed_off.p, A.n);
  const int64_t totals = read_scalar(packed_off.p + A.n);
  const int64_t nq = totals & kPackMask, nr = totals >> kPackShift;
  require(nq < 2147483647LL, CFX_ERR_RUNTIME, "runtime quadrature: more than 2^31 points (int32 offsets)");
  r->nq = nq; r->nr = nr;
  r->points.alloc(nq * tdim); r->weights.alloc(nq); r->offsets.alloc(nr + 1); r->parent_map.alloc(nr);
  dev_fill(r->offsets.p, 0, sizeof(int32_t));
  A.packed_off = packed_off.p; A.points = r->points.p; A.weights = r->weights.p; A.offsets = r->offsets.p;
  A.parent_map = r->parent_map.p;
  run(true);
  CFX_HIP(hipStreamSynchronize(ctx().stream)); // (`cells` and the count arrays die with this frame)
  *out = r.release();
}
} // namespace

namespace
{
// runtime rules of one level set on cell hosts for one or two parts (PART_IN / PART_OUT / PART_IF) of the same cut:
// one count + scan per part, ONE read-back of the totals, one emit launch that stages every cut cell once
void simple_rules(cfx_cut_t cut, int n, const int* parts, int order, cfx_rules_t* out)
{
  cfx_mesh_t mesh = cut->mesh;
  const int tdim = mesh->tdim;
  const DevArray<int32_t>& cutc = locate(cut, "phi=0");
  const DevN ncut_d = cutc.devn();
  const int64_t ncut = ncut_d.cap; // (the capacity of the list while its length is still in HBM: grids and scans run over it)
  const double* phi = cut->ls_values[0].p;
  require(ncut < (1ll << 26), CFX_ERR_RUNTIME, "runtime quadrature: more than 2^26 cut cells");
  std::unique_ptr<cfx_rules_s> r[2];
  DevArray<int64_t> packed[2], packed_off[2];
  CountParts cparts{};
  cparts.n = n;
  for (int k = 0; k < n; ++k)
  {
    r[k] = std::make_unique<cfx_rules_s>();
    r[k]->mesh = mesh; r[k]->tdim = tdim; r[k]->gdim = mesh->gdim;
    packed[k].alloc(ncut);
    packed_off[k].alloc(ncut + 1);
    cparts.part[k] = parts[k];
    cparts.nref[k] = quad_npoints(parts[k] == PART_IF ? tdim - 1 : tdim, order);
    cparts.packed[k] = packed[k].p;
  }
  if (ncut > 0)
  {
    // (both parts from one pass over the cut cells)
    if (tdim == 2)
      launch("cut_count", cut_count_kernel<2>, grid_for(ncut), dim3(kBlock), 0, ncut_d, cutc.p, cut->ls_dofmap.p, phi, cparts);
    else
      launch("cut_count", cut_count_kernel<3>, grid_for(ncut), dim3(kBlock), 0, ncut_d, cutc.p, cut->ls_dofmap.p, phi, cparts);
  }
  // one scan for both totals of a part: a cut cell emits at most 3 sub-simplices x 64 points and 2 rules, so below
  // 2^26 cut cells the point prefix stays under 2^34 and the rule prefix under 2^27 (no carry, no sign bit).
  // the totals of all parts in one round trip (none inside a step: they stay in HBM, published by the last scan)
  const char* names[4] = {"rules.points.0", "rules.rules.0", "rules.points.1", "rules.rules.1"};
  CountSource src[4];
  Count totals[4];
  for (int k = 0; k < n; ++k)
  {
    src[2 * k] = CountSource{packed_off[k].p + ncut, kCountPackedLo, kCountUpTo};
    src[2 * k + 1] = CountSource{packed_off[k].p + ncut, kCountPackedHi, kCountUpTo};
  }
  CountPlan cp(2 * n, names, src);
  if (n == 2) exclusive_scan_pair(packed[0].p, packed_off[0].p, packed[1].p, packed_off[1].p, ncut, &cp);
  else
    for (int k = 0; k < n; ++k) exclusive_scan(packed[k].p, packed_off[k].p, ncut, k == n - 1 ? &cp : nullptr);
  cp.finish(totals);
  EmitJobs jobs{};
  jobs.n = n;
  for (int k = 0; k < n; ++k)
  {
    const int64_t nq = totals[2 * k].cap(), nr = totals[2 * k + 1].cap();
    // int32 offsets are part of the RuntimeQuadrature contract
    require(nq < 2147483647LL, CFX_ERR_RUNTIME, "runtime quadrature: more than 2^31 points (int32 offsets)");
    r[k]->nq = totals[2 * k]; r[k]->nr = totals[2 * k + 1];
    r[k]->points.alloc(nq * tdim);
    r[k]->weights.alloc(nq);
    r[k]->offsets.alloc(nr + 1);
    r[k]->parent_map.alloc(nr + 1); // (+ the sentinel behind the last rule, cut_emit_kernel)
    r[k]->parent_sentinel = true;
    // (offsets[0] = 0 is written by the cell that emits rule 0; an empty rule set gets it here)
    if (ncut == 0 || nr == 0) dev_fill(r[k]->offsets.p, 0, sizeof(int32_t));
    jobs.part[k] = parts[k]; jobs.packed_off[k] = packed_off[k].p;
    jobs.points[k] = r[k]->points.p; jobs.weights[k] = r[k]->weights.p;
    jobs.offsets[k] = r[k]->offsets.p; jobs.parent_map[k] = r[k]->parent_map.p;
  }
  if (ncut > 0)
  {
    const dim3 grid((unsigned)((ncut + kBlock / kEmitLanes - 1) / (kBlock / kEmitLanes)));
    if (tdim == 2)
      launch("cut_emit", cut_emit_kernel<2>, grid, dim3(kBlock), 0, ncut_d, cutc.p, mesh->x.p, mesh->conn.p,
             cut->ls_dofmap.p, phi, order, jobs);
    else
      launch("cut_emit", cut_emit_kernel<3>, grid, dim3(kBlock), 0, ncut_d, cutc.p, mesh->x.p, mesh->conn.p,
             cut->ls_dofmap.p, phi, order, jobs);
  }
  for (int k = 0; k < n; ++k) out[k] = r[k].release();
}
} // namespace

extern "C" {

int cfx_runtime_quadrature(cfx_cut_t cut, const char* selector, int order, const char* backend, cfx_rules_t* out)
{
  CFX_API_BEGIN
  require(cut && selector && out, CFX_ERR_INVALID_ARGUMENT, "cfx_runtime_quadrature: null argument");
  if (backend && strcmp(backend, "straight") != 0)
    throw Error(CFX_ERR_INVALID_ARGUMENT, std::string("unsupported runtime quadrature backend '") + backend
                                              + "' (only 'straight' is implemented)");
  require(order >= 0, CFX_ERR_INVALID_ARGUMENT, "quadrature order must be non-negative");
  require(order <= CFX_QUAD_MAX_DEGREE, CFX_ERR_INVALID_ARGUMENT, "quadrature order exceeds the built-in tables");
  if (cut->host_width != 0)
  {
    facet_runtime_quadrature(cut, selector, order, false, out);
    return CFX_OK;
  }
  cfx_mesh_t mesh = cut->mesh;
  const int tdim = mesh->tdim;
  require(cut->ls_ndofs_cell == tdim + 1, CFX_ERR_INVALID_ARGUMENT,
          "runtime quadrature requires a P1 level set (straight cuts)");
  const Selector sel = parse_selector(selector, cut->nls);
  if (cut->nls > 1 || sel.n > 1)
  {
    multi_runtime_quadrature(cut, sel, order, out);
    return CFX_OK;
  }
  const int m = sel.mask[0];
  const int part = (m == 2) ? PART_IF : ((m & 1) ? PART_IN : PART_OUT);
  simple_rules(cut, 1, &part, order, out);
  CFX_API_END
}

int cfx_runtime_quadratures(cfx_cut_t cut, int n, const char* const* selectors, int order, const char* backend,
                            cfx_rules_t* out)
{
  CFX_API_BEGIN
  require(cut && selectors && out && n >= 0, CFX_ERR_INVALID_ARGUMENT, "cfx_runtime_quadratures: null argument");
  for (int k = 0; k < n; ++k) out[k] = nullptr;
  // pairs of plain selectors of one level set on cell hosts share one pass over the cut cells; everything else is
  // the single call, selector by selector
  int k = 0;
  auto plain_part = [&](const char* text, int& part) -> bool
  {
    if (!text || cut->host_width != 0 || cut->nls != 1 || cut->ls_ndofs_cell != cut->mesh->tdim + 1) return false;
    if (backend && strcmp(backend, "straight") != 0) return false;
    if (order < 0 || order > CFX_QUAD_MAX_DEGREE) return false;
    Selector sel;
    try { sel = parse_selector(text, cut->nls); } catch (const Error&) { return false; }
    if (sel.n != 1) return false;
    const int m = sel.mask[0];
    if (m != 1 && m != 2 && m != 4) return false; // "<", "=", ">" only: the unions keep the single call
    part = (m == 2) ? PART_IF : ((m & 1) ? PART_IN : PART_OUT);
    return true;
  };
  // the handles produced so far are released on EVERY error path (a status code of the single call, or an exception
  // of the pair path: too many points, order out of range)
  auto release = [&]() { for (int q = 0; q < n; ++q) { delete out[q]; out[q] = nullptr; } };
  for (int q = 0; q < n; ++q) out[q] = nullptr;
  try
  {
    while (k < n)
    {
      int parts[2];
      if (k + 1 < n && plain_part(selectors[k], parts[0]) && plain_part(selectors[k + 1], parts[1]))
      {
        simple_rules(cut, 2, parts, order, out + k);
        k += 2;
        continue;
      }
      const int rc = cfx_runtime_quadrature(cut, selectors[k], order, backend, out + k);
      if (rc != CFX_OK)
      {
        release();
        return rc;
      }
      ++k;
    }
  }
  catch (...)
  {
    release();
    throw;
  }
  CFX_API_END
}

int cfx_full_cell_rules(cfx_mesh_t mesh, const int32_t* cells, int64_t n, int order, cfx_rules_t* out)
{
  CFX_API_BEGIN
  require(mesh && out && (cells || n == 0), CFX_ERR_INVALID_ARGUMENT, "cfx_full_cell_rules: null argument");
  require(order >= 0 && order <= CFX_QUAD_MAX_DEGREE, CFX_ERR_INVALID_ARGUMENT, "quadrature order out of range");
  const int tdim = mesh->tdim;
  const int nref = quad_npoints(tdim, order);
  require(n * nref < 2147483647LL, CFX_ERR_RUNTIME, "too many points for int32 offsets");
  DevArray<int32_t> dcells = to_device(cells, n);
  auto r = std::make_unique<cfx_rules_s>();
  r->mesh = mesh; r->tdim = tdim; r->gdim = mesh->gdim; r->nr = n; r->nq = n * nref;
  r->points.alloc(n * nref * tdim); r->weights.alloc(n * nref); r->offsets.alloc(n + 1); r->parent_map.alloc(n);
  dev_fill(r->offsets.p, 0, sizeof(int32_t));
  if (n > 0)
  {
    if (tdim == 2)
      launch("full_rules", full_rules_kernel<2>, grid_for(n), dim3(kBlock), 0, n, dcells.p, mesh->x.p, mesh->conn.p,
             order, r->points.p, r->weights.p, r->offsets.p, r->parent_map.p);
    else
      launch("full_rules", full_rules_kernel<3>, grid_for(n), dim3(kBlock), 0, n, dcells.p, mesh->x.p, mesh->conn.p,
             order, r->points.p, r->weights.p, r->offsets.p, r->parent_map.p);
  }
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  *out = r.release();
  CFX_API_END
}

int cfx_rules_create(cfx_mesh_t mesh, int tdim, int64_t nq, int64_t nr, const double* points, const double* weights,
                     const int32_t* offsets, const int32_t* parent_map, cfx_rules_t* out)
{
  CFX_API_BEGIN
  require(mesh && out, CFX_ERR_INVALID_ARGUMENT, "cfx_rules_create: null argument");
  require(tdim == mesh->tdim, CFX_ERR_INVALID_ARGUMENT, "rules tdim must match the mesh");
  require(nq >= 0 && nr >= 0 && offsets, CFX_ERR_INVALID_ARGUMENT, "cfx_rules_create: invalid sizes");
  auto r = std::make_unique<cfx_rules_s>();
  r->mesh = mesh; r->tdim = tdim; r->gdim = mesh->gdim; r->nq = nq; r->nr = nr;
  r->points = to_device(points, nq * tdim);
  r->weights = to_device(weights, nq);
  r->offsets = to_device(offsets, nr + 1);
  r->parent_map = to_device(parent_map, nr);
  *out = r.release();
  CFX_API_END
}

int cfx_rules_view_get(cfx_rules_t r, cfx_rules_view* v)
{
  CFX_API_BEGIN
  require(r && v, CFX_ERR_INVALID_ARGUMENT, "cfx_rules_view_get: null argument");
  // (capacities while the step that made the rules is open: the arrays are at least that long, cutfemx_amd.h)
  v->tdim = r->tdim; v->gdim = r->gdim; v->nq = r->nq.cap(); v->nr = r->nr.cap();
  v->points = r->points.p; v->weights = r->weights.p; v->offsets = r->offsets.p; v->parent_map = r->parent_map.p;
  v->host_width = r->host_width; v->reserved = 0;
  v->host_rows = r->host_width ? r->host_rows.p : nullptr;
  v->host_verts = r->host_width ? r->host_verts.p : nullptr;
  CFX_API_END
}

int cfx_rules_physical_points(cfx_rules_t r, double* out)
{
  CFX_API_BEGIN
  require(r && out, CFX_ERR_INVALID_ARGUMENT, "cfx_rules_physical_points: null argument");
  const int64_t nq = r->nq.value(), nr = r->nr.value(); // (a host array of nq points: the exact count)
  OutArray<double> o(out, nq * r->gdim, false);
  if (nq > 0 && r->host_width != 0)
  {
    // physical_points_for_host_mesh (cut.cpp:1344-1345)
    if (r->mesh->tdim == 2)
      launch("physical_points", facet_physical_points_kernel<2>, grid_for(nq), dim3(kBlock), 0, nq, nr,
             r->offsets.p, r->host_verts.p, r->points.p, r->mesh->x.p, o.dev);
    else
      launch("physical_points", facet_physical_points_kernel<3>, grid_for(nq), dim3(kBlock), 0, nq, nr,
             r->offsets.p, r->host_verts.p, r->points.p, r->mesh->x.p, o.dev);
  }
  else if (nq > 0)
  {
    if (r->tdim == 2)
      launch("physical_points", physical_points_kernel<2>, grid_for(nq), dim3(kBlock), 0, nq, nr,
             r->offsets.p, r->parent_map.p, r->points.p, r->mesh->x.p, r->mesh->conn.p, o.dev);
    else
      launch("physical_points", physical_points_kernel<3>, grid_for(nq), dim3(kBlock), 0, nq, nr,
             r->offsets.p, r->parent_map.p, r->points.p, r->mesh->x.p, r->mesh->conn.p, o.dev);
  }
  o.finish();
  CFX_API_END
}

int cfx_rules_destroy(cfx_rules_t r)
{
  CFX_API_BEGIN
  delete r;
  CFX_API_END
}

int cfx_evaluate_normals(cfx_cut_t cut, int ls, cfx_rules_t r, double sign, double* out)
{
  CFX_API_BEGIN
  require(cut && r && out, CFX_ERR_INVALID_ARGUMENT, "Cannot evaluate normals without a level set.");
  require(ls >= 0 && ls < cut->nls, CFX_ERR_OUT_OF_RANGE, "level-set index out of range");
  require(r->tdim == cut->mesh->tdim, CFX_ERR_RUNTIME, "Normal evaluation points must have cell reference dimension.");
  require(cut->ls_ndofs_cell == cut->mesh->tdim + 1, CFX_ERR_INVALID_ARGUMENT,
          "normal evaluation is implemented for P1 level sets");
  // a device destination takes the rules' capacity (nq.cap() doubles per component) while their length is in HBM;
  // a host destination needs the exact count
  if (!is_device_pointer(out)) (void)r->nq.value();
  OutArray<double> o(out, r->nq.cap() * r->gdim, false);
  if (r->nq.cap() > 0)
  {
    if (r->tdim == 2)
      launch("evaluate_normals", normals_rule_kernel<2>, grid_for(r->nr.cap()), dim3(kBlock), 0, r->nr, r->offsets.p,
             r->parent_map.p, cut->mesh->x.p, cut->mesh->conn.p, cut->ls_dofmap.p, cut->ls_values[ls].p, sign, o.dev);
    else
      launch("evaluate_normals", normals_rule_kernel<3>, grid_for(r->nr.cap()), dim3(kBlock), 0, r->nr, r->offsets.p,
             r->parent_map.p, cut->mesh->x.p, cut->mesh->conn.p, cut->ls_dofmap.p, cut->ls_values[ls].p, sign, o.dev);
  }
  o.finish();
  CFX_API_END
}

int cfx_evaluate_values(cfx_cut_t cut, int ls, cfx_rules_t r, double* out)
{
  CFX_API_BEGIN
  require(cut && r && out, CFX_ERR_INVALID_ARGUMENT, "Cannot evaluate values without a level set.");
  require(ls >= 0 && ls < cut->nls, CFX_ERR_OUT_OF_RANGE, "level-set index out of range");
  require(cut->ls_ndofs_cell == cut->mesh->tdim + 1, CFX_ERR_INVALID_ARGUMENT,
          "value evaluation is implemented for P1 level sets");
  const int64_t nq = r->nq.value(), nr = r->nr.value();
  OutArray<double> o(out, nq, false);
  if (nq > 0)
  {
    if (r->tdim == 2)
      launch("evaluate_values", values_kernel<2>, grid_for(nq), dim3(kBlock), 0, nq, nr, r->offsets.p,
             r->parent_map.p, r->points.p, cut->ls_dofmap.p, cut->ls_values[ls].p, o.dev);
    else
      launch("evaluate_values", values_kernel<3>, grid_for(nq), dim3(kBlock), 0, nq, nr, r->offsets.p,
             r->parent_map.p, r->points.p, cut->ls_dofmap.p, cut->ls_values[ls].p, o.dev);
  }
  o.finish();
  CFX_API_END
}

int cfx_ghost_penalty_facets(cfx_cut_t cut, const char* selector, const int32_t** rows, int64_t* n)
{
  CFX_API_BEGIN
  require(cut && selector && rows && n, CFX_ERR_INVALID_ARGUMENT, "cfx_ghost_penalty_facets: null argument");
  require(cut->host_width == 0, CFX_ERR_INVALID_ARGUMENT, "ghost_penalty_facets needs a cut hosted by the mesh cells");
  cfx_mesh_t mesh = cut->mesh;
  {
    auto it = cut->ghost_rows.find(selector);
    if (it != cut->ghost_rows.end())
    {
      *rows = it->second.p;
      *n = it->second.count.cell ? it->second.count.cap() : it->second.n / 4;
      return CFX_OK;
    }
  }
  SelectorPred pred{cut->domain.p, mesh->ncells, parse_selector(selector, cut->nls)};
  const DevArray<int32_t>& cutc = locate(cut, "phi=0");
  const DevN ncut_d = cutc.devn();
  const int64_t ncut = ncut_d.cap; // (capacity while the list's length is in HBM: the counts behind it stay zero)
  const DevArray<int32_t>& c2c = mesh->cell_neighbours();
  DevArray<int32_t> counts(ncut), cand(ncut * (int64_t)(mesh->tdim + 1) * 4);
  DevArray<int64_t> offs(ncut + 1);
  Count total(0);
  if (ncut > 0)
  {
    const int64_t nthreads = ncut;
    if (mesh->tdim == 2)
      launch("ghost_facets_find", ghost_facets_find_kernel<2>, grid_for(nthreads), dim3(kBlock), 0, ncut_d, cutc.p,
             mesh->conn.p, c2c.p, cut->domain.p, pred, counts.p, cand.p);
    else
      launch("ghost_facets_find", ghost_facets_find_kernel<3>, grid_for(nthreads), dim3(kBlock), 0, ncut_d, cutc.p,
             mesh->conn.p, c2c.p, cut->domain.p, pred, counts.p, cand.p);
    const char* gname = "ghost_facets";
    const CountSource gsrc{offs.p + ncut, kCountI64, kCountUpTo};
    CountPlan cp(1, &gname, &gsrc);
    exclusive_scan(counts.p, offs.p, ncut, &cp);
    cp.finish(&total);
  }
  DevArray<int32_t>& grows = cut->ghost_rows[selector];
  grows.alloc(total.cap() * 4);
  if (total.cap() > 0)
  {
    // (a total beyond the capacity voids the step before this launch: ncut_d then reads 0 and nothing is written)
    if (mesh->tdim == 2)
      launch("ghost_facets_pack", ghost_facets_pack_kernel<2>, grid_for(ncut), dim3(kBlock), 0, ncut_d, cand.p, offs.p,
             grows.p);
    else
      launch("ghost_facets_pack", ghost_facets_pack_kernel<3>, grid_for(ncut), dim3(kBlock), 0, ncut_d, cand.p, offs.p,
             grows.p);
  }
  if (total.cell)
  {
    grows.count = total;
    list_register(grows.p, total);
  }
  *rows = grows.p;
  *n = total.cap();
  CFX_API_END
}

int cfx_interior_facets_for_cells(cfx_mesh_t mesh, const int32_t* cells, int64_t n, int32_t** rows, int64_t* n_rows)
{
  CFX_API_BEGIN
  require(mesh && (cells || n == 0) && rows && n_rows && n >= 0, CFX_ERR_INVALID_ARGUMENT,
          "cfx_interior_facets_for_cells: null argument");
  const int64_t nc = mesh->ncells;
  DevArray<int32_t> dcells = to_device(cells, n);
  DevArray<uint8_t> inset(nc);
  inset.zero();
  ZeroFlag bad;
  launch("mark_subset", mark_subset_kernel, grid_for(n), dim3(kBlock), 0, n, dcells.p, nc, inset.p, bad.p);
  require(!read_scalar(bad.p), CFX_ERR_OUT_OF_RANGE, "cfx_interior_facets_for_cells: cell index out of range");
  DevArray<int32_t> sorted; // ascending, duplicates removed
  const int64_t m = compact("interior_facets", nc, ByteSet{inset.p}, sorted);
  const int nv = mesh->tdim + 1;
  DevArray<int32_t> counts(m), cand(m * nv * 4);
  DevArray<int64_t> offs(m + 1);
  int64_t total = 0;
  const Adjacency& adj = mesh->vertex_cells();
  if (m > 0)
  {
    counts.zero();
    if (mesh->tdim == 2)
      launch("interior_facets_find", interior_facets_find_kernel<2>, grid_for(m * nv), dim3(kBlock), 0, m, sorted.p,
             mesh->conn.p, adj.offsets.p, adj.cells.p, inset.p, counts.p, cand.p);
    else
      launch("interior_facets_find", interior_facets_find_kernel<3>, grid_for(m * nv), dim3(kBlock), 0, m, sorted.p,
             mesh->conn.p, adj.offsets.p, adj.cells.p, inset.p, counts.p, cand.p);
    exclusive_scan(counts.p, offs.p, m);
    total = read_scalar(offs.p + m);
  }
  int32_t* out = static_cast<int32_t*>(dev_alloc(sizeof(int32_t) * 4 * (size_t)(total > 0 ? total : 1)));
  if (total > 0)
  {
    if (mesh->tdim == 2)
      launch("ghost_facets_pack", ghost_facets_pack_kernel<2>, grid_for(m), dim3(kBlock), 0, m, cand.p, offs.p, out);
    else
      launch("ghost_facets_pack", ghost_facets_pack_kernel<3>, grid_for(m), dim3(kBlock), 0, m, cand.p, offs.p, out);
  }
  *rows = out;
  *n_rows = total;
  CFX_API_END
}

int cfx_cell_aggregation_create(cfx_cut_t cut, const char* selector, double threshold, int root_policy,
                                int max_iterations, int allow_rootless, cfx_aggregation_t* out)
{
  CFX_API_BEGIN
  require(cut && selector && out, CFX_ERR_INVALID_ARGUMENT, "cfx_cell_aggregation_create: null argument");
  require(cut->host_width == 0, CFX_ERR_INVALID_ARGUMENT, "Cell aggregation requires cell-hosted cut data."); // cell_aggregation.cpp:114
  require(threshold >= 0.0 && threshold <= 1.0, CFX_ERR_INVALID_ARGUMENT, "Volume fraction threshold must be in [0, 1].");
  require(root_policy == 0 || root_policy == 1, CFX_ERR_INVALID_ARGUMENT,
          "Unknown root policy. Expected 'interior_only' or 'interior_or_well_cut'.");
  const Selector sel = parse_selector(selector, cut->nls);
  require(sel.n == 1 && (sel.mask[0] == 1 || sel.mask[0] == 4), CFX_ERR_INVALID_ARGUMENT,
          "CellAggregation v1 expects a strict single level-set selector such as 'phi < 0' or 'phi > 0'.");
  cfx_mesh_t mesh = cut->mesh;
  const int64_t nc = mesh->ncells;
  const int tdim = mesh->tdim;
  const int8_t selcode = sel.mask[0] == 1 ? (int8_t)CFX_INSIDE : (int8_t)CFX_OUTSIDE;
  const int8_t* domain = cut->domain.p + (int64_t)sel.ls[0] * nc;
  auto A = std::make_unique<cfx_aggregation_s>();
  A->ncells = nc;
  A->root_cell.alloc(nc); A->aggregate_id.alloc(nc); A->depth.alloc(nc); A->fraction.alloc(nc);
  A->fraction.zero();
  dev_fill(A->root_cell.p, 0xff, sizeof(int32_t) * (size_t)nc);
  dev_fill(A->aggregate_id.p, 0xff, sizeof(int32_t) * (size_t)nc);
  dev_fill(A->depth.p, 0xff, sizeof(int32_t) * (size_t)nc);
  {
    // volume fraction of the selected part: order-1 rules of the cut cells (weights sum to the part's measure)
    cfx_rules_t rules = nullptr;
    const int rc = cfx_runtime_quadrature(cut, selector, 1, "straight", &rules);
    if (rc != CFX_OK) throw Error(rc, cfx_last_error());
    std::unique_ptr<cfx_rules_s> guard(rules);
    const int64_t nrules = rules->nr.value();
    if (nrules > 0)
    {
      if (tdim == 2)
        launch("agg_fraction", agg_fraction_kernel<2>, grid_for(nrules), dim3(kBlock), 0, nrules, rules->offsets.p,
               rules->parent_map.p, rules->weights.p, mesh->x.p, mesh->conn.p, A->fraction.p);
      else
        launch("agg_fraction", agg_fraction_kernel<3>, grid_for(nrules), dim3(kBlock), 0, nrules, rules->offsets.p,
               rules->parent_map.p, rules->weights.p, mesh->x.p, mesh->conn.p, A->fraction.p);
    }
    CFX_HIP(hipStreamSynchronize(ctx().stream)); // the rules die here
  }
  DevArray<int32_t> t(nc);
  DevArray<uint8_t> cls(nc);
  launch("agg_init", agg_init_kernel, grid_for(nc), dim3(kBlock), 0, nc, domain, selcode, root_policy, A->fraction.p,
         threshold, t.p, cls.p);
  const int64_t n_interior = compact("agg_lists", nc, ClsMask{cls.p, 1, 0}, A->interior);
  const int64_t n_cut = compact("agg_lists", nc, ClsMask{cls.p, 2, 0}, A->cut);
  compact("agg_lists", nc, ClsMask{cls.p, 3, 0}, A->active);
  const int64_t n_well = compact("agg_lists", nc, ClsMask{cls.p, 4, 0}, A->well);
  const int64_t n_ill = compact("agg_lists", nc, ClsMask{cls.p, 8, 0}, A->ill);
  (void)n_interior; (void)n_cut;
  launch("agg_roots", agg_roots_kernel, grid_for(n_well), dim3(kBlock), 0, n_well, A->well.p, A->root_cell.p,
         A->aggregate_id.p, A->depth.p);
  const Adjacency& adj = mesh->vertex_cells();
  ZeroFlag changed;
  DevArray<int32_t> parent(n_ill);
  if (n_ill > 0)
  {
    for (int64_t it = 0;; ++it)
    {
      require(it <= nc, CFX_ERR_RUNTIME, "cell aggregation: relaxation did not converge");
      changed.zero();
      if (tdim == 2)
        launch("agg_relax", agg_relax_kernel<2>, grid_for(n_ill), dim3(kBlock), 0, n_ill, A->ill.p, mesh->conn.p,
               adj.offsets.p, adj.cells.p, cls.p, t.p, changed.p);
      else
        launch("agg_relax", agg_relax_kernel<3>, grid_for(n_ill), dim3(kBlock), 0, n_ill, A->ill.p, mesh->conn.p,
               adj.offsets.p, adj.cells.p, cls.p, t.p, changed.p);
      if (!read_scalar(changed.p)) break;
    }
    const int64_t limit = max_iterations < 0 ? nc : max_iterations;
    if (tdim == 2)
      launch("agg_parent", agg_parent_kernel<2>, grid_for(n_ill), dim3(kBlock), 0, n_ill, A->ill.p, mesh->conn.p,
             adj.offsets.p, adj.cells.p, cls.p, t.p, limit, parent.p);
    else
      launch("agg_parent", agg_parent_kernel<3>, grid_for(n_ill), dim3(kBlock), 0, n_ill, A->ill.p, mesh->conn.p,
             adj.offsets.p, adj.cells.p, cls.p, t.p, limit, parent.p);
    for (int64_t it = 0;; ++it)
    {
      require(it <= nc, CFX_ERR_RUNTIME, "cell aggregation: root propagation did not converge");
      changed.zero();
      launch("agg_resolve", agg_resolve_kernel, grid_for(n_ill), dim3(kBlock), 0, n_ill, A->ill.p, parent.p,
             A->root_cell.p, A->aggregate_id.p, A->depth.p, changed.p);
      if (!read_scalar(changed.p)) break;
    }
    launch("agg_flag_rootless", agg_flag_rootless_kernel, grid_for(n_ill), dim3(kBlock), 0, n_ill, A->ill.p,
           A->root_cell.p, cls.p);
  }
  const int64_t n_rootless = compact("agg_lists", nc, ClsMask{cls.p, 16, 0}, A->rootless);
  require(allow_rootless || n_rootless == 0, CFX_ERR_RUNTIME,
          "CellAggregation found active ill-posed cells without an admissible root. Adjust the root policy or "
          "threshold, or explicitly allow rootless aggregation for diagnostics.");
  DevArray<int32_t> bad;
  A->n_pairs = compact("agg_lists", nc, ClsMask{cls.p, 8, 16}, bad);
  A->pairs.alloc(A->n_pairs * 4);
  launch("agg_pairs", agg_pairs_kernel, grid_for(A->n_pairs), dim3(kBlock), 0, A->n_pairs, bad.p, A->root_cell.p,
         A->pairs.p);
  *out = A.release();
  CFX_API_END
}

int cfx_cell_aggregation_view_get(cfx_aggregation_t agg, cfx_aggregation_view* v)
{
  CFX_API_BEGIN
  require(agg && v, CFX_ERR_INVALID_ARGUMENT, "cfx_cell_aggregation_view_get: null argument");
  v->ncells = agg->ncells;
  v->root_cell = agg->root_cell.p; v->aggregate_id = agg->aggregate_id.p; v->propagation_depth = agg->depth.p;
  v->cut_volume_fraction = agg->fraction.p;
  v->active_cells = agg->active.p; v->n_active = agg->active.n;
  v->cut_cells = agg->cut.p; v->n_cut = agg->cut.n;
  v->interior_cells = agg->interior.p; v->n_interior = agg->interior.n;
  v->well_posed_cells = agg->well.p; v->n_well_posed = agg->well.n;
  v->ill_posed_cells = agg->ill.p; v->n_ill_posed = agg->ill.n;
  v->rootless_cells = agg->rootless.p; v->n_rootless = agg->rootless.n;
  v->pairs = agg->pairs.p; v->n_pairs = agg->n_pairs;
  CFX_API_END
}

int cfx_cell_aggregation_destroy(cfx_aggregation_t agg)
{
  CFX_API_BEGIN
  delete agg;
  CFX_API_END
}

int cfx_cut_destroy(cfx_cut_t cut)
{
  CFX_API_BEGIN
  delete cut;
  CFX_API_END
}

} // extern "C"

const cfx::DevArray<int32_t>& cfx_mesh_s::cell_neighbours()
{
  if (c2c_built) return c2c;
  const Adjacency& adj = vertex_cells();
  c2c.alloc(ncells * (int64_t)(tdim + 1));
  if (ncells > 0)
  {
    const dim3 grid = xcd_grid((ncells + kBlock - 1) / kBlock);
    if (tdim == 2)
      launch("cell_neighbours", cell_neighbours_kernel<2, true>, grid, dim3(kBlock), 0, ncells, conn.p, adj.offsets.p,
             adj.cells.p, c2c.p);
    else
      launch("cell_neighbours", cell_neighbours_kernel<3, true>, grid, dim3(kBlock), 0, ncells, conn.p, adj.offsets.p,
             adj.cells.p, c2c.p);
  }
  c2c_built = true;
  cfx::publish_across_lanes();
  return c2c;
}

// ---------------------------------------------------------------------------
// 8f-4 facet hosts
// ---------------------------------------------------------------------------
namespace
{
void facet_runtime_quadrature(cfx_cut_t cut, const char* selector, int order, bool whole_hosts, cfx_rules_t* out)
{
  cfx_mesh_t mesh = cut->mesh;
  const int tdim = mesh->tdim, hd = tdim - 1;
  require(cut->nls == 1, CFX_ERR_INVALID_ARGUMENT, "runtime quadrature for several level sets is not implemented");
  auto r = std::make_unique<cfx_rules_s>();
  r->mesh = mesh; r->tdim = hd; r->gdim = mesh->gdim; r->host_width = cut->host_width;
  DevArray<int32_t> rule_host;
  const double* phi = cut->ls_values[0].p;
  if (whole_hosts)
  {
    const DevArray<int32_t>* hosts = selector ? &locate_exact(cut, selector) : nullptr;
    const int64_t n = hosts ? hosts->n : cut->n_hosts;
    const int nref = quad_npoints(hd, order);
    require(n * nref < 2147483647LL, CFX_ERR_RUNTIME, "too many points for int32 offsets");
    r->nr = n; r->nq = n * nref;
    r->points.alloc(n * nref * hd); r->weights.alloc(n * nref); r->offsets.alloc(n + 1); r->parent_map.alloc(n);
    rule_host.alloc(n);
    dev_fill(r->offsets.p, 0, sizeof(int32_t));
    if (n > 0)
    {
      if (tdim == 2)
        launch("facet_full_rules", facet_full_rules_kernel<2>, grid_for(n), dim3(kBlock), 0, n, hosts ? hosts->p : nullptr,
               mesh->x.p, cut->host_verts.p, cut->host_ids.p, order, r->points.p, r->weights.p, r->offsets.p,
               r->parent_map.p, rule_host.p);
      else
        launch("facet_full_rules", facet_full_rules_kernel<3>, grid_for(n), dim3(kBlock), 0, n, hosts ? hosts->p : nullptr,
               mesh->x.p, cut->host_verts.p, cut->host_ids.p, order, r->points.p, r->weights.p, r->offsets.p,
               r->parent_map.p, rule_host.p);
    }
  }
  else
  {
    const Selector sel = parse_selector(selector, cut->nls);
    require(sel.n == 1, CFX_ERR_INVALID_ARGUMENT, "runtime quadrature expects a single-clause selector");
    const int m = sel.mask[0];
    const int part = (m == 2) ? PART_IF : ((m & 1) ? PART_IN : PART_OUT);
    // phi = 0 on a facet: a point (segment hosts) or a straight segment (triangle hosts)
    const int nref = part == PART_IF ? (hd == 1 ? 1 : quad_npoints(1, order)) : quad_npoints(hd, order);
    const DevArray<int32_t>& cuth = locate_exact(cut, "phi=0");
    const int64_t ncut = cuth.n;
    DevArray<int32_t> n_rules(ncut), n_points(ncut), rule_off(ncut + 1), point_off(ncut + 1);
    if (ncut > 0)
    {
      if (tdim == 2)
        launch("facet_count", facet_count_kernel<2>, grid_for(ncut), dim3(kBlock), 0, ncut, cuth.p, cut->ls_dofmap.p, phi,
               part, nref, n_rules.p, n_points.p);
      else
        launch("facet_count", facet_count_kernel<3>, grid_for(ncut), dim3(kBlock), 0, ncut, cuth.p, cut->ls_dofmap.p, phi,
               part, nref, n_rules.p, n_points.p);
    }
    DevArray<int64_t> point_off64(ncut + 1);
    exclusive_scan(n_points.p, point_off64.p, ncut);
    const int64_t nq = read_scalar(point_off64.p + ncut);
    require(nq < 2147483647LL, CFX_ERR_RUNTIME, "runtime quadrature: more than 2^31 points (int32 offsets)");
    exclusive_scan(n_points.p, point_off.p, ncut);
    exclusive_scan(n_rules.p, rule_off.p, ncut);
    const int64_t nr = read_scalar(rule_off.p + ncut);
    r->nq = nq; r->nr = nr;
    r->points.alloc(nq * hd); r->weights.alloc(nq); r->offsets.alloc(nr + 1); r->parent_map.alloc(nr);
    rule_host.alloc(nr);
    dev_fill(r->offsets.p, 0, sizeof(int32_t));
    if (ncut > 0)
    {
      if (tdim == 2)
        launch("facet_emit", facet_emit_kernel<2>, grid_for(ncut), dim3(kBlock), 0, ncut, cuth.p, mesh->x.p,
               cut->host_verts.p, cut->ls_dofmap.p, phi, cut->host_ids.p, part, order, rule_off.p, point_off.p,
               r->points.p, r->weights.p, r->offsets.p, r->parent_map.p, rule_host.p);
      else
        launch("facet_emit", facet_emit_kernel<3>, grid_for(ncut), dim3(kBlock), 0, ncut, cuth.p, mesh->x.p,
               cut->host_verts.p, cut->ls_dofmap.p, phi, cut->host_ids.p, part, order, rule_off.p, point_off.p,
               r->points.p, r->weights.p, r->offsets.p, r->parent_map.p, rule_host.p);
    }
  }
  // the rules carry their hosts' rows and vertices: they outlive the cut object
  const int64_t nrh = r->nr.value();
  r->host_rows.alloc(nrh * cut->host_width);
  r->host_verts.alloc(nrh * tdim);
  if (nrh > 0)
  {
    launch("facet_gather_rows", gather_rows_kernel, grid_for(nrh * cut->host_width), dim3(kBlock), 0, nrh, rule_host.p,
           cut->host_width, cut->host_rows.p, r->host_rows.p);
    launch("facet_gather_rows", gather_rows_kernel, grid_for(nrh * tdim), dim3(kBlock), 0, nrh, rule_host.p, tdim,
           cut->host_verts.p, r->host_verts.p);
  }
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  *out = r.release();
}
} // namespace

extern "C" {

int cfx_exterior_facets(cfx_mesh_t mesh, int32_t** rows, int64_t* n_rows)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(mesh && rows && n_rows, CFX_ERR_INVALID_ARGUMENT, "cfx_exterior_facets: null argument");
  const int64_t nc = mesh->ncells;
  const Adjacency& adj = mesh->vertex_cells();
  DevArray<uint8_t> mask(nc);
  if (mesh->tdim == 2)
    launch("exterior_facets", exterior_mask_kernel<2>, grid_for(nc), dim3(kBlock), 0, nc, mesh->conn.p, adj.offsets.p,
           adj.cells.p, mask.p);
  else
    launch("exterior_facets", exterior_mask_kernel<3>, grid_for(nc), dim3(kBlock), 0, nc, mesh->conn.p, adj.offsets.p,
           adj.cells.p, mask.p);
  DevArray<int32_t> cells;
  const int64_t m = compact("exterior_facets", nc, ByteSet{mask.p}, cells);
  DevArray<int32_t> counts(m);
  DevArray<int64_t> offs(m + 1);
  int64_t total = 0;
  if (m > 0)
  {
    launch("exterior_facets", exterior_count_kernel, grid_for(m), dim3(kBlock), 0, m, cells.p, mask.p, counts.p);
    exclusive_scan(counts.p, offs.p, m);
    total = read_scalar(offs.p + m);
  }
  int32_t* out = static_cast<int32_t*>(dev_alloc(sizeof(int32_t) * 2 * (size_t)(total > 0 ? total : 1)));
  if (total > 0)
    launch("exterior_facets", exterior_pack_kernel, grid_for(m), dim3(kBlock), 0, m, cells.p, mask.p, offs.p, out);
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  *rows = out;
  *n_rows = total;
  CFX_API_END
}

int cfx_cut_create_facets(cfx_mesh_t mesh, int64_t n, const int32_t* facet_ids, const int32_t* rows, int row_width,
                          const int32_t* entity_geometry, int nls, const int32_t* ls_dofmap, int ls_ndofs_cell,
                          int64_t ls_ndofs, const double* const* ls_values, const cfx_cut_options* opt, cfx_cut_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(mesh && out, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: null mesh/output");
  require(n >= 0 && (rows || n == 0), CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: null facet rows");
  require(row_width == 2 || row_width == 4, CFX_ERR_INVALID_ARGUMENT,
          "cfx_cut_create_facets: rows are (cell, local facet) or (cell0, local facet0, cell1, local facet1)");
  require(nls >= 1 && ls_values, CFX_ERR_INVALID_ARGUMENT, "cutfemx.cut requires at least one level-set function");
  require(nls <= 8, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: at most 8 level sets");
  require(ls_dofmap && ls_ndofs > 0, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: empty level-set dofmap");
  const int tdim = mesh->tdim;
  require(ls_ndofs_cell == tdim + 1, CFX_ERR_INVALID_ARGUMENT,
          "cfx_cut_create_facets: facet hosts take a P1 level set (its dofs on a facet are its vertex dofs)");
  ensure_cases();
  auto cut = std::make_unique<cfx_cut_s>();
  cut->mesh = mesh;
  cut->nls = nls;
  cut->ls_ndofs_cell = tdim; // dofs per host
  cut->ls_ndofs = ls_ndofs;
  if (opt) cut->options = *opt; else cfx_cut_options_default(&cut->options);
  require(cut->options.cut_approximation_order == 1, CFX_ERR_INVALID_ARGUMENT,
          "cfx_cut_create_facets: only straight (order-1) cut approximation is implemented");
  cut->host_width = row_width;
  cut->n_hosts = n;
  cut->host_rows.alloc(n * row_width);
  cut->host_ids.alloc(n);
  cut->host_verts.alloc(n * tdim);
  cut->ls_dofmap.alloc(n * tdim);
  {
    DevArray<int32_t> drows = to_device(rows, n * row_width);
    DevArray<int32_t> dgeom = to_device(entity_geometry, entity_geometry ? n * tdim : 0);
    DevArray<int32_t> dls = to_device(ls_dofmap, mesh->ncells * (int64_t)ls_ndofs_cell);
    ZeroFlag bad;
    if (n > 0)
    {
      CFX_HIP(hipMemcpyAsync(cut->host_rows.p, drows.p, sizeof(int32_t) * (size_t)(n * row_width), hipMemcpyDeviceToDevice,
                             ctx().stream));
      if (facet_ids)
      {
        DevArray<int32_t> dids = to_device(facet_ids, n);
        CFX_HIP(hipMemcpyAsync(cut->host_ids.p, dids.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx().stream));
        CFX_HIP(hipStreamSynchronize(ctx().stream));
      }
      else
        launch("iota", iota_kernel, grid_for(n), dim3(kBlock), 0, n, cut->host_ids.p);
      const int32_t* geom = entity_geometry ? dgeom.p : nullptr;
      if (tdim == 2)
        launch("facet_hosts", facet_hosts_kernel<2>, grid_for(n), dim3(kBlock), 0, n, drows.p, row_width, geom, mesh->conn.p,
               mesh->ncells, dls.p, cut->host_verts.p, cut->ls_dofmap.p, bad.p);
      else
        launch("facet_hosts", facet_hosts_kernel<3>, grid_for(n), dim3(kBlock), 0, n, drows.p, row_width, geom, mesh->conn.p,
               mesh->ncells, dls.p, cut->host_verts.p, cut->ls_dofmap.p, bad.p);
    }
    const int b = read_scalar(bad.p);
    require(b != 1, CFX_ERR_OUT_OF_RANGE, "cfx_cut_create_facets: cell index or local facet out of range");
    require(b != 2, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: entity_geometry names a vertex that is not on the facet");
    require(b != 3, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: the two (cell, local facet) pairs of a row are different facets");
  }
  for (int k = 0; k < nls; ++k)
  {
    require(ls_values[k] != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_cut_create_facets: null level-set values");
    cut->ls_values.push_back(to_device(ls_values[k], ls_ndofs));
  }
  cut->domain.alloc((int64_t)nls * n);
  classify(cut.get());
  *out = cut.release();
  CFX_API_END
}

int cfx_full_facet_rules(cfx_cut_t cut, const char* selector, int order, cfx_rules_t* out)
{
  CFX_API_BEGIN
  require(cut && out, CFX_ERR_INVALID_ARGUMENT, "cfx_full_facet_rules: null argument");
  require(cut->host_width != 0, CFX_ERR_INVALID_ARGUMENT, "cfx_full_facet_rules: the cut is hosted by cells");
  require(order >= 0 && order <= CFX_QUAD_MAX_DEGREE, CFX_ERR_INVALID_ARGUMENT, "quadrature order out of range");
  facet_runtime_quadrature(cut, selector, order, true, out);
  CFX_API_END
}

int cfx_facet_rules_to_cells(cfx_rules_t R, int side, cfx_rules_t* out)
{
  CFX_API_BEGIN
  require(R && out, CFX_ERR_INVALID_ARGUMENT, "cfx_facet_rules_to_cells: null argument");
  require(R->host_width != 0, CFX_ERR_INVALID_ARGUMENT, "cfx_facet_rules_to_cells: the rules are hosted by cells");
  require(side == 0 || (side == 1 && R->host_width == 4), CFX_ERR_INVALID_ARGUMENT,
          "cfx_facet_rules_to_cells: side is 0, or 1 for interior-facet rows");
  cfx_mesh_t mesh = R->mesh;
  const int tdim = mesh->tdim;
  const int64_t nr = R->nr.value(), nq = R->nq.value();
  auto r = std::make_unique<cfx_rules_s>();
  r->mesh = mesh; r->tdim = tdim; r->gdim = mesh->gdim; r->nr = nr; r->nq = nq;
  r->points.alloc(nq * tdim); r->weights.alloc(nq); r->offsets.alloc(nr + 1); r->parent_map.alloc(nr);
  dev_fill(r->offsets.p, 0, sizeof(int32_t));
  if (nr > 0)
  {
    // cell-hosted rules are consumed in ascending parent order (runs of one parent are contiguous)
    DevArray<int32_t> keys(nr), perm(nr), counts(nr);
    ZeroFlag unsorted;
    launch("facet_to_cells", rule_cell_keys_kernel, grid_for(nr), dim3(kBlock), 0, nr, R->host_rows.p, R->host_width, side,
           keys.p, unsorted.p);
    launch("iota", iota_kernel, grid_for(nr), dim3(kBlock), 0, nr, perm.p);
    if (read_scalar(unsorted.p))
    {
      DevArray<int32_t> keys_out(nr), perm_out(nr);
      size_t tmp_bytes = 0;
      CFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.p, keys_out.p, perm.p, perm_out.p, (size_t)nr, 0, 32,
                                        ctx().stream));
      DevArray<uint8_t> tmp((int64_t)tmp_bytes);
      CFX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.p, keys_out.p, perm.p, perm_out.p, (size_t)nr, 0, 32,
                                        ctx().stream));
      CFX_HIP(hipStreamSynchronize(ctx().stream));
      perm = std::move(perm_out);
    }
    launch("facet_to_cells", permuted_counts_kernel, grid_for(nr), dim3(kBlock), 0, nr, perm.p, R->offsets.p, counts.p);
    exclusive_scan(counts.p, r->offsets.p, nr);
    if (nq > 0)
    {
      if (tdim == 2)
        launch("facet_to_cells", facet_to_cell_kernel<2>, grid_for(nq), dim3(kBlock), 0, nq, nr, r->offsets.p, perm.p,
               R->offsets.p, R->host_rows.p, R->host_width, side, R->host_verts.p, R->points.p, R->weights.p, mesh->conn.p,
               r->points.p, r->weights.p, r->parent_map.p);
      else
        launch("facet_to_cells", facet_to_cell_kernel<3>, grid_for(nq), dim3(kBlock), 0, nq, nr, r->offsets.p, perm.p,
               R->offsets.p, R->host_rows.p, R->host_width, side, R->host_verts.p, R->points.p, R->weights.p, mesh->conn.p,
               r->points.p, r->weights.p, r->parent_map.p);
    }
    CFX_HIP(hipStreamSynchronize(ctx().stream));
  }
  *out = r.release();
  CFX_API_END
}

} // extern "C"
