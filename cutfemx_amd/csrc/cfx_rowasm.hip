// cutfemx_amd: row-centric kernels, part 1 -- the form's row plan and the CSR
// sparsity (the gather assembly that uses them is in cfx_gather.hip).
//
// Decomposition: a group of G lanes owns one matrix row (one dof r).  The
// "items" of the row are the marked cells incident to r (static dof->cells
// incidence of the space) and the ghost-penalty facets incident to r (dof->
// facets incidence of the form).  Each lane forms the local row of one item in
// registers -- the same (entity, local row) work unit the reference's cell loop
// produces one after the other (assemble_matrix_impl.h:103-188, :462-606) --
// and the group reduces the items into the row's CSR slots in LDS in item
// order, so no global atomics are issued and the sums do not depend on timing.
#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>

#include "cfx_elem.h"

using namespace cfx;

namespace
{

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------
__global__ void plan_mark_cells_kernel(DevN n_d, const int32_t* __restrict__ cells, int stride, uint8_t bit,
                                       uint8_t* mark)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // Launches are serialised on the stream and every thread of one launch ORs the
  // same bit, so a repeated cell (interface rules) is a benign same-value race.
  const int64_t c = cells[i * stride];
  mark[c] = mark[c] | bit;
}

// bit `bit` of 64 consecutive cell marks -> one word; its popcount for the rank scan
// `marked_tiles` (optional): marked cells (any bit) per compaction tile of kByteTile cells = the 64 words of one
// wavefront -- the count pass of the active-cell list (cfx_active_domain) for free
__global__ void __launch_bounds__(kBlock) plan_pack_bits_kernel(int64_t ncells, const uint8_t* __restrict__ mark, uint8_t bit,
                                                                unsigned long long* __restrict__ words,
                                                                int32_t* __restrict__ pop, int32_t* __restrict__ marked_tiles)
{
  static_assert(kByteTile == 64 * 64, "one wavefront of 64-cell words per compaction tile");
  const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = w * 64 < ncells;
  unsigned long long v = 0;
  int marked = 0;
  const int64_t base = w * 64;
  if (live && base + 64 <= ncells)
  {
    const uint4* p = reinterpret_cast<const uint4*>(mark + base);
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      const uint4 u = p[q];
      const unsigned x[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int k = 0; k < 16; ++k)
      {
        const unsigned byte = (x[k >> 2] >> (8 * (k & 3))) & 0xffu;
        if (byte & bit) v |= 1ull << (16 * q + k);
        marked += byte ? 1 : 0;
      }
    }
  }
  else if (live)
    for (int k = 0; base + k < ncells; ++k)
    {
      if (mark[base + k] & bit) v |= 1ull << k;
      marked += mark[base + k] ? 1 : 0;
    }
  if (live)
  {
    words[w] = v;
    pop[w] = __popcll(v);
  }
  if (marked_tiles)
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) marked += __shfl_xor(marked, o, 64);
    if ((threadIdx.x & 63) == 0 && live) marked_tiles[w >> 6] = marked;
  }
}

// the same words for the uncut entities of a located list, from the classification bytes: bit k of word w = (domain of
// cell 64 w + k == value).  A block of kClassBlock cells that the culled classification found uniform is not read.
__device__ __forceinline__ void pack_bits_domain_body(int64_t w, int64_t ncells, const int8_t* __restrict__ domain, int8_t value,
                                                      const uint8_t* __restrict__ block_class,
                                                      unsigned long long* __restrict__ words, int32_t* __restrict__ pop,
                                                      const int64_t* __restrict__ poison)
{
  const int64_t base = w * 64;
  if (base >= ncells) return;
  unsigned long long v = 0;
  int known = -1;
  if (poison != nullptr && *poison != 0) known = 0; // (a void step: no entities, as when the list has length 0)
  else if (block_class)
  {
    const unsigned c = block_class[base / kClassBlock];
    if (c == 1u) known = -1 == value ? 1 : 0;
    else if (c == 2u) known = 1 == value ? 1 : 0;
  }
  if (known == 1 && base + 64 <= ncells) v = ~0ull;
  else if (known == 0) {}
  else if (base + 64 <= ncells)
  {
    const uint4* p = reinterpret_cast<const uint4*>(domain + base);
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      const uint4 u = p[q];
      const unsigned x[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if ((int8_t)((x[k >> 2] >> (8 * (k & 3))) & 0xffu) == value) v |= 1ull << (16 * q + k);
    }
  }
  else
    for (int k = 0; base + k < ncells; ++k)
      if (domain[base + k] == value) v |= 1ull << k;
  words[w] = v;
  pop[w] = __popcll(v);
}
__global__ void __launch_bounds__(kBlock) pack_bits_domain_kernel(int64_t ncells, const int8_t* __restrict__ domain, int8_t value,
                                                                  const uint8_t* __restrict__ block_class,
                                                                  unsigned long long* __restrict__ words, int32_t* __restrict__ pop,
                                                                  const int64_t* __restrict__ poison)
{
  pack_bits_domain_body((int64_t)blockIdx.x * kBlock + threadIdx.x, ncells, domain, value, block_class, words, pop, poison);
}

// uncut entities of a cell integral, nd <= 4 dofs per cell, in one pass over the list: cell mark, row marks
// (dofmap row as one 16 B load when nd == 4) and the ascending check
template <int ND>
__global__ void __launch_bounds__(kBlock) plan_mark_entities_kernel(DevN n_d, const int32_t* __restrict__ cells,
                                                                    const int32_t* __restrict__ dofmap, uint8_t bit,
                                                                    uint8_t* mark, uint8_t* rowmark, int* flag)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cells[i];
  if (i + 1 < n && cells[i + 1] <= c) atomicOr(flag, 1); // strictly ascending (plan_check_sorted_kernel, strict)
  mark[c] = mark[c] | bit;
  int32_t d[ND];
  if constexpr (ND == 4)
  {
    const int4 v = *reinterpret_cast<const int4*>(dofmap + c * 4);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  else
  {
#pragma unroll
    for (int j = 0; j < ND; ++j) d[j] = dofmap[c * ND + j];
  }
#pragma unroll
  for (int j = 0; j < ND; ++j) rowmark[d[j]] = 1;
}

// runtime rules of a cell integral, nd <= 4: one pass over the parent list -- ascending check, and at the first
// rule of every parent its cell mark, row marks (special rows) and hash-map entry parent -> first rule
// (the rule sets of all cell integrals of a form in one launch: job k covers the threads [start[k], start[k + 1]))
struct RuleJobs
{
  int n;
  int64_t start[5];
  DevN nr[4];
  const int32_t* parent[4];
  uint8_t bit[4];
  uint32_t hmask[4];
  int32_t* keys[4];
  int32_t* first[4];
  // the rows of the form's one interior-facet list ride in the same launch (threads from start[n] on): null = none
  DevN nf;
  const int32_t* facet_rows;
};
template <int ND>
__device__ __forceinline__ void plan_facet_rows_body(int64_t f, DevN nf_d, const int32_t* __restrict__ rows,
                                                     const int32_t* __restrict__ dofmap, uint8_t* rowmark, uint8_t* special, int* flag);
template <int ND>
__global__ void __launch_bounds__(kBlock) plan_rules_kernel(RuleJobs J, const int32_t* __restrict__ dofmap, uint8_t* mark,
                                                            uint8_t* rowmark, uint8_t* special, int* flag)
{
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (J.facet_rows != nullptr && t >= J.start[J.n])
  {
    plan_facet_rows_body<ND>(t - J.start[J.n], J.nf, J.facet_rows, dofmap, rowmark, special, flag);
    return;
  }
  int k = 0;
  while (k + 1 < J.n && t >= J.start[k + 1]) ++k;
  const int64_t e = t - J.start[k];
  const int64_t nr = dev_n(J.nr[k]);
  if (e >= nr) return;
  const int32_t* __restrict__ parent = J.parent[k];
  const uint8_t bit = J.bit[k];
  const uint32_t hmask = J.hmask[k];
  int32_t* __restrict__ keys = J.keys[k];
  int32_t* __restrict__ first = J.first[k];
  const int32_t c = parent[e];
  if (e > 0)
  {
    const int32_t prev = parent[e - 1];
    if (prev > c) atomicOr(flag, 1);
    if (prev == c) return;
  }
  // (two jobs of this launch may mark the same cell with different bits: a word-wide atomic OR on the byte's word;
  // the mark array starts on a 16 B boundary and is padded to a multiple of four)
  atomicOr(reinterpret_cast<unsigned int*>(mark + (c & ~3)), (unsigned int)bit << (8 * (c & 3)));
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    const int32_t dof = dofmap[(int64_t)c * ND + j];
    rowmark[dof] = 1;
    special[dof] = 1;
  }
  uint32_t h = cfx_hash32((uint32_t)c) & hmask;
  while (true)
  {
    const int32_t old = atomicCAS(&keys[h], -1, c);
    if (old == -1 || old == c) break;
    h = (h + 1) & hmask;
  }
  first[h] = (int32_t)e;
}

// interior-facet entities, nd <= 4: row marks (special rows) of both cells of every row
template <int ND>
__device__ __forceinline__ void plan_facet_rows_body(int64_t f, DevN nf_d, const int32_t* __restrict__ rows,
                                                     const int32_t* __restrict__ dofmap, uint8_t* rowmark, uint8_t* special, int* flag)
{
  const int64_t nf = dev_n(nf_d);
  if (f >= nf) return;
  const int4 r = *reinterpret_cast<const int4*>(rows + 4 * f);
  int32_t d0[ND], d1[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    d0[j] = dofmap[(int64_t)r.x * ND + j]; d1[j] = dofmap[(int64_t)r.z * ND + j];
    rowmark[d0[j]] = 1; special[d0[j]] = 1;
    rowmark[d1[j]] = 1; special[d1[j]] = 1;
  }
  // P1 facet folding needs all dofs but one per cell shared (plan_check_fold_kernel): flag bit 1 otherwise
  int nfree = 0;
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    bool shared = false;
#pragma unroll
    for (int i = 0; i < ND; ++i) shared = shared || d1[j] == d0[i];
    nfree += shared ? 0 : 1;
  }
  if (nfree != 1) atomicOr(flag, 2);
}
template <int ND>
__global__ void __launch_bounds__(kBlock) plan_facet_rows_kernel(DevN nf_d, const int32_t* __restrict__ rows,
                                                                 const int32_t* __restrict__ dofmap, uint8_t* rowmark,
                                                                 uint8_t* special, int* flag)
{
  plan_facet_rows_body<ND>((int64_t)blockIdx.x * kBlock + threadIdx.x, nf_d, rows, dofmap, rowmark, special, flag);
}

// `special` (may be null): rows that receive something other than uncut-cell items
__global__ void plan_mark_rows_cells_kernel(DevN n_d, const int32_t* __restrict__ cells, int stride,
                                            const int32_t* __restrict__ dofmap, int nd, uint8_t* rowmark,
                                            uint8_t* special)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * nd) return;
  const int64_t c = cells[(i / nd) * stride];
  const int32_t dof = dofmap[c * nd + (int)(i % nd)];
  rowmark[dof] = 1;
  if (special) special[dof] = 1;
}


// ---------------------------------------------------------------------------
// Bulk rows (cfx_row_plan::rowcls).  The reference marks cell by cell and row by row (deactivate.h:103-183,
// assembler.h:442-560).  Where the uncut entities of a form are "the cells on one side of level set 0" and the space's
// dofs are the level set's (P1 on the geometry dofmap), a dof on that side with no cut cell around it has ONLY such
// cells around it -- active, whole stencil, one mark -- and a dof on the other side with no cut cell around it has none:
// neither needs a mark to be written or read.  The classification leaves the two bytes that say so per dof (sign code,
// touch byte: cfx_cut.hip).
// ---------------------------------------------------------------------------
constexpr uint8_t kRowOut = 0, kRowIn = 1, kRowMix = 2;

// 16 rows per thread: class of every row (`sel`: sign code of the entities' side), and the rows' initial marks --
// rowmark = 1 on a bulk row, 0 elsewhere; special = 0; the segment offsets of a linear form's staging = 0 ("none")
// wherever they can be read (rows that can lie on an entity).  All arrays are padded to a multiple of 16 rows.
__device__ __forceinline__ void row_class_body(int64_t thread, int64_t ndofs, const uint8_t* __restrict__ codes,
                                               const uint8_t* __restrict__ touch, uint8_t sel,
                                               uint8_t* __restrict__ rowcls, uint8_t* __restrict__ rowmark,
                                               uint8_t* __restrict__ special, int32_t* __restrict__ t2off,
                                               const int64_t* __restrict__ poison)
{
  const int64_t r0 = thread * 16;
  if (r0 >= ndofs) return;
  // (a void step: every list-driven kernel sees length 0 and marks nothing -- the rows must not be marked either, or the
  // kernels that walk the marks would read lengths nobody wrote)
  const bool void_step = poison != nullptr && *poison != 0;
  unsigned c[4] = {0u, 0u, 0u, 0u}, t[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
  if (void_step) { c[0] = c[1] = c[2] = c[3] = (3u - sel) * 0x01010101u; t[0] = t[1] = t[2] = t[3] = 0u; }
  else if (r0 + 16 <= ndofs && ((reinterpret_cast<uintptr_t>(codes + r0) | reinterpret_cast<uintptr_t>(touch + r0)) & 15) == 0)
  {
    const uint4 cv = *reinterpret_cast<const uint4*>(codes + r0), tv = *reinterpret_cast<const uint4*>(touch + r0);
    c[0] = cv.x; c[1] = cv.y; c[2] = cv.z; c[3] = cv.w;
    t[0] = tv.x; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w;
  }
  else
    for (int k = 0; k < 16 && r0 + k < ndofs; ++k) // (rows beyond the last one stay "touched, code 0": class 2, mark 0)
    {
      c[k >> 2] |= (unsigned)codes[r0 + k] << (8 * (k & 3));
      t[k >> 2] = (t[k >> 2] & ~(0xffu << (8 * (k & 3)))) | ((unsigned)touch[r0 + k] << (8 * (k & 3)));
    }
  unsigned cls[4], rm[4];
  bool any_in = false, all_out = true;
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    cls[q] = 0u; rm[q] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const unsigned code = (c[q] >> (8 * k)) & 0xffu, tch = (t[q] >> (8 * k)) & 0xffu;
      const unsigned cl = tch ? kRowMix : (code == sel ? kRowIn : (code == 3u - sel ? kRowOut : kRowMix));
      cls[q] |= cl << (8 * k);
      rm[q] |= (cl == kRowIn ? 1u : 0u) << (8 * k);
      any_in = any_in || cl == kRowIn;
      all_out = all_out && cl == kRowOut;
    }
  }
  *reinterpret_cast<uint4*>(rowcls + r0) = make_uint4(cls[0], cls[1], cls[2], cls[3]);
  *reinterpret_cast<uint4*>(rowmark + r0) = make_uint4(rm[0], rm[1], rm[2], rm[3]);
  *reinterpret_cast<uint4*>(special + r0) = make_uint4(0u, 0u, 0u, 0u);
  if (t2off && !all_out)
  {
    if (r0 + 16 <= ndofs)
    {
      uint4* q = reinterpret_cast<uint4*>(t2off + r0);
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    else
      for (int64_t r = r0; r < ndofs; ++r) t2off[r] = 0;
  }
}

// cell marks of the uncut entities of a located list straight from the classification bytes: mark = bits where
// domain == value, 0 elsewhere -- 16 cells per thread; a block of kClassBlock cells the culled classification found
// uniform (block_class 1: all inside, 2: all outside) is written without being read.  Covers the padded array.
__device__ __forceinline__ void cellmark_from_domain_body(int64_t thread, int64_t ncells, int64_t npad,
                                                          const int8_t* __restrict__ domain, int8_t value, uint8_t bits,
                                                          const uint8_t* __restrict__ block_class, uint8_t* __restrict__ cellmark,
                                                          const int64_t* __restrict__ poison)
{
  const int64_t base = thread * 16;
  if (base >= npad) return;
  unsigned out[4] = {0u, 0u, 0u, 0u};
  int known = -1; // the block's class says what every cell of it is
  if (poison != nullptr && *poison != 0) known = 0; // (a void step: no marks, as when the list has length 0)
  else if (block_class && base < ncells)
  {
    const unsigned c = block_class[base / kClassBlock];
    if (c == 1u) known = -1 == value ? 1 : 0;
    else if (c == 2u) known = 1 == value ? 1 : 0;
  }
  if (known == 1 && base + 16 <= ncells)
  {
    const unsigned w = bits * 0x01010101u;
    out[0] = out[1] = out[2] = out[3] = w;
  }
  else if (known == 0) {}
  else if (base + 16 <= ncells)
  {
    const uint4 d = *reinterpret_cast<const uint4*>(domain + base); // (domain of level set 0 and base: 16 B aligned)
    const unsigned x[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((int8_t)((x[q] >> (8 * k)) & 0xffu) == value) out[q] |= (unsigned)bits << (8 * k);
  }
  else
    for (int k = 0; base + k < ncells; ++k)
      if (domain[base + k] == value) out[k >> 2] |= (unsigned)bits << (8 * (k & 3));
  *reinterpret_cast<uint4*>(cellmark + base) = make_uint4(out[0], out[1], out[2], out[3]);
}

// the rows next to the interface (class 2): active iff a cell around them carries a mark (the rows of rule cells and
// facets were marked by the kernels that walk those lists).  16 rows per thread: most threads read 16 class bytes and
// leave; the others walk the dof -> cells lists of their class-2 rows.
__global__ void __launch_bounds__(kBlock) mix_rowmark_kernel(int64_t ndofs, const uint8_t* __restrict__ rowcls,
                                                             const int64_t* __restrict__ d2c_off, const int32_t* __restrict__ d2c,
                                                             const uint8_t* __restrict__ cellmark, uint8_t* __restrict__ rowmark)
{
  const int64_t r0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 16;
  if (r0 >= ndofs) return;
  const uint4 cv = *reinterpret_cast<const uint4*>(rowcls + r0); // (padded to a multiple of 16 rows)
  const unsigned c[4] = {cv.x, cv.y, cv.z, cv.w};
  if (((c[0] | c[1] | c[2] | c[3]) & 0x02020202u) == 0u) return;
  for (int k = 0; k < 16 && r0 + k < ndofs; ++k)
  {
    if (((c[k >> 2] >> (8 * (k & 3))) & 0xffu) != kRowMix) continue;
    const int64_t r = r0 + k;
    if (rowmark[r]) continue; // (a dof of a rule cell or of a facet: marked by the kernel that walked that list)
    const int64_t cb = d2c_off[r], ce = d2c_off[r + 1];
    unsigned any = 0;
    for (int64_t t = cb; t < ce; t += 8)
    {
      int32_t cell[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) cell[j] = t + j < ce ? d2c[t + j] : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) any |= cell[j] >= 0 ? (unsigned)cellmark[cell[j]] : 0u;
    }
    if (any) rowmark[r] = 1;
  }
}

// degree 2: the two vertices of every dof, ascending (every cell that holds the dof writes the same pair)
__global__ void __launch_bounds__(kBlock) dof_verts_kernel(int64_t ncells, int tdim, const int32_t* __restrict__ conn,
                                                           const int32_t* __restrict__ dofmap, int32_t* __restrict__ dof_verts)
{
  const int nv = tdim + 1, nd = tdim == 2 ? 6 : 10;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t c = t / nd;
  if (c >= ncells) return;
  const int j = (int)(t - c * nd);
  const int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  const int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
  const int a = j < nv ? j : (tdim == 2 ? ea2[j - nv] : ea3[j - nv]), b = j < nv ? j : (tdim == 2 ? eb2[j - nv] : eb3[j - nv]);
  const int32_t va = conn[c * nv + a], vb = conn[c * nv + b];
  const int64_t dof = dofmap[c * nd + j];
  *reinterpret_cast<int2*>(dof_verts + 2 * dof) = make_int2(min(va, vb), max(va, vb));
}

// row classes of a degree-2 space from the vertex codes: a dof with an end vertex on the entities' side that no cut
// cell touches has only entities around it (the cells around an edge are cells around either of its vertices); likewise
// on the other side; else the marks decide.  Four dofs per thread; also the rows' initial marks (row_class_kernel).
__device__ __forceinline__ void row_class_p2_body(int64_t thread, int64_t ndofs, const int32_t* __restrict__ dof_verts,
                                                  const uint8_t* __restrict__ codes, const uint8_t* __restrict__ touch,
                                                  uint8_t sel, uint8_t* __restrict__ rowcls, uint8_t* __restrict__ rowmark,
                                                  uint8_t* __restrict__ special, const int64_t* __restrict__ poison)
{
  const int64_t r0 = thread * 4;
  if (r0 >= ndofs) return;
  const bool void_step = poison != nullptr && *poison != 0; // (mark nothing: row_class_kernel)
  unsigned cls = 0u, rm = 0u;
  int32_t v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = r0 + k / 2 < ndofs ? dof_verts[2 * r0 + k] : -1;
  unsigned cd[8], tc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { cd[k] = v[k] >= 0 ? codes[v[k]] : 0u; tc[k] = v[k] >= 0 ? touch[v[k]] : 1u; }
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    unsigned c = kRowMix;
    if (void_step) c = kRowOut;
    else if (r0 + q < ndofs)
    {
      const bool in = (!tc[2 * q] && cd[2 * q] == sel) || (!tc[2 * q + 1] && cd[2 * q + 1] == sel);
      const bool out = (!tc[2 * q] && cd[2 * q] == 3u - sel) || (!tc[2 * q + 1] && cd[2 * q + 1] == 3u - sel);
      c = in ? kRowIn : (out ? kRowOut : kRowMix);
    }
    cls |= c << (8 * q);
    rm |= (c == kRowIn ? 1u : 0u) << (8 * q);
  }
  *reinterpret_cast<unsigned*>(rowcls + r0) = cls; // (the three arrays are padded to a multiple of 16 rows)
  *reinterpret_cast<unsigned*>(rowmark + r0) = rm;
  *reinterpret_cast<unsigned*>(special + r0) = 0u;
}

// The marks a bulk row plan starts from, in ONE launch (four independent passes, each over its own range of workgroups:
// a launch costs ~6 us whatever it does, and at 32^3 the whole step is forty such floors): row classes + initial row
// marks (kind 1: P1, 16 rows per thread; kind 2: degree 2, four), cell marks from the classification bytes, the bit
// words + counts of the uncut entities, and the 0xff fill of the rule-key tables.
struct BulkInit
{
  unsigned blocks_cls, blocks_cm, blocks_pack; // (the fill takes the rest of the grid)
  int kind;
  int64_t ndofs;
  const int32_t* dof_verts;
  const uint8_t* codes;
  const uint8_t* touch;
  uint8_t sel;
  uint8_t* rowcls;
  uint8_t* rowmark;
  uint8_t* special;
  int32_t* t2off;
  int64_t ncells, npad;
  const int8_t* domain;
  int8_t value;
  uint8_t bits;
  const uint8_t* block_class;
  uint8_t* cellmark;
  unsigned long long* words;
  int32_t* pop;
  int32_t* keys; // rule-key tables: n_keys words of -1 (n_keys a multiple of 64: power-of-two tables of >= 64 keys)
  int64_t n_keys;
  const int64_t* poison;
};
__global__ void __launch_bounds__(kBlock) plan_bulk_init_kernel(BulkInit B)
{
  unsigned b = blockIdx.x;
  if (b < B.blocks_cls)
  {
    const int64_t t = (int64_t)b * kBlock + threadIdx.x;
    if (B.kind == 1) row_class_body(t, B.ndofs, B.codes, B.touch, B.sel, B.rowcls, B.rowmark, B.special, B.t2off, B.poison);
    else row_class_p2_body(t, B.ndofs, B.dof_verts, B.codes, B.touch, B.sel, B.rowcls, B.rowmark, B.special, B.poison);
    return;
  }
  b -= B.blocks_cls;
  if (b < B.blocks_cm)
  {
    cellmark_from_domain_body((int64_t)b * kBlock + threadIdx.x, B.ncells, B.npad, B.domain, B.value, B.bits, B.block_class,
                              B.cellmark, B.poison);
    return;
  }
  b -= B.blocks_cm;
  if (b < B.blocks_pack)
  {
    pack_bits_domain_body((int64_t)b * kBlock + threadIdx.x, B.ncells, B.domain, B.value, B.block_class, B.words, B.pop, B.poison);
    return;
  }
  b -= B.blocks_pack;
  const int64_t i = ((int64_t)b * kBlock + threadIdx.x) * 4; // (16 B per thread; the block is 16 B aligned)
  if (i < B.n_keys) *reinterpret_cast<int4*>(B.keys + i) = make_int4(-1, -1, -1, -1);
}

// active-row positions whose CSR row is at most / longer than `limit` columns
struct RowLenTest
{
  const int32_t* rows;
  const int64_t* indptr;
  int limit;
  bool longer;
  __device__ bool operator()(int64_t i) const
  {
    const int64_t r = rows[i];
    return ((indptr[r + 1] - indptr[r]) > limit) == longer;
  }
};
struct RowLenRange // lo < length <= hi
{
  const int32_t* rows;
  const int64_t* indptr;
  int lo, hi;
  __device__ bool operator()(int64_t i) const
  {
    const int64_t r = rows[i];
    const int64_t len = indptr[r + 1] - indptr[r];
    return len > lo && len <= hi;
  }
};

// positions in the active-row list -> row ids, in place
__global__ void map_rows_kernel(DevN n_d, const int32_t* __restrict__ rows, int32_t* __restrict__ idx)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = rows[idx[i]];
}

// hash map parent cell -> first rule index (one entry per run of equal parents)
__global__ void plan_rule_hash_kernel(DevN nr_d, const int32_t* __restrict__ parent, uint32_t mask,
                                      int32_t* __restrict__ keys, int32_t* __restrict__ first)
{
  const int64_t nr = dev_n(nr_d);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nr) return;
  const int32_t c = parent[e];
  if (e > 0 && parent[e - 1] == c) return;
  uint32_t h = cfx_hash32((uint32_t)c) & mask;
  while (true)
  {
    const int32_t old = atomicCAS(&keys[h], -1, c);
    if (old == -1 || old == c) break; // (a repeated run can only come from an unsorted list, flagged elsewhere)
    h = (h + 1) & mask;
  }
  first[h] = (int32_t)e;
}

// flags a list that is not ascending (strict: not strictly ascending)
__global__ void plan_check_sorted_kernel(DevN n_d, const int32_t* __restrict__ a, int strict, int* flag)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 < n && (a[i] > a[i + 1] || (strict && a[i] == a[i + 1]))) atomicOr(flag, 1);
}

// P1 facet folding (assemble_rows_kernel) needs every facet row to join two cells that share all dofs but one
// each -- a continuous P1 space on a conforming mesh.  Flags bit 1 otherwise (DG spaces, extension pairs).
template <int ND>
__global__ void __launch_bounds__(kBlock) plan_check_fold_kernel(DevN nf_d, const int32_t* __restrict__ rows,
                                                                 const int32_t* __restrict__ dofmap, int nx, int* flag)
{
  const int64_t nf = dev_n(nf_d);
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int64_t c0 = rows[4 * f], c1 = rows[4 * f + 2];
  int32_t d0[ND], d1[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) { d0[j] = dofmap[c0 * ND + j]; d1[j] = dofmap[c1 * ND + j]; }
  int nfree = 0;
#pragma unroll
  for (int j = 0; j < ND; ++j)
  {
    bool shared = false;
#pragma unroll
    for (int i = 0; i < ND; ++i) shared = shared || d1[j] == d0[i];
    nfree += shared ? 0 : 1;
  }
  if (nfree != nx) atomicOr(flag, 2); // nx: dofs of a cell that are not on a given facet (continuous space)
}

// (counts / offsets / cursors are indexed by the dof's position in the special-row list: the
// incidence only exists next to the interface, a scan over all dofs would cost more than the rest)
__global__ void plan_scatter_pos_kernel(DevN n_d, const int32_t* __restrict__ rows, int32_t* __restrict__ pos)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[rows[i]] = (int32_t)i;
}

// count + fill with the workgroup's atomics combined in LDS (cfx_device.h: adj_lds_insert): consecutive facets of the
// list share most of their dofs (the list follows the cut cells), so a run of ~1000 (facet, dof) pairs reaches memory as
// ~150 atomics.  (counts / offsets / cursors are indexed by the dof's position in the special-row list.)  A thread owns
// one (facet, side) unit: the dof row of its cell, and for side 1 the dof row of cell 0 to drop the dofs both cells hold
// (a shared dof lists the facet once) -- 2 nd loads per unit where one thread per pair paid 1 + nd loads per pair.
// Round 3 sorted the pairs instead from 6 M pairs on (13 launches, 1.1 ms at 512^3 against 0.5 ms now); CFX_FACET_SORT=1
// keeps that path.
constexpr int kFacetMaxNd = 10;
__host__ __device__ inline int facet_units_per_block(int nd) { return kAdjRun / nd < kBlock ? kAdjRun / nd : kBlock; }

__device__ __forceinline__ void facet_unit_keys(const int32_t* __restrict__ rows, const int32_t* __restrict__ dofmap, int nd,
                                                const int32_t* __restrict__ pos, int64_t unit, int32_t* key)
{
  const int64_t f = unit >> 1;
  const int side = (int)(unit & 1);
  const int64_t c = rows[4 * f + 2 * side];
  int32_t d[kFacetMaxNd], d0[kFacetMaxNd];
#pragma unroll
  for (int j = 0; j < kFacetMaxNd; ++j) d[j] = j < nd ? dofmap[c * nd + j] : -1;
  if (side)
  {
    const int64_t c0 = rows[4 * f];
#pragma unroll
    for (int j = 0; j < kFacetMaxNd; ++j) d0[j] = j < nd ? dofmap[c0 * nd + j] : -2;
  }
#pragma unroll
  for (int j = 0; j < kFacetMaxNd; ++j)
  {
    bool skip = j >= nd;
    if (side)
    {
#pragma unroll
      for (int i = 0; i < kFacetMaxNd; ++i) skip = skip || d0[i] == d[j];
    }
    key[j] = skip ? -1 : pos[d[j]];
  }
}

// (nkeys: entries of `counts` = capacity of the special-row list; poison: the step's poison word or null.  The facet
// count of a form with several facet lists is exact on the host, so it does not drop to 0 when a LATER site -- the row
// lists, whose capacity sizes `counts` -- makes the step void: positions beyond the capacity are skipped and a void
// step is left alone (the DG loop of tools/soak_fuzz.py under forced overflow)
__global__ void __launch_bounds__(kBlock) facet_dof_count_kernel(DevN nf_d, const int32_t* __restrict__ rows,
                                                                 const int32_t* __restrict__ dofmap, int nd,
                                                                 const int32_t* __restrict__ pos, int32_t* counts,
                                                                 int64_t nkeys, const int64_t* __restrict__ poison)
{
  if (poison != nullptr && *poison != 0) return;
  __shared__ int32_t s_key[kAdjSlots], s_cnt[kAdjSlots];
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock) { s_key[k] = -1; s_cnt[k] = 0; }
  __syncthreads();
  const int per = facet_units_per_block(nd);
  const int64_t unit = (int64_t)blockIdx.x * per + threadIdx.x;
  if (threadIdx.x < per && unit < 2 * dev_n(nf_d))
  {
    int32_t key[kFacetMaxNd];
    facet_unit_keys(rows, dofmap, nd, pos, unit, key);
#pragma unroll
    for (int j = 0; j < kFacetMaxNd; ++j)
    {
      int rank;
      if (key[j] >= 0 && key[j] < nkeys) (void)adj_lds_insert(s_key, s_cnt, key[j], rank);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock)
    if (s_key[k] >= 0) atomicAdd(&counts[s_key[k]], s_cnt[k]);
}

__global__ void __launch_bounds__(kBlock) facet_dof_fill_kernel(DevN nf_d, const int32_t* __restrict__ rows,
                                                                const int32_t* __restrict__ dofmap, int nd,
                                                                const int32_t* __restrict__ pos,
                                                                const int64_t* __restrict__ offs, int32_t* cursor,
                                                                int32_t* facets, int64_t nkeys, const int64_t* __restrict__ poison)
{
  if (poison != nullptr && *poison != 0) return;
  __shared__ int32_t s_key[kAdjSlots], s_cnt[kAdjSlots];
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock) { s_key[k] = -1; s_cnt[k] = 0; }
  __syncthreads();
  const int per = facet_units_per_block(nd);
  const int64_t unit = (int64_t)blockIdx.x * per + threadIdx.x;
  const bool live = threadIdx.x < per && unit < 2 * dev_n(nf_d);
  int32_t key[kFacetMaxNd];
  int slot[kFacetMaxNd], rank[kFacetMaxNd];
#pragma unroll
  for (int j = 0; j < kFacetMaxNd; ++j) { key[j] = -1; slot[j] = 0; rank[j] = 0; }
  if (live)
  {
    facet_unit_keys(rows, dofmap, nd, pos, unit, key);
#pragma unroll
    for (int j = 0; j < kFacetMaxNd; ++j)
    {
      if (key[j] >= nkeys) key[j] = -1;
      if (key[j] >= 0) slot[j] = adj_lds_insert(s_key, s_cnt, key[j], rank[j]);
    }
  }
  __syncthreads();
  // (cursor[q] holds the row's count from the count pass: slots are handed out from the back, so the counters need no
  // second zero fill; the workgroup's entries of a row take one contiguous run)
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock)
    if (s_key[k] >= 0) s_cnt[k] = atomicSub(&cursor[s_key[k]], s_cnt[k]) - s_cnt[k];
  __syncthreads();
  const int32_t f = (int32_t)(unit >> 1);
#pragma unroll
  for (int j = 0; j < kFacetMaxNd; ++j)
    if (key[j] >= 0) facets[offs[key[j]] + s_cnt[slot[j]] + rank[j]] = f;
}

// dof -> facets incidence by sorting: one (special-row position, facet) pair per dof of a facet's two cells (a dof
// of both cells once: the cell-1 copy gets the sentinel key `nkeys`, which sorts behind everything).  A stable radix
// sort over the few key bits replaces ~35 M returning integer atomics on ~2 M counters (count + fill passes) and
// leaves every list in ascending facet order, whatever the schedule.
__global__ void __launch_bounds__(kBlock) facet_dof_pairs_kernel(DevN nf_d, const int32_t* __restrict__ rows,
                                                                 const int32_t* __restrict__ dofmap, int nd,
                                                                 const int32_t* __restrict__ pos, int32_t nkeys,
                                                                 int32_t* __restrict__ keys, int32_t* __restrict__ vals)
{
  const int64_t nf = dev_n(nf_d);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nf * 2 * nd)
  {
    // (a facet list shorter than its capacity: the sort runs over the capacity, the tail sorts behind everything)
    if (i < nf_d.cap * 2 * nd) { keys[i] = nkeys; vals[i] = 0; }
    return;
  }
  const int64_t f = i / (2 * nd);
  const int k = (int)(i - f * 2 * nd);
  const int64_t c = rows[4 * f + (k < nd ? 0 : 2)];
  const int32_t dof = dofmap[c * nd + (k < nd ? k : k - nd)];
  bool skip = false;
  if (k >= nd)
  {
    const int64_t c0 = rows[4 * f];
    for (int j = 0; j < nd; ++j) skip = skip || dofmap[c0 * nd + j] == dof;
  }
  keys[i] = skip ? nkeys : pos[dof];
  vals[i] = (int32_t)f;
}

// offsets[k] = first sorted position whose key is >= k (k = 0 .. nkeys)
__global__ void __launch_bounds__(kBlock) sorted_key_offsets_kernel(DevN nkeys_d, int64_t n, const int32_t* __restrict__ keys,
                                                                    int64_t* __restrict__ offsets)
{
  const int64_t nkeys = dev_n(nkeys_d);
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k > nkeys) return;
  int64_t lo = 0, hi = n;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (int32_t)k) lo = mid + 1; else hi = mid;
  }
  offsets[k] = lo;
}

// sort each listed dof's facet list so the gather order is reproducible
__global__ void seg_sort_kernel(DevN nseg_d, const int64_t* __restrict__ offsets, int32_t* vals)
{
  const int64_t nseg = dev_n(nseg_d);
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nseg) return;
  const int64_t b = offsets[r], e = offsets[r + 1];
  for (int64_t i = b + 1; i < e; ++i)
  {
    const int32_t v = vals[i];
    int64_t j = i - 1;
    while (j >= b && vals[j] > v) { vals[j + 1] = vals[j]; --j; }
    vals[j + 1] = v;
  }
}

struct FlagSetU8
{
  const uint8_t* f;
  __device__ bool operator()(int64_t i) const { return f[i] != 0; }
};

// ---------------------------------------------------------------------------
// a9 sparsity.  Group of G lanes per active row: every candidate column (dofs
// of the marked incident cells, dofs of both cells of the incident facets, the
// diagonal) goes into a per-row hash set in LDS; the set is then ranked so the
// row comes out sorted.  Rows outside the active set hold only their diagonal
// (assembler.h:538-560).
// ---------------------------------------------------------------------------
struct PatArgs
{
  DevN n_active;    // rows of this launch (length in HBM inside a sync-free step)
  const int32_t* active_rows;
  int nd, bs;
  const int32_t* dofmap;
  const int64_t* d2c_off;
  const int32_t* d2c;
  const uint8_t* cellmark;
  const int64_t* d2f_off; // or null
  const int32_t* d2f;
  const int32_t* facet_rows;
  int32_t* tmp;     // [n_active * T] sorted unique columns of each active row (narrow path), or null
  const int64_t* indptr; // wide path, second pass: write rows directly
  int32_t* indices;
  int all_cells;    // stencil build: every incident cell counts (cellmark is null), active_rows null = all rows
  const uint8_t* special_mark; // dof->facets incidence: d2f_off is indexed by special_pos[r] where special_mark[r]
  const int32_t* special_pos;
  int32_t* len;     // [n_active]
  int32_t* counts;  // [ndofs*bs] expanded row lengths
  int* overflow;
  int* maxlen;      // longest scalar-dof row
  // rectangular forms (test space != trial space): the row dof is a dof of another space -- it is not a column of its
  // own row (no_self), and a column dof expands to bs_col entries while a row dof owns bs rows (bs_col = 0: bs)
  int no_self, bs_col;
  // position of active_rows[i] in the list whose first n_first entries are the plan's special rows in plan order (the
  // dof -> facets incidence is indexed by that position): saves the special_mark -> special_pos hops; or null
  const int32_t* row_pos;
  int64_t n_first;
};

template <int T>
__device__ __forceinline__ bool hash_insert(int32_t* tab, int32_t v)
{
#ifdef CFX_PATTERN_MURMUR
  unsigned slot = cfx_hash32((uint32_t)v);
#else
  unsigned slot = ((unsigned)v * 2654435761u) >> 7;
#endif
  for (int probe = 0; probe < T; ++probe)
  {
    slot &= (T - 1);
    // most candidates are duplicates: a plain read settles them without an LDS atomic
    int32_t old = tab[slot];
    if (old == v) return true;
    if (old == -1)
    {
      old = atomicCAS(&tab[slot], -1, v);
      if (old == -1 || old == v) return true;
    }
    ++slot;
  }
  return false;
}

template <int G, int T>
__global__ void __launch_bounds__(kWave) pattern_rows_kernel(PatArgs P)
{
  constexpr int RPW = kWave / G;
  constexpr int kSrc = G >= 16 ? 128 : 8; // source cells staged per row and chunk (degree-2 / vector / 2-D rows)
  __shared__ int32_t s_tab[RPW][T];
  __shared__ __align__(16) int32_t s_list[RPW][T];
  __shared__ int32_t s_src[RPW][kSrc];
  // incident cells kept for the facet cells' look-up (rows with more skip the look-up beyond); the 4-lane form serves
  // the P1 rows, whose own path never reads the list: 2.5 KB of LDS less per wavefront (512^3: 0.90 -> 0.76 ms)
  constexpr int kInc = G >= 8 ? 32 : 4;
  __shared__ int32_t s_inc[RPW][kInc];
  __shared__ uint8_t s_incm[RPW][kInc];
  __shared__ int s_cnt[RPW];
  __shared__ int s_nsrc[RPW]; // sources of the current chunk that add candidates (the others are not iterated at all)
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  // one pass for every launch that fits HIP's 2^32-thread limit; the grid is capped beyond it
  const int64_t n_active = dev_n(P.n_active);
  for (int64_t blk = blockIdx.x; blk * RPW < n_active; blk += gridDim.x)
  {
  const int64_t ri = blk * RPW + grp; // (XCD-contiguous chunks measured slower: 5.95 vs 6.0 ms here, 12.6 vs 14.5 ms in the gather)
  const bool live = ri < n_active;
  const int64_t r = live ? (P.active_rows ? (int64_t)P.active_rows[ri] : ri) : 0;
  for (int k = gl; k < T; k += G) s_tab[grp][k] = -1;
  if (gl == 0) s_cnt[grp] = 0;
  __syncthreads();
  bool ok = true;
  if (live)
  {
    int32_t* tab = s_tab[grp];
    if (gl == 0 && !P.no_self) ok = hash_insert<T>(tab, (int32_t)r);
    if (P.cellmark || P.all_cells)
    {
      // R incident cells per lane per pass; incidence, marks and dof rows of a pass
      // are requested together so that the three dependent levels overlap
      constexpr int R = G <= 4 ? 6 : (G <= 8 ? 4 : 2);
      const int64_t cb = P.d2c_off[r];
      const int nc = (int)(P.d2c_off[r + 1] - cb);
      if (P.nd != 4) {} // handled below together with the facets (source cells staged in LDS)
      else
      for (int base = 0; base < nc; base += R * G)
      {
        int64_t cell[R];
        uint8_t mk[R];
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
          const int t = base + k * G + gl;
          cell[k] = t < nc ? (int64_t)P.d2c[cb + t] : -1;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) mk[k] = cell[k] >= 0 ? (P.all_cells ? (uint8_t)1 : P.cellmark[cell[k]]) : (uint8_t)0;
        if (P.nd == 4)
        {
          int4 dr[R];
#pragma unroll
          for (int k = 0; k < R; ++k)
            dr[k] = mk[k] ? *reinterpret_cast<const int4*>(P.dofmap + cell[k] * 4) : make_int4(-1, -1, -1, -1);
#pragma unroll
          for (int k = 0; k < R; ++k)
            if (mk[k])
            {
              ok = hash_insert<T>(tab, dr[k].x) && ok;
              ok = hash_insert<T>(tab, dr[k].y) && ok;
              ok = hash_insert<T>(tab, dr[k].z) && ok;
              ok = hash_insert<T>(tab, dr[k].w) && ok;
            }
        }
        else
        {
#pragma unroll
          for (int k = 0; k < R; ++k)
            if (mk[k])
              for (int j = 0; j < P.nd; ++j) ok = hash_insert<T>(tab, P.dofmap[cell[k] * P.nd + j]) && ok;
        }
      }
    }
    const int64_t fpos = (P.d2f_off && P.special_mark[r]) ? (int64_t)P.special_pos[r] : -1;
    if (fpos >= 0 && P.nd == 4)
    {
      // P1: one facet per lane (11 loads for 8 candidates; the triple form below costs 24), kF facets of a lane per trip
      // with each of the three dependent levels -- incidence entry, facet row, dof rows -- requested for all of them at
      // once: one facet per trip waited three memory latencies per trip, ten trips for a row with forty facets
      constexpr int kF = 4;
      const int64_t fe = P.d2f_off[fpos + 1];
      for (int64_t k0 = P.d2f_off[fpos] + gl; k0 < fe; k0 += (int64_t)G * kF)
      {
        int64_t f[kF];
#pragma unroll
        for (int u = 0; u < kF; ++u) f[u] = k0 + (int64_t)u * G < fe ? (int64_t)P.d2f[k0 + (int64_t)u * G] : -1;
        int64_t c[kF][2];
#pragma unroll
        for (int u = 0; u < kF; ++u)
        {
          const int4 row = f[u] >= 0 ? *reinterpret_cast<const int4*>(P.facet_rows + 4 * f[u]) : make_int4(-1, 0, -1, 0);
          c[u][0] = row.x; c[u][1] = row.z;
        }
        int4 v[kF][2];
#pragma unroll
        for (int u = 0; u < kF; ++u)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            v[u][s2] = c[u][s2] >= 0 ? *reinterpret_cast<const int4*>(P.dofmap + c[u][s2] * 4) : make_int4(-1, -1, -1, -1);
#pragma unroll
        for (int u = 0; u < kF; ++u)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            if (c[u][s2] >= 0)
            {
              ok = hash_insert<T>(tab, v[u][s2].x) && ok;
              ok = hash_insert<T>(tab, v[u][s2].y) && ok;
              ok = hash_insert<T>(tab, v[u][s2].z) && ok;
              ok = hash_insert<T>(tab, v[u][s2].w) && ok;
            }
      }
    }
  }
  if (P.nd != 4)
  {
    // Source cells of the row -- its marked incident cells, then both cells of its facets -- go to LDS first (two
    // dependent gather levels, once, spread over the lanes); the (source, local dof) pairs are then one dofmap load
    // each.  A lane that walks incidence -> mark / facet row -> dofmap per candidate pays three dependent levels
    // per candidate: 33 ms instead of 17 ms for the 12 M short rows of BASELINE config 4.
    int32_t* tab = s_tab[grp];
    const bool cells_on = live && (P.cellmark || P.all_cells);
    const int64_t cb = cells_on ? P.d2c_off[r] : 0;
    const int nc = cells_on ? (int)(P.d2c_off[r + 1] - cb) : 0;
    int64_t fpos = -1;
    if (live && P.d2f_off)
    {
      if (P.row_pos) { const int32_t rp = P.row_pos[ri]; fpos = rp < P.n_first ? (int64_t)rp : -1; }
      else if (P.special_mark[r]) fpos = (int64_t)P.special_pos[r];
    }
    const int64_t fb = fpos >= 0 ? P.d2f_off[fpos] : 0;
    const int nsrc = nc + (fpos >= 0 ? 2 * (int)(P.d2f_off[fpos + 1] - fb) : 0);
    // the row's incident cells (ascending), kept for the facet cells to look themselves up in: a facet cell that is a
    // marked incident cell of the row is a source already (on the facets that contain the row's dof both cells are,
    // ~36 of the ~60 ghost facets around a vertex dof of a Kuhn mesh, the cell on the row's side of the other 24 as
    // well: 1440 candidate columns shrink to ~480)
    const int ninc = min(nc, kInc);
    for (int t = gl; t < ninc; t += G)
    {
      const int32_t c = P.d2c[cb + t];
      s_inc[grp][t] = c;
      s_incm[grp][t] = (P.all_cells || P.cellmark[c]) ? 1 : 0;
    }
    __syncthreads();
    int nchunk = (nsrc + kSrc - 1) / kSrc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nchunk = max(nchunk, __shfl_xor(nchunk, o, 64)); // the groups loop together
    for (int ch = 0; ch < nchunk; ++ch)
    {
      const int base = ch * kSrc;
      const int m = min(kSrc, nsrc - base);
      if (gl == 0) s_nsrc[grp] = 0;
      __syncthreads();
      for (int t = gl; t < m; t += G)
      {
        const int idx = base + t;
        int32_t cell;
        if (idx < nc)
        {
          if (idx < ninc) cell = s_incm[grp][idx] ? s_inc[grp][idx] : -1;
          else
          {
            const int32_t c = P.d2c[cb + idx];
            cell = (P.all_cells || P.cellmark[c]) ? c : -1;
          }
        }
        else
        {
          const int64_t f = P.d2f[fb + ((idx - nc) >> 1)];
          cell = P.facet_rows[4 * f + 2 * ((idx - nc) & 1)];
          int lo = 0, hi = ninc;
          while (lo < hi)
          {
            const int mid = (lo + hi) >> 1;
            if (s_inc[grp][mid] < cell) lo = mid + 1; else hi = mid;
          }
          if (lo < ninc && s_inc[grp][lo] == cell && s_incm[grp][lo]) cell = -1;
        }
        // only the sources that add candidates are kept (their order does not matter: the set is ranked afterwards): the
        // candidate loop below ran over every (source, local dof) pair, skipped ones included -- 1440 trips for a vertex
        // dof of a degree-2 space of which ~480 loaded anything
        if (cell >= 0) s_src[grp][atomicAdd(&s_nsrc[grp], 1)] = cell;
      }
      __syncthreads();
      const int mv = s_nsrc[grp];
      // the (source, local dof) pairs, lane after lane: the pair index advances by G, (q, j) follow without a division
      // (t / nd with a run-time nd was ~20 instructions per candidate).  kU dof-row entries are requested together and
      // inserted afterwards: with one load per loop trip every trip waited for its own gather (~30 trips x 1.5 us for
      // a vertex dof of a degree-2 space)
      constexpr int kU = 8;
      const int npairs = mv * P.nd;
      int q = gl / P.nd, j = gl - q * P.nd;
      const int dq = G / P.nd, dj = G - dq * P.nd;
      for (int t0 = gl; t0 < npairs; t0 += G * kU)
      {
        int32_t dv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
          const bool in = t0 + u * G < npairs;
          const int64_t cell = in ? (int64_t)s_src[grp][q] : -1;
          dv[u] = cell >= 0 ? P.dofmap[cell * P.nd + j] : -1;
          q += dq; j += dj;
          if (j >= P.nd) { j -= P.nd; ++q; }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (dv[u] >= 0) ok = hash_insert<T>(tab, dv[u]) && ok;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (!ok) *P.overflow = 1;
  // compact the set, then rank inside the compact list (cnt^2/G reads instead of T^2/G)
  for (int k = gl; k < T; k += G)
  {
    const int32_t v = s_tab[grp][k];
    if (v >= 0) s_list[grp][atomicAdd(&s_cnt[grp], 1)] = v;
  }
  __syncthreads();
  const int cnt = s_cnt[grp];
  // rank = number of smaller entries, four entries per LDS read and trip (the list is padded with INT_MAX to a multiple
  // of four): the scalar loop -- one read, one compare, one add, counter and branch per entry -- was a third of the
  // kernel's ~4000 instructions per wavefront, and the kernel is bound by VALU + SALU issue (profiles/r03: 1917 vector
  // and 1771 scalar instructions per wavefront against 34 vector-memory ones)
  if (cnt < T)
    for (int k = cnt + gl; k < ((cnt + 3) & ~3); k += G) s_list[grp][k] = 0x7fffffff;
  __syncthreads();
  if (P.tmp || P.indices)
    for (int k = gl; k < cnt; k += G)
    {
      const int32_t v = s_list[grp][k];
      int rank = 0;
      const int4* l4 = reinterpret_cast<const int4*>(s_list[grp]);
      for (int m = 0; m < (cnt + 3) / 4; ++m)
      {
        const int4 q4 = l4[m];
        rank += (q4.x < v ? 1 : 0) + (q4.y < v ? 1 : 0) + (q4.z < v ? 1 : 0) + (q4.w < v ? 1 : 0);
      }
      if (P.indices)
      {
        // second pass of the wide path: the row goes straight into the CSR arrays
        const int bc = P.bs_col ? P.bs_col : P.bs;
        for (int a = 0; a < P.bs; ++a)
          for (int b = 0; b < bc; ++b)
            P.indices[P.indptr[r * P.bs + a] + (int64_t)rank * bc + b] = v * bc + b;
      }
      else
        P.tmp[ri * T + rank] = v;
    }
  // a full table cannot be told from an overflowing one: keep one slot free
  if (cnt >= T) *P.overflow = 1;
  if (live && gl == 0 && !P.indices)
  {
    // a racy pre-check keeps 10^7 rows from serialising on one address
    if (cnt > *reinterpret_cast<volatile int*>(P.maxlen)) atomicMax(P.maxlen, cnt);
    P.len[ri] = cnt;
    for (int a = 0; a < P.bs; ++a) P.counts[r * P.bs + a] = cnt * (P.bs_col ? P.bs_col : P.bs);
  }
  __syncthreads(); // the LDS tables are reused by the next pass
  }
}

__global__ void fill_i32_kernel(int64_t n, int32_t v, int32_t* out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

// diagonal block of the rows that are not active
__global__ void pattern_diag_kernel(int64_t ndofs, int bs, const uint8_t* __restrict__ rowmark,
                                    const int64_t* __restrict__ indptr, int32_t* __restrict__ indices)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ndofs || rowmark[r]) return;
  for (int a = 0; a < bs; ++a)
    for (int b = 0; b < bs; ++b) indices[indptr[r * bs + a] + b] = (int32_t)(r * bs + b);
}

// indptr of the whole matrix in two passes over the rows, the inactive rows' diagonal blocks included: the length of
// an active row comes from the pattern kernels (counts), an inactive row holds its bs x bs diagonal block
// (assembler.h:538-560) -- no counts fill, no separate diagonal pass, one read of the row marks per pass.
__global__ void __launch_bounds__(kBlock) indptr_reduce_kernel(int64_t nrows, int bs, const uint8_t* __restrict__ rowmark,
                                                               const int32_t* __restrict__ counts, int64_t* __restrict__ tile_sums)
{
  const int64_t tile = (int64_t)blockIdx.x * kTile;
  int64_t s = 0;
  if (bs == 1 && tile + kTile <= nrows)
  {
    // scalar space, whole tile: the marks of a thread's eight consecutive rows in one 8 B load (one byte per lane and
    // load cost 0.18 ms for 135 M rows); the lengths are read for marked rows only
    static_assert(kScanItems == 8, "eight rows per thread");
    const int64_t r0 = tile + 8 * threadIdx.x;
    const uint2 m = *reinterpret_cast<const uint2*>(rowmark + r0); // (tile and 8 t: 8 B aligned)
    if ((m.x | m.y) == 0u) s = 8;
    else
    {
#pragma unroll
      for (int q = 0; q < 8; ++q)
      {
        const unsigned byte = ((q < 4 ? m.x : m.y) >> (8 * (q & 3))) & 0xffu;
        s += byte ? counts[r0 + q] : 1;
      }
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
    {
      const int64_t R = tile + k * kBlock + threadIdx.x;
      if (R < nrows) s += rowmark[R / bs] ? counts[R] : bs;
    }
  }
  int64_t total;
  (void)block_exclusive_scan<int64_t>(s, total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one tile of kTile rows: indptr, and the diagonal entries of its inactive rows.  first(total): the tile's first entry,
// asked once the tile's own entries are summed (all threads call it: a table look-up, or the look-back of a chained
// launch).  Returns first + total (all threads).  Nothing is written to `indices` at or beyond nnz_cap.
template <typename First>
__device__ __forceinline__ int64_t indptr_tile(int64_t nrows, int bs, const uint8_t* __restrict__ rowmark,
                                               const int32_t* __restrict__ counts, const int64_t tile, First first,
                                               int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                               const int64_t nnz_cap)
{
  __shared__ int64_t s_v[kTile];
  // A whole tile of inactive rows of a scalar space (87 % of the rows at 512^3): indptr is an arithmetic progression and
  // the diagonal entries are the row numbers -- eight rows per thread as 16 B stores, no marks gathered row by row, no
  // LDS scan.
  if (bs == 1 && tile + kTile <= nrows)
  {
    static_assert(kScanItems == 8, "eight rows per thread");
    const uint2 m = *reinterpret_cast<const uint2*>(rowmark + tile + 8 * threadIdx.x); // (tile and 8 t: 8 B aligned)
    if (__syncthreads_or((m.x | m.y) != 0u) == 0)
    {
      const int64_t t0 = first((int64_t)kTile);
      const int64_t p0 = t0 + 8 * threadIdx.x;
      const int64_t r0 = tile + 8 * threadIdx.x;
      if ((reinterpret_cast<uintptr_t>(indptr + r0) & 15) == 0)
      {
        longlong2* ip = reinterpret_cast<longlong2*>(indptr + r0);
#pragma unroll
        for (int q = 0; q < 4; ++q) ip[q] = make_longlong2(p0 + 2 * q, p0 + 2 * q + 1);
      }
      else
      {
#pragma unroll
        for (int q = 0; q < 8; ++q) indptr[r0 + q] = p0 + q;
      }
      // (the diagonal entries one by one: as 16 B words from the first aligned position on -- t0 is whatever the active
      // rows before the tile add up to -- the kernel ran 0.63 instead of 0.51 ms at 512^3)
      // (p0 < 0: a void step -- the lengths of marked rows before this tile were never written and summed to garbage)
      if (p0 >= 0 && p0 + 8 <= nnz_cap)
      {
#pragma unroll
        for (int q = 0; q < 8; ++q) indices[p0 + q] = (int32_t)(r0 + q);
      }
      else
      {
        for (int q = 0; q < 8; ++q)
          if (p0 + q >= 0 && p0 + q + 1 <= nnz_cap) indices[p0 + q] = (int32_t)(r0 + q);
      }
      return t0 + kTile;
    }
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int i = k * kBlock + threadIdx.x;
    const int64_t R = tile + i;
    s_v[i] = R < nrows ? (rowmark[R / bs] ? (int64_t)counts[R] : (int64_t)bs) : 0;
  }
  __syncthreads();
  int64_t v[kScanItems], s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { v[k] = s_v[threadIdx.x * kScanItems + k]; s += v[k]; }
  int64_t total;
  const int64_t local = block_exclusive_scan<int64_t>(s, total);
  const int64_t t0 = first(total);
  int64_t off = local + t0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { s_v[threadIdx.x * kScanItems + k] = off; off += v[k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int i = k * kBlock + threadIdx.x;
    const int64_t R = tile + i;
    if (R >= nrows) continue;
    const int64_t p = s_v[i];
    indptr[R] = p;
    const int64_t dof = R / bs;
    if (!rowmark[dof] && p >= 0 && p + bs <= nnz_cap)
      for (int b = 0; b < bs; ++b) indices[p + b] = (int32_t)(dof * bs + b);
  }
  return t0 + total;
}

__global__ void __launch_bounds__(kBlock) indptr_write_kernel(int64_t nrows, int bs, const uint8_t* __restrict__ rowmark,
                                                              const int32_t* __restrict__ counts,
                                                              const int64_t* __restrict__ tile_offsets,
                                                              int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                                              DevN nnz_d)
{
  // (indices sized by the previous step's nnz: nothing is written beyond the published total -- 0 in a void step)
  const int64_t nnz_cap = nnz_d.dev ? dev_n(nnz_d) : INT64_MAX;
  const int64_t end = indptr_tile(nrows, bs, rowmark, counts, (int64_t)blockIdx.x * kTile,
                                  [&](int64_t) { return tile_offsets[blockIdx.x]; }, indptr, indices, nnz_cap);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) indptr[nrows] = end;
}

// reduce, offsets and write in ONE launch (tiles chained by look-back, cfx_device.h) -- inside a sync-free step, where
// `indices` is sized by the previous step's nnz before the rows are summed.  The last tile leaves nnz in `total_out` and
// publishes the step's counts; nothing is written beyond the capacity of `indices`, and a total beyond it voids the
// step there.
__global__ void __launch_bounds__(kBlock) indptr_chained_kernel(int64_t nrows, int bs, const uint8_t* __restrict__ rowmark,
                                                                const int32_t* __restrict__ counts,
                                                                int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                                                int64_t nnz_cap, ChainState chain,
                                                                int64_t* __restrict__ total_out, CountJobs after)
{
  const unsigned int tile = chain_take_tile(chain.ticket);
  const int64_t end = indptr_tile(nrows, bs, rowmark, counts, (int64_t)tile * kTile,
                                  [&](int64_t total) { return (int64_t)chain_exclusive_prefix(chain.state, tile, (unsigned long long)total); },
                                  indptr, indices, nnz_cap);
  if (tile == gridDim.x - 1 && threadIdx.x == kBlock - 1)
  {
    indptr[nrows] = end;
    *total_out = end;
    if (after.n > 0) count_publish(after);
  }
}

template <int T>
__global__ void pattern_write_kernel(DevN n_active_d, const int32_t* __restrict__ active_rows, int bs,
                                     const int32_t* __restrict__ tmp, const int32_t* __restrict__ len,
                                     const int64_t* __restrict__ indptr, int32_t* __restrict__ indices)
{
  const int64_t n_active = dev_n(n_active_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ri = i / 8;
  if (ri >= n_active) return;
  const int n = len[ri];
  const int64_t r = active_rows ? (int64_t)active_rows[ri] : ri; // (no list: all rows, the mesh-static stencil)
  if (bs == 1)
  {
    // scalar spaces: 16 B per lane from the (aligned) staging row, 32 entries per trip of the row's eight lanes
    const int4* src = reinterpret_cast<const int4*>(tmp + ri * T);
    const int64_t ob = indptr[r];
    for (int k4 = (int)(i - ri * 8); 4 * k4 < n; k4 += 8)
    {
      const int4 v = src[k4];
      const int k = 4 * k4;
      indices[ob + k] = v.x;
      if (k + 1 < n) indices[ob + k + 1] = v.y;
      if (k + 2 < n) indices[ob + k + 2] = v.z;
      if (k + 3 < n) indices[ob + k + 3] = v.w;
    }
    return;
  }
  for (int k = (int)(i - ri * 8); k < n; k += 8)
  {
    const int32_t col = tmp[ri * T + k];
    for (int a = 0; a < bs; ++a)
      for (int b = 0; b < bs; ++b) indices[indptr[r * bs + a] + (int64_t)k * bs + b] = col * bs + b;
  }
}

// ---------------------------------------------------------------------------
// Mesh-static stencil (cfx::Stencil) and the rows that are subsets of it
// ---------------------------------------------------------------------------
// positions of the dofs of every incident cell inside the dof's neighbour list.  4 lanes per dof, 16 dofs per
// wavefront: the (sorted) neighbour list is staged in LDS once, every lane then takes incident cells t = gl, gl + 4,
// ... and finds the positions of their dofs by binary search in LDS (one thread per dof with the searches in
// global memory took 257 ms at 512^3)
__global__ void __launch_bounds__(kWave) stencil_slots_kernel(int64_t ndofs, const int64_t* __restrict__ d2c_off,
                                                              const int32_t* __restrict__ d2c, const int32_t* __restrict__ dofmap, int nd,
                                                              const int64_t* __restrict__ off, const int32_t* __restrict__ nbr,
                                                              uint32_t* __restrict__ slot4, uint8_t* __restrict__ diagpos,
                                                              uint8_t* __restrict__ cpos)
{
  constexpr int G = 4, RPW = kWave / G;
  __shared__ int32_t s_nbr[RPW][64];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  for (int64_t blk = blockIdx.x; blk * RPW < ndofs; blk += gridDim.x)
  {
    const int64_t r = blk * RPW + grp;
    const bool live = r < ndofs;
    const int64_t b = live ? off[r] : 0;
    const int len = live ? (int)(off[r + 1] - b) : 0; // <= 64 (space_stencil gives up otherwise)
    for (int k = gl; k < len; k += G) s_nbr[grp][k] = nbr[b + k];
    __syncthreads();
    auto pos_of = [&](int32_t v) -> uint32_t
    {
      int lo = 0, hi = len;
      while (lo < hi)
      {
        const int mid = (lo + hi) >> 1;
        if (s_nbr[grp][mid] < v) lo = mid + 1; else hi = mid;
      }
      return (uint32_t)lo;
    };
    if (live)
    {
      if (gl == 0) diagpos[r] = (uint8_t)pos_of((int32_t)r);
      const int64_t cb = d2c_off[r], ce = d2c_off[r + 1];
      for (int64_t t = cb + gl; t < ce; t += G)
      {
        const int64_t c = d2c[t];
        uint32_t w = 0;
        for (int j = 0; j < nd; ++j)
        {
          const int32_t v = dofmap[c * nd + j];
          w |= pos_of(v) << (8 * j);
          if (v == (int32_t)r) cpos[c * nd + j] = (uint8_t)(t - cb);
        }
        slot4[t] = w;
      }
    }
    __syncthreads(); // s_nbr is reused by the next block of rows
  }
}

// The three row lists of a plan -- active (rowmark), special (rowmark & special) and plain (rowmark & !special),
// all ascending -- from one count pass and one write pass over the two byte arrays.  Tile counts of the special
// and plain rows are packed into one int64 for a single scan; active = special + plain.
__global__ void __launch_bounds__(kBlock) plan_row_lists_count_kernel(int64_t n, const uint8_t* __restrict__ rowmark,
                                                                      const uint8_t* __restrict__ special,
                                                                      int64_t* __restrict__ tile_counts)
{
  const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kByteItems;
  unsigned fa = 0, fs = 0;
  if (base < n)
  {
    fa = byte_flags(rowmark, base, n, ByteNonZero{});
    fs = byte_flags(special, base, n, ByteNonZero{}) & fa;
  }
  int tot_s, tot_p;
  (void)block_exclusive_scan<int>(__popc(fs), tot_s);
  (void)block_exclusive_scan<int>(__popc(fa & ~fs), tot_p);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = (int64_t)tot_s | ((int64_t)tot_p << 32);
}

// one tile of kByteTile dofs: its part of the three lists, nothing at or beyond the capacities.  first(mine): where the
// tile's lists start (special | plain << 32), asked once the tile's own packed counts are known (all threads call it: a
// table look-up, or the look-back of a chained launch).  Returns first + mine (all threads).
template <typename First>
__device__ __forceinline__ int64_t row_lists_tile(int64_t n, const uint8_t* __restrict__ rowmark,
                                                  const uint8_t* __restrict__ special, const int64_t tile, First first,
                                                  int32_t* __restrict__ active, int32_t* __restrict__ special_rows,
                                                  int32_t* __restrict__ plain_rows, const int64_t cap_s, const int64_t cap_p,
                                                  int32_t* __restrict__ special_pos)
{
  const int64_t base = (tile * kBlock + threadIdx.x) * kByteItems;
  unsigned fa = 0, fs = 0;
  if (base < n)
  {
    fa = byte_flags(rowmark, base, n, ByteNonZero{});
    fs = byte_flags(special, base, n, ByteNonZero{}) & fa;
  }
  const unsigned fp = fa & ~fs;
  int tot_s, tot_p;
  const int os = block_exclusive_scan<int>(__popc(fs), tot_s);
  const int op = block_exclusive_scan<int>(__popc(fp), tot_p);
  const int64_t mine = (int64_t)tot_s | ((int64_t)tot_p << 32);
  const int64_t t = first(mine);
  int64_t s = (t & 0xffffffffll) + os, p = (t >> 32) + op;
#pragma unroll
  for (int k = 0; k < kByteItems; ++k)
  {
    const int32_t row = (int32_t)(base + k);
    if (fs & (1u << k))
    {
      // (special_pos: position of a special row in its list, the index of the dof -> facets incidence)
      if (s < cap_s && p <= cap_p) { active[s + p] = row; special_rows[s] = row; if (special_pos) special_pos[row] = (int32_t)s; }
      ++s;
    }
    else if (fp & (1u << k)) { if (p < cap_p && s <= cap_s) { active[s + p] = row; if (plain_rows) plain_rows[p] = row; } ++p; }
  }
  return t + mine;
}

// the counters of the dof -> facets incidence (one per special row, capacity zero_n <= about the dofs): zeroed by the
// kernel that writes the row lists, a launch less than a fill of their own
__device__ __forceinline__ void row_lists_zero(int32_t* __restrict__ zero_counts, int64_t zero_n, int64_t tile)
{
  const int64_t base = (tile * kBlock + threadIdx.x) * kByteItems;
#pragma unroll
  for (int k = 0; k < kByteItems; ++k)
    if (base + k < zero_n) zero_counts[base + k] = 0;
  if (tile == gridDim.x - 1 && threadIdx.x == kBlock - 1)
    for (int64_t i = (int64_t)gridDim.x * kBlock * kByteItems; i < zero_n; ++i) zero_counts[i] = 0;
}

__global__ void __launch_bounds__(kBlock) plan_row_lists_write_kernel(int64_t n, const uint8_t* __restrict__ rowmark,
                                                                      const uint8_t* __restrict__ special,
                                                                      const int64_t* __restrict__ tile_offsets,
                                                                      int32_t* __restrict__ active,
                                                                      int32_t* __restrict__ special_rows,
                                                                      int32_t* __restrict__ plain_rows, DevN n_special_d,
                                                                      DevN n_plain_d, int32_t* __restrict__ special_pos,
                                                                      int32_t* __restrict__ zero_counts, int64_t zero_n)
{
  // (lists sized by the previous step: nothing is written beyond their published lengths -- 0 in a void step)
  const int64_t cap_s = n_special_d.dev ? dev_n(n_special_d) : INT64_MAX, cap_p = n_plain_d.dev ? dev_n(n_plain_d) : INT64_MAX;
  if (zero_counts) row_lists_zero(zero_counts, zero_n, blockIdx.x);
  (void)row_lists_tile(n, rowmark, special, blockIdx.x, [&](int64_t) { return tile_offsets[blockIdx.x]; }, active,
                       special_rows, plain_rows, cap_s, cap_p, special_pos);
}

// count, offsets and lists in ONE launch (tiles chained by look-back, cfx_device.h) -- inside a sync-free step, where the
// lists are sized by the previous step before anything is counted.  The per-tile counts go to `tile_counts` as
// plan_row_lists_count_kernel leaves them; the last tile leaves the packed totals in `total_out` and publishes the step's
// counts; nothing is written beyond the capacities, and a total beyond them voids the step there.
__global__ void __launch_bounds__(kBlock) plan_row_lists_chained_kernel(int64_t n, const uint8_t* __restrict__ rowmark,
                                                                        const uint8_t* __restrict__ special,
                                                                        int32_t* __restrict__ active,
                                                                        int32_t* __restrict__ special_rows,
                                                                        int32_t* __restrict__ plain_rows, int64_t cap_s,
                                                                        int64_t cap_p, int32_t* __restrict__ special_pos,
                                                                        int32_t* __restrict__ zero_counts, int64_t zero_n,
                                                                        ChainState chain, int64_t* __restrict__ tile_counts,
                                                                        int64_t* __restrict__ total_out, CountJobs after)
{
  const unsigned int tile = chain_take_tile(chain.ticket);
  if (zero_counts) row_lists_zero(zero_counts, zero_n, tile);
  const int64_t end = row_lists_tile(n, rowmark, special, tile,
                                     [&](int64_t mine)
                                     {
                                       if (threadIdx.x == 0) tile_counts[tile] = mine;
                                       return (int64_t)chain_exclusive_prefix(chain.state, tile, (unsigned long long)mine);
                                     },
                                     active, special_rows, plain_rows, cap_s, cap_p, special_pos);
  if (tile == gridDim.x - 1 && threadIdx.x == 0)
  {
    *total_out = end;
    if (after.n > 0) count_publish(after);
  }
}

// 0 inactive, 1 plain (uncut-cell items only), 2 special
__global__ void plan_row_class_kernel(int64_t n, const uint8_t* __restrict__ rowmark, const uint8_t* __restrict__ special,
                                      uint8_t* __restrict__ cls)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cls[i] = rowmark[i] ? (special[i] ? 2 : 1) : 0;
}

struct ByteIs
{
  uint8_t v;
  __device__ bool operator()(uint8_t x) const { return x == v; }
};

#ifndef CFX_MASKS_G
#define CFX_MASKS_G 4
#endif
// stencil mask of a plain row: OR over its marked incident cells of the positions of their dofs, and
// the mark byte shared by all its incident cells (0: not uniform).  G lanes per row.
__device__ __forceinline__ void plain_masks_rows(const int64_t n_plain, const int64_t i, const int32_t* __restrict__ rows,
                                                            const int64_t* __restrict__ d2c_off,
                                                            const int32_t* __restrict__ d2c,
                                                            const uint32_t* __restrict__ slot4,
                                                            const uint8_t* __restrict__ cellmark, int nd,
                                                            const int64_t* __restrict__ st_off,
                                                            unsigned long long* __restrict__ masks,
                                                            uint8_t* __restrict__ uniform, int32_t* __restrict__ counts,
                                                            int* maxlen, const uint8_t* __restrict__ rowcls, uint8_t bulk_bits)
{
  constexpr int G = CFX_MASKS_G;
  const int lane = threadIdx.x, gl = lane % G;
  const bool live = i < n_plain;
  const int64_t r = live ? rows[i] : 0;
  unsigned long long m = 0;
  unsigned all_or = 0, all_and = 0xffu; // over the incident cells: equal iff every cell has the same mark
  bool need_slots = false; // this lane met an unmarked cell
  int full_cells = 0;      // marked cells this lane did not expand into stencil positions
  // a bulk row: every cell around it is an uncut entity -- the whole stencil, one mark, nothing to gather (no cell
  // around a PLAIN row carries a rule mark: the dofs of a rule cell are special rows)
  const bool bulk = live && rowcls && rowcls[r] == kRowIn;
  if (bulk) { all_or = all_and = bulk_bits; full_cells = 1; }
  else if (live)
  {
    const int64_t cb = d2c_off[r];
    const int nc = (int)(d2c_off[r + 1] - cb);
    constexpr int R = 24 / G;
    for (int base = 0; base < nc; base += R * G)
    {
      int32_t cell[R];
      uint8_t mk[R];
#pragma unroll
      for (int k = 0; k < R; ++k)
      {
        const int t = base + k * G + gl;
        cell[k] = t < nc ? d2c[cb + t] : -1;
      }
#pragma unroll
      for (int k = 0; k < R; ++k) mk[k] = cell[k] >= 0 ? cellmark[cell[k]] : (uint8_t)0;
      bool all_marked = true;
#pragma unroll
      for (int k = 0; k < R; ++k)
      {
        if (cell[k] >= 0) { all_or |= mk[k]; all_and &= mk[k]; all_marked = all_marked && mk[k] != 0; }
      }
      // the stencil positions are only needed around an unmarked cell: a row whose incident cells are all
      // marked has every stencil neighbour (the stencil is the union of its incident cells' dofs)
      if (!all_marked)
      {
        need_slots = true;
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (mk[k])
          {
            const uint32_t s4 = slot4[cb + base + k * G + gl];
            for (int j = 0; j < nd; ++j) m |= 1ull << ((s4 >> (8 * j)) & 0xffu);
          }
      }
      else
      {
#pragma unroll
        for (int k = 0; k < R; ++k) full_cells += cell[k] >= 0 ? 1 : 0;
      }
    }
  }
  int any_unmarked = need_slots ? 1 : 0;
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1)
  {
    m |= __shfl_xor(m, o, G);
    all_or |= __shfl_xor(all_or, o, G);
    all_and &= __shfl_xor(all_and, o, G);
    any_unmarked |= __shfl_xor(any_unmarked, o, G);
    full_cells += __shfl_xor(full_cells, o, G);
  }
  if (live && (any_unmarked == 0 || full_cells == 0) == false)
  {
    // mixed row: some lanes skipped the positions of their (all marked) chunk -- expand those cells now
    const int64_t cb = d2c_off[r];
    const int nc = (int)(d2c_off[r + 1] - cb);
    m = 0;
    for (int t = gl; t < nc; t += G)
      if (cellmark[d2c[cb + t]])
      {
        const uint32_t s4 = slot4[cb + t];
        for (int j = 0; j < nd; ++j) m |= 1ull << ((s4 >> (8 * j)) & 0xffu);
      }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) m |= __shfl_xor(m, o, G);
  }
  else if (live && any_unmarked == 0)
  {
    const int len = (int)(st_off[r + 1] - st_off[r]);
    m = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
  }
  if (live && gl == 0)
  {
    masks[i] = m;
    uniform[i] = (all_or == all_and) ? (uint8_t)all_or : (uint8_t)0;
    if (counts) // (the sparsity build asked for the row lengths in the same pass: pattern_plain_len_kernel otherwise)
    {
      const int cnt = __popcll(m);
      counts[r] = cnt;
      if (cnt > *reinterpret_cast<volatile int*>(maxlen)) atomicMax(maxlen, cnt);
    }
  }
}


// 64 consecutive plain rows per wavefront.  All of them bulk rows (every cell around them an uncut entity -- most
// wavefronts away from the interface): one lane per row, the whole stencil, one mark, nothing gathered; else four passes
// of 16 rows with G lanes per row.
__global__ void __launch_bounds__(kWave) plain_masks_kernel(DevN n_plain_d, const int32_t* __restrict__ rows,
                                                            const int64_t* __restrict__ d2c_off,
                                                            const int32_t* __restrict__ d2c,
                                                            const uint32_t* __restrict__ slot4,
                                                            const uint8_t* __restrict__ cellmark, int nd,
                                                            const int64_t* __restrict__ st_off,
                                                            unsigned long long* __restrict__ masks,
                                                            uint8_t* __restrict__ uniform, int32_t* __restrict__ counts,
                                                            int* maxlen, const uint8_t* __restrict__ rowcls, uint8_t bulk_bits)
{
  static_assert(kWave / CFX_MASKS_G == 16, "sixteen rows per pass");
  const int64_t n_plain = dev_n(n_plain_d);
  const int lane = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kWave;
  if (i0 >= n_plain) return;
  if (rowcls)
  {
    const int64_t i = i0 + lane;
    const bool live = i < n_plain;
    const int64_t r = live ? rows[i] : 0;
    const bool bulk = live && rowcls[r] == kRowIn;
    if (__ballot(live && !bulk) == 0ull)
    {
      int cnt = 0;
      if (live)
      {
        cnt = (int)(st_off[r + 1] - st_off[r]);
        masks[i] = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        uniform[i] = bulk_bits;
        if (counts) counts[r] = cnt;
      }
      if (counts)
      {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt = max(cnt, __shfl_xor(cnt, o, 64));
        if (lane == 0 && cnt > *reinterpret_cast<volatile int*>(maxlen)) atomicMax(maxlen, cnt);
      }
      return;
    }
  }
  for (int pass = 0; pass < CFX_MASKS_G; ++pass)
    plain_masks_rows(n_plain, i0 + pass * 16 + lane / CFX_MASKS_G, rows, d2c_off, d2c, slot4, cellmark, nd, st_off, masks, uniform,
                     counts, maxlen, rowcls, bulk_bits);
}

// sparsity of the plain rows, pass 1: row length = popcount of the mask
__global__ void pattern_plain_len_kernel(DevN n_plain_d, const int32_t* __restrict__ rows,
                                         const unsigned long long* __restrict__ masks, int32_t* __restrict__ counts,
                                         int* maxlen)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_plain) return;
  const int cnt = __popcll(masks[i]);
  counts[rows[i]] = cnt;
  if (cnt > *reinterpret_cast<volatile int*>(maxlen)) atomicMax(maxlen, cnt);
}

// pass 2: the set bits of the mask select the row's columns from the (sorted) stencil
#ifndef CFX_PPW_LANES
#define CFX_PPW_LANES 4 // lanes per plain row (512^3: 16 -> 1304 us, 8 -> 933, 4 -> 793, 2 -> 1457)
#endif
__global__ void __launch_bounds__(kBlock) pattern_plain_write_kernel(DevN n_plain_d, const int32_t* __restrict__ rows,
                                                                     const unsigned long long* __restrict__ masks,
                                                                     const int64_t* __restrict__ off,
                                                                     const int32_t* __restrict__ nbr,
                                                                     const int64_t* __restrict__ indptr,
                                                                     int32_t* __restrict__ indices)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / CFX_PPW_LANES;
  if (i >= n_plain) return;
  const int64_t r = rows[i];
  const unsigned long long m = masks[i];
  const int64_t b = off[r], ob = indptr[r];
  const int len = (int)(off[r + 1] - b);
  for (int p = (int)(t - i * CFX_PPW_LANES); p < len; p += CFX_PPW_LANES)
    if ((m >> p) & 1ull) indices[ob + __popcll(m & ((1ull << p) - 1ull))] = nbr[b + p];
}

// ---------------------------------------------------------------------------
// Row tiles of the stencil (Stencil::tile_verts / st_loc): one wavefront per tile of kRowTile dofs.
// The rows' neighbour lists are ascending, so the rank of an entry in the merged list is the sum of its
// lower bounds in the kRowTile lists; copies of one vertex share a rank, different vertices never do:
// writing every entry to slot[rank] and packing the used slots gives the sorted union without a sort.
// ---------------------------------------------------------------------------
constexpr int kTileMaxSt = kRowTile * 64; // a stencil holds at most 64 neighbours

// MODE 0: count the union of every tile; 1: write tile_verts / st_loc at the scanned offsets (the two-pass form);
// 2: one pass -- st_loc final, the union into a staging row of kTileStage entries per tile (tile_verts then points at
// the staging array and tile_voff is unused), compacted by stencil_tiles_pack_kernel once the offsets are known
constexpr int kTileStage = 256;
template <int MODE>
__global__ void __launch_bounds__(kWave) stencil_tiles_kernel(int64_t ndofs, int64_t ntiles,
                                                              const int64_t* __restrict__ st_off,
                                                              const int32_t* __restrict__ nbr,
                                                              const int64_t* __restrict__ d2c_off,
                                                              int32_t* __restrict__ counts,
                                                              const int64_t* __restrict__ tile_voff,
                                                              int32_t* __restrict__ tile_verts,
                                                              uint16_t* __restrict__ st_loc, int* maxima)
{
  __shared__ int32_t s_v[kTileMaxSt];
  __shared__ int32_t s_m[kTileMaxSt];
  __shared__ uint16_t s_rank[kTileMaxSt];
  __shared__ int s_off[kRowTile + 1];
  const int lane = threadIdx.x;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x)
  {
    const int64_t r0 = t * kRowTile;
    const int64_t b0 = st_off[r0];
    if (lane <= kRowTile) s_off[lane] = (int)(st_off[r0 + lane < ndofs ? r0 + lane : ndofs] - b0);
    __syncthreads();
    const int n = s_off[kRowTile];
    if (n > kTileMaxSt)
    {
      if (lane == 0) atomicMax(&maxima[1], n);
      __syncthreads();
      continue;
    }
    for (int e = lane; e < n; e += kWave) { s_v[e] = nbr[b0 + e]; s_m[e] = -1; }
    __syncthreads();
    for (int e = lane; e < n; e += kWave)
    {
      const int32_t v = s_v[e];
      int rank = 0;
#pragma unroll 4
      for (int j = 0; j < kRowTile; ++j)
      {
        int lo = s_off[j], hi = s_off[j + 1];
        const int b = lo;
        while (lo < hi)
        {
          const int mid = (lo + hi) >> 1;
          if (s_v[mid] < v) lo = mid + 1; else hi = mid;
        }
        rank += lo - b;
      }
      s_rank[e] = (uint16_t)rank;
      s_m[rank] = v; // copies of v write the same value to the same slot
    }
    __syncthreads();
    // pack the used slots in order: slot -> position in the union
    int total = 0;
    constexpr bool WRITE = MODE != 0;
    const int64_t vb = MODE == 1 ? tile_voff[t] : (MODE == 2 ? t * kTileStage : 0);
    for (int c = 0; c < n; c += kWave)
    {
      const int e = c + lane;
      const int32_t v = e < n ? s_m[e] : -1;
      const unsigned long long used = __ballot(v >= 0);
      const int pos = total + __popcll(used & ((1ull << lane) - 1ull));
      if (v >= 0)
      {
        if constexpr (MODE == 1) tile_verts[vb + pos] = v;
        if constexpr (MODE == 2) { if (pos < kTileStage) tile_verts[vb + pos] = v; }
        s_m[e] = pos;
      }
      total += __popcll(used);
    }
    __syncthreads();
    if constexpr (WRITE)
    {
      for (int e = lane; e < n; e += kWave) st_loc[b0 + e] = (uint16_t)s_m[s_rank[e]];
    }
    if (MODE != 1 && lane == 0)
    {
      counts[t] = total;
      const int64_t rl = r0 + kRowTile < ndofs ? r0 + kRowTile : ndofs;
      const int items = (int)(d2c_off[rl] - d2c_off[r0]);
      if (total > *reinterpret_cast<volatile int*>(&maxima[0])) atomicMax(&maxima[0], total);
      if (n > *reinterpret_cast<volatile int*>(&maxima[1])) atomicMax(&maxima[1], n);
      if (items > *reinterpret_cast<volatile int*>(&maxima[2])) atomicMax(&maxima[2], items);
    }
    __syncthreads(); // the LDS arrays are reused by the next tile
  }
}

__global__ void __launch_bounds__(kBlock) stencil_tiles_pack_kernel(int64_t ntiles, const int32_t* __restrict__ counts,
                                                                    const int64_t* __restrict__ tile_voff,
                                                                    const int32_t* __restrict__ staged,
                                                                    int32_t* __restrict__ tile_verts)
{
  const int64_t t = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (t >= ntiles) return;
  const int n = counts[t];
  const int64_t vb = tile_voff[t];
  for (int k = threadIdx.x % kWave; k < n; k += kWave) tile_verts[vb + k] = staged[t * kTileStage + k];
}

// plain-list positions at which a new row tile starts
struct TileStart
{
  const int32_t* rows;
  __device__ bool operator()(int64_t i) const { return i == 0 || (rows[i] / kRowTile) != (rows[i - 1] / kRowTile); }
};

struct TileIdEmit
{
  const int32_t* rows;
  int32_t* ids;
  __device__ void operator()(int64_t o, int64_t i) const { ids[o] = rows[i] / kRowTile; }
};

__global__ void tile_ids_kernel(DevN n_d, const int32_t* __restrict__ first, const int32_t* __restrict__ rows,
                                int32_t* __restrict__ ids)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = rows[first[i]] / kRowTile;
}

// pass 2 by row tile (Stencil tiles, plan.plain_tile_first / _id): the rows of a tile are neighbours in the stencil
// and in the CSR arrays.  A tile whose kRowTile rows are all plain and hold their whole stencil -- the bulk of the
// domain -- is one contiguous copy nbr[st_off[r0] ...] -> indices[indptr[r0] ...]; other tiles go row by row.
__global__ void __launch_bounds__(kWave) pattern_plain_tiles_kernel(DevN n_tiles_d, const int32_t* __restrict__ tile_first,
                                                                    const int32_t* __restrict__ tile_id, DevN n_plain_d,
                                                                    const int32_t* __restrict__ rows,
                                                                    const unsigned long long* __restrict__ masks,
                                                                    const int64_t* __restrict__ off,
                                                                    const int32_t* __restrict__ nbr,
                                                                    const int64_t* __restrict__ indptr,
                                                                    int32_t* __restrict__ indices, int64_t ndofs)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const int64_t n_tiles = dev_n(n_tiles_d);
  constexpr int G = kWave / kRowTile;
  const int lane = threadIdx.x, g = lane / G, gl = lane % G;
  const int64_t w = blockIdx.x;
  if (w >= n_tiles) return;
  const int64_t i0 = tile_first[w], t = tile_id[w], r0 = t * kRowTile;
  int32_t rl = -1;
  if (lane < kRowTile && i0 + lane < n_plain) rl = rows[i0 + lane];
  unsigned pm = (rl >= 0 && rl / kRowTile == t) ? 1u << (rl % kRowTile) : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pm |= __shfl_xor(pm, o, 64);
  const int64_t rr = r0 + (lane < kRowTile ? lane : kRowTile);
  const int64_t so = off[rr < ndofs ? rr : ndofs], io = indptr[rr < ndofs ? rr : ndofs];
  const bool live = (pm >> g) & 1u;
  const unsigned long long m = live ? masks[i0 + __popc(pm & ((1u << g) - 1u))] : 0ull;
  const int64_t sb = __shfl(so, g, 64), ob = __shfl(io, g, 64);
  const int slen = (int)(__shfl(so, g + 1, 64) - sb);
  const bool full = live && (m & (m + 1ull)) == 0ull && __popcll(m) == slen;
  const int64_t sb0 = __shfl(so, 0, 64), ob0 = __shfl(io, 0, 64);
  const int nst = (int)(__shfl(so, kRowTile, 64) - sb0);
  if (pm == 0xffffu && __ballot(!full) == 0ull && __shfl(io, kRowTile, 64) - ob0 == nst)
  {
    // (eight loads in flight per lane -- a tile of a P1 space on a Kuhn mesh holds ~430 entries: all of them -- where one
    // load -> store per trip left the wavefront seven dependent latencies long)
    for (int j0 = 0; j0 < nst; j0 += 8 * kWave)
    {
      int32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j0 + lane + u * kWave < nst ? nbr[sb0 + j0 + lane + u * kWave] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + lane + u * kWave < nst) indices[ob0 + j0 + lane + u * kWave] = v[u];
    }
    return;
  }
  if (!live) return;
  for (int p = gl; p < slen; p += G)
    if ((m >> p) & 1ull) indices[ob + __popcll(m & ((1ull << p) - 1ull))] = nbr[sb + p];
}

// Spaces with neighbour lists only (degree 2, vector-valued, DG): a plain row whose incident cells are ALL uncut
// entities of the form couples exactly its static list -- its columns are a copy, no hash set.  full[i] = 1 for
// such rows (and their expanded row lengths go to counts); the others join the hashed rows.
// (64 consecutive plain rows per wavefront: all of them bulk rows -- nearly every wavefront away from the interface -- one
// lane per row and nothing gathered; else kWave / G passes of G lanes per row)
template <int G>
__global__ void __launch_bounds__(kWave) plain_full_kernel(int64_t n_plain, const int32_t* __restrict__ rows,
                                                            const int64_t* __restrict__ d2c_off, const int32_t* __restrict__ d2c,
                                                            const uint8_t* __restrict__ cellmark,
                                                            const int64_t* __restrict__ st_off, int bs,
                                                            uint8_t* __restrict__ full, int32_t* __restrict__ counts,
                                                            const uint8_t* __restrict__ rowcls)
{
  const int lane = threadIdx.x, gl = lane % G;
  const int64_t i0 = (int64_t)blockIdx.x * kWave;
  if (i0 >= n_plain) return;
  if (rowcls)
  {
    // a bulk row (cfx_row_plan::rowcls): every cell around it is an uncut entity
    const int64_t i = i0 + lane;
    const bool live = i < n_plain;
    const int64_t r = live ? rows[i] : 0;
    const bool bulk = live && rowcls[r] == kRowIn && d2c_off[r + 1] > d2c_off[r];
    if (__ballot(live && !bulk) == 0ull)
    {
      if (live)
      {
        full[i] = 1;
        const int len = (int)(st_off[r + 1] - st_off[r]);
        for (int a = 0; a < bs; ++a) counts[r * bs + a] = len * bs;
      }
      return;
    }
  }
  for (int pass = 0; pass < G; ++pass)
  {
    const int64_t i = i0 + pass * (kWave / G) + lane / G;
    const bool live = i < n_plain;
    const int64_t r = live ? rows[i] : 0;
    const int64_t cb = live ? d2c_off[r] : 0;
    int nc = live ? (int)(d2c_off[r + 1] - cb) : 0;
    const bool bulk = live && rowcls && rowcls[r] == kRowIn && nc > 0;
    const int nc_all = nc;
    if (bulk) nc = 0;
    int miss = 0;
    for (int t = gl; t < nc; t += G) miss |= (cellmark[d2c[cb + t]] & 0x0Fu) == 0;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) miss |= __shfl_xor(miss, o, G);
    if (!live || gl != 0) continue;
    const bool f = miss == 0 && nc_all > 0;
    full[i] = f ? 1 : 0;
    if (f)
    {
      const int len = (int)(st_off[r + 1] - st_off[r]);
      for (int a = 0; a < bs; ++a) counts[r * bs + a] = len * bs;
    }
  }
}

// 12-byte slot records of a degree-2 scalar space (Stencil::slotn): 16 lanes per dof, the dof's neighbour list staged
// in LDS, (incident cell, local dof) pairs spread over the lanes, one binary search each
__global__ void __launch_bounds__(kWave) stencil_slotn_kernel(int64_t ndofs, const int64_t* __restrict__ d2c_off,
                                                              const int32_t* __restrict__ d2c, const int32_t* __restrict__ dofmap, int nd,
                                                              const int64_t* __restrict__ off, const int32_t* __restrict__ nbr,
                                                              uint8_t* __restrict__ slotn)
{
  constexpr int G = 16, RPW = kWave / G;
  __shared__ int32_t s_nbr[RPW][256];
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  for (int64_t blk = blockIdx.x; blk * RPW < ndofs; blk += gridDim.x)
  {
    const int64_t r = blk * RPW + grp;
    const bool live = r < ndofs;
    const int64_t b = live ? off[r] : 0;
    const int len = live ? (int)(off[r + 1] - b) : 0; // < 256 (space_stencil_slotn gives up otherwise)
    for (int k = gl; k < len; k += G) s_nbr[grp][k] = nbr[b + k];
    __syncthreads();
    if (live)
    {
      const int64_t cb = d2c_off[r];
      const int npairs = (int)(d2c_off[r + 1] - cb) * nd;
      for (int t = gl; t < npairs; t += G)
      {
        const int q = t / nd, j = t - q * nd;
        const int32_t v = dofmap[(int64_t)d2c[cb + q] * nd + j];
        int lo = 0, hi = len;
        while (lo < hi)
        {
          const int mid = (lo + hi) >> 1;
          if (s_nbr[grp][mid] < v) lo = mid + 1; else hi = mid;
        }
        uint8_t* rec = slotn + (cb + q) * 12;
        rec[j] = (uint8_t)lo;
        if (v == (int32_t)r) rec[10] = (uint8_t)j;
      }
    }
    __syncthreads();
  }
}

struct StaticLenTest
{
  const int32_t* rows;
  const int64_t* st_off;
  int limit;
  bool above;
  __device__ bool operator()(int64_t i) const
  {
    const int64_t r = rows[i];
    return ((int)(st_off[r + 1] - st_off[r]) > limit) == above;
  }
};

__global__ void gather2_kernel(const int64_t* a, const int* b, int64_t* out)
{
  out[0] = *a; out[1] = *b;
}

__global__ void gather3_kernel(const int* a, const int* b, const int64_t* c, int64_t* out)
{
  out[0] = *a; out[1] = *b; out[2] = *c;
}

struct FlagSet8
{
  const uint8_t* f;
  __device__ bool operator()(int64_t i) const { return f[i] != 0; }
};

struct FlagIsZero
{
  const uint8_t* f;
  __device__ bool operator()(int64_t i) const { return f[i] == 0; }
};

template <int G>
__global__ void __launch_bounds__(kBlock) pattern_plain_copy_kernel(int64_t n_plain, const int32_t* __restrict__ rows,
                                                                    const uint8_t* __restrict__ full,
                                                                    const int64_t* __restrict__ st_off,
                                                                    const int32_t* __restrict__ nbr, int bs,
                                                                    const int64_t* __restrict__ indptr, int32_t* __restrict__ indices)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / G;
  if (i >= n_plain || !full[i]) return;
  const int gl = (int)(t - i * G);
  const int64_t r = rows[i];
  const int64_t sb = st_off[r];
  const int len = (int)(st_off[r + 1] - sb);
  if (bs == 1)
  {
    const int64_t ob = indptr[r];
    for (int k = gl; k < len; k += G) indices[ob + k] = nbr[sb + k];
    return;
  }
  for (int k = gl; k < len; k += G)
  {
    const int32_t col = nbr[sb + k];
    for (int a = 0; a < bs; ++a)
    {
      const int64_t ob = indptr[r * bs + a] + (int64_t)k * bs;
      for (int b = 0; b < bs; ++b) indices[ob + b] = col * bs + b;
    }
  }
}

// ... block spaces (bs > 1): the bs rows of a dof are one contiguous run of bs x (len x bs) entries, each row the dof's
// static list expanded by the block size.  A wavefront takes 64 entries of the plain-row list and writes the runs of the
// dofs that copy their list one after the other, one entry per lane and store: whole lines instead of the 4 B stores a
// stride of 32 B apart that eight lanes per dof issued (configs[4] share: 2.9 ms for 5.7 GB of indices).
template <int BS>
__global__ void __launch_bounds__(kWave) pattern_plain_copy_block_kernel(int64_t n_plain, const int32_t* __restrict__ rows,
                                                                        const uint8_t* __restrict__ full,
                                                                        const int64_t* __restrict__ st_off,
                                                                        const int32_t* __restrict__ nbr,
                                                                        const int64_t* __restrict__ indptr,
                                                                        int32_t* __restrict__ indices)
{
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kWave + lane;
  const bool f = i < n_plain && full[i] != 0;
  const int64_t r = f ? (int64_t)rows[i] : 0;
  const int64_t sb = f ? st_off[r] : 0;
  const int len = f ? (int)(st_off[r + 1] - sb) : 0;
  const int64_t ob = f ? indptr[r * BS] : 0;
  for (int j = 0; j < kWave; ++j)
  {
    const int lj = __shfl(len, j, kWave);
    if (lj == 0) continue;
    const int64_t sj = __shfl(sb, j, kWave), oj = __shfl(ob, j, kWave);
    const int L = lj * BS; // entries of one of the dof's rows
    for (int e = lane; e < L; e += kWave)
    {
      const int k = e / BS, b = e - k * BS;
      const int32_t v = nbr[sj + k] * BS + b;
#pragma unroll
      for (int a = 0; a < BS; ++a) indices[oj + (int64_t)a * L + e] = v;
    }
  }
}

// ... scalar spaces, by chunks of 64 entries of the plain-row list: when the chunk is 64 CONSECUTIVE dofs that all
// copy their list (the bulk of the domain), their lists are one contiguous span of `nbr` and their rows one contiguous
// span of `indices` of the same length -- a straight wave-wide copy, 256 B per instruction instead of 32 B segments
// behind three dependent loads per row (configs[3]: 5.5 ms for 2 x 7.7 GB).  Other chunks go row by row.
__global__ void __launch_bounds__(kWave) pattern_plain_copy_runs_kernel(int64_t n_plain, const int32_t* __restrict__ rows,
                                                                       const uint8_t* __restrict__ full,
                                                                       const int64_t* __restrict__ st_off,
                                                                       const int32_t* __restrict__ nbr,
                                                                       const int64_t* __restrict__ indptr,
                                                                       int32_t* __restrict__ indices)
{
  const int lane = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kWave + lane;
  const bool in = i < n_plain;
  const int64_t r = in ? (int64_t)rows[i] : -1;
  const bool f = in && full[i] != 0;
  const int64_t r0 = __shfl(r, 0, kWave);
  const bool run = __ballot(f && r == r0 + lane) == ~0ull;
  if (run)
  {
    // (lane 0's row starts the span, lane 63's ends it)
    const int64_t sb0 = st_off[r0], se = st_off[r0 + kWave], ob0 = indptr[r0];
    const int64_t n = se - sb0;
    // (four loads in flight per lane: one load -> store per trip ran the copy at 2 TB/s)
    for (int64_t k = lane; k < n; k += 4 * kWave)
    {
      int32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = k + u * kWave < n ? nbr[sb0 + k + u * kWave] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + u * kWave < n) indices[ob0 + k + u * kWave] = v[u];
    }
    return;
  }
  const int64_t sb = f ? st_off[r] : 0;
  const int len = f ? (int)(st_off[r + 1] - sb) : 0;
  const int64_t ob = f ? indptr[r] : 0;
  for (int j = 0; j < kWave; ++j)
  {
    const int lj = __shfl(len, j, kWave);
    if (lj == 0) continue;
    const int64_t sj = __shfl(sb, j, kWave), oj = __shfl(ob, j, kWave);
    for (int k = lane; k < lj; k += kWave) indices[oj + k] = nbr[sj + k];
  }
}

} // namespace

namespace cfx
{

// A form points at entity lists and rules it does not own.  Lists that live in library blocks (located lists, ghost
// rows: dropped by cfx_cut_update) and rules handles carry serial numbers taken at form creation: if the block was
// released or handed out again, or the rules handle destroyed, the form is stale and every use of it is refused.
void validate_form(const cfx_form_s* a)
{
  for (const cfx_integral_dev& I : a->integrals)
  {
    if (I.entities_serial != 0 && I.n_entities.cap() > 0 && !I.entities.owned && dev_block_serial(I.entities.p) != I.entities_serial)
      throw Error(CFX_ERR_RUNTIME, "stale form: an entity list it refers to was released or rebuilt (cfx_cut_update drops the "
                                   "located lists and ghost rows of a cut); create the form again from the new lists");
    if (I.rules_serial != 0 && !rules_serial_is_live(I.rules_serial))
      throw Error(CFX_ERR_RUNTIME, "stale form: the runtime rules it refers to were destroyed; create the form again");
  }
}

cfx_row_plan& row_plan(cfx_form_s* a)
{
  if (a->plan) return *a->plan;
  cfx_space_s* V = a->V;
  // identity of this form's entity lists: address and count, plus the serial of the library block behind the address
  // and of the rules handle -- serials are never reused, so a list rebuilt at a recycled address is a different key
  std::vector<std::array<int64_t, 7>> key_cells, key_facets;
  for (size_t ii = 0; ii < a->integrals.size(); ++ii)
  {
    const cfx_integral_dev& I = a->integrals[ii];
    // (Count::key(): a count resolved between the plan of the bilinear form and the look-up of the linear form of the
    // same step must not change the key -- degree-2 steps built their plan twice, +4.2 ms at configs[3])
    const int64_t ne = I.n_entities.cap();
    const std::array<int64_t, 7> k{(int64_t)ii, ne > 0 ? (int64_t)(uintptr_t)I.entities.p : 0, ne > 0 ? I.n_entities.key() : 0,
                                   (int64_t)(uintptr_t)I.rules, I.rules ? I.rules->nr.key() : 0,
                                   ne > 0 ? (int64_t)I.entities_serial : 0, (int64_t)I.rules_serial};
    (I.type == CFX_CELL ? key_cells : key_facets).push_back(k);
  }
  // Share the plan of another LIVE form of this space built from the same lists (caller-owned entity arrays,
  // serial 0, must not change while a form that references them is alive: for them address identity stands for
  // content identity).  A form without facet integrals may use a plan that has some: the extra active rows receive
  // zeros.
  for (auto it = V->plans.begin(); it != V->plans.end();)
  {
    std::shared_ptr<cfx_row_plan> p = it->lock();
    if (!p) { it = V->plans.erase(it); continue; }
    if (p->key_cells == key_cells && (key_facets.empty() || p->key_facets == key_facets)
        && (a->rank == 1 || p->key_facets == key_facets))
    {
      a->plan = p;
      return *a->plan;
    }
    ++it;
  }
  a->plan = std::make_shared<cfx_row_plan>();
  cfx_row_plan& P = *a->plan;
  P.serial = next_serial();
  P.key_cells = key_cells;
  P.key_facets = key_facets;
  V->plans.push_back(a->plan);
  const int nd = V->ndofs_cell;
  const int64_t nc = V->mesh->ncells;
  P.usable = true;
  // the three mark arrays in one block, zeroed by one fill (16 B aligned parts: the kernels read them 16 B per lane)
  // (... and, for P1 spaces on the geometry dofmap, the segment offsets of a linear form's row-ordered staging, which
  // start out as "no segment" = 0: cfx::plain_vec_offsets)
  const int64_t n_cm = (nc + 15) & ~15LL, n_rm = (V->ndofs + 15) & ~15LL;
  const int64_t n_t2 = space_stencil(V).usable ? ((4 * V->ndofs + 15) & ~15LL) : 0;
  P.mark_block.alloc(n_cm + 2 * n_rm + n_t2); // (zeroed below, in one launch with the rule-key tables)
  // Bulk rows: the uncut entities of every cell integral that has any are ONE list, and that list is what
  // cfx_locate_entities made of the current classification of a cut whose level set lives on this space's dofmap
  // (= the geometry dofmap: the space is P1 on it).  CFX_BULK_ROWS=0: marks from the lists, as the reference walks them.
  const cfx_cut_s* bulk_cut = nullptr;
  int bulk_value = 0;
  {
    const char* be = getenv("CFX_BULK_ROWS");
    const void* list = nullptr;
    int64_t list_n = 0;
    bool one_list = true;
    int slot = 0;
    uint8_t bits = 0;
    for (const cfx_integral_dev& I : a->integrals)
    {
      if (I.type != CFX_CELL) continue;
      if (slot >= 4) { one_list = false; break; }
      if (I.n_entities.cap() > 0)
      {
        if (list && (list != I.entities.p || list_n != I.n_entities.cap())) one_list = false;
        list = I.entities.p; list_n = I.n_entities.cap();
        bits |= (uint8_t)(1u << slot);
      }
      ++slot;
    }
    // kind 1: P1 on the geometry dofmap (a dof IS a level-set vertex); kind 2: degree 2, any block size (a dof sits
    // between two vertices: cfx_space_s::dof_verts)
    const bool p1 = space_stencil(V).usable && V->bs == 1 && (nd == 4 || nd == 3);
    const bool p2 = V->degree == 2 && (nd == 10 || nd == 6) && nd == (V->mesh->tdim == 3 ? 10 : 6);
    const ListProvenance* pv = (!(be && be[0] == '0') && one_list && list && (p1 || p2)) ? provenance_lookup(list) : nullptr;
    const int64_t nvert = V->mesh->nnodes;
    if (pv && pv->n == list_n && pv->cut->gen == pv->gen && pv->cut->mesh == V->mesh && pv->cut->ls_dofmap.p == V->mesh->conn.p
        && pv->cut->ls_ndofs == nvert && pv->cut->codes0.n == nvert && pv->cut->touch_valid && pv->cut->touch0.n == nvert
        && (pv->value == -1 || pv->value == 1) && (p1 ? V->ndofs == nvert : space_dof_verts(V)))
    {
      P.bulk = true;
      P.bulk_kind = p1 ? 1 : 2;
      P.bulk_bits = bits;
      bulk_cut = pv->cut;
      bulk_value = pv->value;
    }
  }
  if (n_t2 > 0)
  {
    P.vec_t2off.p = reinterpret_cast<int32_t*>(P.mark_block.p + n_cm + 2 * n_rm); P.vec_t2off.n = V->ndofs; P.vec_t2off.owned = false;
  }
  P.cellmark.p = P.mark_block.p; P.cellmark.n = (nc + 3) & ~3LL; P.cellmark.owned = false;
  P.rowmark.p = P.mark_block.p + n_cm; P.rowmark.n = V->ndofs; P.rowmark.owned = false;
  DevArray<uint8_t> special; // rows touched by a runtime-rule cell or a facet
  DevArray<int32_t> bulk_pop;  // bulk: entities per bit word of slot bulk_packed_slot (plan_bulk_init), scanned below
  int bulk_packed_slot = -1;
  special.p = P.mark_block.p + n_cm + n_rm; special.n = V->ndofs; special.owned = false;
  ZeroFlag flag;
  int plan_flags = 0; // the flag word, read together with the row totals
  // Entity counts may still be in HBM (lists made inside a sync-free step): grids then cover the capacity of a list
  // and the kernels take its length from the device (DevN).
  int n_facet_lists = 0;
  // hash maps parent cell -> first rule of the cell integrals with runtime rules: key tables of all slots in one block
  int64_t key_off[4] = {0, 0, 0, 0}, key_total = 0;
  {
    int slot = 0;
    for (const cfx_integral_dev& I : a->integrals)
    {
      if (I.type != CFX_CELL || slot >= 4) continue;
      const int64_t nr = I.rules ? I.rules->nr.cap() : 0;
      if (nr > 0)
      {
        uint32_t size = 64;
        while (size < 2 * (uint64_t)nr) size <<= 1;
        key_off[slot] = key_total;
        key_total += size;
      }
      ++slot;
    }
    if (key_total > 0) P.rule_key_block.alloc(key_total);
    if (P.bulk)
    {
      // no zero fill: the row classes initialise the row marks, the cell marks come from the classification bytes; the
      // bit words of the first slot with entities and the emptied rule-key tables come out of the same launch
      P.rowcls.alloc(n_rm);
      const int64_t nb = (nc + kClassBlock - 1) / kClassBlock;
      const int64_t nwords = (nc + 63) / 64;
      bulk_packed_slot = __builtin_ctz((unsigned)P.bulk_bits);
      P.std_bits[bulk_packed_slot].alloc(nwords);
      bulk_pop.alloc(nwords);
      BulkInit B{};
      B.kind = P.bulk_kind;
      B.blocks_cls = grid_for(P.bulk_kind == 1 ? n_rm / 16 : n_rm / 4).x;
      B.blocks_cm = grid_for(n_cm / 16).x;
      B.blocks_pack = grid_for(nwords).x;
      B.ndofs = V->ndofs; B.dof_verts = P.bulk_kind == 2 ? V->dof_verts.p : nullptr;
      B.codes = bulk_cut->codes0.p; B.touch = bulk_cut->touch0.p; B.sel = (uint8_t)(bulk_value < 0 ? 1 : 2);
      B.rowcls = P.rowcls.p; B.rowmark = P.rowmark.p; B.special = special.p;
      B.t2off = (P.bulk_kind == 1 && n_t2 > 0) ? P.vec_t2off.p : nullptr;
      B.ncells = nc; B.npad = n_cm; B.domain = bulk_cut->domain.p; B.value = (int8_t)bulk_value; B.bits = P.bulk_bits;
      B.block_class = bulk_cut->block_class.n == nb ? bulk_cut->block_class.p : nullptr;
      B.cellmark = P.cellmark.p;
      B.words = reinterpret_cast<unsigned long long*>(P.std_bits[bulk_packed_slot].p); B.pop = bulk_pop.p;
      B.keys = key_total > 0 ? P.rule_key_block.p : nullptr; B.n_keys = key_total;
      B.poison = step_poison();
      const unsigned blocks_fill = key_total > 0 ? grid_for((key_total + 3) / 4).x : 0u;
      launch("plan_bulk_init", plan_bulk_init_kernel, dim3(B.blocks_cls + B.blocks_cm + B.blocks_pack + blocks_fill), dim3(kBlock), 0, B);
      P.any_cells = true;
    }
    else if (key_total > 0)
      dev_fill2(P.mark_block.p, 0, (size_t)P.mark_block.n, P.rule_key_block.p, 0xff, sizeof(int32_t) * (size_t)key_total);
    else
      P.mark_block.zero();
  }
  RuleJobs rjobs{};
  for (size_t ii = 0; ii < a->integrals.size(); ++ii)
  {
    const cfx_integral_dev& I = a->integrals[ii];
    const int64_t ne = I.n_entities.cap();
    if (I.type == CFX_CELL)
    {
      if (P.n_cell_slots >= 4) { P.usable = false; continue; }
      const int slot = P.n_cell_slots++;
      P.cell_slot_integral[slot] = (int)ii;
      if (ne > 0 && P.bulk) {} // (marked from the classification above)
      else if (ne > 0)
      {
        if (nd == 4)
          launch("plan_mark_entities", plan_mark_entities_kernel<4>, grid_for(ne), dim3(kBlock), 0, I.n_entities,
                 I.entities.p, V->dofmap.p, (uint8_t)(1u << slot), P.cellmark.p, P.rowmark.p, flag.p);
        else if (nd == 3)
          launch("plan_mark_entities", plan_mark_entities_kernel<3>, grid_for(ne), dim3(kBlock), 0, I.n_entities,
                 I.entities.p, V->dofmap.p, (uint8_t)(1u << slot), P.cellmark.p, P.rowmark.p, flag.p);
        else
        {
          launch("plan_mark_cells", plan_mark_cells_kernel, grid_for(ne), dim3(kBlock), 0, I.n_entities,
                 I.entities.p, 1, (uint8_t)(1u << slot), P.cellmark.p);
          launch("plan_mark_rows", plan_mark_rows_cells_kernel, grid_for(ne * nd), dim3(kBlock), 0,
                 I.n_entities, I.entities.p, 1, V->dofmap.p, nd, P.rowmark.p, (uint8_t*)nullptr);
          launch("plan_check_sorted", plan_check_sorted_kernel, grid_for(ne), dim3(kBlock), 0, I.n_entities,
                 I.entities.p, 1, flag.p);
        }
        P.any_cells = true;
      }
      const int64_t nr = I.rules ? I.rules->nr.cap() : 0;
      if (nr > 0)
      {
        uint32_t size = 64;
        while (size < 2 * (uint64_t)nr) size <<= 1;
        P.rule_mask[slot] = size - 1;
        // (the key tables of all slots in one block, emptied by one fill: see rule_key_block above)
        P.rule_keys[slot].p = P.rule_key_block.p + key_off[slot]; P.rule_keys[slot].n = size; P.rule_keys[slot].owned = false;
        P.rule_first[slot].alloc(size);
        if (nd == 4 || nd == 3)
        {
          // (queued: the rule sets of all slots go in one launch behind this loop)
          const int k = rjobs.n++;
          rjobs.start[k + 1] = rjobs.start[k] + ((nr + kBlock - 1) / kBlock) * kBlock;
          rjobs.nr[k] = I.rules->nr; rjobs.parent[k] = I.rules->parent_map.p; rjobs.bit[k] = (uint8_t)(16u << slot);
          rjobs.hmask[k] = P.rule_mask[slot]; rjobs.keys[k] = P.rule_keys[slot].p; rjobs.first[k] = P.rule_first[slot].p;
        }
        else
        {
          launch("plan_mark_cells", plan_mark_cells_kernel, grid_for(nr), dim3(kBlock), 0, I.rules->nr,
                 I.rules->parent_map.p, 1, (uint8_t)(16u << slot), P.cellmark.p);
          launch("plan_mark_rows", plan_mark_rows_cells_kernel, grid_for(nr * nd), dim3(kBlock), 0,
                 I.rules->nr, I.rules->parent_map.p, 1, V->dofmap.p, nd, P.rowmark.p, special.p);
          launch("plan_check_sorted", plan_check_sorted_kernel, grid_for(nr), dim3(kBlock), 0, I.rules->nr,
                 I.rules->parent_map.p, 0, flag.p);
          launch("plan_rule_hash", plan_rule_hash_kernel, grid_for(nr), dim3(kBlock), 0, I.rules->nr,
                 I.rules->parent_map.p, P.rule_mask[slot], P.rule_keys[slot].p, P.rule_first[slot].p);
        }
        P.any_cells = true;
      }
    }
    else if (I.type == CFX_INTERIOR_FACET)
    {
      if (P.n_facet_slots >= 2) { P.usable = false; continue; }
      P.facet_slot_integral[P.n_facet_slots++] = (int)ii;
      if (ne > 0) ++n_facet_lists;
    }
  }
  // (one launch for the rule sets; the rows of the form's only facet list join it)
  bool facets_with_rules = false;
  if (rjobs.n > 0)
  {
    int64_t threads = rjobs.start[rjobs.n];
    if (n_facet_lists == 1)
      for (int s = 0; s < P.n_facet_slots; ++s)
      {
        const cfx_integral_dev& I = a->integrals[P.facet_slot_integral[s]];
        if (I.n_entities.cap() == 0) continue;
        rjobs.nf = I.n_entities; rjobs.facet_rows = I.entities.p;
        threads += I.n_entities.cap();
        facets_with_rules = true;
      }
    if (nd == 4)
      launch("plan_rules", plan_rules_kernel<4>, grid_for(threads), dim3(kBlock), 0, rjobs, V->dofmap.p,
             P.cellmark.p, P.rowmark.p, special.p, flag.p);
    else
      launch("plan_rules", plan_rules_kernel<3>, grid_for(threads), dim3(kBlock), 0, rjobs, V->dofmap.p,
             P.cellmark.p, P.rowmark.p, special.p, flag.p);
  }
  // the facet rows of all facet integrals, concatenated: one list keeps its (possibly pending) length, several are
  // joined at their exact lengths
  P.nfacets = Count(0);
  if (n_facet_lists == 1)
  {
    for (int s = 0; s < P.n_facet_slots; ++s)
      if (a->integrals[P.facet_slot_integral[s]].n_entities.cap() > 0) P.nfacets = a->integrals[P.facet_slot_integral[s]].n_entities;
  }
  else if (n_facet_lists > 1)
  {
    int64_t total = 0;
    for (int s = 0; s < P.n_facet_slots; ++s) total += a->integrals[P.facet_slot_integral[s]].n_entities.value();
    P.nfacets = Count(total);
  }
  const int64_t nf_cap = P.nfacets.cap();
  if (nf_cap > 0)
  {
    // one list: the plan refers to the integral's own rows (the form's entity array: alive and unchanged while a form
    // that shares this plan is used, cfx::validate_form) and needs no slot array -- every facet belongs to slot 0 of
    // the facet integrals that have entities
    const bool single_list = n_facet_lists == 1;
    if (!single_list)
    {
      P.facet_rows.alloc(nf_cap * 4);
      P.facet_slot.alloc(nf_cap);
    }
    int64_t o = 0;
    for (int s = 0; s < P.n_facet_slots; ++s)
    {
      const cfx_integral_dev& I = a->integrals[P.facet_slot_integral[s]];
      const int64_t ne = I.n_entities.cap(); // (exact unless this is the only list)
      if (ne == 0) continue;
      if (single_list)
      {
        P.facet_rows.p = I.entities.p; P.facet_rows.n = ne * 4; P.facet_rows.owned = false;
      }
      else
      {
        CFX_HIP(hipMemcpyAsync(P.facet_rows.p + 4 * o, I.entities.p, sizeof(int32_t) * 4 * (size_t)ne,
                               hipMemcpyDeviceToDevice, ctx().stream));
        dev_fill(P.facet_slot.p + o, s, (size_t)ne);
      }
      if (facets_with_rules) {} // (marked by plan_rules)
      else if (nd == 4)
        launch("plan_facet_rows", plan_facet_rows_kernel<4>, grid_for(ne), dim3(kBlock), 0, I.n_entities,
               I.entities.p, V->dofmap.p, P.rowmark.p, special.p, flag.p);
      else if (nd == 3)
        launch("plan_facet_rows", plan_facet_rows_kernel<3>, grid_for(ne), dim3(kBlock), 0, I.n_entities,
               I.entities.p, V->dofmap.p, P.rowmark.p, special.p, flag.p);
      else
      {
        launch("plan_mark_rows", plan_mark_rows_cells_kernel, grid_for(ne * nd), dim3(kBlock), 0,
               I.n_entities, I.entities.p, 4, V->dofmap.p, nd, P.rowmark.p, special.p);
        launch("plan_mark_rows", plan_mark_rows_cells_kernel, grid_for(ne * nd), dim3(kBlock), 0,
               I.n_entities, I.entities.p + 2, 4, V->dofmap.p, nd, P.rowmark.p, special.p);
      }
      o += ne;
    }
  }
  bool no_fold = false;
  if (nf_cap > 0 && nd > 4 && V->degree == 2)
  {
    // degree 2: the two cells of a facet share the facet's dofs (6 in 3-D, 3 in 2-D) when the space is continuous
    const int ns = V->mesh->tdim == 3 ? 6 : 3;
    if (nd == 10)
      launch("plan_check_fold", plan_check_fold_kernel<10>, grid_for(nf_cap), dim3(kBlock), 0, P.nfacets, P.facet_rows.p,
             V->dofmap.p, nd - ns, flag.p);
    else if (nd == 6)
      launch("plan_check_fold", plan_check_fold_kernel<6>, grid_for(nf_cap), dim3(kBlock), 0, P.nfacets, P.facet_rows.p,
             V->dofmap.p, nd - ns, flag.p);
    else
      no_fold = true;
  }
  if (P.bulk)
  {
    // the rows next to the interface, from the marks of the cells around them
    const Adjacency& adj = V->dof_cells();
    launch("plan_mix_rows", mix_rowmark_kernel, grid_for(n_rm / 16), dim3(kBlock), 0, V->ndofs, P.rowcls.p, adj.offsets.p,
           adj.cells.p, P.cellmark.p, P.rowmark.p);
  }
  Count n_plain_all;
  // counters of the dof -> facets incidence (zeroed by the kernel that writes the row lists)
  DevArray<int32_t> fcount;
  const bool facets_by_sort = [&]() { const char* fs = getenv("CFX_FACET_SORT"); return fs && fs[0] == '1'; }()
                              && nf_cap * 2 * nd < 2147483647LL;
  {
    const int64_t ntiles = (V->ndofs + kByteTile - 1) / kByteTile;
    DevArray<int64_t> tcounts(ntiles), toffs(ntiles + 1);
    // the row totals and the plan's flag word (every kernel that sets a flag has been launched) in one read-back --
    // or, inside a step, left in HBM: the flag word (what the host branches on) must then repeat the last step's
    const char* names[4] = {"plan.special_rows", "plan.plain_rows", "plan.active_rows", "plan.flags"};
    CountSource src[4];
    src[0].src = toffs.p + ntiles; src[0].kind = kCountLo32;
    src[1].src = toffs.p + ntiles; src[1].kind = kCountHi32;
    src[2].src = toffs.p + ntiles; src[2].kind = kCountSum32;
    src[3].src = flag.p; src[3].kind = kCountI32; src[3].mode = kCountMustEqual;
    Count tot[4];
    CountPlan cp(4, names, src);
    // inside a step the capacities are known before anything is counted: count + offsets + lists in one chained launch
    const ChainState chain = fused_chain(cp.publish, ntiles);
    CountJobs after{};
    if (chain.state) after = cp.take_jobs();
    else
    {
      launch("plan_row_lists", plan_row_lists_count_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, V->ndofs, P.rowmark.p,
             special.p, tcounts.p);
      exclusive_scan(tcounts.p, toffs.p, ntiles, &cp);
    }
    cp.finish(tot);
    plan_flags = (int)tot[3].cap();
    const bool want_plain = space_stencil(V).lists;
    if (getenv("CFX_PLAN_DEBUG"))
      fprintf(stderr, "cutfemx_amd: plan rows special %lld plain %lld of %lld dofs\n", (long long)tot[0].cap(),
              (long long)tot[1].cap(), (long long)V->ndofs);
    P.n_special_rows = tot[0];
    n_plain_all = tot[1];
    P.n_active_rows = tot[2];
    P.n_plain_rows = want_plain ? tot[1] : Count(0);
    // (the active list holds both classes: its capacity is the sum of theirs)
    P.active_rows.alloc(tot[0].cap() + tot[1].cap());
    P.special_rows.alloc(tot[0].cap());
    if (want_plain) P.plain_rows.alloc(tot[1].cap());
    if (nf_cap > 0) P.special_pos.alloc(V->ndofs);
    if (nf_cap > 0 && !facets_by_sort) fcount.alloc(tot[0].cap());
    if (chain.state)
      launch("plan_row_lists", plan_row_lists_chained_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, V->ndofs,
             P.rowmark.p, special.p, P.active_rows.p, P.special_rows.p, want_plain ? P.plain_rows.p : (int32_t*)nullptr,
             tot[0].cap(), tot[1].cap(), nf_cap > 0 ? P.special_pos.p : (int32_t*)nullptr, fcount.p, fcount.n, chain,
             tcounts.p, toffs.p + ntiles, after);
    else
      launch("plan_row_lists", plan_row_lists_write_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, V->ndofs,
             P.rowmark.p, special.p, toffs.p, P.active_rows.p, P.special_rows.p,
             want_plain ? P.plain_rows.p : (int32_t*)nullptr, tot[0].devn(), tot[1].devn(),
             nf_cap > 0 ? P.special_pos.p : (int32_t*)nullptr, fcount.p, fcount.n);
    P.row_tile_counts = std::move(tcounts); // the inactive dofs of a tile are the rest (cfx_active_domain)
  }
  P.special_mark = std::move(special);
  const int64_t ns_cap = P.n_special_rows.cap();
  if (nf_cap > 0)
  {
    // dof -> facets incidence of the rows that have facets (all of them special; special_pos came with the row lists)
    const char* fs = getenv("CFX_FACET_SORT");
    const int64_t npairs = nf_cap * 2 * nd;
    // count + scan + fill with integer atomics combined per workgroup in LDS: 5 launches where the radix sort of
    // rounds 2-3 takes 13 (a kernel boundary costs ~10 us on this chip, profiles/r04_launch_gaps.txt).  CFX_FACET_SORT=1
    // keeps the sort (every list then comes out in ascending facet order without the deterministic mode's list sort).
    (void)fs;
    if (facets_by_sort && ns_cap < 2147483647LL)
    {
      // (lengths still in HBM: the sort covers the capacity of the pair list, pairs behind the last facet carry the
      // sentinel key = the capacity of the special-row list, which no row position reaches)
      DevArray<int32_t> keys(npairs), vals(npairs), keys_out(npairs);
      P.d2f.alloc(npairs); // the sorted values: the entries behind the last offset (sentinel keys) are never read
      launch("facet_dof_pairs", facet_dof_pairs_kernel, grid_for(npairs), dim3(kBlock), 0, P.nfacets, P.facet_rows.p,
             V->dofmap.p, nd, P.special_pos.p, (int32_t)ns_cap, keys.p, vals.p);
      int bits = 1;
      while ((1ll << bits) <= ns_cap) ++bits; // keys 0 .. ns_cap (the sentinel)
      size_t tmp_bytes = 0;
      CFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.p, keys_out.p, vals.p, P.d2f.p, (size_t)npairs, 0, bits,
                                        ctx().stream));
      DevArray<uint8_t> tmp((int64_t)tmp_bytes);
      CFX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.p, keys_out.p, vals.p, P.d2f.p, (size_t)npairs, 0, bits,
                                        ctx().stream));
      P.d2f_offsets.alloc(ns_cap + 1);
      launch("facet_dof_offsets", sorted_key_offsets_kernel, grid_for(ns_cap + 1), dim3(kBlock), 0,
             P.n_special_rows, npairs, keys_out.p, P.d2f_offsets.p);
      P.d2f_sorted = true;
    }
    else
    {
      // (lengths may still be in HBM: the counters cover the capacity of the special-row list -- the tail stays zero --,
      // the facet list is sized by its upper bound, one entry per (facet, dof of its two cells) pair; the fill reuses
      // the counters as cursors counting DOWN, so that they need no second zero fill)
      if (fcount.n != ns_cap || !fcount.p) { fcount.alloc(ns_cap); fcount.zero(); }
      require(nd <= kFacetMaxNd, CFX_ERR_RUNTIME, "dof -> facets incidence: more than 10 dofs per cell");
      const int per = facet_units_per_block(nd);
      const dim3 run_grid((unsigned)((2 * nf_cap + per - 1) / per));
      launch("facet_dof_count", facet_dof_count_kernel, run_grid, dim3(kBlock), 0, P.nfacets,
             P.facet_rows.p, V->dofmap.p, nd, P.special_pos.p, fcount.p, ns_cap, step_poison());
      P.d2f_offsets.alloc(ns_cap + 1);
      exclusive_scan(fcount.p, P.d2f_offsets.p, ns_cap);
      P.d2f.alloc(npairs);
      launch("facet_dof_fill", facet_dof_fill_kernel, run_grid, dim3(kBlock), 0, P.nfacets,
             P.facet_rows.p, V->dofmap.p, nd, P.special_pos.p, P.d2f_offsets.p, fcount.p, P.d2f.p, ns_cap, step_poison());
    }
  }
  // rank structure of every uncut entity list: entity index of cell c =
  // rank[c/64] + popcount(bits[c/64] below c), two cached loads instead of a
  // binary search over the (10^8-entry) list
  const int64_t nwords = (nc + 63) / 64;
  for (int slot = 0; slot < P.n_cell_slots; ++slot)
  {
    const cfx_integral_dev& I = a->integrals[P.cell_slot_integral[slot]];
    if (I.n_entities.cap() == 0) continue;
    P.std_rank[slot].alloc(nwords + 1);
    if (P.bulk && slot == bulk_packed_slot)
    {
      // (packed by plan_bulk_init; the marked cells per tile -- the count pass of the active-cell list -- are counted
      // when that list is asked for)
      exclusive_scan(bulk_pop.p, P.std_rank[slot].p, nwords);
      continue;
    }
    P.std_bits[slot].alloc(nwords);
    DevArray<int32_t> pop(nwords);
    int32_t* tiles = nullptr;
    if (!P.bulk && P.cell_tile_counts.n == 0)
    {
      P.cell_tile_counts.alloc((nc + kByteTile - 1) / kByteTile);
      tiles = P.cell_tile_counts.p;
    }
    if (P.bulk)
    {
      // (the entities are the cells of one domain value: no pass over the mark bytes)
      const int64_t nb = (nc + kClassBlock - 1) / kClassBlock;
      launch("plan_pack_bits", pack_bits_domain_kernel, grid_for(nwords), dim3(kBlock), 0, nc, bulk_cut->domain.p, (int8_t)bulk_value,
             bulk_cut->block_class.n == nb ? bulk_cut->block_class.p : (const uint8_t*)nullptr,
             reinterpret_cast<unsigned long long*>(P.std_bits[slot].p), pop.p, step_poison());
    }
    else
      launch("plan_pack_bits", plan_pack_bits_kernel, grid_for(nwords), dim3(kBlock), 0, nc, P.cellmark.p,
             (uint8_t)(1u << slot), reinterpret_cast<unsigned long long*>(P.std_bits[slot].p), pop.p, tiles);
    exclusive_scan(pop.p, P.std_rank[slot].p, nwords);
  }
  const char* det = getenv("CFX_DETERMINISTIC");
  if (nf_cap > 0 && det && det[0] == '1' && !P.d2f_sorted)
  {
    // reproducible gather order (the lists were filled through an atomic cursor)
    launch("plan_sort_d2f", seg_sort_kernel, grid_for(ns_cap), dim3(kBlock), 0, P.n_special_rows,
           P.d2f_offsets.p, P.d2f.p);
  }
  // unsorted / repeated caller-supplied entity lists: the gather path cannot look them up
  const int flags = plan_flags;
  if (flags & 1) P.usable = false;
  P.fold_ok = !(flags & 2) && !no_fold;
  P.built = true;
  publish_across_lanes(); // a second form on the other lane may adopt this plan
  return P;
}

const Stencil& space_stencil(cfx_space_s* V)
{
  Stencil& S = V->stencil;
  if (S.built) return S;
  S.built = true;
  const char* env = getenv("CFX_STENCIL");
  if (env && env[0] == '0') return S;
  // P1 scalar space on the geometry dofmap (dofs are mesh vertices): lists + slots; every other space (degree 2,
  // vector-valued, DG): the neighbour lists alone, which is what the sparsity of the plain rows needs
  if (V->degree != 1 || V->bs != 1 || V->ndofs_cell > 4 || V->dofmap.p != V->mesh->conn.p || V->ndofs != V->mesh->nnodes)
  {
    const char* le = getenv("CFX_STENCIL_LISTS");
    if (le && le[0] == '0') return S;
    const Adjacency& adj = V->dof_cells();
    PatArgs A{};
    A.n_active = V->ndofs; A.active_rows = nullptr; A.all_cells = 1;
    A.nd = V->ndofs_cell; A.bs = 1; A.dofmap = V->dofmap.p; // lists of scalar dofs: a block space expands them by bs
    A.d2c_off = adj.offsets.p; A.d2c = adj.cells.p;
    DevArray<int32_t> counts(V->ndofs), len(V->ndofs);
    ZeroFlag overflow, maxlen;
    A.len = len.p; A.counts = counts.p; A.overflow = overflow.p; A.maxlen = maxlen.p;
    launch("stencil_rows", pattern_rows_kernel<64, 512>, wave_grid(V->ndofs), dim3(kWave), 0, A);
    if (read_scalar(overflow.p)) return S;
    S.max_len = read_scalar(maxlen.p);
    S.offsets.alloc(V->ndofs + 1);
    exclusive_scan(counts.p, S.offsets.p, V->ndofs);
    S.nbr.alloc(read_scalar(S.offsets.p + V->ndofs));
    A.indptr = S.offsets.p; A.indices = S.nbr.p;
    launch("stencil_rows_write", pattern_rows_kernel<64, 512>, wave_grid(V->ndofs), dim3(kWave), 0, A);
    S.lists = true;
    publish_across_lanes();
    return S;
  }
  const Adjacency& adj = V->dof_cells();
  PatArgs A{};
  A.n_active = V->ndofs; A.active_rows = nullptr; A.all_cells = 1;
  A.nd = V->ndofs_cell; A.bs = 1; A.dofmap = V->dofmap.p;
  A.d2c_off = adj.offsets.p; A.d2c = adj.cells.p;
  DevArray<int32_t> counts(V->ndofs), len(V->ndofs);
  ZeroFlag overflow, maxlen;
  A.len = len.p; A.counts = counts.p; A.overflow = overflow.p; A.maxlen = maxlen.p;
  // One pass when the card has room for ndofs x 64 staged columns (34 GB at 512^3, released right after): every set is
  // ranked into its staging row and copied once the offsets are known; else count, scan, build every set again and
  // write it in place (30 + 31 ms at 512^3 against 30 + 7).
  DevArray<int32_t> staged;
  {
    size_t free_b = 0, total_b = 0;
    CFX_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t live_b = 0, cached_b = 0, peak_b = 0;
    device_memory_stats(live_b, cached_b, peak_b);
    const size_t need = (size_t)V->ndofs * 64 * sizeof(int32_t);
    const char* sv = getenv("CFX_STENCIL_STAGED");
    if (need < (free_b + cached_b) / 3 && !(sv && sv[0] == '0')) { staged.alloc(V->ndofs * 64); A.tmp = staged.p; }
  }
  launch("stencil_rows", pattern_rows_kernel<4, 64>, wave_grid((V->ndofs + 15) / 16), dim3(kWave), 0, A);
  if (read_scalar(overflow.p)) return S; // a vertex with more than 63 neighbours: keep the hashed paths
  S.max_len = read_scalar(maxlen.p);
  S.offsets.alloc(V->ndofs + 1);
  exclusive_scan(counts.p, S.offsets.p, V->ndofs);
  S.nbr.alloc(read_scalar(S.offsets.p + V->ndofs));
  if (staged.p)
  {
    launch("stencil_rows_write", pattern_write_kernel<64>, grid_for(V->ndofs * 8), dim3(kBlock), 0, DevN(V->ndofs),
           (const int32_t*)nullptr, 1, staged.p, len.p, S.offsets.p, S.nbr.p);
    // the staging block becomes the stencil's arena (Stencil::arena): the tables below are carved from it
    S.arena.p = reinterpret_cast<uint8_t*>(staged.p); S.arena.n = staged.n * (int64_t)sizeof(int32_t); S.arena.owned = staged.owned;
    staged.p = nullptr; staged.n = 0; staged.owned = false;
    S.arena_used = 0;
  }
  else
  {
    A.indptr = S.offsets.p; A.indices = S.nbr.p;
    launch("stencil_rows_write", pattern_rows_kernel<4, 64>, wave_grid((V->ndofs + 15) / 16), dim3(kWave), 0, A);
  }
  S.take(S.slot4, adj.cells.n);
  S.take(S.diagpos, V->ndofs);
  S.take(S.cpos, V->mesh->ncells * (int64_t)V->ndofs_cell);
  launch("stencil_slots", stencil_slots_kernel, wave_grid((V->ndofs + 15) / 16), dim3(kWave), 0, V->ndofs, adj.offsets.p,
         adj.cells.p, V->dofmap.p, V->ndofs_cell, S.offsets.p, S.nbr.p, S.slot4.p, S.diagpos.p, S.cpos.p);
  S.usable = true;
  S.lists = true;
  publish_across_lanes();
  return S;
}

const Stencil& space_stencil_tiles(cfx_space_s* V)
{
  Stencil& S = const_cast<Stencil&>(space_stencil(V));
  if (S.tiles_built || !S.usable) return S;
  S.tiles_built = true;
  const char* env = getenv("CFX_TILES");
  if (env && env[0] == '0') return S;
  const Adjacency& adj = V->dof_cells();
  const int64_t ntiles = (V->ndofs + kRowTile - 1) / kRowTile;
  DevArray<int32_t> counts(ntiles);
  DevArray<int> maxima(3);
  maxima.zero();
  const dim3 grid = wave_grid(ntiles);
  // one pass when there is room for kTileStage staged vertices per tile (8.6 GB at 512^3: the tail of the stencil's
  // arena, else a block of its own) and no tile's union outgrows its staging row; else count, scan and build every
  // tile a second time (42 + 52 ms)
  DevArray<int32_t> staged;
  {
    const int64_t need = ntiles * kTileStage * (int64_t)sizeof(int32_t);
    const int64_t perm = (((int64_t)sizeof(uint16_t) * S.nbr.n + 255) & ~255LL);     // st_loc goes in front of it
    const char* sv = getenv("CFX_STENCIL_STAGED");
    const bool on = !(sv && sv[0] == '0');
    if (on && S.arena.p && S.arena_used + perm + need <= S.arena.n)
    {
      // (the tail of the arena; the permanent tables grow from the front and are checked against it below)
      staged.p = reinterpret_cast<int32_t*>(S.arena.p + ((S.arena.n - need) & ~255LL)); staged.n = ntiles * kTileStage; staged.owned = false;
    }
    else if (on)
    {
      size_t free_b = 0, total_b = 0;
      CFX_HIP(hipMemGetInfo(&free_b, &total_b));
      size_t live_b = 0, cached_b = 0, peak_b = 0;
      device_memory_stats(live_b, cached_b, peak_b);
      if ((size_t)(need + perm) < (free_b + cached_b) / 3) staged.alloc(ntiles * kTileStage);
    }
  }
  const uint8_t* tail = (staged.p && !staged.owned) ? reinterpret_cast<const uint8_t*>(staged.p) : nullptr; // arena tail in use
  if (staged.p)
  {
    S.take(S.st_loc, S.nbr.n);
    launch("stencil_tiles", stencil_tiles_kernel<2>, grid, dim3(kWave), 0, V->ndofs, ntiles, S.offsets.p, S.nbr.p,
           adj.offsets.p, counts.p, (const int64_t*)nullptr, staged.p, S.st_loc.p, maxima.p);
  }
  else
    launch("stencil_tiles", stencil_tiles_kernel<0>, grid, dim3(kWave), 0, V->ndofs, ntiles, S.offsets.p, S.nbr.p,
           adj.offsets.p, counts.p, (const int64_t*)nullptr, (int32_t*)nullptr, (uint16_t*)nullptr, maxima.p);
  const std::vector<int> mx = download(maxima.p, 3);
  S.max_tile_verts = mx[0]; S.max_tile_st = mx[1]; S.max_tile_items = mx[2];
  if (S.max_tile_st > kTileMaxSt) { staged.release(); return S; }
  S.tile_voff.alloc(ntiles + 1);
  exclusive_scan(counts.p, S.tile_voff.p, ntiles);
  const int64_t n_verts = read_scalar(S.tile_voff.p + ntiles);
  // (tile_verts from the arena only if it ends below the scratch in the arena's tail)
  if (tail && S.arena.p + S.arena_used + (((int64_t)sizeof(int32_t) * n_verts + 255) & ~255LL) > tail) S.tile_verts.alloc(n_verts);
  else S.take(S.tile_verts, n_verts);
  if (staged.p && S.max_tile_verts <= kTileStage)
  {
    launch("stencil_tiles_write", stencil_tiles_pack_kernel, grid_for(ntiles * kWave), dim3(kBlock), 0, ntiles, counts.p,
           S.tile_voff.p, staged.p, S.tile_verts.p);
    staged.release();
  }
  else
  {
    staged.release();
    if (!S.st_loc.p) S.take(S.st_loc, S.nbr.n);
    launch("stencil_tiles_write", stencil_tiles_kernel<1>, grid, dim3(kWave), 0, V->ndofs, ntiles, S.offsets.p, S.nbr.p,
           adj.offsets.p, (int32_t*)nullptr, S.tile_voff.p, S.tile_verts.p, S.st_loc.p, maxima.p);
  }
  S.tiles_usable = true;
  publish_across_lanes();
  return S;
}

bool space_dof_verts(cfx_space_s* V)
{
  if (V->dof_verts_built) return V->dof_verts_ok;
  V->dof_verts_built = true;
  const int tdim = V->mesh->tdim, nd = tdim == 3 ? 10 : 6;
  if (V->degree != 2 || V->ndofs_cell != nd || (tdim != 2 && tdim != 3)) return false;
  V->dof_verts.alloc(2 * V->ndofs);
  dev_fill(V->dof_verts.p, 0xff, sizeof(int32_t) * 2 * (size_t)V->ndofs); // (-1: a dof no cell holds)
  launch("space_dof_verts", dof_verts_kernel, grid_for(V->mesh->ncells * nd), dim3(kBlock), 0, V->mesh->ncells, tdim,
         V->mesh->conn.p, V->dofmap.p, V->dof_verts.p);
  V->dof_verts_ok = true;
  publish_across_lanes();
  return true;
}

const Stencil& space_stencil_slotn(cfx_space_s* V)
{
  Stencil& S = const_cast<Stencil&>(space_stencil(V));
  if (S.slotn_built) return S;
  S.slotn_built = true;
  const char* env = getenv("CFX_P2_PLAIN");
  if ((env && env[0] == '0') || !S.lists || S.usable || V->degree != 2 || V->ndofs_cell > 10 || S.max_len > 255)
    return S;
  const Adjacency& adj = V->dof_cells();
  S.slotn.alloc(adj.cells.n * 12);
  S.slotn.zero();
  launch("stencil_slotn", stencil_slotn_kernel, wave_grid((V->ndofs + 3) / 4), dim3(kWave), 0, V->ndofs, adj.offsets.p,
         adj.cells.p, V->dofmap.p, V->ndofs_cell, S.offsets.p, S.nbr.p, S.slotn.p);
  S.slotn_ok = true;
  publish_across_lanes();
  return S;
}

bool plain_row_masks(cfx_form_s* a, int32_t* counts, int* maxlen)
{
  cfx_row_plan& plan = row_plan(a);
  if (plan.plain_masks_built) return false;
  plan.plain_masks_built = true;
  cfx_space_s* V = a->V;
  const Stencil& st = space_stencil(V);
  const int64_t np = plan.n_plain_rows.cap(); // (capacity of the list while its length is in HBM)
  if (!st.usable || np == 0 || !plan.any_cells) return false;
  const Adjacency& adj = V->dof_cells();
  plan.plain_masks.alloc(np);
  plan.plain_uniform.alloc(np);
  launch("plan_plain_masks", plain_masks_kernel,
         dim3((unsigned)((np + kWave - 1) / kWave)), dim3(kWave), 0,
         plan.n_plain_rows, plan.plain_rows.p, adj.offsets.p, adj.cells.p, st.slot4.p, plan.cellmark.p, V->ndofs_cell,
         st.offsets.p, plan.plain_masks.p, plan.plain_uniform.p, counts, maxlen, plan.bulk ? plan.rowcls.p : (const uint8_t*)nullptr,
         plan.bulk_bits);
  if (space_stencil_tiles(V).tiles_usable)
  {
    // work list of the tile kernels: one entry per row tile that holds a plain row (inside a step: positions and tile
    // numbers from one chained launch; else the numbers follow the compaction's read-back)
    DevArray<int32_t> first;
    TileIdEmit emit{plan.plain_rows.p, nullptr};
    bool emitted = false;
    auto pre = [&](int64_t cap) { plan.plain_tile_id.alloc(cap); emit.ids = plan.plain_tile_id.p; emitted = true; };
    plan.n_plain_tiles = compact_count("plan_plain_tiles", "plan.plain_tiles", plan.n_plain_rows.devn(),
                                       TileStart{plan.plain_rows.p}, first, &emit, pre);
    plan.plain_tile_first = std::move(first);
    if (!emitted)
    {
      plan.plain_tile_id.alloc(plan.n_plain_tiles.cap());
      launch("plan_plain_tiles", tile_ids_kernel, grid_for(plan.n_plain_tiles.cap()), dim3(kBlock), 0, plan.n_plain_tiles,
             plan.plain_tile_first.p, plan.plain_rows.p, plan.plain_tile_id.p);
    }
  }
  publish_across_lanes(); // the masks belong to the plan, which the other lane's form may share
  return counts != nullptr;
}

// first rule of every cut cell: rule e opens a new parent -> slot of that parent in the cut-cell list
__global__ void __launch_bounds__(kBlock) plan_cut_first_kernel(int64_t nr, const int32_t* __restrict__ parent,
                                                                const unsigned long long* __restrict__ bits,
                                                                const int32_t* __restrict__ rank, int32_t* __restrict__ first)
{
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nr) return;
  const int32_t c = parent[e];
  if (e > 0 && parent[e - 1] == c) return;
  const int64_t w = c >> 6;
  first[(int64_t)rank[w] + __popcll(bits[w] & ((1ull << (c & 63)) - 1ull))] = (int32_t)e;
}

struct ByteRuleMark
{
  __device__ bool operator()(uint8_t v) const { return (v & 0xF0u) != 0; }
};

void plan_cut_cells(cfx_form_s* a)
{
  cfx_row_plan& plan = row_plan(a);
  if (plan.cut_cells_built) return;
  plan.cut_cells_built = true;
  const int64_t nc = a->V->mesh->ncells;
  plan.n_cut_cells = compact_bytes("plan_cut_cells", nc, plan.cellmark.p, ByteRuleMark{}, plan.cut_cells);
  const int64_t n_cut_cells = plan.n_cut_cells;
  const int64_t nwords = (nc + 63) / 64;
  plan.cut_bits.alloc(nwords);
  plan.cut_rank.alloc(nwords + 1);
  DevArray<int32_t> pop(nwords);
  launch("plan_pack_bits", plan_pack_bits_kernel, grid_for(nwords), dim3(kBlock), 0, nc, plan.cellmark.p, (uint8_t)0xF0u,
         reinterpret_cast<unsigned long long*>(plan.cut_bits.p), pop.p, (int32_t*)nullptr);
  exclusive_scan(pop.p, plan.cut_rank.p, nwords);
  for (int slot = 0; slot < plan.n_cell_slots; ++slot)
  {
    const cfx_integral_dev& I = a->integrals[plan.cell_slot_integral[slot]];
    const int64_t nrl = I.rules ? I.rules->nr.value() : 0;
    if (nrl == 0 || n_cut_cells == 0) continue;
    plan.cut_first[slot].alloc(n_cut_cells);
    dev_fill(plan.cut_first[slot].p, 0xff, sizeof(int32_t) * (size_t)n_cut_cells);
    launch("plan_cut_cells", plan_cut_first_kernel, grid_for(nrl), dim3(kBlock), 0, nrl, I.rules->parent_map.p,
           reinterpret_cast<const unsigned long long*>(plan.cut_bits.p), plan.cut_rank.p, plan.cut_first[slot].p);
  }
  publish_across_lanes();
}

// lengths of the dof->cells lists of the plain rows whose incident cells all carry `mark` (0 for the others:
// rows at the edge of a restricted entity list, e.g. a rank's owned cells, keep the per-cell records)
// (packed for one scan: segment entries in the low 32 bits, "this row has no segment" in the high ones)
__global__ void vec_plain_len_kernel(DevN n_plain_d, const int32_t* __restrict__ rows, const int64_t* __restrict__ d2c_off,
                                     const uint8_t* __restrict__ uniform, uint8_t mark, int64_t* __restrict__ len)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_plain)
  {
    if (i < n_plain_d.cap) len[i] = 0; // (the scan runs over the capacity of the list)
    return;
  }
  const int64_t r = rows[i];
  const int64_t l = uniform[i] == mark ? d2c_off[r + 1] - d2c_off[r] : 0;
  len[i] = l > 0 ? l : (1ll << 32);
}

__global__ void vec_plain_scatter_kernel(DevN n_plain_d, const int32_t* __restrict__ rows, const int64_t* __restrict__ off,
                                         int32_t* __restrict__ t2off)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (entries in the low 32 bits of the packed prefix)
  if (i < n_plain && (off[i + 1] & 0xffffffffll) > (off[i] & 0xffffffffll)) t2off[rows[i]] = (int32_t)(off[i] & 0xffffffffll) + 1;
}


// the two kernels above and the scan between them in ONE launch (tiles chained by look-back, cfx_device.h): inside a
// sync-free step, where nothing is sized by the totals before they are published.  Eight consecutive plain rows per thread.
__global__ void __launch_bounds__(kBlock) vec_plain_offsets_chained_kernel(DevN n_plain_d, const int32_t* __restrict__ rows,
                                                                           const int64_t* __restrict__ d2c_off,
                                                                           const uint8_t* __restrict__ uniform, uint8_t mark,
                                                                           int32_t* __restrict__ t2off, ChainState chain,
                                                                           int64_t* __restrict__ total_out, CountJobs after)
{
  const int64_t n_plain = dev_n(n_plain_d);
  const unsigned int tile = chain_take_tile(chain.ticket);
  const int64_t base = (int64_t)tile * kTile + (int64_t)threadIdx.x * kScanItems;
  int32_t r[kScanItems];
  int64_t l[kScanItems], s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    const int64_t i = base + k;
    r[k] = 0; l[k] = 0;
    if (i < n_plain)
    {
      r[k] = rows[i];
      const int64_t len = uniform[i] == mark ? d2c_off[(int64_t)r[k] + 1] - d2c_off[r[k]] : 0;
      l[k] = len > 0 ? len : (1ll << 32);
    }
    s += l[k];
  }
  int64_t total;
  const int64_t local = block_exclusive_scan<int64_t>(s, total);
  const int64_t prefix = (int64_t)chain_exclusive_prefix(chain.state, tile, (unsigned long long)total);
  int64_t off = prefix + local;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
  {
    // (entries in the low 32 bits of the packed prefix)
    if ((l[k] & 0xffffffffll) > 0) t2off[r[k]] = (int32_t)(off & 0xffffffffll) + 1;
    off += l[k];
  }
  if (tile == gridDim.x - 1 && threadIdx.x == kBlock - 1)
  {
    *total_out = prefix + total;
    if (after.n > 0) count_publish(after);
  }
}

__global__ void gather_i32_kernel(DevN n_d, const int32_t* __restrict__ idx, const int32_t* __restrict__ src,
                                  int32_t* __restrict__ dst)
{
  const int64_t n = dev_n(n_d);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// Layout of the row-ordered staging of a linear form's uncut-cell element vectors (see cfx_row_plan::vec_t2off).
// A plain row takes part when all its incident cells are uncut entities of the one integral whose mark bit is
// `mark` (always so for the volume terms of a single-level-set problem: a vertex without a cut cell around it
// has only inside cells around it; not so at the edge of a restricted entity list).  False: no such row, the
// caller keeps the per-cell staging for all rows.
bool plain_vec_offsets(cfx_form_s* L, uint8_t mark)
{
  cfx_row_plan& plan = row_plan(L);
  if (plan.vec_fast >= 0 && plan.vec_mark == mark) return plan.vec_fast == 1;
  plan.vec_mark = mark;
  plan.vec_fast = 0;
  cfx_space_s* V = L->V;
  const Stencil& st = space_stencil(V);
  const int64_t n = plan.n_plain_rows.cap(); // (capacity while the length is in HBM)
  if (!st.usable || n == 0 || !plan.any_cells) return false;
  plain_row_masks(L);
  if (plan.plain_uniform.n != n) return false;
  const Adjacency& adj = V->dof_cells();
  DevArray<int64_t> len, off(n + 1);
  // one scan: the entries of all segments, and the plain rows without one.  Inside a step both totals stay in HBM
  // (published by the scan); "no such row" / "too many entries" are then the recorded step's answers
  Count tot[2];
  const char* names[2] = {"vec.segment_entries", "vec.odd_rows"};
  CountSource src[2];
  src[0].src = off.p + n; src[0].kind = kCountLo32;
  src[1].src = off.p + n; src[1].kind = kCountHi32;
  CountPlan cp(2, names, src);
  // (inside a step nothing waits for the totals: lengths + offsets + scatter in one chained launch)
  const int64_t ntiles = (n + kTile - 1) / kTile;
  // (... when the recorded step took this layout: else the host leaves below, before any kernel that could publish)
  const bool fast_before = cp.publish && cp.counts[0].cap() > 0 && cp.counts[0].cap() < 2147483647LL;
  const ChainState chain = fused_chain(fast_before, ntiles);
  CountJobs after{};
  if (chain.state) after = cp.take_jobs();
  else
  {
    len.alloc(n);
    launch("vec_plain_offsets", vec_plain_len_kernel, grid_for(n), dim3(kBlock), 0, plan.n_plain_rows, plan.plain_rows.p,
           adj.offsets.p, plan.plain_uniform.p, mark, len.p);
    exclusive_scan(len.p, off.p, n, &cp);
  }
  cp.finish(tot);
  if (tot[0].cap() == 0 || tot[0].cap() >= 2147483647LL) return false;
  // segment offsets are stored + 1 (0: the row has no segment): the array came zeroed with the plan's mark block
  // (cfx::row_plan: one fill for all of them) unless this is a second layout of the same plan
  if (plan.vec_t2off.n != V->ndofs || plan.vec_t2off_used)
  {
    plan.vec_t2off.alloc(V->ndofs);
    dev_fill(plan.vec_t2off.p, 0, sizeof(int32_t) * (size_t)V->ndofs);
  }
  plan.vec_t2off_used = true;
  if (chain.state)
    launch("vec_plain_offsets", vec_plain_offsets_chained_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, plan.n_plain_rows,
           plan.plain_rows.p, adj.offsets.p, plan.plain_uniform.p, mark, plan.vec_t2off.p, chain, off.p + n, after);
  else
    launch("vec_plain_offsets", vec_plain_scatter_kernel, grid_for(n), dim3(kBlock), 0, plan.n_plain_rows, plan.plain_rows.p,
           off.p, plan.vec_t2off.p);
  // everything else reads the per-cell records: the special rows, and (a second pass that skips the rows with a
  // segment) the few plain rows whose cells do not all carry the mark
  plan.n_vec_odd_rows = tot[1];
  plan.vec_t2_total = tot[0];
  plan.vec_fast = 1;
  publish_across_lanes();
  return true;
}

// ---------------------------------------------------------------------------
// Cell blocks (VecBlocks): one workgroup per block sorts the block's (dof, entry) pairs in LDS (bitonic, up to
// kVbEntries keys); the sorted position of an entry is its slot, the first position of every dof its segment.
// Pass 1 counts the dofs of each union, pass 2 writes slot / seg / the union itself.
// ---------------------------------------------------------------------------
template <bool WRITE>
__global__ void __launch_bounds__(kBlock) vec_blocks_kernel(int64_t ncells, int nd, int B, const int32_t* __restrict__ dofmap,
                                                            int32_t* __restrict__ counts, const int64_t* __restrict__ u_off,
                                                            int32_t* __restrict__ u_dofs, uint16_t* __restrict__ slot,
                                                            uint16_t* __restrict__ seg)
{
  __shared__ unsigned long long s_key[kVbEntries];
  const int tid = threadIdx.x;
  const int64_t k = blockIdx.x, c0 = k * B;
  const int nb = (int)min((int64_t)B, ncells - c0), n = nb * nd;
  int N = kBlock; // keys sorted: a power of two >= n (and >= the block, every thread owns N / kBlock of them)
  while (N < n) N <<= 1;
  for (int e = tid; e < N; e += kBlock)
    s_key[e] = e < n ? (((unsigned long long)(uint32_t)dofmap[c0 * nd + e] << 11) | (unsigned)e) : ~0ull;
  __syncthreads();
  for (int size = 2; size <= N; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1)
    {
      for (int i = tid; i < N / 2; i += kBlock)
      {
        const int pos = 2 * i - (i & (stride - 1)), other = pos + stride;
        const unsigned long long a = s_key[pos], b = s_key[other];
        const bool up = (pos & size) == 0;
        if ((a > b) == up) { s_key[pos] = b; s_key[other] = a; }
      }
      __syncthreads();
    }
  // heads of the runs of equal dofs; thread t owns the sorted positions [t E, (t + 1) E)
  const int E = N / kBlock;
  int heads = 0;
  for (int q = 0; q < E; ++q)
  {
    const int e = tid * E + q;
    if (e < n && (e == 0 || (s_key[e] >> 11) != (s_key[e - 1] >> 11))) ++heads;
  }
  int total;
  int t = block_exclusive_scan<int>(heads, total);
  if constexpr (!WRITE)
  {
    if (tid == 0) counts[k] = total;
  }
  else
  {
    const int64_t ub = u_off[k];
    for (int q = 0; q < E; ++q)
    {
      const int e = tid * E + q;
      if (e >= n) break;
      const unsigned long long key = s_key[e];
      slot[c0 * nd + (int)(key & 2047u)] = (uint16_t)e;
      if (e == 0 || (key >> 11) != (s_key[e - 1] >> 11))
      {
        seg[ub + t] = (uint16_t)e;
        u_dofs[ub + t] = (int32_t)(key >> 11);
        ++t;
      }
    }
  }
}

__global__ void vb_count_kernel(int64_t n, const int32_t* __restrict__ u_dofs, int32_t* __restrict__ cnt)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&cnt[u_dofs[i]], 1);
}

__global__ void __launch_bounds__(kBlock) vb_fill_kernel(const int64_t* __restrict__ u_off, const int32_t* __restrict__ u_dofs,
                                                         const int64_t* __restrict__ p_off, int32_t* __restrict__ cursor,
                                                         int64_t* __restrict__ p_pos)
{
  const int64_t k = blockIdx.x, ub = u_off[k];
  const int nu = (int)(u_off[k + 1] - ub);
  for (int t = threadIdx.x; t < nu; t += kBlock)
  {
    const int32_t dof = u_dofs[ub + t];
    p_pos[p_off[dof] + atomicAdd(&cursor[dof], 1)] = (k << 11) | t;
  }
}

// the pairs of a dof in ascending block order (the fill above is in arrival order); the lists are a handful long
__global__ void vb_sort_kernel(int64_t ndofs, const int64_t* __restrict__ p_off, int64_t* __restrict__ p_pos)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ndofs) return;
  const int64_t b = p_off[r], e = p_off[r + 1];
  for (int64_t i = b + 1; i < e; ++i)
  {
    const int64_t v = p_pos[i];
    int64_t j = i;
    while (j > b && p_pos[j - 1] > v) { p_pos[j] = p_pos[j - 1]; --j; }
    p_pos[j] = v;
  }
}

const VecBlocks& space_vec_blocks(cfx_space_s* V)
{
  VecBlocks& S = V->vblocks;
  if (S.built) return S;
  S.built = true;
  const char* env = getenv("CFX_VEC_BLOCKS");
  if (env && env[0] == '0') return S;
  const int nd = V->ndofs_cell;
  const int64_t nc = V->mesh->ncells;
  if (V->bs != 1 || nc == 0) return S;
  int B = 256;
  while (B * nd > kVbEntries) B >>= 1;
  if (B < 32) return S;
  S.B = B;
  S.nblocks = (nc + B - 1) / B;
  if (S.nblocks > 0x7fffffffLL) return S;
  DevArray<int32_t> counts(S.nblocks);
  launch("vec_blocks", vec_blocks_kernel<false>, dim3((unsigned)S.nblocks), dim3(kBlock), 0, nc, nd, B, V->dofmap.p, counts.p,
         (const int64_t*)nullptr, (int32_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr);
  S.u_off.alloc(S.nblocks + 1);
  exclusive_scan(counts.p, S.u_off.p, S.nblocks);
  S.u_total = read_scalar(S.u_off.p + S.nblocks);
  DevArray<int32_t> u_dofs(S.u_total);
  S.slot.alloc(nc * nd);
  S.seg.alloc(S.u_total);
  launch("vec_blocks_write", vec_blocks_kernel<true>, dim3((unsigned)S.nblocks), dim3(kBlock), 0, nc, nd, B, V->dofmap.p,
         (int32_t*)nullptr, S.u_off.p, u_dofs.p, S.slot.p, S.seg.p);
  // dof -> (block, position in the union) pairs: count, scan, fill, order
  DevArray<int32_t> cnt(V->ndofs);
  dev_fill(cnt.p, 0, sizeof(int32_t) * (size_t)V->ndofs);
  launch("vec_blocks_lists", vb_count_kernel, grid_for(S.u_total), dim3(kBlock), 0, S.u_total, u_dofs.p, cnt.p);
  S.p_off.alloc(V->ndofs + 1);
  exclusive_scan(cnt.p, S.p_off.p, V->ndofs);
  S.p_pos.alloc(S.u_total);
  dev_fill(cnt.p, 0, sizeof(int32_t) * (size_t)V->ndofs);
  launch("vec_blocks_lists", vb_fill_kernel, dim3((unsigned)S.nblocks), dim3(kBlock), 0, S.u_off.p, u_dofs.p, S.p_off.p, cnt.p,
         S.p_pos.p);
  launch("vec_blocks_lists", vb_sort_kernel, grid_for(V->ndofs), dim3(kBlock), 0, V->ndofs, S.p_off.p, S.p_pos.p);
  S.usable = true;
  publish_across_lanes();
  return S;
}

// per block: the size of its union if one of its cells carries a bit of `mark`, else 0, and which kinds of cells it
// holds (bit 0: uncut entities, mark & 0x0F; bit 1: parents of runtime rules, mark & 0xF0); one wavefront per block
__global__ void __launch_bounds__(kBlock) vb_flag_kernel(int64_t nblocks, int B, int64_t ncells, const uint8_t* __restrict__ cellmark,
                                                         uint8_t mark, const int64_t* __restrict__ u_off, int32_t* __restrict__ len,
                                                         uint8_t* __restrict__ kind)
{
  const int lane = threadIdx.x & 63;
  const int64_t k = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (k >= nblocks) return;
  const int64_t c0 = k * B;
  uint32_t acc = 0;
  for (int o = lane * 4; o < B; o += kWave * 4) // B is a multiple of 32 and the cell array of a block 4 B aligned
  {
    uint32_t w = 0;
    if (c0 + o + 4 <= ncells) w = *reinterpret_cast<const uint32_t*>(cellmark + c0 + o);
    else
      for (int q = 0; q < 4; ++q)
        if (c0 + o + q < ncells) w |= (uint32_t)cellmark[c0 + o + q] << (8 * q);
    acc |= w & (0x01010101u * mark);
  }
  const bool has_std = __ballot((acc & 0x0F0F0F0Fu) != 0) != 0, has_cut = __ballot((acc & 0xF0F0F0F0u) != 0) != 0;
  if (lane == 0)
  {
    len[k] = (has_std || has_cut) ? (int32_t)(u_off[k + 1] - u_off[k]) : 0;
    kind[k] = (uint8_t)((has_std ? 1 : 0) | (has_cut ? 2 : 0));
  }
}

// base[k] = -1 for the blocks without partials; a block of the cut list that holds no uncut entity gets bit 31 (its
// partials are stored, not added to the ones of the uncut cells)
__global__ void vb_base_kernel(int64_t nblocks, const int32_t* __restrict__ len, int64_t* __restrict__ base, int64_t n_cut,
                               int32_t* __restrict__ cut_list, const uint8_t* __restrict__ kind)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nblocks && len[k] == 0) base[k] = -1;
  if (k < n_cut && !(kind[cut_list[k]] & 1)) cut_list[k] |= (int32_t)0x80000000;
}

struct KindBit
{
  const uint8_t* kind;
  uint8_t bit;
  __device__ bool operator()(int64_t i) const { return (kind[i] & bit) != 0; }
};

// The blocks a linear form runs over this step (uncut entities with mark bits `mark & 0x0F`, rule parents with
// `mark & 0xF0`) and where their partials go
bool vec_block_plan(cfx_form_s* L, uint8_t mark, bool merged)
{
  cfx_row_plan& plan = row_plan(L);
  if (plan.vb_state >= 0 && plan.vb_mark == mark && plan.vb_merged == merged) return plan.vb_state == 1;
  plan.vb_mark = mark;
  plan.vb_merged = merged;
  plan.vb_state = 0;
  cfx_space_s* V = L->V;
  if (!plan.usable || !plan.any_cells) return false;
  const VecBlocks& S = space_vec_blocks(V);
  if (!S.usable) return false;
  DevArray<int32_t> len(S.nblocks);
  DevArray<uint8_t> kind(S.nblocks);
  launch("vec_block_plan", vb_flag_kernel, dim3((unsigned)((S.nblocks + 3) / 4)), dim3(kBlock), 0, S.nblocks, S.B, V->mesh->ncells,
         plan.cellmark.p, mark, S.u_off.p, len.p, kind.p);
  plan.vb_base.alloc(S.nblocks + 1);
  exclusive_scan(len.p, plan.vb_base.p, S.nblocks);
  // merged: one list, a block's uncut entities and rule parents in one pass; else the two kinds of blocks apart
  plan.n_vb_active = compact("vec_block_plan", S.nblocks, KindBit{kind.p, (uint8_t)(merged ? 3 : 1)}, plan.vb_active);
  plan.n_vb_cut = (!merged && (mark & 0xF0u)) ? compact("vec_block_plan", S.nblocks, KindBit{kind.p, 2}, plan.vb_cut) : 0;
  plan.vb_total = read_scalar(plan.vb_base.p + S.nblocks);
  launch("vec_block_plan", vb_base_kernel, grid_for(S.nblocks), dim3(kBlock), 0, S.nblocks, len.p, plan.vb_base.p, plan.n_vb_cut,
         plan.vb_cut.p, kind.p);
  plan.vb_state = 1;
  publish_across_lanes();
  return true;
}

__global__ void mark_cells_u8_kernel(int64_t n, const int32_t* __restrict__ cells, uint8_t* mark)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mark[cells[i]] = 1;
}

// ---------------------------------------------------------------------------
// Row reuse between the patterns of consecutive steps (cfx_pattern_cache)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) cellsig_marks_kernel(int64_t n, const uint8_t* __restrict__ cellmark, uint8_t* __restrict__ sig)
{
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c < n) sig[c] = cellmark[c] ? 1 : 0;
}
// side lf of cell c is a facet of the form: bit 1 + lf of the cell's signature (word-wide atomic OR on the byte)
__global__ void __launch_bounds__(kBlock) cellsig_facets_kernel(int64_t nf, const int32_t* __restrict__ rows, uint8_t* sig)
{
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2 * nf) return;
  const int64_t c = rows[4 * (t >> 1) + 2 * (t & 1)];
  const int lf = rows[4 * (t >> 1) + 2 * (t & 1) + 1];
  atomicOr(reinterpret_cast<unsigned int*>(sig + (c & ~3)), (2u << lf) << (8 * (c & 3)));
}
// clean[i] = every incident cell of hashed row i has the signature it had when the previous pattern was built; a clean
// row's length is the one it had there (expanded rows of a block space: bs rows per dof)
__global__ void __launch_bounds__(kBlock) row_clean_kernel(int64_t n, const int32_t* __restrict__ rows,
                                                           const int64_t* __restrict__ d2c_off, const int32_t* __restrict__ d2c,
                                                           const uint8_t* __restrict__ sig, const uint8_t* __restrict__ prev_sig,
                                                           const int64_t* __restrict__ prev_indptr, int bs,
                                                           uint8_t* __restrict__ clean, int32_t* __restrict__ counts, int* maxlen)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t r = rows[i];
  bool same = true;
  for (int64_t k = d2c_off[r]; k < d2c_off[r + 1]; ++k)
  {
    const int32_t c = d2c[k];
    same = same && sig[c] == prev_sig[c];
  }
  clean[i] = same ? 1 : 0;
  if (!same) return;
  for (int a = 0; a < bs; ++a)
  {
    const int len = (int)(prev_indptr[r * bs + a + 1] - prev_indptr[r * bs + a]);
    counts[r * bs + a] = len;
    if (a == 0 && len / bs > *reinterpret_cast<volatile int*>(maxlen)) atomicMax(maxlen, len / bs);
  }
}
// the columns of the clean rows from the previous pattern: G lanes per dof
template <int G>
__global__ void __launch_bounds__(kBlock) pattern_copy_prev_kernel(int64_t n, const int32_t* __restrict__ rows, int bs,
                                                                   const int64_t* __restrict__ prev_indptr,
                                                                   const int32_t* __restrict__ prev_indices,
                                                                   const int64_t* __restrict__ indptr, int32_t* __restrict__ indices)
{
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t i = t / G;
  if (i >= n) return;
  const int gl = (int)(t - i * G);
  const int64_t r = rows[i];
  for (int a = 0; a < bs; ++a)
  {
    const int64_t pb = prev_indptr[r * bs + a], ob = indptr[r * bs + a];
    const int len = (int)(prev_indptr[r * bs + a + 1] - pb);
    for (int k = gl; k < len; k += G) indices[ob + k] = prev_indices[pb + k];
  }
}

// Sparsity of a form whose test and trial spaces differ (assembler.h:442-529 with two dofmaps): row dof r of the test
// space couples to the trial-space dofs of every cell of a cell integral (standard entity or rule parent) that
// contains r.  No all-rows diagonal (assembler.h:537-560: only when the two index maps coincide).  One wavefront per
// test-space dof, a 512-slot LDS set of trial dofs, count pass + write pass: the rectangular blocks are a small part
// of a system's assembly, the kernel is the general one of the square patterns.
// what the column set of a row depends on, cell by cell: "carries a mark of the form" and "which sides are facets of it"
void plan_cell_signature(cfx_form_s* a)
{
  cfx_row_plan& plan = row_plan(a);
  if (plan.cellsig.n > 0) return;
  const int64_t nc = a->V->mesh->ncells;
  plan.cellsig.alloc((nc + 3) & ~3LL);
  launch("pattern_reuse", cellsig_marks_kernel, grid_for(nc), dim3(kBlock), 0, nc, plan.cellmark.p, plan.cellsig.p);
  const int64_t nf = plan.nfacets.value();
  if (nf > 0)
    launch("pattern_reuse", cellsig_facets_kernel, grid_for(2 * nf), dim3(kBlock), 0, nf, plan.facet_rows.p, plan.cellsig.p);
}

// (type, kernel class) of a form's integrals: facet rows must be sides of mesh facets -- the (bad, root) pairs of the
// extension penalty are not -- and the previous pattern must come from a form of the same structure
static std::vector<int> pattern_form_key(const cfx_form_s* a, bool& ok)
{
  std::vector<int> key;
  ok = true;
  for (const auto& I : a->integrals)
  {
    if (I.type == CFX_INTERIOR_FACET && I.kernel == CFX_K_EXTENSION_L2) ok = false;
    key.push_back(I.type);
  }
  return key;
}

bool pattern_reuse_ok(cfx_form_s* a)
{
  const char* e = getenv("CFX_PATTERN_REUSE");
  if (e && e[0] == '0') return false;
  cfx_space_s* V = a->V;
  const cfx_pattern_cache& pc = V->pcache;
  bool ok = true;
  const std::vector<int> key = pattern_form_key(a, ok);
  return ok && pc.valid && pc.nrows == V->ndofs * V->bs && pc.form_key == key && (pc.live != nullptr || pc.indptr.n > 0);
}

// the pattern just built becomes the space's previous pattern
void pattern_remember(cfx_form_s* a, cfx_pattern_s* P)
{
  const char* e = getenv("CFX_PATTERN_REUSE");
  cfx_space_s* V = a->V;
  cfx_row_plan& plan = row_plan(a);
  const Stencil& st = space_stencil(V);
  bool ok = true;
  const std::vector<int> key = pattern_form_key(a, ok);
  // (the stencil path -- P1 on the geometry dofmap -- describes its rows by masks and has no use for the cache)
  if ((e && e[0] == '0') || st.usable || !plan.any_cells || !ok || a->rectangular()) return;
  plan_cell_signature(a);
  cfx_pattern_cache& pc = V->pcache;
  pc.drop();
  pc.sig.alloc(plan.cellsig.n);
  CFX_HIP(hipMemcpyAsync(pc.sig.p, plan.cellsig.p, (size_t)plan.cellsig.n, hipMemcpyDeviceToDevice, ctx().stream));
  pc.form_key = key;
  pc.nrows = P->nrows;
  pc.live = P;
  P->cache_owner = V;
  pc.valid = true;
  // built inside a speculative step: provisional until the step ends -- a void step leaves `indptr` / `indices` partial
  // or uninitialised (every kernel after the poisoning saw length 0) while the cell signature is mostly the right one,
  // and the repeat would copy "clean" rows out of it
  if (step_speculative()) step_on_void(V, [V]() { V->pcache.drop(); });
}

void build_pattern_rectangular(cfx_form_s* a, cfx_pattern_s* P)
{
  cfx_space_s* V0 = a->V;
  cfx_space_s* V1 = a->V1;
  const int64_t nc = V0->mesh->ncells;
  DevArray<uint8_t> mark((nc + 3) & ~3LL);
  mark.zero();
  bool facets = false;
  for (const auto& I : a->integrals)
  {
    if (I.type == CFX_INTERIOR_FACET) { facets = true; continue; }
    require(I.type == CFX_CELL, CFX_ERR_INVALID_ARGUMENT, "forms with different test and trial spaces take cell and interior-facet integrals");
    const int64_t ne = I.n_entities.value(), nrl = I.rules ? I.rules->nr.value() : 0; // (rectangular blocks: exact lengths)
    if (ne > 0)
      launch("pattern2_mark", mark_cells_u8_kernel, grid_for(ne), dim3(kBlock), 0, ne, I.entities.p, mark.p);
    if (nrl > 0)
      launch("pattern2_mark", mark_cells_u8_kernel, grid_for(nrl), dim3(kBlock), 0, nrl, I.rules->parent_map.p,
             mark.p);
  }
  const Adjacency& adj = V0->dof_cells();
  P->nrows = V0->ndofs * V0->bs;
  P->ncols = V1->ndofs * V1->bs;
  PatArgs S{};
  S.n_active = V0->ndofs; S.active_rows = nullptr;
  S.nd = V1->ndofs_cell; S.bs = V0->bs; S.bs_col = V1->bs; S.no_self = 1; S.dofmap = V1->dofmap.p;
  S.d2c_off = adj.offsets.p; S.d2c = adj.cells.p; S.cellmark = mark.p;
  if (facets)
  {
    // interior-facet terms: a row of either cell couples the TRIAL dofs of both cells (assembler.h:442-529 with two
    // dofmaps).  The facet rows and the dof -> facets incidence are the TEST space's: its row plan has them.
    cfx_row_plan& plan = row_plan(a);
    if (plan.nfacets.value() > 0)
    {
      (void)plan.n_special_rows.value();
      S.d2f_off = plan.d2f_offsets.p; S.d2f = plan.d2f.p; S.facet_rows = plan.facet_rows.p;
      S.special_mark = plan.special_mark.p; S.special_pos = plan.special_pos.p;
    }
  }
  DevArray<int32_t> counts(P->nrows), len(V0->ndofs);
  ZeroFlag overflow, maxlen;
  S.len = len.p; S.counts = counts.p; S.overflow = overflow.p; S.maxlen = maxlen.p;
  launch("pattern2_rows", pattern_rows_kernel<64, 512>, wave_grid(V0->ndofs), dim3(kWave), 0, S);
  require(!read_scalar(overflow.p), CFX_ERR_RUNTIME, "sparsity: a row couples more than 511 dofs");
  P->max_row_len = std::max(read_scalar(maxlen.p), 1);
  P->indptr.alloc(P->nrows + 1);
  exclusive_scan(counts.p, P->indptr.p, P->nrows);
  const int64_t nnz2 = read_scalar(P->indptr.p + P->nrows);
  P->nnz = nnz2;
  P->indices.alloc(nnz2);
  S.indptr = P->indptr.p; S.indices = P->indices.p;
  launch("pattern2_rows_write", pattern_rows_kernel<64, 512>, wave_grid(V0->ndofs), dim3(kWave), 0, S);
  P->built_plan = 0; P->stencil_plan = 0; P->split_plan = 0; P->full_plan = 0; P->odd_plan = 0;
}

void build_pattern(cfx_form_s* a, cfx_pattern_s* P)
{
  if (a->rectangular()) { build_pattern_rectangular(a, P); return; }
  cfx_space_s* V = a->V;
  cfx_row_plan& plan = row_plan(a);
  const Stencil& st = space_stencil(V);
  // plain rows (uncut-cell items only) are subsets of the static stencil: mask + popcount;
  // the hash-set path then only sees the rows next to the interface
  const bool use_stencil = st.usable && plan.n_plain_rows.cap() > 0 && plan.any_cells;
  // lists only (degree 2, vector, DG spaces): the plain rows whose cells are all marked copy their static list
  const bool use_lists = !st.usable && st.lists && plan.n_plain_rows.cap() > 0 && plan.any_cells;
  // Row counts that are still in HBM (plan built inside a sync-free step) stay there on the stencil path (P1 on the
  // geometry dofmap), whose kernels take their lengths from the device; the other paths size host-side work by the
  // counts and read them back first.
  const Count n_h_c = use_stencil ? plan.n_special_rows : Count(plan.n_active_rows.value());
  if (!use_stencil) { (void)plan.n_special_rows.value(); (void)plan.n_plain_rows.value(); }
  int64_t n_h = n_h_c.cap();
  const int64_t n_special_x = plan.n_special_rows.cap(), n_plain_x = plan.n_plain_rows.cap(); // exact off the stencil path
  const int32_t* rows_h = use_stencil ? plan.special_rows.p : plan.active_rows.p;
  P->nrows = V->ndofs * V->bs;
  P->ncols = P->nrows;
  DevArray<int32_t> counts(P->nrows);
  DevArray<uint8_t> full;
  DevArray<int32_t> hashed;
  bool any_full = false;
  if (use_lists)
  {
    const Adjacency& adj = V->dof_cells();
    constexpr int G = 4;
    full.alloc(n_plain_x);
    launch("pattern_plain_full", plain_full_kernel<G>, dim3((unsigned)((n_plain_x + kWave - 1) / kWave)),
           dim3(kWave), 0, n_plain_x, plan.plain_rows.p, adj.offsets.p, adj.cells.p, plan.cellmark.p, st.offsets.p,
           V->bs, full.p, counts.p, plan.bulk ? plan.rowcls.p : (const uint8_t*)nullptr);
    DevArray<int32_t> odd;
    const int64_t n_odd = compact("pattern_plain_full", n_plain_x, FlagIsZero{full.p}, odd);
    any_full = n_odd < n_plain_x;
    n_h = n_special_x + n_odd;
    hashed.alloc(n_h);
    if (n_special_x > 0)
      CFX_HIP(hipMemcpyAsync(hashed.p, plan.special_rows.p, sizeof(int32_t) * (size_t)n_special_x,
                             hipMemcpyDeviceToDevice, ctx().stream));
    if (n_odd > 0)
      launch("pattern_plain_full", gather_i32_kernel, grid_for(n_odd), dim3(kBlock), 0, n_odd, odd.p, plan.plain_rows.p,
             hashed.p + n_special_x);
    rows_h = hashed.p;
  }
  // Moving-domain loops: a hashed row whose incident cells kept their signature since the previous pattern of this space
  // couples the same columns as before -- it copies them (below, once the row offsets are known) instead of building
  // and ranking its hash set again.  Only the rows around cells that changed go through the hash sets.
  int64_t n_d = n_h;                 // rows that are hashed in this build
  const int32_t* rows_d = rows_h;
  DevArray<int32_t> dirty_rows, clean_rows;
  int64_t n_clean = 0;
  ZeroFlag reuse_maxlen;
  const bool reuse = !use_stencil && n_h > 0 && plan.any_cells && pattern_reuse_ok(a);
  P->n_hashed_rows = use_stencil ? 0 : n_h;
  P->n_reused_rows = 0;
  if (reuse)
  {
    const Adjacency& adj = V->dof_cells();
    plan_cell_signature(a);
    const cfx_pattern_cache& pc = V->pcache;
    const int64_t* prev_indptr = pc.live ? pc.live->indptr.p : pc.indptr.p;
    DevArray<uint8_t> clean(n_h);
    launch("pattern_reuse", row_clean_kernel, grid_for(n_h), dim3(kBlock), 0, n_h, rows_h, adj.offsets.p, adj.cells.p,
           plan.cellsig.p, pc.sig.p, prev_indptr, V->bs, clean.p, counts.p, reuse_maxlen.p);
    DevArray<int32_t> idx_d, idx_c;
    n_d = compact("pattern_reuse", n_h, FlagIsZero{clean.p}, idx_d);
    n_clean = n_h - n_d;
    dirty_rows.alloc(n_d);
    if (n_d > 0)
      launch("pattern_reuse", gather_i32_kernel, grid_for(n_d), dim3(kBlock), 0, n_d, idx_d.p, rows_h, dirty_rows.p);
    if (n_clean > 0)
    {
      (void)compact("pattern_reuse", n_h, FlagSet8{clean.p}, idx_c);
      clean_rows.alloc(n_clean);
      launch("pattern_reuse", gather_i32_kernel, grid_for(n_clean), dim3(kBlock), 0, n_clean, idx_c.p, rows_h, clean_rows.p);
    }
    rows_d = dirty_rows.p;
    P->n_reused_rows = n_clean;
    if (getenv("CFX_PLAN_DEBUG"))
      fprintf(stderr, "cutfemx_amd: pattern reuse: %lld of %lld hashed rows copy their columns\n", (long long)n_clean, (long long)n_h);
  }
  PatArgs S{};
  S.n_active = use_stencil ? n_h_c.devn() : DevN(n_d); S.active_rows = rows_d;
  S.nd = V->ndofs_cell; S.bs = V->bs; S.dofmap = V->dofmap.p;
  if (plan.any_cells)
  {
    const Adjacency& adj = V->dof_cells();
    S.d2c_off = adj.offsets.p; S.d2c = adj.cells.p; S.cellmark = plan.cellmark.p;
  }
  if (plan.nfacets.cap() > 0)
  {
    S.d2f_off = plan.d2f_offsets.p; S.d2f = plan.d2f.p; S.facet_rows = plan.facet_rows.p;
    S.special_mark = plan.special_mark.p; S.special_pos = plan.special_pos.p;
  }
  DevArray<int32_t> len(n_d), tmp;
  ZeroFlag overflow, maxlen;
  // (the rows off the active set are never read from `counts`: indptr_*_kernel knows their length)
  S.len = len.p; S.counts = counts.p; S.overflow = overflow.p; S.maxlen = maxlen.p;
  int T = 64;
  // lists path: the hashed rows split by the length of their static list -- short rows (the edge dofs of a degree-2
  // space: ~85 % of the rows) take 16 lanes and a 128-slot set, 4 rows per wavefront, the others one wavefront and
  // 512 slots; both count first and build each set again to write it in place
  constexpr int kShortLen = 40;
  DevArray<int32_t> short_rows, long_rows, short_idx, long_idx, tmp_short, tmp_long;
  bool staged_sets = false;
  int64_t n_short = 0, n_long = 0;
  bool split_hashed = false;
  if (use_lists && n_d > 0 && !V->lists_short_overflow)
  {
    n_short = compact("pattern_split", n_d, StaticLenTest{rows_d, st.offsets.p, kShortLen, false}, short_idx);
    n_long = n_d - n_short;
    short_rows.alloc(n_short);
    long_rows.alloc(n_long);
    if (n_short > 0)
      launch("pattern_split", gather_i32_kernel, grid_for(n_short), dim3(kBlock), 0, n_short, short_idx.p, rows_d, short_rows.p);
    if (n_long > 0)
    {
      compact("pattern_split", n_d, StaticLenTest{rows_d, st.offsets.p, kShortLen, true}, long_idx);
      launch("pattern_split", gather_i32_kernel, grid_for(n_long), dim3(kBlock), 0, n_long, long_idx.p, rows_d, long_rows.p);
    }
    split_hashed = true;
    T = 512;
    PatArgs S1 = S, S2 = S;
    // one pass: every set is ranked into a staging row (128 / 512 columns per row: 6 + 4.5 GB at BASELINE config 4),
    // copied into the CSR arrays once the row offsets are known -- building each set a second time cost as much again
    size_t free_b = 0, total_b = 0;
    CFX_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t need = ((size_t)n_short * 128 + (size_t)n_long * 512) * sizeof(int32_t);
    staged_sets = need < free_b / 2;
    if (staged_sets)
    {
      tmp_short.alloc(n_short * 128);
      tmp_long.alloc(n_long * 512);
    }
    S1.n_active = n_short; S1.active_rows = short_rows.p; S1.tmp = staged_sets ? tmp_short.p : nullptr;
    S2.n_active = n_long; S2.active_rows = long_rows.p; S2.tmp = staged_sets ? tmp_long.p : nullptr; S2.len = len.p + n_short;
    // rows_d = [the plan's special rows in plan order | other rows]: the position in rows_d is the facet-incidence index
    if (rows_d == hashed.p && plan.nfacets.cap() > 0)
    {
      S1.row_pos = short_idx.p; S1.n_first = n_special_x;
      if (n_long > 0) { S2.row_pos = long_idx.p; S2.n_first = n_special_x; }
    }
    if (n_short > 0) launch("pattern_rows_short", pattern_rows_kernel<16, 128>, wave_grid((n_short + 3) / 4), dim3(kWave), 0, S1);
    if (n_long > 0) launch("pattern_rows_wide", pattern_rows_kernel<64, 512>, wave_grid(n_long), dim3(kWave), 0, S2);
    if (read_scalar(overflow.p))
    {
      // a short static list with more than 127 - 40 facet couplings, or a row beyond 511: all rows wide from now on
      V->lists_short_overflow = true;
      split_hashed = false;
      overflow.zero();
      maxlen.zero();
    }
  }
  bool deferred = false;
  if (n_d > 0 && !split_hashed)
  {
    bool wide = V->long_rows;
    if (!wide)
    {
      tmp.alloc(n_d * 64);
      S.tmp = tmp.p;
      launch("pattern_rows", pattern_rows_kernel<4, 64>, wave_grid((n_d + 15) / 16), dim3(kWave), 0, S);
      // stencil path (P1): the overflow flag travels with the longest row and nnz in ONE read-back further down; an
      // overflow (a row with more than 63 columns) then restarts the build on the wide path
      if (use_stencil) deferred = true;
      else wide = read_scalar(overflow.p) != 0;
    }
    if (wide)
    {
      V->long_rows = true; // P2 / vector spaces: skip the narrow attempt next time (150 ms of 450 at config 4)
      // long rows (P2, vector spaces, many facet couplings): one wavefront per row with a
      // 512-slot set.  n_active * 512 staged columns would be ~150 GB for config 4, so the
      // wide path counts first and builds each set again to write it in place.
      T = 512;
      overflow.zero();
      maxlen.zero();
      tmp.release();
      S.tmp = nullptr;
      // (the stencil path may have come here with its row count still in HBM: the wide kernels take it exact; the
      // other paths hash the list they were given -- n_d rows, exact already)
      if (use_stencil) { S.n_active = DevN(n_h_c.value()); n_d = S.n_active.cap; }
      else S.n_active = DevN(n_d);
      launch("pattern_rows_wide", pattern_rows_kernel<64, 512>, wave_grid(n_d), dim3(kWave), 0, S);
      require(!read_scalar(overflow.p), CFX_ERR_RUNTIME, "sparsity: a row couples more than 511 dofs");
    }
  }
  if (use_stencil)
  {
    // (the row lengths come out of the mask pass when the masks are built here; a plan whose masks exist already --
    // a second pattern of the same lists -- takes the separate pass)
    if (!plain_row_masks(a, counts.p, maxlen.p))
      launch("pattern_plain", pattern_plain_len_kernel, grid_for(plan.n_plain_rows.cap()), dim3(kBlock), 0, plan.n_plain_rows,
             plan.plain_rows.p, plan.plain_masks.p, counts.p, maxlen.p);
  }
  if (!deferred) P->max_row_len = plan.n_active_rows.cap() > 0 ? read_scalar(maxlen.p) : 1;
  if (any_full) P->max_row_len = std::max(P->max_row_len, st.max_len); // a copied row is at most the longest static list
  if (reuse && n_clean > 0) P->max_row_len = std::max(P->max_row_len, read_scalar(reuse_maxlen.p)); // ... or of the previous pattern
  if (getenv("CFX_PLAN_DEBUG")) fprintf(stderr, "cutfemx_amd: pattern max row length %d (static lists %d)\n", P->max_row_len, st.max_len);
  P->indptr.alloc(P->nrows + 1);
  {
    const int64_t ntiles = (P->nrows + kTile - 1) / kTile;
    DevArray<int64_t> sums(ntiles), offs(ntiles + 1);
    ChainState chain{};
    CountJobs after{};
    if (deferred)
    {
      // overflow flag, longest row and nnz in one round trip -- or, inside a step, none: nnz stays in HBM, the flag
      // has to repeat the last step's answer and the longest row its size class (what the host picks kernels by)
      const char* names[3] = {"pattern.overflow", "pattern.max_row_len", "pattern.nnz"};
      CountSource src[3];
      src[0].src = overflow.p; src[0].kind = kCountI32; src[0].mode = kCountMustEqual;
      src[1].src = maxlen.p; src[1].kind = kCountI32; src[1].mode = kCountSizeClass;
      src[2].src = offs.p + ntiles; src[2].kind = kCountI64;
      Count t[3];
      CountPlan cp(3, names, src);
      // (inside a step nnz's capacity is known before the rows are summed: reduce + offsets + write in one chained
      // launch -- unless the recorded step found a long row: the host then leaves below, before any kernel that could
      // publish)
      chain = fused_chain(cp.publish && cp.counts[0].cap() == 0, ntiles);
      if (chain.state) after = cp.take_jobs();
      else
      {
        launch("pattern_indptr", indptr_reduce_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, P->nrows, V->bs,
               plan.rowmark.p, counts.p, sums.p);
        exclusive_scan(sums.p, offs.p, ntiles, &cp);
      }
      cp.finish(t);
      if (t[0].cap() != 0)
      {
        V->long_rows = true; // rows beyond 63 columns: build again, wide
        build_pattern(a, P);
        return;
      }
      P->max_row_len = std::max((int)t[1].cap(), 1);
      P->nnz = t[2];
    }
    else
    {
      launch("pattern_indptr", indptr_reduce_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, P->nrows, V->bs, plan.rowmark.p,
             counts.p, sums.p);
      exclusive_scan(sums.p, offs.p, ntiles);
      P->nnz = Count(read_scalar(offs.p + ntiles));
    }
    P->indices.alloc(P->nnz.cap());
    if (chain.state)
      launch("pattern_indptr", indptr_chained_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, P->nrows, V->bs,
             plan.rowmark.p, counts.p, P->indptr.p, P->indices.p, P->nnz.cap(), chain, offs.p + ntiles, after);
    else
      launch("pattern_indptr", indptr_write_kernel, dim3((unsigned)ntiles), dim3(kBlock), 0, P->nrows, V->bs,
             plan.rowmark.p, counts.p, offs.p, P->indptr.p, P->indices.p, P->nnz.devn());
  }
  if (use_stencil && space_stencil_tiles(V).tiles_usable && plan.n_plain_tiles.cap() > 0)
    launch("pattern_plain_write", pattern_plain_tiles_kernel, wave_grid(plan.n_plain_tiles.cap()), dim3(kWave), 0, plan.n_plain_tiles,
           plan.plain_tile_first.p, plan.plain_tile_id.p, plan.n_plain_rows, plan.plain_rows.p, plan.plain_masks.p,
           st.offsets.p, st.nbr.p, P->indptr.p, P->indices.p, V->ndofs);
  else if (use_stencil)
    launch("pattern_plain_write", pattern_plain_write_kernel, grid_for(plan.n_plain_rows.cap() * CFX_PPW_LANES), dim3(kBlock), 0,
           plan.n_plain_rows, plan.plain_rows.p, plan.plain_masks.p, st.offsets.p, st.nbr.p, P->indptr.p, P->indices.p);
  if (reuse && n_clean > 0)
  {
    const cfx_pattern_cache& pc = V->pcache;
    launch("pattern_reuse", pattern_copy_prev_kernel<8>, grid_for(n_clean * 8), dim3(kBlock), 0, n_clean, clean_rows.p, V->bs,
           pc.live ? pc.live->indptr.p : pc.indptr.p, pc.live ? pc.live->indices.p : pc.indices.p, P->indptr.p, P->indices.p);
  }
  if (any_full && V->bs == 1)
    launch("pattern_plain_write", pattern_plain_copy_runs_kernel, wave_grid((n_plain_x + kWave - 1) / kWave), dim3(kWave), 0,
           n_plain_x, plan.plain_rows.p, full.p, st.offsets.p, st.nbr.p, P->indptr.p, P->indices.p);
  else if (any_full && V->bs == 3)
    launch("pattern_plain_write", pattern_plain_copy_block_kernel<3>, wave_grid((n_plain_x + kWave - 1) / kWave), dim3(kWave), 0,
           n_plain_x, plan.plain_rows.p, full.p, st.offsets.p, st.nbr.p, P->indptr.p, P->indices.p);
  else if (any_full && V->bs == 2)
    launch("pattern_plain_write", pattern_plain_copy_block_kernel<2>, wave_grid((n_plain_x + kWave - 1) / kWave), dim3(kWave), 0,
           n_plain_x, plan.plain_rows.p, full.p, st.offsets.p, st.nbr.p, P->indptr.p, P->indices.p);
  else if (any_full)
    launch("pattern_plain_write", pattern_plain_copy_kernel<8>, grid_for(n_plain_x * 8), dim3(kBlock), 0,
           n_plain_x, plan.plain_rows.p, full.p, st.offsets.p, st.nbr.p, V->bs, P->indptr.p, P->indices.p);
  if (split_hashed && staged_sets)
  {
    if (n_short > 0)
      launch("pattern_write", pattern_write_kernel<128>, grid_for(n_short * 8), dim3(kBlock), 0, n_short, short_rows.p, V->bs,
             tmp_short.p, len.p, P->indptr.p, P->indices.p);
    if (n_long > 0)
      launch("pattern_write", pattern_write_kernel<512>, grid_for(n_long * 8), dim3(kBlock), 0, n_long, long_rows.p, V->bs,
             tmp_long.p, len.p + n_short, P->indptr.p, P->indices.p);
  }
  else if (split_hashed)
  {
    PatArgs S1 = S, S2 = S;
    S1.n_active = n_short; S1.active_rows = short_rows.p; S1.tmp = nullptr; S1.indptr = P->indptr.p; S1.indices = P->indices.p;
    S2.n_active = n_long; S2.active_rows = long_rows.p; S2.tmp = nullptr; S2.indptr = P->indptr.p; S2.indices = P->indices.p;
    if (n_short > 0) launch("pattern_rows_short_write", pattern_rows_kernel<16, 128>, wave_grid((n_short + 3) / 4), dim3(kWave), 0, S1);
    if (n_long > 0) launch("pattern_rows_wide_write", pattern_rows_kernel<64, 512>, wave_grid(n_long), dim3(kWave), 0, S2);
  }
  else if (n_d > 0)
  {
    if (T == 64)
      launch("pattern_write", pattern_write_kernel<64>, grid_for(n_d * 8), dim3(kBlock), 0, S.n_active, rows_d, V->bs, tmp.p,
             len.p, P->indptr.p, P->indices.p);
    else
    {
      S.indptr = P->indptr.p; S.indices = P->indices.p;
      launch("pattern_rows_wide_write", pattern_rows_kernel<64, 512>, wave_grid(n_d), dim3(kWave), 0, S);
    }
  }
  // full_rows = [dofs with at most 32 neighbours | the others]: the edge dofs of a degree-2 space (7 of 8 dofs, at most
  // 27 neighbours on Kuhn meshes) need a third of the LDS accumulators of the vertex dofs (65) -- more dofs in flight
  auto split_full_rows = [&]()
  {
    P->n_full_short = 0;
    if (P->n_full_rows == 0) return;
    DevArray<int32_t> idx_s, idx_l, sorted(P->n_full_rows);
    const int64_t ns = compact("pattern_full_rows", P->n_full_rows, StaticLenTest{P->full_rows.p, st.offsets.p, 32, false}, idx_s);
    if (ns > 0 && ns < P->n_full_rows)
    {
      (void)compact("pattern_full_rows", P->n_full_rows, StaticLenTest{P->full_rows.p, st.offsets.p, 32, true}, idx_l);
      launch("pattern_full_rows", gather_i32_kernel, grid_for(ns), dim3(kBlock), 0, ns, idx_s.p, P->full_rows.p, sorted.p);
      launch("pattern_full_rows", gather_i32_kernel, grid_for(P->n_full_rows - ns), dim3(kBlock), 0, P->n_full_rows - ns,
             idx_l.p, P->full_rows.p, sorted.p + ns);
      P->full_rows = std::move(sorted);
    }
    P->n_full_short = ns;
  };
  P->built_plan = plan.serial;
  P->stencil_plan = use_stencil ? plan.serial : 0;
  // spaces with long rows (degree 2): the gather assembly runs the short rows 8 lanes per row
  P->split_plan = 0;
  P->full_plan = 0;
  P->odd_plan = 0;
  if (any_full && V->degree == 2 && V->bs > 1 && a->rank == 2)
  {
    // vector-valued degree 2 with slot records and one uncut cell integral: the copied rows (dofs) take
    // assemble_rows_block_plain_kernel, the other active rows keep the searching block kernel
    int n_std = 0;
    for (const auto& I : a->integrals)
      if (I.type == CFX_CELL && I.n_entities.cap() > 0) ++n_std;
    if (n_std == 1 && space_stencil_slotn(V).slotn_ok)
    {
      DevArray<int32_t> pos;
      P->n_full_rows = compact("pattern_full_rows", n_plain_x, FlagSet8{full.p}, pos);
      P->full_rows.alloc(P->n_full_rows);
      launch("pattern_full_rows", gather_i32_kernel, grid_for(P->n_full_rows), dim3(kBlock), 0, P->n_full_rows, pos.p,
             plan.plain_rows.p, P->full_rows.p);
      split_full_rows();
      P->n_rest_rows = n_h;
      P->rest_rows.alloc(n_h);
      if (n_h > 0)
        CFX_HIP(hipMemcpyAsync(P->rest_rows.p, rows_h, sizeof(int32_t) * (size_t)n_h, hipMemcpyDeviceToDevice, ctx().stream));
      P->full_plan = plan.serial;
    }
  }
  if (P->max_row_len > 64 && V->bs == 1 && plan.n_active_rows.cap() > 0)
  {
    // degree 2 with slot records, one uncut stiffness integral: the copied rows get their own gather kernel
    // (assemble_rows_p2_plain_kernel); the split below then covers the other active rows only
    const int32_t* base_rows = plan.active_rows.p;
    int64_t n_base = plan.n_active_rows.value();
    if (any_full && V->degree == 2)
    {
      int n_std = 0;
      bool closed = true;
      for (const auto& I : a->integrals)
        if (I.type == CFX_CELL && I.n_entities.cap() > 0)
        {
          ++n_std;
          closed = closed && I.kernel == CFX_K_STIFFNESS && I.coefficient.n == 0;
        }
      const char* cf = getenv("CFX_P2_CLOSED");
      if (n_std == 1 && closed && a->rank == 2 && !(cf && cf[0] == '0') && space_stencil_slotn(V).slotn_ok)
      {
        DevArray<int32_t> pos;
        P->n_full_rows = compact("pattern_full_rows", n_plain_x, FlagSet8{full.p}, pos);
        P->full_rows.alloc(P->n_full_rows);
        launch("pattern_full_rows", gather_i32_kernel, grid_for(P->n_full_rows), dim3(kBlock), 0, P->n_full_rows, pos.p,
               plan.plain_rows.p, P->full_rows.p);
        P->full_plan = plan.serial;
        base_rows = rows_h;
        n_base = n_h;
        // rows_h = [interface rows | plain rows without a copied list]: the second part on its own
        P->n_odd_rows = n_h - n_special_x;
        P->odd_rows.alloc(P->n_odd_rows);
        if (P->n_odd_rows > 0)
          CFX_HIP(hipMemcpyAsync(P->odd_rows.p, rows_h + n_special_x, sizeof(int32_t) * (size_t)P->n_odd_rows,
                                 hipMemcpyDeviceToDevice, ctx().stream));
        P->odd_plan = plan.serial;
      }
    }
    P->n_short_rows = compact("pattern_short_rows", n_base, RowLenTest{base_rows, P->indptr.p, 64, false}, P->short_rows);
    // (three classes: a row set of 256 columns leaves the interface kernel two wavefronts per SIMD, and the vertex dofs
    // of a degree-2 space -- 65 columns in a Kuhn mesh, ~90 with ghost-penalty couplings -- do not need it)
    P->n_mid_rows = compact("pattern_long_rows", n_base, RowLenRange{base_rows, P->indptr.p, 64, 128}, P->mid_rows);
    P->n_long_rows = compact("pattern_long_rows", n_base, RowLenTest{base_rows, P->indptr.p, 128, true}, P->long_rows);
    launch("pattern_map_rows", map_rows_kernel, grid_for(P->n_short_rows), dim3(kBlock), 0, P->n_short_rows,
           base_rows, P->short_rows.p);
    launch("pattern_map_rows", map_rows_kernel, grid_for(P->n_mid_rows), dim3(kBlock), 0, P->n_mid_rows,
           base_rows, P->mid_rows.p);
    launch("pattern_map_rows", map_rows_kernel, grid_for(P->n_long_rows), dim3(kBlock), 0, P->n_long_rows,
           base_rows, P->long_rows.p);
    if (getenv("CFX_PLAN_DEBUG"))
      fprintf(stderr, "cutfemx_amd: hashed row classes: <= 64 columns %lld, <= 128 %lld, longer %lld\n",
              (long long)P->n_short_rows, (long long)P->n_mid_rows, (long long)P->n_long_rows);
    P->split_plan = plan.serial;
  }
  pattern_remember(a, P);
}

} // namespace cfx

// the pattern a space's cache points at dies: its arrays move to the cache (no copy)
cfx_pattern_s::~cfx_pattern_s()
{
  if (cache_owner && cache_owner->pcache.live == this)
  {
    cfx_pattern_cache& pc = cache_owner->pcache;
    pc.indptr = std::move(indptr);
    pc.indices = std::move(indices);
    pc.live = nullptr;
  }
}

void cfx_pattern_cache::drop()
{
  if (live) live->cache_owner = nullptr;
  live = nullptr;
  indptr.release();
  indices.release();
  sig.release();
  valid = false;
}
