// cutfemx_amd: shared host/device plumbing for the HIP engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/cutfemx_amd.h"

namespace cfx
{

// ---------------------------------------------------------------------------
// errors: C++ exceptions inside, status code + thread-local message outside
// ---------------------------------------------------------------------------
struct Error : std::runtime_error
{
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

extern thread_local std::string g_last_error;
// An error raised while a sync-free step is open may be an artefact of a VOID step (a count did not fit: every kernel
// after that point did nothing, and whatever the host then reads back -- an empty active-cell list, say -- is not a
// property of the problem).  The status of such a call becomes CFX_ERR_STEP_VOID: end the step, repeat it.
int error_in_step(int code);
void step_resolve_all(); // inside a step: every count published so far comes to the host in one read-back (cfx_step_resolve)

#define CFX_HIP(expr)                                                                   \
  do                                                                                    \
  {                                                                                     \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      throw ::cfx::Error(CFX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define CFX_API_BEGIN try {
#define CFX_API_END                                                   \
  return CFX_OK;                                                      \
  }                                                                   \
  catch (const ::cfx::Error& e)                                       \
  {                                                                   \
    ::cfx::g_last_error = e.what();                                   \
    return ::cfx::error_in_step(e.code);                              \
  }                                                                   \
  catch (const std::exception& e)                                     \
  {                                                                   \
    ::cfx::g_last_error = e.what();                                   \
    return ::cfx::error_in_step(CFX_ERR_RUNTIME);                     \
  }

// closes a CFX_API_BEGIN block whose success path continues below it (argument checks ahead of a delegated call)
#define CFX_API_END_NO_RETURN                                         \
  }                                                                   \
  catch (const ::cfx::Error& e)                                       \
  {                                                                   \
    ::cfx::g_last_error = e.what();                                   \
    return ::cfx::error_in_step(e.code);                              \
  }                                                                   \
  catch (const std::exception& e)                                     \
  {                                                                   \
    ::cfx::g_last_error = e.what();                                   \
    return ::cfx::error_in_step(CFX_ERR_RUNTIME);                     \
  }

inline void require(bool ok, int code, const char* msg)
{
  if (!ok) throw Error(code, msg);
}

// ---------------------------------------------------------------------------
// context: device, stream, per-kernel profile
// ---------------------------------------------------------------------------
struct ProfileEntry
{
  std::string name;
  double total_ms = 0.0;
  int64_t launches = 0;
};

struct Context
{
  int device = -1;
  bool initialised = false;
  hipStream_t stream = nullptr;      // the stream launches go to: main_stream, or side_stream inside cfx_overlap_side()
  hipStream_t main_stream = nullptr; // what cfx_set_stream chose (default: the null stream)
  hipStream_t side_stream = nullptr; // second lane of an overlap section (created on first use, non-blocking)
  bool overlap = false;              // inside cfx_overlap_begin / cfx_overlap_end: block frees are deferred
  std::vector<void*> deferred_free;
  bool profile = false;
  const char* last_launch = "(none)"; // name of the latest kernel launch (CFX_COUNT_SYNC=2 names the read-backs by it)
  bool trace_sync = false;
  std::vector<ProfileEntry> entries;
  std::map<std::string, int> entry_index;
  struct Pending { int entry; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;

  void ensure();
  hipEvent_t get_event();
  int entry(const char* name);
  void flush_profile();
};

Context& ctx();
// A table that is built lazily and then shared (incidence lists, row stencils, tiles, slot records, the row plan of
// two forms) is enqueued on the lane that first needs it while the host flag says "built" at once: inside an
// overlap section the OTHER lane must not run ahead of that build.  Call after enqueuing the build: the other
// lane waits (device-side) for everything queued so far on this one.  No-op outside overlap sections.
void publish_across_lanes();

// launch with optional HIP-event bracketing on the library stream
template <typename K, typename... Args>
inline void launch(const char* name, K kernel, dim3 grid, dim3 block, size_t shmem, Args... args)
{
  Context& c = ctx();
  if (grid.x == 0) return;
  c.last_launch = name;
  static const bool trace = getenv("CFX_LAUNCH_TRACE") != nullptr; // the launch sequence of a step on stderr
  if (trace) fprintf(stderr, "cutfemx_amd: launch %s grid %u\n", name, grid.x);
  // HIP launches at most 2^32 - 1 work-items per dimension; a larger grid is cut short silently
  if ((uint64_t)grid.x * block.x > 0xffffffffull)
    throw Error(CFX_ERR_RUNTIME, std::string(name) + ": launch exceeds 2^32 threads");
  if (c.profile)
  {
    hipEvent_t a = c.get_event(), b = c.get_event();
    CFX_HIP(hipEventRecord(a, c.stream));
    hipLaunchKernelGGL(kernel, grid, block, shmem, c.stream, args...);
    CFX_HIP(hipEventRecord(b, c.stream));
    c.pending.push_back({c.entry(name), a, b});
  }
  else
  {
    hipLaunchKernelGGL(kernel, grid, block, shmem, c.stream, args...);
  }
  CFX_HIP(hipGetLastError());
  static const bool sync_each = getenv("CFX_LAUNCH_SYNC") != nullptr; // diagnostics: a fault surfaces at its own launch
  if (sync_each) CFX_HIP(hipStreamSynchronize(c.stream));
}

// grid of one-wavefront blocks whose kernel loops `for (blk = blockIdx.x; ...; blk += gridDim.x)`
inline dim3 wave_grid(int64_t nblocks)
{
  const int64_t cap = (0xffffffffll / 64) / 8 * 8;
  return dim3((unsigned)std::min<int64_t>(std::max<int64_t>(nblocks, 1), cap));
}

inline dim3 grid_for(int64_t n, int block = 256)
{
  int64_t g = (n + block - 1) / block;
  if (g > 2147483647LL) throw Error(CFX_ERR_RUNTIME, "grid too large");
  return dim3((unsigned)g);
}

// ---------------------------------------------------------------------------
// HBM block cache.  All work is issued in order on one stream, so a block
// released on the host can be handed to a later launch without an event.
// (ROCm 7.2's stream-ordered pool gives memory back at every synchronisation
// even with the release threshold at its maximum: measured 1.2 ms per 1.6 GB
// of hipMallocAsync/hipFreeAsync per step, hence this cache.)
// ---------------------------------------------------------------------------
void* dev_alloc(size_t bytes);
void dev_free(void* p);
void dev_cache_release(); // hipFree every cached block
// Identity of library-owned device arrays and handles.  Every block handed out by dev_alloc carries a serial number
// that is never reused: a list that was released and rebuilt at the same address (cfx_cut_update drops the located
// lists of a cut) has a different serial, so a row plan keyed on it cannot be adopted by mistake and a form that
// still points at the old list is refused instead of reading recycled memory.  0 = not a live library block
// (caller-owned memory: identity by address, the caller keeps it unchanged while a form refers to it).
uint64_t dev_block_serial(const void* p);
uint64_t next_serial();                    // process-wide counter shared by blocks, rules handles and plans
void rules_serial_live(uint64_t s, bool live);
bool rules_serial_is_live(uint64_t s);
void device_memory_stats(size_t& live, size_t& cached, size_t& peak); // bytes held from HIP: in use / cached / high-water
void device_memory_reset_peak();

// ---------------------------------------------------------------------------
// Lengths that live in HBM (sync-free steps).
//
// Inside cfx_step_begin / cfx_step_end the size of a data-dependent list (located cells, rule points, ghost facets,
// row classes, nnz ...) is not read back where it is produced: the producer PUBLISHES it in a slot of a small device
// pool, the host continues with a capacity taken from the same site's count in the previous step of the loop
// (x 1.03125 + slack), every consumer sizes its grid by the capacity and takes the length from the slot.  One read-back
// at cfx_step_end fetches all slots: it resolves the counts on the host, feeds the next step's capacities and says
// whether some count did not fit (then the step is void and the caller repeats it; the repeat sizes everything by
// read-backs, like the first step of a loop).  A count that does not fit sets the pool's POISON word: every kernel
// that takes a DevN sees length 0 from then on, so the rest of a void step touches no memory beyond what was sized.
// Outside a step -- and at every site that has not been taught to wait -- a count is read back at once as before.
// ---------------------------------------------------------------------------
struct DevN
{
  int64_t cap = 0;              // host bound: the exact length when dev == nullptr
  const int64_t* dev = nullptr; // published length (entry of the count pool), or nullptr
  DevN() = default;
  DevN(int64_t n) : cap(n) {}
  DevN(int64_t c, const int64_t* d) : cap(c), dev(d) {}
};
constexpr int kCountEntries = 1024;                          // pool entries of {published, raw} pairs; entry 0 = poison
constexpr uintptr_t kCountPoolBytes = kCountEntries * 16;    // the pool is aligned to its size: poison = base of any slot
constexpr int kCountFirstSlot = 128;                         // entries 1..127: error words of the step's assembly calls

#if defined(__HIPCC__)
// (one scalar load: the kernel that finds a count beyond its capacity sets the poison word AND zeroes every published
// length of the pool, and a length published later in a void step is published as 0 -- count_publish_kernel)
__device__ __forceinline__ int64_t dev_n(const DevN& nn)
{
  if (nn.dev == nullptr) return nn.cap;
  const int64_t v = *nn.dev;
  return v < nn.cap ? v : nn.cap;
}
__device__ __forceinline__ int64_t dev_len(const DevN& nn) { return dev_n(nn); }
#endif

struct CountCell
{
  int slot = -1;          // pool entry while the value is only on the device
  int64_t cap = 0;        // capacity the host works with until then
  int64_t hint = 0;       // the same site's value in the previous step (what host-side heuristics compare)
  int64_t value = 0;      // exact value once resolved
  bool resolved = true;
  uint64_t step_serial = 0; // the speculative step that published it (its pool slot is meaningful while that step is open)
  ~CountCell();
};

// the length of a library-owned list: exact on the host, or capacity + device slot until the step ends
struct Count
{
  std::shared_ptr<CountCell> cell;
  int64_t exact_n = 0; // used when cell == nullptr
  Count() = default;
  Count(int64_t n) : exact_n(n) {}
  bool pending() const { return cell && !cell->resolved; }
  int64_t cap() const { return cell ? (cell->resolved ? cell->value : cell->cap) : exact_n; } // allocation / grid bound
  int64_t hint() const { return cell ? (cell->resolved ? cell->value : cell->hint) : exact_n; } // best guess of the value
  // the same number whether the count is still pending or has been read back since: what a cache key may hold (cap()
  // turns from the capacity into the value when a count is resolved in mid-step)
  int64_t key() const { return cell ? cell->cap : exact_n; }
  int64_t value() const;        // exact: a pending count is read back now (one counted round trip)
  DevN devn() const;            // what kernels take
  operator DevN() const { return devn(); }
};

// what a site hands to publish: where queued work leaves the total, and how to read it
enum CountKind { kCountI64 = 0, kCountI32 = 1, kCountPackedLo = 2, kCountPackedHi = 3,
                 kCountLo32 = 4, kCountHi32 = 5, kCountSum32 = 6 }; // halves of an int64 of two packed 32-bit totals, their sum
constexpr int kCountPackShift = 34; // packed totals: low bits | high bits << kCountPackShift (the rule scans of cfx_cut.hip)
// how a site's capacity follows from its value in the previous step: a length that may grow (x margin + slack), or a
// quantity the host branches on (a flag word, a size class) that has to come out EQUAL or the step is void
// kCountSizeClass: a maximum the host only compares with 32 / 64 / 128 / 256 (row-length classes that select kernel
// instantiations): the capacity is the class bound of the previous value
enum CountMode { kCountUpTo = 0, kCountMustEqual = 1, kCountSizeClass = 2 };
struct CountSource
{
  const void* src = nullptr;
  int kind = kCountI64;
  int mode = kCountUpTo;
  const int64_t* plus = nullptr; // value = read(src, kind) + *plus + add (lengths of concatenated lists); src may be null
  int64_t add = 0;
};
// n sites at once (one read-back outside a step / one publish launch inside one); `names` identify the sites in the
// step's history (the k-th site of a step must be the k-th site of the previous one, else it reads back as before)
void count_sites(int n, const char* const* names, const CountSource* src, Count* out);
// The same in two phases around the scan that produces the totals: inside a speculative step the scan's last thread
// publishes them itself (no publish launch); otherwise finish() reads them back after the scan.
//   CountPlan cp(n, names, src);  exclusive_scan(in, out, len, &cp);  cp.finish(counts);
constexpr int kMaxCountJobs = 6;
struct CountJobs
{
  int64_t* pool;   // the count pool; n_slots: slots handed out so far in this step (what a void step zeroes)
  int n_slots;
  int n;
  const void* src[kMaxCountJobs];
  int kind[kMaxCountJobs];
  int mode[kMaxCountJobs];
  const int64_t* plus[kMaxCountJobs];
  int64_t add[kMaxCountJobs];
  int64_t cap[kMaxCountJobs];
  int slot[kMaxCountJobs];
};
struct CountPlan
{
  int n = 0;
  bool publish = false, fused = false, finished = false;
  std::vector<std::string> names;
  std::vector<CountSource> src;
  std::vector<Count> counts;         // publish mode: the pending counts, made by the constructor
  std::shared_ptr<CountJobs> jobs;   // publish mode: what the publishing thread does
  CountPlan(int n, const char* const* names, const CountSource* src);
  void finish(Count* out);
  // publish mode, for a kernel of the caller's that writes the totals itself (its last thread then runs count_publish):
  // the jobs to hand to it; finish() then launches nothing and may be called BEFORE that kernel to get the capacities
  CountJobs take_jobs() { fused = true; return *jobs; }
};
#if defined(__HIPCC__)
__device__ __forceinline__ int64_t count_read(const void* src, int kind)
{
  int64_t v;
  if (src == nullptr) return 0;
  if (kind == kCountI32) v = *static_cast<const int32_t*>(src);
  else
  {
    v = *static_cast<const int64_t*>(src);
    if (kind == kCountPackedLo) v &= (1ll << kCountPackShift) - 1;
    else if (kind == kCountPackedHi) v >>= kCountPackShift;
    else if (kind == kCountLo32) v &= 0xffffffffll;
    else if (kind == kCountHi32) v >>= 32;
    else if (kind == kCountSum32) v = (v & 0xffffffffll) + (v >> 32);
  }
  return v;
}
// speculative step: raw total next to the published one; a total beyond its capacity poisons the step
// Run by ONE thread.  A total beyond its capacity (or a must-equal word that differs) voids the step: the poison word is
// set and EVERY published length of the pool becomes 0, so that each kernel launched from now on -- whatever list drives
// it -- sees length 0 with the one load of dev_n; lengths published later in the void step are published as 0.
__device__ __forceinline__ void count_publish(const CountJobs& J)
{
  int64_t* pool = J.pool;
  bool void_step = pool[0] != 0;
  int64_t raw[kMaxCountJobs];
  for (int k = 0; k < J.n; ++k)
  {
    raw[k] = count_read(J.src[k], J.kind[k]) + (J.plus[k] ? *J.plus[k] : 0) + J.add[k];
    pool[2 * J.slot[k] + 1] = raw[k];
    void_step = void_step || (J.mode[k] == kCountMustEqual ? raw[k] != J.cap[k] : raw[k] > J.cap[k]);
  }
  for (int k = 0; k < J.n; ++k) pool[2 * J.slot[k]] = void_step ? 0 : raw[k];
  if (void_step)
  {
    pool[0] = 1;
    for (int e = kCountFirstSlot; e < J.n_slots; ++e) pool[2 * e] = 0;
  }
}
#endif
inline Count count_site(const char* name, const void* src, int kind = kCountI64, int mode = kCountUpTo)
{
  CountSource s{src, kind, mode};
  Count c;
  count_sites(1, &name, &s, &c);
  return c;
}
// library lists whose length is still in HBM, by address: the ABI hands (pointer, capacity) out and gets it back in
// cfx_integral / cfx_cut_* arguments -- the library recognises its own list and takes the published length
void list_register(const void* p, const Count& c);
void list_unregister(const void* p);
// Provenance of a located entity list: "the cells of cut `cut` (classification number `gen`) whose domain value of
// level set 0 is `value`" -- what lets a row plan derive the marks of such a list from the classification instead of
// walking the list (cfx::row_plan, bulk tiles).  Keyed by the list's address; `serial` = the library block behind it
// (addresses are recycled, serials are not).  Forgotten when the cut classifies again or dies.
struct ListProvenance
{
  const struct ::cfx_cut_s* cut = nullptr;
  uint64_t gen = 0;
  int value = 0;
  uint64_t serial = 0;
  int64_t n = 0; // elements of the list as handed out (its capacity while the length is in HBM)
};
void provenance_register(const void* p, const struct ::cfx_cut_s* cut, uint64_t gen, int value, int64_t n);
const ListProvenance* provenance_lookup(const void* p); // nullptr: unknown, or the block behind p is another one now
void provenance_forget_cut(const struct ::cfx_cut_s* cut);
Count list_lookup(const void* p, int64_t n_given);
void step_record(const char* name, int64_t value); // a total the host obtained by other means: keeps the step's site
                                                   // sequence equal to the one of a step that publishes it
// a count that must not be zero: checked now when it is on the host, else when the step's read-back resolves it
void step_require_positive(const Count& c, int code, const char* message);
Count count_sum(const char* name, const Count& a, const Count& b); // a + b: exact, or published when one is pending
bool step_speculative();      // inside a step whose sizes come from the previous one
const int64_t* step_poison(); // the pool's poison word (nullptr outside speculative steps)
// `fn` runs if the open speculative step ends void or is aborted (state kept across steps that the step wrote must not
// survive it); `step_forget_owner` withdraws an owner's hooks when the owner dies first
void step_on_void(const void* owner, std::function<void()> fn);
void step_forget_owner(const void* owner);
// error words of assembly calls: inside a step they are read with the slots at cfx_step_end (the call returns at
// once), outside they are read back by the caller as before
int* step_error_flag(int code, const char* message, void (*decode)(int) = nullptr); // nullptr outside a step

// ---------------------------------------------------------------------------
// device arrays: owning (cached blocks) or aliasing a caller pointer
// ---------------------------------------------------------------------------
// memset on the library stream in ONE launch (hipMemsetAsync splits unaligned sizes into up to three fill kernels,
// ~100 tiny launches per step); cfx_runtime.hip
void dev_fill(void* p, int byte, size_t bytes);
void dev_fill2(void* pa, int byte_a, size_t bytes_a, void* pb, int byte_b, size_t bytes_b); // two fills, one launch

// a zero-initialised device int from a pool that is cleared with one fill per 4096 flags (error / overflow flags:
// a step took a dozen 4-byte memsets for them)
int* zero_flag(); // cfx_runtime.hip
struct ZeroFlag
{
  int* p;
  ZeroFlag() : p(zero_flag()) {}
  void zero() { dev_fill(p, 0, sizeof(int)); } // a flag that is reused in a loop
};

template <typename T>
struct DevArray
{
  T* p = nullptr;
  int64_t n = 0;   // elements allocated; the length of the list unless `count` says otherwise
  bool owned = false;
  Count count;     // lists sized by a previous step: n is their capacity, count the published length
  DevN devn(int64_t per = 1) const { return count.cell ? count.devn() : DevN(n / per); }

  DevArray() = default;
  explicit DevArray(int64_t elements) { alloc(elements); }
  DevArray(const DevArray&) = delete;
  DevArray& operator=(const DevArray&) = delete;
  DevArray(DevArray&& o) noexcept : p(o.p), n(o.n), owned(o.owned), count(std::move(o.count))
  {
    o.p = nullptr; o.n = 0; o.owned = false; o.count = Count();
  }
  DevArray& operator=(DevArray&& o) noexcept
  {
    if (this != &o)
    {
      release();
      p = o.p; n = o.n; owned = o.owned; count = std::move(o.count);
      o.p = nullptr; o.n = 0; o.owned = false; o.count = Count();
    }
    return *this;
  }
  ~DevArray() { release(); }

  void alloc(int64_t elements)
  {
    release();
    n = elements;
    owned = true;
    // never hand out a null pointer for an empty array
    p = static_cast<T*>(dev_alloc(sizeof(T) * (size_t)(elements > 0 ? elements : 1)));
  }
  void release()
  {
    if (p && count.cell) list_unregister(p);
    if (p && owned) dev_free(p);
    p = nullptr; n = 0; owned = false; count = Count();
  }
  void zero() { if (n > 0) dev_fill(p, 0, sizeof(T) * (size_t)n); }
  T* get() const { return p; }
};

bool is_device_pointer(const void* p);

// device copy of an input array: aliases device pointers, uploads host ones
template <typename T>
inline DevArray<T> to_device(const T* src, int64_t n)
{
  DevArray<T> a;
  if (src == nullptr || n == 0) { a.alloc(0); return a; }
  if (is_device_pointer(src))
  {
    a.p = const_cast<T*>(src); a.n = n; a.owned = false;
    return a;
  }
  a.alloc(n);
  CFX_HIP(hipMemcpyAsync(a.p, src, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, ctx().stream));
  CFX_HIP(hipStreamSynchronize(ctx().stream)); // the host buffer may die right after the call
  return a;
}

// connectivity / dofmap tables: kernels read their rows with 16-byte (8-byte) loads, so a caller's device array that
// starts off a 16-byte boundary (a view into a larger tensor) is copied once instead of aliased
template <typename T>
inline DevArray<T> to_device_aligned(const T* src, int64_t n)
{
  if (src != nullptr && n > 0 && is_device_pointer(src) && (reinterpret_cast<uintptr_t>(src) & 15) != 0)
  {
    DevArray<T> a;
    a.alloc(n);
    CFX_HIP(hipMemcpyAsync(a.p, src, sizeof(T) * (size_t)n, hipMemcpyDeviceToDevice, ctx().stream));
    return a;
  }
  return to_device(src, n);
}

// error word of an assembly call (an entry missing from the pattern, a deactivated row without a diagonal ...): read
// back by check() outside a step; inside a step it is one of the step's error words, read with the slots at
// cfx_step_end, which raises the error -- the call itself returns without waiting
struct ErrorFlag
{
  int* p;
  bool deferred;
  // decode (optional): raises the error that a non-zero value of the word stands for
  ErrorFlag(int code, const char* message, void (*decode)(int) = nullptr);
  void check(int code, const char* message) const;
};
// count of a caller-supplied output buffer: the capacity serves for HBM destinations (written in place up to the
// published length), a host destination is filled from a staging copy and needs the exact count
int64_t count_for_buffer(const Count& c, const void* user);
void end_of_call_sync(); // stream synchronisation at the end of an API call -- skipped inside a step

int64_t& sync_counter(); // host round trips so far (cfx_runtime.hip), reported by CFX_COUNT_SYNC=1

void* pinned_scratch(); // 256 B of page-locked host memory (cfx_runtime.hip): target of the size read-backs

// A value read back inside a SPECULATIVE step is only meaningful while the step is not void: once a count did not fit,
// every kernel after it did nothing and what the host reads -- a total of lengths nobody wrote, say -- is garbage that
// must not size anything (a 3 GB `nnz` and a fault followed, tests/test_gpu_step.py: the long loop).  Every read-back
// of such a step therefore brings the poison word along in the same round trip and throws CFX_ERR_STEP_VOID when it is
// set: the call returns that status, the caller ends the step and repeats it.
struct StepVoidGuard
{
  int64_t* host = nullptr; // pinned word (null: no speculative step open)
  StepVoidGuard();         // queues the copy of the poison word on ctx().stream (cfx_runtime.hip)
  void check() const;      // after the synchronisation: throws Error(CFX_ERR_STEP_VOID) when the step is void
};

template <typename T>
inline T read_scalar(const T* dev)
{
  // into page-locked memory: a pageable destination makes hipMemcpyAsync stage the copy and block longer
  static_assert(sizeof(T) <= 256, "read_scalar reads small values");
  T* h = static_cast<T*>(pinned_scratch());
  CFX_HIP(hipMemcpyAsync(h, dev, sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
  const StepVoidGuard guard;
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  const T v = *h;
  ++sync_counter();
  if (ctx().trace_sync) fprintf(stderr, "cutfemx_amd: read-back after %s\n", ctx().last_launch);
  guard.check();
  return v;
}

// two scalars, one round trip
template <typename T>
inline void read_two(const T* dev_a, const T* dev_b, T& a, T& b)
{
  static_assert(2 * sizeof(T) <= 256, "read_two reads small values");
  T* h = static_cast<T*>(pinned_scratch());
  CFX_HIP(hipMemcpyAsync(h, dev_a, sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
  CFX_HIP(hipMemcpyAsync(h + 1, dev_b, sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
  const StepVoidGuard guard;
  CFX_HIP(hipStreamSynchronize(ctx().stream));
  a = h[0]; b = h[1];
  ++sync_counter();
  if (ctx().trace_sync) fprintf(stderr, "cutfemx_amd: read-back after %s\n", ctx().last_launch);
  guard.check();
}

template <typename T>
inline std::vector<T> download(const T* dev, int64_t n)
{
  std::vector<T> v((size_t)n);
  if (n > 0)
  {
    CFX_HIP(hipMemcpyAsync(v.data(), dev, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, ctx().stream));
    const StepVoidGuard guard;
    CFX_HIP(hipStreamSynchronize(ctx().stream));
    guard.check();
  }
  return v;
}

// an output pointer supplied by the caller: device pointers are written in
// place, host pointers through a staging buffer copied back at the end
template <typename T>
struct OutArray
{
  T* user = nullptr;
  DevArray<T> staging;
  T* dev = nullptr;
  int64_t n = 0;
  OutArray(T* dst, int64_t count, bool copy_in) : user(dst), n(count)
  {
    if (is_device_pointer(dst)) { dev = dst; return; }
    staging.alloc(count);
    dev = staging.p;
    if (copy_in && count > 0)
      CFX_HIP(hipMemcpyAsync(dev, dst, sizeof(T) * (size_t)count, hipMemcpyHostToDevice, ctx().stream));
  }
  void finish()
  {
    if (dev != user && n > 0)
    {
      CFX_HIP(hipMemcpyAsync(user, dev, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, ctx().stream));
      CFX_HIP(hipStreamSynchronize(ctx().stream));
    }
  }
};

// ---------------------------------------------------------------------------
// device primitives (cfx_primitives.hip)
// ---------------------------------------------------------------------------
// out[0..n] = exclusive scan of in[0..n-1]; out[n] = total
// `after` (optional): counts whose sources are complete once this scan has written its total (cfx::CountPlan)
void exclusive_scan(const int32_t* in, int64_t* out, int64_t n, CountPlan* after = nullptr);
void exclusive_scan(const int32_t* in, int32_t* out, int64_t n, CountPlan* after = nullptr);
void exclusive_scan(const int64_t* in, int64_t* out, int64_t n, CountPlan* after = nullptr);
// two scans of one length in one launch (the counts of `after` are published once both totals are written)
void exclusive_scan_pair(const int32_t* inA, int64_t* outA, const int32_t* inB, int64_t* outB, int64_t n, CountPlan* after = nullptr);
void exclusive_scan_pair(const int64_t* inA, int64_t* outA, const int64_t* inB, int64_t* outB, int64_t n, CountPlan* after = nullptr);

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------
struct Adjacency
{
  // item -> cells incidence (CSR), cells ascending inside each segment
  DevArray<int64_t> offsets; // [nitems+1]
  DevArray<int32_t> cells;   // [ncells*width]
  bool built = false;
};

void build_adjacency(const int32_t* map, int64_t ncells, int width, int64_t nitems, Adjacency& adj);

// Mesh-static row stencil of a P1 space on the geometry dofmap: the sorted neighbour list of
// every dof (the CSR row it has when all its cells are present) and, for every (dof, incident
// cell) pair of the dof->cells incidence, the positions of the cell's dofs inside that list.
// A row whose items are uncut cells only is then a SUBSET of its stencil: the sparsity and
// assembly kernels describe it by a 64-bit mask and find CSR slots by popcount, with no hash
// set, no sort and no column search.
struct Stencil
{
  DevArray<int64_t> offsets; // [ndofs+1]
  DevArray<int32_t> nbr;     // neighbours incl. the dof itself, ascending
  DevArray<uint32_t> slot4;  // parallel to dof_cells().cells: byte j = position of the cell's j-th dof
  DevArray<uint8_t> diagpos; // position of the dof in its own list
  DevArray<uint8_t> cpos;    // [ncells*nd]: position of the cell in the dof->cells list of its j-th dof
  bool built = false, usable = false;
  bool lists = false; // offsets + nbr are valid (any space); usable: slot4 / diagpos / cpos too (P1 on the geometry dofmap)
  // degree-2 scalar spaces: per dof->cells entry a 12-byte record -- bytes 0..9 (0..5 in 2-D) the positions of the cell's
  // dofs in the dof's neighbour list, byte 10 the local index of the dof in the cell (built on first use by
  // cfx::space_stencil_slotn; needs every list shorter than 256)
  DevArray<uint8_t> slotn;
  bool slotn_built = false, slotn_ok = false;
  int max_len = 0; // longest neighbour list
  // Row tiles: kRowTile consecutive dofs.  Mesh-static per tile: the sorted union of its rows' neighbour lists
  // (tile_verts) and, parallel to nbr, the position of every neighbour in that union (st_loc).  The gather
  // kernels stage a tile's coordinates, slot4 range and st_loc range in LDS with coalesced loads -- one
  // request per 128 B line instead of one per (row, lane) gather.  tiles_usable: every tile fits the LDS budget.
  DevArray<int64_t> tile_voff;  // [ntiles+1]
  DevArray<int32_t> tile_verts;
  DevArray<uint16_t> st_loc;
  bool tiles_built = false, tiles_usable = false;
  int max_tile_verts = 0, max_tile_items = 0, max_tile_st = 0; // items: dof->cells entries, st: neighbour entries
  // The single-pass build of the neighbour lists stages 64 columns per dof (34 GB at 512^3).  Giving that block back to
  // the driver made the NEXT large hipMalloc of the process take 1.5 - 2.6 s on this stack (tools/time_malloc.py: 0.2 ms
  // for the first 34 GB, 1.8 s for the same request after a hipFree of 34 GB); left in the block cache it would sit
  // there unused.  It stays with the stencil instead and hosts the tables built after it (slot4, diagpos, cpos, st_loc,
  // tile_verts are views into it; the tile build's scratch is its tail): `take` hands out the next 256 B-aligned piece,
  // or allocates when the arena is absent or full.
  DevArray<uint8_t> arena;
  int64_t arena_used = 0;
  template <typename T>
  void take(DevArray<T>& a, int64_t n)
  {
    const int64_t bytes = ((int64_t)sizeof(T) * (n > 0 ? n : 1) + 255) & ~255LL;
    if (arena.p && arena_used + bytes <= arena.n)
    {
      a.release();
      a.p = reinterpret_cast<T*>(arena.p + arena_used); a.n = n; a.owned = false;
      arena_used += bytes;
    }
    else
      a.alloc(n);
  }
};
constexpr int kRowTile = 16;

// Cell blocks of a space (mesh-static, built on first use by cfx::space_vec_blocks): B consecutive cells and the
// sorted union of their dofs.  A linear form's element vectors are summed per (block, dof) in LDS and leave the
// block as one partial per dof of the union -- coalesced 8 B stores instead of one scattered store (or gather) per
// (cell, local dof) pair -- and a row adds the partials of the few blocks around it.  Every order is static: the
// entries of a dof inside a block in ascending (cell, local index) order, a row's partials in ascending block order.
struct VecBlocks
{
  bool built = false, usable = false;
  int B = 0;                  // cells per block: B * ndofs_cell <= 2048 entries
  int64_t nblocks = 0, u_total = 0;
  DevArray<int64_t> u_off;    // [nblocks+1] first dof of each block's union
  DevArray<uint16_t> slot;    // [ncells*nd] position of entry (cell, j) in its block's list sorted by (dof, entry)
  DevArray<uint16_t> seg;     // [u_total] first sorted position of each dof of the union
  DevArray<int64_t> p_off;    // [ndofs+1] dof -> its (block, dof of the union) pairs
  DevArray<int64_t> p_pos;    // (block << 11) | position in the union, ascending
};
constexpr int kVbEntries = 2048;

} // namespace cfx

struct cfx_mesh_s
{
  int tdim = 0, gdim = 0;
  int64_t nnodes = 0, ncells = 0;
  cfx::DevArray<double> x;     // [nnodes*3]
  cfx::DevArray<int32_t> conn; // [ncells*(tdim+1)]
  cfx::Adjacency v2c;          // vertex -> cells
  const cfx::Adjacency& vertex_cells()
  {
    if (!v2c.built) cfx::build_adjacency(conn.p, ncells, tdim + 1, nnodes, v2c);
    return v2c;
  }
  // cell -> cell across local facet lf ([ncells*(tdim+1)], -1 on the boundary): the mesh's facet connectivity
  // (dolfinx topology.create_connectivity(tdim-1, tdim)), built on first use by cfx::build_cell_neighbours
  cfx::DevArray<int32_t> c2c;
  bool c2c_built = false;
  // generated box / slab mesh (cfx_mesh_create_box / _slab): box_n cubes per edge, Kuhn split, vertex id
  // ix + (n+1)(iy + (n+1) iz): the connectivity is a function of the cell id ("implicit-structured")
  int box_n = 0;
  const cfx::DevArray<int32_t>& cell_neighbours();
  // Vertex summary of every block of kClassBlock consecutive cells (mesh-static, built on the first classification of a
  // level set that lives on the geometry dofmap): the distinct vertices of the block's cells as at most kClassRuns runs
  // of consecutive ids (start, length); nruns < 0: the block has more runs than that (an unordered mesh) and is always
  // classified cell by cell.  A block whose vertices all carry one sign is classified without reading its connectivity.
  cfx::DevArray<int32_t> class_nruns; // [nblocks]
  cfx::DevArray<int2> class_runs;     // [nblocks * kClassRuns]
  // ... and of its kClassSub quarter blocks (kClassSubRuns runs each): a block with vertices on both sides is decided
  // quarter by quarter before any cell is looked at
  cfx::DevArray<int2> class_sub_runs; // [nblocks * kClassSub * kClassSubRuns]
  bool class_built = false;
};
constexpr int kClassBlock = 1024, kClassRuns = 32; // (one wavefront classifies a block: 16 cells per lane)
constexpr int kClassSub = 4, kClassSubRuns = 16;

struct cfx_rules_s
{
  const uint64_t serial = cfx::next_serial(); // identity of this handle (row-plan keys, stale-form detection)
  cfx_rules_s() { cfx::rules_serial_live(serial, true); }
  ~cfx_rules_s() { cfx::rules_serial_live(serial, false); }
  cfx_rules_s(const cfx_rules_s&) = delete;
  cfx_rules_s& operator=(const cfx_rules_s&) = delete;
  cfx_mesh_t mesh = nullptr;
  int tdim = 0, gdim = 0;
  cfx::Count nq, nr; // total points / rules (capacities while a step is open, see cfx::Count)
  cfx::DevArray<double> points, weights;
  cfx::DevArray<float> points_f32, weights_f32; // rounded copies behind cfx_rules_view_get_f32 (made on first use)
  cfx::DevArray<int32_t> offsets, parent_map;
  bool parent_sentinel = false; // parent_map[nr] = -1 (rules made by cfx_runtime_quadrature: walks over "the rules of a cell"
                                // end there and need not load the number of rules while it is still in HBM)
  // rules hosted by facets (cut(level_set, facets, tdim-1)): tdim == mesh tdim - 1, points are coordinates of
  // the host facet's reference simplex spanned by host_verts, parent_map = the caller's facet ids
  int host_width = 0;                 // 0: hosted by cells; 2: (cell, lf) rows; 4: (c0, lf0, c1, lf1) rows
  cfx::DevArray<int32_t> host_rows;   // [nr*host_width]
  cfx::DevArray<int32_t> host_verts;  // [nr*(tdim+1)] mesh vertices of the host facet, in host order
};

struct cfx_cut_s
{
  cfx_mesh_t mesh = nullptr;
  int nls = 0;
  int ls_ndofs_cell = 0;
  int64_t ls_ndofs = 0;
  cfx_cut_options options{};
  cfx::DevArray<int32_t> ls_dofmap;
  std::vector<cfx::DevArray<double>> ls_values;
  cfx::DevArray<int8_t> domain; // [nls*ncells]
  cfx::DevArray<uint8_t> host_mask; // cut(level_set, cells, tdim): 1 on the candidate cells; empty = all cells
  // inside / cut cells per compaction tile of level set 0, counted by the classification itself (empty: not available)
  cfx::DevArray<int32_t> tile_block; // both arrays in one block: one zero fill per classification
  cfx::DevArray<int32_t> tiles_inside, tiles_cut;
  // per block of kClassBlock cells of level set 0: 1 all inside, 2 all outside, 0 mixed (written by the culled
  // classification; empty when the cell-by-cell kernel ran): the selector scan skips what it names
  cfx::DevArray<uint8_t> block_class;
  // sign codes of level set 0 (1 negative, 2 positive, 0 zero: sign_codes_kernel), kept after the classification when
  // the level set lives on the geometry dofmap: a row plan classifies whole row tiles from them (cfx::row_plan)
  cfx::DevArray<uint8_t> codes0;
  // ... and one byte per dof of level set 0, set on the dofs of every cut cell (by the classification kernels that
  // visit the cut cells one by one; touch_valid says whether the last classification did)
  cfx::DevArray<uint8_t> touch0;
  bool touch_valid = false;
  uint64_t gen = 0; // classification number of this cut (provenance of its located lists)
  ~cfx_cut_s() { cfx::provenance_forget_cut(this); }
  std::map<std::string, cfx::DevArray<int32_t>> located;
  std::map<std::string, cfx::DevArray<int32_t>> ghost_rows;
  // facet hosts (cut(level_set, facets, tdim-1), cut.cpp:540-591): the hosts are n_hosts facets of the mesh;
  // ls_dofmap then holds the level-set dofs of each host ([n_hosts*tdim]) and domain has n_hosts entries per level set
  int host_width = 0;                 // 0: the hosts are the mesh cells
  int64_t n_hosts = 0;
  cfx::DevArray<int32_t> host_ids;    // parent_entities
  cfx::DevArray<int32_t> host_rows;   // [n_hosts*host_width]
  cfx::DevArray<int32_t> host_verts;  // [n_hosts*tdim]
  std::map<std::string, cfx::DevArray<int32_t>> located_ids; // located host indices mapped to parent entity ids
  int64_t nhosts() const { return host_width ? n_hosts : mesh->ncells; }
  int host_dim() const { return host_width ? mesh->tdim - 1 : mesh->tdim; }
};

// The previous sparsity pattern of a space (moving-domain loops, spaces whose patterns are built from neighbour lists
// and hash sets: degree 2, vector-valued, DG).  A row of the new pattern whose incident cells kept their SIGNATURE --
// "carries a mark of the form" and "which of its sides are facets of the form" -- couples exactly the columns it coupled
// before: it copies them from the previous pattern instead of going through the hash sets again (cfx::build_pattern).
// The arrays stay with the pattern they belong to while it is alive and move here when it is destroyed.
struct cfx_pattern_cache
{
  struct cfx_pattern_s* live = nullptr;
  cfx::DevArray<int64_t> indptr;
  cfx::DevArray<int32_t> indices;
  cfx::DevArray<uint8_t> sig;   // [ncells] signature of every cell in the step the pattern was built
  std::vector<int> form_key;    // (type, kernel class) of the integrals: the same form structure only
  int64_t nrows = 0;
  bool valid = false;
  void drop();
};

struct cfx_space_s
{
  cfx_mesh_t mesh = nullptr;
  int degree = 1, bs = 1, ndofs_cell = 0;
  int64_t ndofs = 0;
  cfx::DevArray<int32_t> dofmap;
  cfx::Adjacency d2c; // dof -> cells
  std::vector<std::weak_ptr<struct cfx_row_plan>> plans; // plans of the live forms on this space
  cfx::Stencil stencil; // built on first use by cfx::space_stencil()
  cfx::VecBlocks vblocks; // built on first use by cfx::space_vec_blocks()
  // degree 2: the two mesh vertices every dof sits between (a vertex dof: twice its vertex), ascending -- mesh-static,
  // built on first use by cfx::space_dof_verts(); lets a row plan class the dofs of a degree-2 space from the level
  // set's vertex codes (bulk rows)
  cfx::DevArray<int32_t> dof_verts;
  bool dof_verts_built = false, dof_verts_ok = false;
  cfx_pattern_cache pcache;          // previous pattern of this space (row reuse in moving-domain loops)
  ~cfx_space_s() { cfx::step_forget_owner(this); pcache.drop(); }
  bool lists_short_overflow = false; // a short-list row overflowed the 128-slot set once: hashed rows all go wide
  bool long_rows = false; // a sparsity build of this space overflowed the 63-entry row sets: start with the wide kernel
  const cfx::Adjacency& dof_cells()
  {
    // a P1 space whose dofmap aliases the geometry dofmap shares the mesh's vertex->cells table
    if (dofmap.p == mesh->conn.p && ndofs == mesh->nnodes) return mesh->vertex_cells();
    if (!d2c.built) cfx::build_adjacency(dofmap.p, mesh->ncells, ndofs_cell, ndofs, d2c);
    return d2c;
  }
};

struct cfx_integral_dev
{
  int type = 0, kernel = 0, qdegree = 0, point_stride = 0;
  cfx::DevArray<int32_t> entities;
  cfx::Count n_entities;
  cfx_rules_t rules = nullptr;
  uint64_t entities_serial = 0, rules_serial = 0; // identity at form creation (0: caller-owned entity array)
  int64_t n_std = -1; // interior-facet integrals with facet-hosted rules: entities [n_std, n_entities) are the rules' rows (-1: none)
  cfx::DevArray<double> point_data;
  cfx::DevArray<double> coefficient; // dof values of a CFX_F_COEFFICIENT field
  double params[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Row plan of a form: everything the row-centric kernels (sparsity, gather
// assembly, active domain) need that depends on the form's entity lists.
struct cfx_row_plan
{
  bool built = false;
  bool usable = false;             // row-gather assembly is legal for this form
  cfx::DevArray<uint8_t> mark_block; // cellmark | rowmark | special_mark in one block (one zero fill per plan)
  // Bulk rows (P1 on the geometry dofmap, the uncut entities = a located list of the cut whose level set lives on the
  // same dofmap): class of every row from its vertex's sign code and the cut's touch byte -- 1: every cell around the row
  // is an uncut entity (row active, stencil complete, nothing gathered), 0: no cell around it is an entity or cut, 2:
  // look at the marks.  Cell marks then come from the classification bytes, not from a walk over the 10^8-entry list.
  cfx::DevArray<uint8_t> rowcls;
  bool bulk = false;
  int bulk_kind = 0; // 1: P1 on the geometry dofmap (dof = vertex); 2: degree 2 (a dof between two vertices, cfx_space_s::dof_verts)
  uint8_t bulk_bits = 0; // mark bits of the cell integrals whose uncut entities are that list
  cfx::DevArray<uint8_t> cellmark; // bit i: uncut entity of cell integral slot i; bit 4+i: parent of its rules
  cfx::DevArray<uint8_t> cellsig;  // bit 0: cellmark != 0; bit 1 + lf: side lf of the cell is a facet of the form (built on
                                   // first use by cfx::plan_cell_signature: what a row's column set depends on)
  cfx::DevArray<uint8_t> rowmark;  // dof touched by any entity of the form
  cfx::DevArray<int32_t> active_rows;
  cfx::Count n_active_rows;
  cfx::DevArray<int32_t> special_rows; // active rows next to the interface: touched by a runtime-rule cell or a facet
  cfx::Count n_special_rows;
  cfx::DevArray<uint8_t> special_mark; // [ndofs] 1 on the special rows
  cfx::DevArray<int32_t> special_pos;  // [ndofs] position of a special row in special_rows (undefined elsewhere);
                                       // the dof->facets incidence (d2f_offsets) is indexed by that position
  cfx::DevArray<int32_t> plain_rows;   // the other active rows: uncut-cell items only
  cfx::Count n_plain_rows;
  uint64_t serial = 0;                 // identity of this plan (a pattern remembers the plan it was built from)
  // per plain row (built on first use by cfx::plain_row_masks): its stencil mask, and the mark byte that ALL
  // its incident cells carry (0 if they differ or one is unmarked) -- such rows need no mark gathers
  cfx::DevArray<unsigned long long> plain_masks;
  cfx::DevArray<uint8_t> plain_uniform;
  bool plain_masks_built = false;
  // positions in plain_rows where a new row tile (dof id / kRowTile) starts: the work list of the tile kernels
  cfx::DevArray<int32_t> plain_tile_first, plain_tile_id; // ... and the tile's number
  cfx::Count n_plain_tiles;
  // linear forms, P1: the element vectors of the uncut cells are staged in the order the plain rows read them
  // (cfx::plain_vec_offsets): entry k of plain row r lives at vec_t2off[r] - 1 + k, k = position of the cell in the
  // row's dof->cells list.  vec_t2off[dof] = 0 off the plain rows (offsets are stored + 1).  vec_fast: -1 not decided, 0 no, 1 yes
  cfx::DevArray<int32_t> vec_t2off;
  bool vec_t2off_used = false; // plain_vec_offsets wrote segment offsets into it (stored + 1; 0 = none)
  cfx::Count n_vec_odd_rows; // plain rows without a segment (their cells do not all carry the mark): with the special rows
                             // they gather the per-cell records -- the second pass of assemble_vec_rows skips the others
  cfx::Count vec_t2_total;
  int vec_fast = -1;
  uint8_t vec_mark = 0;
  // linear forms by cell block (cfx::vec_block_plan, VecBlocks): the blocks that hold a cell with mark `vb_mark`, and
  // for every block the first of its partials in the step's compact partial array (-1: no such cell, no partials)
  cfx::DevArray<int32_t> vb_active; // blocks with an uncut entity
  cfx::DevArray<int32_t> vb_cut;    // blocks with a rule parent (bit 31: no uncut entity in the block)
  cfx::DevArray<int64_t> vb_base;
  int64_t n_vb_active = 0, n_vb_cut = 0, vb_total = 0;
  int vb_state = -1; // -1 not decided, 0 no, 1 yes
  uint8_t vb_mark = 0;
  bool vb_merged = false; // vb_active lists every block with a marked cell (one pass), vb_cut is empty
  bool any_cells = false;
  // interior facets of all facet integrals, concatenated
  cfx::Count nfacets;
  cfx::DevArray<int32_t> cell_tile_counts; // marked cells per compaction tile of the cells (empty: not counted)
  cfx::DevArray<int64_t> row_tile_counts; // per compaction tile of the dofs: special rows | plain rows << 32
  bool fold_ok = true;                // P1: every facet row shares all dofs but one per cell (continuous space)
  cfx::DevArray<int32_t> facet_rows;  // [nfacets*4]
  cfx::DevArray<uint8_t> facet_slot;  // facet integral slot of each row
  cfx::DevArray<int64_t> d2f_offsets; // dof -> facets incidence
  bool d2f_sorted = false; // built by the sort path: every list already in ascending facet order
  cfx::DevArray<int32_t> d2f;
  int n_cell_slots = 0, n_facet_slots = 0;
  // per cell slot: bitset of its uncut entities + exclusive popcount ranks (entity index lookup)
  cfx::DevArray<int64_t> std_bits[4];
  cfx::DevArray<int32_t> std_rank[4];
  // per cell slot: open-addressing map parent cell -> first rule of the cell (rules of a cell are
  // consecutive); keys -1 = empty, size = mask + 1 >= 2 * distinct parents
  cfx::DevArray<int32_t> rule_key_block; // the key tables of all slots (one 0xff fill)
  cfx::DevArray<int32_t> rule_keys[4], rule_first[4];
  uint32_t rule_mask[4] = {0, 0, 0, 0};
  // cells that host a runtime rule of any cell integral (cellmark & 0xF0), ascending, with the bitset + rank
  // structure that maps a cell to its position in the list (built on first use by cfx::plan_cut_cells): the key of
  // the per-cut-cell tensors of the degree-2 gather
  cfx::DevArray<int32_t> cut_cells;
  int64_t n_cut_cells = 0;
  cfx::DevArray<int64_t> cut_bits;
  cfx::DevArray<int32_t> cut_rank;
  cfx::DevArray<int32_t> cut_first[4]; // per cell slot: first rule of every cut cell of the list (-1: none), [n_cut_cells]
  bool cut_cells_built = false;
  int cell_slot_integral[4] = {0, 0, 0, 0};
  int facet_slot_integral[2] = {0, 0};
  // identity of the entity lists the plan was built from: (integral index, entities ptr, count, rules handle,
  // rules count, serial of the entity block, serial of the rules handle) per integral.  Forms of one space whose
  // cell integrals have the same identity share the plan (e.g. the bilinear and the linear form of one problem).
  std::vector<std::array<int64_t, 7>> key_cells, key_facets;
};

struct cfx_form_s
{
  cfx_space_t V = nullptr;  // test space (the space of a linear form)
  cfx_space_t V1 = nullptr; // trial space: V for square forms, another space for cfx_form_create2 forms
  bool rectangular() const { return V1 != nullptr && V1 != V; }
  int rank = 2;
  std::vector<cfx_integral_dev> integrals;
  std::shared_ptr<cfx_row_plan> plan; // built lazily, possibly shared with another live form
  // complex128 forms: the integrals that share one complex constant as a real form of their own (cfx_c128.hip), kept
  // so that the group's row plan survives from one assembly to the next
  std::map<std::vector<int>, std::unique_ptr<cfx_form_s>> sub_forms;
};

namespace cfx
{
bool assemble_rect_rows(cfx_form_s* a, cfx_pattern_s* P, const int8_t* bc0, const int8_t* bc1, double* values, int* error); // cfx_gather.hip
cfx_row_plan& row_plan(cfx_form_s* a);                                  // cfx_rowasm.hip
void validate_form(const cfx_form_s* a);                                // cfx_rowasm.hip: stale entity lists / rules -> Error
const Stencil& space_stencil(cfx_space_s* V);                           // cfx_rowasm.hip
const Stencil& space_stencil_slotn(cfx_space_s* V);                     // cfx_rowasm.hip
// (counts / maxlen: the sparsity build's row lengths, written by the same kernel when the masks are built now: true)
bool plain_row_masks(cfx_form_s* a, int32_t* counts = nullptr, int* maxlen = nullptr); // cfx_rowasm.hip
void plan_cut_cells(cfx_form_s* a);                                     // cfx_rowasm.hip
const Stencil& space_stencil_tiles(cfx_space_s* V);                     // cfx_rowasm.hip
bool space_dof_verts(cfx_space_s* V);                                   // cfx_rowasm.hip (cfx_space_s::dof_verts)
bool plain_vec_offsets(cfx_form_s* L, uint8_t mark);                    // cfx_rowasm.hip
const VecBlocks& space_vec_blocks(cfx_space_s* V);                      // cfx_rowasm.hip
bool vec_block_plan(cfx_form_s* L, uint8_t mark, bool merged);          // cfx_rowasm.hip
void build_pattern(cfx_form_s* a, cfx_pattern_s* P);                    // cfx_rowasm.hip
void build_pattern_rectangular(cfx_form_s* a, cfx_pattern_s* P);        // cfx_rowasm.hip
void plan_cell_signature(cfx_form_s* a);                                // cfx_rowasm.hip
bool pattern_reuse_ok(cfx_form_s* a);                                   // cfx_rowasm.hip
void pattern_remember(cfx_form_s* a, cfx_pattern_s* P);                 // cfx_rowasm.hip
void prepare_form_tables(cfx_form_s* a);                                // cfx_gather.hip
bool assemble_matrix_rows(cfx_form_s* a, cfx_pattern_s* P, const int8_t* bc0, const int8_t* bc1, double* values,
                          bool fresh = false);
bool assemble_vector_rows(cfx_form_s* L, double* b);
// integrands compiled at run time (cfx_rtc.hip)
bool user_integrand_known(int kernel);
int user_integrand_rank(int kernel);
int user_integrand_kind(int kernel); // 0: cell integrand, 1: interior-facet integrand
void user_stage1_facets(const cfx_form_s* a, const cfx_integral_dev& I, double* out, int64_t only_index = -1);
void user_stage1(const cfx_form_s* a, const cfx_integral_dev& I, bool runtime, double* out, int out_mode, int64_t out_stride,
                 int64_t only_index = -1);
} // namespace cfx

struct cfx_pattern_s
{
  cfx_space_s* cache_owner = nullptr; // the space whose pattern cache points at this pattern (cfx_pattern_cache::live)
  ~cfx_pattern_s();                   // hands the arrays to that cache (cfx_rowasm.hip)
  int64_t nrows = 0;
  cfx::Count nnz;
  int64_t ncols = 0; // = nrows unless the form is rectangular
  int max_row_len = 0; // upper bound on the scalar-dof row length
  int64_t n_hashed_rows = 0, n_reused_rows = 0; // rows that needed a hash set / of which: copied from the space's previous pattern
  uint64_t built_plan = 0; // serial of the plan the pattern was built from (0: rectangular).  Only then are the rows off
                           // that plan's active set known to hold one diagonal entry each (the fused zero fill relies on it)
  uint64_t stencil_plan = 0; // serial of the plan whose plain rows were laid out as stencil subsets (0: none)
  // long-row spaces: the plan's active rows split by row length (<= 64 columns / longer)
  uint64_t split_plan = 0;
  cfx::DevArray<int32_t> short_rows, mid_rows, long_rows; // <= 64 columns, <= 128, longer
  int64_t n_short_rows = 0, n_mid_rows = 0, n_long_rows = 0;
  // degree-2 spaces with neighbour lists: the plain rows that copied their static list (every incident cell an uncut
  // entity of the form's one stiffness integral); short_rows / long_rows then hold the other active rows only
  uint64_t full_plan = 0;
  cfx::DevArray<int32_t> full_rows;
  int64_t n_full_rows = 0;
  int64_t n_full_short = 0; // vector-valued spaces: full_rows = [dofs with at most 32 neighbours | the others] (0: not split)
  // degree-2 scalar spaces: the hashed rows that are not interface rows (plain rows that could not copy their list);
  // odd_plan = serial of the plan they were split for: the interface rows then take ONE gather pass of their own
  uint64_t odd_plan = 0;
  cfx::DevArray<int32_t> odd_rows;
  int64_t n_odd_rows = 0;
  cfx::DevArray<int32_t> rest_rows; // vector-valued spaces: the active rows that are not in full_rows
  int64_t n_rest_rows = 0;
  cfx::DevArray<int64_t> indptr;
  cfx::DevArray<int32_t> indices;
};

struct cfx_aggregation_s
{
  int64_t ncells = 0;
  cfx::DevArray<int32_t> root_cell, aggregate_id, depth;
  cfx::DevArray<double> fraction;
  cfx::DevArray<int32_t> active, cut, interior, well, ill, rootless, pairs;
  int64_t n_pairs = 0;
};

struct cfx_active_s
{
  cfx_space_t V = nullptr;
  // The two indicators of deactivate.h:103-183 are the marks of the form's row plan (cells of every cell integral,
  // dofs touched by any entity): the domain keeps the plan alive and deactivation works from the marks, row tile by
  // row tile.  The LISTS (active cells, inactive dofs: ActiveDomain's public arrays) are compacted from the marks on
  // first request (cfx::active_lists) -- a moving-domain step never asks for them.
  std::shared_ptr<cfx_row_plan> plan;
  cfx::DevArray<uint8_t> cell_indicator; // only when a facet integral has a cell outside every cell integral
  bool lists_built = false;
  cfx::DevArray<int32_t> active_cells, inactive_dofs;
  cfx::Count n_active, n_inactive;
};
namespace cfx
{
void active_lists(cfx_active_s* d);   // cfx_fem.hip
}
