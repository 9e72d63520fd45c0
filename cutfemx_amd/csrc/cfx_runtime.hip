// cutfemx_amd: runtime (device/stream/profile), scans, incidence inversion.
#include "cfx_device.h"
#include <chrono>

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
  if (bytes < (1u << 20)) return (bytes + 4095) / 4096 * 4096; // 4 KiB granules
  // 2 MiB granules, and from 16 MiB on eight size classes per octave: the lists of a moving-domain loop drift by a few
  // per cent from step to step, and with fixed granules every step that outgrew its predecessor's block took a new
  // hipMalloc (milliseconds per GB) and left the old block in the cache until the card was full (measured: one
  // create_matrix of 2.6 s in a loop of 39 ms ones at configs[3]).  <= 12.5 % of a block is slack.
  const int lg = 63 - __builtin_clzll((unsigned long long)bytes);
  size_t g = (size_t)1 << (lg - 3);
  if (g < ((size_t)2 << 20)) g = (size_t)2 << 20;
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

// ---------------------------------------------------------------------------
// counts in HBM / sync-free steps (see cfx_common.h)
// ---------------------------------------------------------------------------
namespace
{
struct StepHistory
{
  std::vector<std::string> names;
  std::vector<int64_t> values;
  bool valid = false;
  // the valid record before this one (same sites): a count that grew from there to `values` is expected to grow on --
  // a front that advances steadily outruns a fixed 3 % margin every few steps otherwise
  std::vector<std::string> before_names;
  std::vector<int64_t> before_values;
};
struct StepState
{
  bool active = false, spec = false, mismatch = false;
  std::string key;
  size_t cursor = 0;                 // next site of the step
  std::vector<std::string> names;    // this step's sites ...
  std::vector<int64_t> values;       // ... and their raw totals (filled at the read-back for published sites)
  struct Pending { std::shared_ptr<CountCell> cell; size_t index; };
  std::vector<Pending> pending;
  struct Err { int entry; int code; std::string message; void (*decode)(int); };
  std::vector<Err> errors;
  struct Positive { std::shared_ptr<CountCell> cell; int code; std::string message; };
  std::vector<Positive> positives;
  int next_err = 1, next_slot = kCountFirstSlot;
  uint64_t serial = 0;               // of this step (cells remember it)
  int64_t published = 0, read_back = 0; // sites of this step by kind (diagnostics)
  // state kept ACROSS steps that this step wrote while speculative (a space's remembered pattern): undone when the
  // step turns out void or is aborted -- what a void step built is partial or uninitialised by design
  struct Undo { const void* owner; std::function<void()> fn; };
  std::vector<Undo> undo;
};
StepState& step()
{
  static StepState s;
  return s;
}
std::map<std::string, StepHistory>& histories()
{
  static std::map<std::string, StepHistory> h;
  return h;
}
bool step_debug()
{
  static const bool on = getenv("CFX_STEP_DEBUG") != nullptr;
  return on;
}
double g_margin = 1.03125; // capacity = previous count x margin + slack
int64_t g_slack = 256;

int64_t* count_pool()
{
  // aligned to its size, so that the poison word is the base of any slot address (dev_n)
  static int64_t* pool = []() {
    void* raw = nullptr;
    CFX_HIP(hipMalloc(&raw, 2 * kCountPoolBytes));
    const uintptr_t a = (reinterpret_cast<uintptr_t>(raw) + kCountPoolBytes - 1) & ~(kCountPoolBytes - 1);
    CFX_HIP(hipMemset(reinterpret_cast<void*>(a), 0, kCountPoolBytes));
    return reinterpret_cast<int64_t*>(a);
  }();
  return pool;
}
int64_t* count_mirror()
{
  static int64_t* h = []() {
    void* q = nullptr;
    CFX_HIP(hipHostMalloc(&q, kCountPoolBytes, hipHostMallocDefault));
    return static_cast<int64_t*>(q);
  }();
  return h;
}

__global__ void count_publish_kernel(CountJobs J)
{
  if (threadIdx.x == 0) count_publish(J);
}
// read-back path: the totals side by side for one copy
__global__ void count_gather_kernel(CountJobs J, int64_t* out)
{
  const int k = threadIdx.x;
  if (k < J.n) out[k] = count_read(J.src[k], J.kind[k]) + (J.plus[k] ? *J.plus[k] : 0) + J.add[k];
}
} // namespace

CountCell::~CountCell() = default;

static void step_resolve_pending()
{
  StepState& st = step();
  if (st.active && !st.pending.empty())
  {
    // every count published so far comes to the host (one read-back); the step stays open and later sites publish again
    int64_t* host = count_mirror();
    CFX_HIP(hipMemcpyAsync(host, count_pool(), sizeof(int64_t) * 2 * (size_t)st.next_slot, hipMemcpyDeviceToHost, ctx().main_stream));
    CFX_HIP(hipStreamSynchronize(ctx().main_stream));
    ++sync_counter();
    if (ctx().trace_sync) fprintf(stderr, "cutfemx_amd: read-back by cfx_step_resolve\n");
    // (a void step: the published slots were zeroed and later totals mean nothing -- nothing is resolved, the caller ends
    // the step and repeats it)
    if (st.spec && host[0] != 0)
      throw Error(CFX_ERR_STEP_VOID, "the step is void (a count did not fit the capacity taken from the previous step): end it "
                                     "and repeat it");
    for (auto& p : st.pending)
    {
      if (!p.cell->resolved) { p.cell->value = host[2 * p.cell->slot]; p.cell->resolved = true; }
      st.values[p.index] = host[2 * p.cell->slot + 1];
    }
    st.pending.clear();
  }
}

void step_resolve_all() { step_resolve_pending(); }

int64_t Count::value() const
{
  if (!cell) return exact_n;
  if (!cell->resolved)
  {
    // inside the step that published it: every count published so far comes along in the same round trip (the paths
    // that ask for one size usually ask for the next one right after: 20 -> 16 read-backs per configs[3]-style step)
    if (step().active) step_resolve_pending();
    if (!cell->resolved)
    {
      cell->value = read_scalar(count_pool() + 2 * cell->slot);
      cell->resolved = true;
    }
  }
  return cell->value;
}

DevN Count::devn() const
{
  if (!cell) return DevN(exact_n);
  // A count that was read back in mid-step (a size asked for by the caller, a path that sizes host-side work) still
  // follows its slot while the step that published it is open: the step may turn void AFTER the read-back, and a
  // kernel driven by the exact length would then fill arrays whose capacity belongs to the site that overflowed
  // (the DG loop of tools/soak_fuzz.py: rule points written behind their 528-entry capacity).  min(slot, value) is
  // the value while the step is sound and 0 once it is void.
  if (cell->resolved)
    return (step().active && step().spec && cell->step_serial == step().serial && cell->slot >= 0)
               ? DevN(cell->value, count_pool() + 2 * cell->slot)
               : DevN(cell->value);
  return DevN(cell->cap, count_pool() + 2 * cell->slot);
}

namespace
{
std::map<const void*, Count>& list_registry()
{
  static std::map<const void*, Count> m;
  return m;
}
} // namespace
void list_register(const void* p, const Count& c) { if (p && c.cell) list_registry()[p] = c; }
void list_unregister(const void* p) { list_registry().erase(p); }

static std::map<const void*, ListProvenance>& provenance_registry()
{
  static std::map<const void*, ListProvenance> r;
  return r;
}
void provenance_register(const void* p, const cfx_cut_s* cut, uint64_t gen, int value, int64_t n)
{
  if (!p) return;
  provenance_registry()[p] = ListProvenance{cut, gen, value, dev_block_serial(p), n};
}
const ListProvenance* provenance_lookup(const void* p)
{
  auto& r = provenance_registry();
  auto it = r.find(p);
  if (it == r.end()) return nullptr;
  if (it->second.serial == 0 || dev_block_serial(p) != it->second.serial) { r.erase(it); return nullptr; }
  return &it->second;
}
void provenance_forget_cut(const cfx_cut_s* cut)
{
  auto& r = provenance_registry();
  for (auto it = r.begin(); it != r.end();)
    if (it->second.cut == cut) it = r.erase(it); else ++it;
}
Count list_lookup(const void* p, int64_t n_given)
{
  auto it = list_registry().find(p);
  if (it == list_registry().end()) return Count(n_given);
  // the caller passes back what the ABI gave it: the capacity while the step was open, the exact length afterwards;
  // a shorter prefix of a resolved list is the caller's own choice
  const Count& c = it->second;
  if (n_given == c.cap() || (c.cell && n_given == c.cell->cap)) return c;
  if (!c.pending() && n_given <= c.cap()) return Count(n_given);
  throw Error(CFX_ERR_INVALID_ARGUMENT, "a list whose length is still in HBM (open cfx_step) was passed back with another "
                                        "count: pass the count the library returned, or end the step first");
}

ErrorFlag::ErrorFlag(int code, const char* message, void (*decode)(int))
{
  p = step_error_flag(code, message, decode);
  deferred = p != nullptr;
  if (!p) p = zero_flag();
}
void ErrorFlag::check(int code, const char* message) const
{
  if (deferred) return;
  require(!read_scalar(p), code, message);
}
int64_t count_for_buffer(const Count& c, const void* user) { return is_device_pointer(user) ? c.cap() : c.value(); }
void end_of_call_sync()
{
  if (step().active) return;
  CFX_HIP(hipStreamSynchronize(ctx().stream));
}

Count count_sum(const char* name, const Count& a, const Count& b)
{
  if (!a.pending() && !b.pending())
  {
    const int64_t v = a.cap() + b.cap();
    step_record(name, v);
    return Count(v);
  }
  const DevN da = a.devn(), db = b.devn();
  CountSource s;
  s.src = da.dev; s.kind = kCountI64; s.plus = db.dev;
  s.add = (da.dev ? 0 : da.cap) + (db.dev ? 0 : db.cap);
  Count c;
  count_sites(1, &name, &s, &c);
  return c;
}

void step_record(const char* name, int64_t value)
{
  StepState& st = step();
  if (!st.active) return;
  st.names.push_back(name);
  st.values.push_back(value);
  if (step_debug()) fprintf(stderr, "cutfemx_amd: step site %s recorded: %lld\n", name, (long long)value);
  ++st.cursor;
  ++st.read_back;
}
void step_require_positive(const Count& c, int code, const char* message)
{
  if (c.pending() && step().active) { step().positives.push_back({c.cell, code, message}); return; }
  require(c.value() > 0, code, message);
}

int error_in_step(int code)
{
  StepState& st = step();
  if (!st.active || !st.spec || code == CFX_ERR_STEP_VOID) return code;
  // ordered after the publishing kernels still queued on the engine's stream (a caller's stream is non-blocking: a
  // null-stream copy would not wait for them), as cfx_step_end reads the pool
  int64_t* host = count_mirror();
  if (hipMemcpyAsync(host, count_pool(), sizeof(int64_t), hipMemcpyDeviceToHost, ctx().main_stream) != hipSuccess) return code;
  if (hipStreamSynchronize(ctx().main_stream) != hipSuccess) return code;
  const int64_t poison = host[0];
  if (poison == 0) return code;
  g_last_error = "the step is void (a count did not fit the capacity taken from the previous step): end it and repeat it"
                 " [while void: " + g_last_error + "]";
  return CFX_ERR_STEP_VOID;
}

bool step_speculative() { return step().active && step().spec && !ctx().overlap; }

StepVoidGuard::StepVoidGuard()
{
  if (!step().active || !step().spec) return;
  static int64_t* word = []() {
    void* q = nullptr;
    CFX_HIP(hipHostMalloc(&q, 64, hipHostMallocDefault));
    return static_cast<int64_t*>(q);
  }();
  host = word;
  *host = 0;
  CFX_HIP(hipMemcpyAsync(host, count_pool(), sizeof(int64_t), hipMemcpyDeviceToHost, ctx().stream));
}
void StepVoidGuard::check() const
{
  if (host && *host != 0)
    throw Error(CFX_ERR_STEP_VOID, "the step is void (a count did not fit the capacity taken from the previous step): a size "
                                   "read back after that point means nothing -- end the step and repeat it");
}
void step_on_void(const void* owner, std::function<void()> fn)
{
  if (step().active && step().spec) step().undo.push_back({owner, std::move(fn)});
}
void step_forget_owner(const void* owner)
{
  auto& u = step().undo;
  for (size_t k = u.size(); k-- > 0;)
    if (u[k].owner == owner) u.erase(u.begin() + (long)k);
}
const int64_t* step_poison() { return step_speculative() ? count_pool() : nullptr; }

int* step_error_flag(int code, const char* message, void (*decode)(int))
{
  StepState& st = step();
  if (!st.active || ctx().overlap || st.next_err >= kCountFirstSlot) return nullptr;
  const int e = st.next_err++;
  st.errors.push_back({e, code, message, decode});
  return reinterpret_cast<int*>(count_pool() + 2 * e); // (entries 0..kCountFirstSlot-1 were zeroed by cfx_step_begin)
}

CountPlan::CountPlan(int n_, const char* const* names_, const CountSource* src_) : n(n_)
{
  require(n >= 1 && n <= kMaxCountJobs, CFX_ERR_RUNTIME, "count_sites: too many sites in one call");
  StepState& st = step();
  jobs = std::make_shared<CountJobs>();
  CountJobs& J = *jobs;
  J.n = n;
  for (int k = 0; k < n; ++k)
  {
    names.push_back(names_[k]);
    src.push_back(src_[k]);
    J.src[k] = src_[k].src; J.kind[k] = src_[k].kind; J.mode[k] = src_[k].mode; J.plus[k] = src_[k].plus; J.add[k] = src_[k].add;
  }
  publish = step_speculative() && st.next_slot + n <= kCountEntries;
  if (publish)
  {
    const StepHistory& h = histories()[st.key];
    for (int k = 0; k < n; ++k)
      publish = publish && st.cursor + k < h.names.size() && h.names[st.cursor + k] == names[k];
    if (!publish) st.mismatch = true; // another call sequence than last time: this step's record replaces the history
  }
  if (!publish) return;
  const StepHistory& h = histories()[st.key];
  counts.resize(n);
  for (int k = 0; k < n; ++k)
  {
    const int64_t prev = h.values[st.cursor + k];
    auto cell = std::make_shared<CountCell>();
    cell->slot = st.next_slot++;
    cell->step_serial = st.serial;
    // (a list that was empty stays empty or the step is void: the host then takes the branches of the recorded step)
    if (J.mode[k] == kCountSizeClass)
      cell->cap = prev <= 32 ? 32 : (prev <= 64 ? 64 : (prev <= 128 ? 128 : (prev <= 256 ? 256 : 512)));
    else
    {
      // linear trend over the last two valid steps (never below the last count itself)
      int64_t expect = prev;
      if (h.before_values.size() == h.values.size() && h.before_names[st.cursor + k] == names[k])
      {
        const int64_t grown = prev - h.before_values[st.cursor + k];
        if (grown > 0 && h.before_values[st.cursor + k] > 0) expect = prev + std::min(grown, prev);
      }
      cell->cap = (J.mode[k] == kCountMustEqual || prev == 0) ? prev : (int64_t)((double)expect * g_margin) + g_slack;
    }
    cell->resolved = false;
    cell->hint = prev;
    J.cap[k] = cell->cap; J.slot[k] = cell->slot;
    counts[k].cell = cell;
    st.pending.push_back({cell, st.names.size()});
    st.names.push_back(names[k]);
    st.values.push_back(0);
    if (step_debug())
      fprintf(stderr, "cutfemx_amd: step site %s published, capacity %lld (previous %lld)\n", names[k].c_str(), (long long)J.cap[k], (long long)prev);
  }
  J.pool = count_pool();
  J.n_slots = st.next_slot;
  st.cursor += n;
  st.published += n;
}

void CountPlan::finish(Count* out)
{
  require(!finished, CFX_ERR_RUNTIME, "CountPlan::finish called twice");
  finished = true;
  StepState& st = step();
  CountJobs& J = *jobs;
  if (publish)
  {
    // (not taken along by a scan: its own one-thread launch)
    if (!fused) launch("count_publish", count_publish_kernel, dim3(1), dim3(1), 0, J);
    for (int k = 0; k < n; ++k) out[k] = counts[k];
    return;
  }
  bool plain = true;
  for (int k = 0; k < n; ++k) plain = plain && src[k].src && !src[k].plus && src[k].add == 0;
  int64_t v[kMaxCountJobs];
  if (n == 1 && plain && J.kind[0] == kCountI64)
    v[0] = read_scalar(static_cast<const int64_t*>(J.src[0]));
  else if (n == 1 && plain && J.kind[0] == kCountI32)
    v[0] = read_scalar(static_cast<const int32_t*>(J.src[0]));
  else
  {
    // (the staging words live past the slots of any step: the last entries of the pool)
    int64_t* stage = count_pool() + 2 * kCountEntries - kMaxCountJobs;
    launch("count_gather", count_gather_kernel, dim3(1), dim3(64), 0, J, stage);
    struct Many { int64_t v[kMaxCountJobs]; };
    const Many m = read_scalar(reinterpret_cast<const Many*>(stage));
    for (int k = 0; k < n; ++k) v[k] = m.v[k];
  }
  for (int k = 0; k < n; ++k)
  {
    out[k] = Count(v[k]);
    if (st.active) { st.names.push_back(names[k]); st.values.push_back(v[k]); }
    if (st.active && step_debug()) fprintf(stderr, "cutfemx_amd: step site %s read back: %lld\n", names[k].c_str(), (long long)v[k]);
  }
  if (st.active) { st.cursor += n; st.read_back += n; }
}

void count_sites(int n, const char* const* names, const CountSource* src, Count* out)
{
  CountPlan cp(n, names, src);
  cp.finish(out);
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
  static const bool trace = getenv("CFX_ALLOC_TRACE") != nullptr; // every hipMalloc of the block cache, with its duration
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, want);
  if (trace)
    fprintf(stderr, "cutfemx_amd: hipMalloc %.3f GB: %.1f ms (in use %.1f GB, cached %.1f GB)\n", want / 1073741824.0,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), c.in_use / 1073741824.0,
            c.cached / 1073741824.0);
  if (e != hipSuccess)
  {
    (void)hipGetLastError();
    dev_cache_release(); // out of memory: give the cached blocks back and retry once
    e = hipMalloc(&p, want);
    if (e != hipSuccess)
    {
      (void)hipGetLastError(); // (the failure is reported here: it must not come back from the next launch's check)
      throw Error(CFX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
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
                                                            const Tout* __restrict__ tile_offsets, Tout* out, CountJobs after)
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
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1)
  {
    out[n] = tile_offsets[blockIdx.x] + total;
    if (after.n > 0) count_publish(after);
  }
}

template <typename Tin, typename Tout>
__device__ __forceinline__ void scan_small_body(const Tin* in, int64_t n, Tout* out)
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

template <typename Tin, typename Tout>
__global__ void scan_small_kernel(const Tin* in, int64_t n, Tout* out, CountJobs after)
{
  scan_small_body<Tin, Tout>(in, n, out);
  if (threadIdx.x == 0 && after.n > 0) count_publish(after); // (reads out[n] back: this thread's own store)
}

// two scans of one length in one launch (the lists of a step come in pairs: inside / cut cells, volume / interface
// rules -- and a kernel boundary costs ~10 us): the same block scans A, then B; thread 0 wrote both totals
template <typename Tin, typename Tout>
__global__ void scan_small_pair_kernel(const Tin* inA, const Tin* inB, int64_t n, Tout* outA, Tout* outB, CountJobs after)
{
  scan_small_body<Tin, Tout>(inA, n, outA);
  __syncthreads();
  scan_small_body<Tin, Tout>(inB, n, outB);
  if (threadIdx.x == 0 && after.n > 0) count_publish(after);
}

// Single-pass scan (chained tiles with wave-wide look-back, cfx_device.h): one launch, the input is read once.
// one tile of a chained scan; true for the thread that wrote the grand total out[n]
template <typename Tin, typename Tout>
__device__ __forceinline__ bool scan_chained_tile(const Tin* __restrict__ in, int64_t n, unsigned long long* __restrict__ state,
                                                  const unsigned int tile, Tout* __restrict__ out)
{
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
  const unsigned long long s_prefix = chain_exclusive_prefix(state, tile, (unsigned long long)total);
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
  if ((int64_t)(tile + 1) * kTile >= n && threadIdx.x == kBlock - 1)
  {
    out[n] = (Tout)s_prefix + total;
    return true;
  }
  return false;
}

template <typename Tin, typename Tout>
__global__ void __launch_bounds__(kBlock) scan_chained_kernel(const Tin* __restrict__ in, int64_t n,
                                                              unsigned long long* __restrict__ state,
                                                              unsigned int* __restrict__ ticket, Tout* __restrict__ out,
                                                              CountJobs after)
{
  __shared__ unsigned int s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  if (scan_chained_tile<Tin, Tout>(in, n, state, s_tile, out) && after.n > 0) count_publish(after);
}

// the pair form: a block takes ONE ticket and scans its tile of A, then of B (the predecessors it waits for hold lower
// tickets in both); the block of the last tile wrote both totals, which the published counts may combine
template <typename Tin, typename Tout>
__global__ void __launch_bounds__(kBlock) scan_chained_pair_kernel(const Tin* __restrict__ inA, const Tin* __restrict__ inB,
                                                                   int64_t n, unsigned long long* __restrict__ stateA,
                                                                   unsigned long long* __restrict__ stateB,
                                                                   unsigned int* __restrict__ ticket, Tout* __restrict__ outA,
                                                                   Tout* __restrict__ outB, CountJobs after)
{
  __shared__ unsigned int s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  (void)scan_chained_tile<Tin, Tout>(inA, n, stateA, s_tile, outA);
  __syncthreads();
  if (scan_chained_tile<Tin, Tout>(inB, n, stateB, s_tile, outB) && after.n > 0) count_publish(after);
}


// tile states + ticket of a chained launch: a zeroed slice of a pool that is cleared with one fill when it wraps (a step
// runs a dozen short scans: one memset each otherwise)
ChainState chain_state(int64_t ntiles)
{
  constexpr int64_t kPoolWords = 1 << 20;
  // (one pool per stream: the refill at wrap-around is ordered against earlier users by the stream it runs on)
  struct ScanPool { unsigned long long* pool = nullptr; int64_t next = kPoolWords; };
  static std::map<hipStream_t, ScanPool> pools;
  ChainState ch;
  if (ntiles <= 0 || ntiles + 1 > kPoolWords / 4) return ch;
  ScanPool& sp = pools[ctx().stream];
  if (!sp.pool) sp.pool = static_cast<unsigned long long*>(dev_alloc(sizeof(unsigned long long) * kPoolWords));
  if (sp.next + ntiles + 1 > kPoolWords)
  {
    cfx::dev_fill(sp.pool, 0, sizeof(unsigned long long) * kPoolWords); // stream order keeps earlier users ahead of it
    sp.next = 0;
  }
  ch.state = sp.pool + sp.next;
  ch.ticket = reinterpret_cast<unsigned int*>(sp.pool + sp.next + ntiles);
  sp.next += ntiles + 1;
  return ch;
}

template <typename Tin, typename Tout>
static void scan_impl(const Tin* in, Tout* out, int64_t n, CountPlan* plan = nullptr)
{
  // counts published by the scan's last thread (speculative steps); else an empty job list
  CountJobs after{};
  if (plan && plan->publish && !plan->fused && n > 0) { after = *plan->jobs; plan->fused = true; }
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
    ChainState ch = chain_state(ntiles);
    DevArray<unsigned long long> own;
    if (!ch.state)
    {
      own.alloc(ntiles + 1);
      own.zero();
      ch.state = own.p;
      ch.ticket = reinterpret_cast<unsigned int*>(own.p + ntiles);
    }
    launch("scan_chained", scan_chained_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, in, n, ch.state,
           ch.ticket, out, after);
    return;
  }
  if (ntiles == 1)
  {
    launch("scan_top", scan_small_kernel<Tin, Tout>, dim3(1), dim3(kBlock), 0, in, n, out, after);
    return;
  }
  DevArray<Tout> sums(ntiles), offs(ntiles + 1);
  launch("scan_reduce", scan_reduce_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, in, n, sums.p);
  if (ntiles <= kTile)
    launch("scan_top", scan_small_kernel<Tout, Tout>, dim3(1), dim3(kBlock), 0, sums.p, ntiles, offs.p, CountJobs{});
  else
    scan_impl<Tout, Tout>(sums.p, offs.p, ntiles);
  launch("scan_write", scan_write_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, in, n, offs.p, out, after);
}

// two scans of one length: one launch where one scan would take one (a single tile, or chained tiles); else one after
// the other.  The counts of `plan` are published when both totals are written.
template <typename Tin, typename Tout>
static void scan_pair_impl(const Tin* inA, Tout* outA, const Tin* inB, Tout* outB, int64_t n, CountPlan* plan)
{
  const int64_t ntiles = (n + kTile - 1) / kTile;
  static const int64_t chained_max = getenv("CFX_SCAN_CHAINED_TILES") ? atoll(getenv("CFX_SCAN_CHAINED_TILES")) : 8192;
  static const bool off = getenv("CFX_SCAN_PAIRS") && getenv("CFX_SCAN_PAIRS")[0] == '0';
  constexpr int64_t kPairWords = 1 << 16;
  if (off || n == 0 || ntiles > chained_max || 2 * ntiles + 1 > kPairWords / 4)
  {
    scan_impl<Tin, Tout>(inA, outA, n, nullptr);
    scan_impl<Tin, Tout>(inB, outB, n, plan);
    return;
  }
  CountJobs after{};
  if (plan && plan->publish && !plan->fused) { after = *plan->jobs; plan->fused = true; }
  if (ntiles == 1)
  {
    launch("scan_top", scan_small_pair_kernel<Tin, Tout>, dim3(1), dim3(kBlock), 0, inA, inB, n, outA, outB, after);
    return;
  }
  // tile states of both scans + the ticket: a zeroed slice of a pool of its own (cleared with one fill when it wraps)
  struct PairPool { unsigned long long* pool = nullptr; int64_t next = kPairWords; };
  static std::map<hipStream_t, PairPool> pools;
  PairPool& sp = pools[ctx().stream];
  if (!sp.pool) sp.pool = static_cast<unsigned long long*>(dev_alloc(sizeof(unsigned long long) * kPairWords));
  if (sp.next + 2 * ntiles + 1 > kPairWords)
  {
    cfx::dev_fill(sp.pool, 0, sizeof(unsigned long long) * kPairWords);
    sp.next = 0;
  }
  unsigned long long* state = sp.pool + sp.next;
  sp.next += 2 * ntiles + 1;
  launch("scan_chained", scan_chained_pair_kernel<Tin, Tout>, dim3((unsigned)ntiles), dim3(kBlock), 0, inA, inB, n, state,
         state + ntiles, reinterpret_cast<unsigned int*>(state + 2 * ntiles), outA, outB, after);
}

void exclusive_scan_pair(const int32_t* inA, int64_t* outA, const int32_t* inB, int64_t* outB, int64_t n, CountPlan* after)
{
  scan_pair_impl<int32_t, int64_t>(inA, outA, inB, outB, n, after);
}
void exclusive_scan_pair(const int64_t* inA, int64_t* outA, const int64_t* inB, int64_t* outB, int64_t n, CountPlan* after)
{
  scan_pair_impl<int64_t, int64_t>(inA, outA, inB, outB, n, after);
}

void exclusive_scan(const int32_t* in, int64_t* out, int64_t n, CountPlan* after) { scan_impl<int32_t, int64_t>(in, out, n, after); }
void exclusive_scan(const int32_t* in, int32_t* out, int64_t n, CountPlan* after) { scan_impl<int32_t, int32_t>(in, out, n, after); }
void exclusive_scan(const int64_t* in, int64_t* out, int64_t n, CountPlan* after) { scan_impl<int64_t, int64_t>(in, out, n, after); }

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

// The same two passes with the atomics of a workgroup combined in LDS first.  Consecutive cells of a mesh share most
// of their items (a vertex of a Kuhn mesh sits in 24 tets, about ten of them in the same run of 256 cells), and a
// global integer atomic is a memory-side read-modify-write of its own: 3.2 G of them were 58 + 77 ms of the 512^3
// setup.  A workgroup takes kAdjRun consecutive entries, counts them per distinct item in an LDS hash table
// (<= kAdjRun distinct keys in 2 kAdjRun slots) and issues ONE global atomic per distinct item; in the fill pass that
// atomic reserves the item's slots for the whole workgroup and an entry's place among them is its arrival rank in
// LDS (the lists are sorted afterwards either way).  Meshes without any locality pay the LDS pass on top (~10 %).
__global__ void __launch_bounds__(kBlock) adj_count_lds_kernel(const int32_t* __restrict__ map, int64_t nentries, int32_t* counts)
{
  __shared__ int32_t s_key[kAdjSlots], s_cnt[kAdjSlots];
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock) { s_key[k] = -1; s_cnt[k] = 0; }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kAdjRun;
#pragma unroll
  for (int q = 0; q < kAdjPer; ++q)
  {
    const int64_t i = base + q * kBlock + threadIdx.x;
    int rank;
    if (i < nentries) (void)adj_lds_insert(s_key, s_cnt, map[i], rank);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock)
    if (s_key[k] >= 0) atomicAdd(&counts[s_key[k]], s_cnt[k]);
}

__global__ void __launch_bounds__(kBlock) adj_fill_lds_kernel(const int32_t* __restrict__ map, int64_t nentries, int width,
                                                              const int64_t* __restrict__ offsets, int32_t* cursor,
                                                              int32_t* cells)
{
  __shared__ int32_t s_key[kAdjSlots], s_cnt[kAdjSlots];
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock) { s_key[k] = -1; s_cnt[k] = 0; }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kAdjRun;
  int32_t item[kAdjPer];
  int slot[kAdjPer], rank[kAdjPer];
#pragma unroll
  for (int q = 0; q < kAdjPer; ++q)
  {
    const int64_t i = base + q * kBlock + threadIdx.x;
    item[q] = i < nentries ? map[i] : -1;
    slot[q] = 0; rank[q] = 0;
    if (item[q] >= 0) slot[q] = adj_lds_insert(s_key, s_cnt, item[q], rank[q]);
  }
  __syncthreads();
  // the workgroup's slots in every item's list: the count becomes the first position
  for (int k = threadIdx.x; k < kAdjSlots; k += kBlock)
    if (s_key[k] >= 0) s_cnt[k] = atomicAdd(&cursor[s_key[k]], s_cnt[k]);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kAdjPer; ++q)
  {
    const int64_t i = base + q * kBlock + threadIdx.x;
    if (item[q] >= 0) cells[offsets[item[q]] + s_cnt[slot[q]] + rank[q]] = (int32_t)(i / width);
  }
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
  const char* av = getenv("CFX_ADJ_LDS"); // '0': one global atomic per entry (the form of rounds 1-3)
  const bool lds = !(av && av[0] == '0');
  const dim3 run_grid((unsigned)((nentries + kAdjRun - 1) / kAdjRun));
  require((nentries + kAdjRun - 1) / kAdjRun < 2147483647LL, CFX_ERR_RUNTIME, "grid too large");
  if (lds) launch("adj_count", adj_count_lds_kernel, run_grid, dim3(kBlock), 0, map, nentries, counts.p);
  else launch("adj_count", adj_count_kernel, grid_for(nentries), dim3(kBlock), 0, map, nentries, counts.p);
  adj.offsets.alloc(nitems + 1);
  exclusive_scan(counts.p, adj.offsets.p, nitems);
  adj.cells.alloc(nentries);
  counts.zero();
  if (lds) launch("adj_fill", adj_fill_lds_kernel, run_grid, dim3(kBlock), 0, map, nentries, width, adj.offsets.p, counts.p, adj.cells.p);
  else launch("adj_fill", adj_fill_kernel, grid_for(nentries), dim3(kBlock), 0, map, nentries, width, adj.offsets.p, counts.p, adj.cells.p);
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

// two fills in one launch (the mark block and the rule-key tables of a row plan: 0x00 and 0xff)
__global__ void __launch_bounds__(kBlock) fill16_pair_kernel(unsigned char* __restrict__ pa, size_t bytes_a, unsigned word_a,
                                                             unsigned blocks_a, unsigned char* __restrict__ pb, size_t bytes_b,
                                                             unsigned word_b)
{
  const bool second = blockIdx.x >= blocks_a;
  unsigned char* p = second ? pb : pa;
  const size_t bytes = second ? bytes_b : bytes_a;
  const unsigned word = second ? word_b : word_a;
  const unsigned blk = second ? blockIdx.x - blocks_a : blockIdx.x;
  const int64_t n16 = (int64_t)(bytes / 16);
  const uint4 v = make_uint4(word, word, word, word);
  const int64_t base = (int64_t)blk * (kBlock * 4) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const int64_t i = base + (int64_t)k * kBlock;
    if (i < n16) reinterpret_cast<uint4*>(p)[i] = v;
  }
  if (blk == 0 && threadIdx.x < (bytes & 15)) p[(size_t)n16 * 16 + threadIdx.x] = (unsigned char)word;
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
  static const bool trace = getenv("CFX_LAUNCH_TRACE") != nullptr;
  if (trace) fprintf(stderr, "cutfemx_amd: fill of %zu bytes\n", bytes);
  launch("fill", fill16_kernel, grid_for(std::max<int64_t>(n16, 1), kBlock * 4), dim3(kBlock), 0,
         static_cast<unsigned char*>(p), bytes, word);
}

void dev_fill2(void* pa, int byte_a, size_t bytes_a, void* pb, int byte_b, size_t bytes_b)
{
  if (bytes_a == 0 || bytes_b == 0 || ((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 15) != 0)
  {
    dev_fill(pa, byte_a, bytes_a);
    dev_fill(pb, byte_b, bytes_b);
    return;
  }
  auto word = [](int byte) { const unsigned b = (unsigned)byte & 0xffu; return b | (b << 8) | (b << 16) | (b << 24); };
  const dim3 ga = grid_for(std::max<int64_t>((int64_t)(bytes_a / 16), 1), kBlock * 4);
  const dim3 gb = grid_for(std::max<int64_t>((int64_t)(bytes_b / 16), 1), kBlock * 4);
  launch("fill", fill16_pair_kernel, dim3(ga.x + gb.x), dim3(kBlock), 0, static_cast<unsigned char*>(pa), bytes_a, word(byte_a),
         ga.x, static_cast<unsigned char*>(pb), bytes_b, word(byte_b));
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

int cfx_step_begin(const char* key)
{
  CFX_API_BEGIN
  ctx().ensure();
  StepState& st = step();
  require(!st.active, CFX_ERR_RUNTIME, "cfx_step_begin: the previous step was not ended");
  require(!ctx().overlap, CFX_ERR_RUNTIME, "cfx_step_begin inside an overlap section");
  st = StepState();
  static uint64_t step_counter = 0;
  st.serial = ++step_counter;
  st.key = key ? key : "";
  st.active = true;
  const char* off = getenv("CFX_STEP_SPECULATE");
  st.spec = histories()[st.key].valid && !(off && off[0] == '0');
  // poison word + error words of the step (entries below the first slot): one fill
  dev_fill(count_pool(), 0, sizeof(int64_t) * 2 * kCountFirstSlot);
  CFX_API_END
}

int cfx_step_end(int* redo, int64_t* published, int64_t* read_back)
{
  CFX_API_BEGIN
  StepState& st = step();
  require(st.active, CFX_ERR_RUNTIME, "cfx_step_end without cfx_step_begin");
  int64_t* host = count_mirror();
  // the one read-back of a step: poison, error words, every slot (published + raw)
  CFX_HIP(hipMemcpyAsync(host, count_pool(), sizeof(int64_t) * 2 * (size_t)st.next_slot, hipMemcpyDeviceToHost, ctx().main_stream));
  CFX_HIP(hipStreamSynchronize(ctx().main_stream));
  ++sync_counter();
  if (ctx().trace_sync) fprintf(stderr, "cutfemx_amd: read-back at cfx_step_end (%lld sites published, %lld read back)\n",
                                (long long)st.published, (long long)st.read_back);
  const bool poisoned = host[0] != 0;
  if (step_debug())
  {
    fprintf(stderr, "cutfemx_amd: step '%s' ends: %s\n", st.key.c_str(), poisoned ? "VOID (a count did not fit)" : "ok");
    for (auto& p : st.pending)
      fprintf(stderr, "  %s: capacity %lld, value %lld\n", st.names[p.index].c_str(), (long long)p.cell->cap, (long long)host[2 * p.cell->slot + 1]);
  }
  for (auto& p : st.pending)
  {
    if (!p.cell->resolved) { p.cell->value = host[2 * p.cell->slot]; p.cell->resolved = true; }
    st.values[p.index] = host[2 * p.cell->slot + 1];
  }
  StepHistory& h = histories()[st.key];
  if (h.valid) { h.before_names = std::move(h.names); h.before_values = std::move(h.values); } // (a void step's record is not kept)
  h.names = st.names;
  h.values = st.values;
  h.valid = !poisoned; // a void step: the repeat sizes everything by read-backs and records afresh
  if (redo) *redo = poisoned ? 1 : 0;
  if (published) *published = st.published;
  if (read_back) *read_back = st.read_back;
  const std::vector<StepState::Err> errors = std::move(st.errors);
  const std::vector<StepState::Positive> positives = std::move(st.positives);
  const std::vector<StepState::Undo> undo = std::move(st.undo);
  st = StepState();
  if (poisoned)
    for (const auto& u : undo) u.fn();
  if (!poisoned)
  {
    for (const auto& e : errors)
      if (host[2 * e.entry] != 0)
      {
        if (e.decode) e.decode((int)host[2 * e.entry]);
        throw Error(e.code, e.message);
      }
    for (const auto& q : positives)
      if (q.cell->value <= 0) throw Error(q.code, q.message);
  }
  CFX_API_END
}

int cfx_step_resolve(void)
{
  CFX_API_BEGIN
  cfx::step_resolve_all();
  CFX_API_END
}

int cfx_step_abort(void)
{
  CFX_API_BEGIN
  StepState& st = step();
  if (st.active)
  {
    CFX_HIP(hipStreamSynchronize(ctx().main_stream));
    int64_t* host = count_mirror();
    CFX_HIP(hipMemcpy(host, count_pool(), sizeof(int64_t) * 2 * (size_t)st.next_slot, hipMemcpyDeviceToHost));
    for (auto& p : st.pending)
      if (!p.cell->resolved) { p.cell->value = host[2 * p.cell->slot]; p.cell->resolved = true; }
    histories()[st.key].valid = false;
    const std::vector<StepState::Undo> undo = std::move(st.undo);
    st = StepState();
    for (const auto& u : undo) u.fn();
  }
  CFX_API_END
}

int cfx_step_set_margin(double factor, int64_t slack)
{
  CFX_API_BEGIN
  require(factor > 0.0 && slack >= 0, CFX_ERR_INVALID_ARGUMENT, "cfx_step_set_margin: factor > 0, slack >= 0");
  g_margin = factor;
  g_slack = slack;
  CFX_API_END
}

int cfx_step_forget(const char* key)
{
  CFX_API_BEGIN
  if (key) histories().erase(key); else histories().clear();
  CFX_API_END
}

int cfx_list_count(const void* list, int64_t n_given, int64_t* n)
{
  CFX_API_BEGIN
  require(n != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_list_count: null argument");
  // a list handed out with its length still in HBM: the capacity while that step is open, the exact length once it has
  // ended; any other pointer (a list with an exact length, the caller's own array, a list that is gone): n_given.
  // Nothing is recomputed.
  *n = n_given;
  auto it = list_registry().find(list);
  if (list != nullptr && it != list_registry().end()) *n = it->second.pending() ? it->second.cap() : it->second.value();
  CFX_API_END
}

int cfx_sync_count(int64_t* n)
{
  CFX_API_BEGIN
  require(n != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_sync_count: null argument");
  *n = sync_counter();
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
