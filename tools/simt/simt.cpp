// tools/simt/simt.cpp -- the scheduler and the stand-in runtime of the lock-step wave64 emulator (see include/hip/hip_runtime.h:
// development / test infrastructure, never part of the product path).
//
// One fiber per lane; the fibers of a workgroup share one OS thread (so `__shared__` = `static thread_local` is the workgroup's
// LDS), workgroups of a launch are spread over a small pool of OS threads.  Every launch runs to completion before it returns:
// streams and events are accepted and ignored (the order in which a correct host program ENQUEUES work is one valid order of
// executing it).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------ context switch (x86-64 SysV)
extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(R"(
  .text
  .hidden simt_switch
  .globl simt_switch
  .type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
  .size simt_switch, .-simt_switch
)");

namespace simt {
thread_local Lane *cur = nullptr;

enum { RUNNABLE = 0, WAVE_WAIT, BAR_WAIT, SPIN, DONE };
struct Fiber {
  void *sp;
  Lane L;
  int state;
  const char *what, *file;
  int line;
  uint64_t pub;
  int depth;            // L.depth when it parked
};
struct Job {
  const char *name;
  dim3 grid, block;
  size_t dyn_lds;
  void (*tramp)(void *);
  void *closure;
};
static const size_t STACK = 512 * 1024;
struct Worker {
  std::vector<Fiber> fibers;
  std::vector<char *> stacks;
  std::vector<Xch> xch;
  std::vector<char> dyn;
  std::vector<int> worder;
  void *sched_sp = nullptr;
  Fiber *running = nullptr;
  const Job *job = nullptr;
  int bar_or = 0;
};
static thread_local Worker *W = nullptr;
static Worker *worker()
{
  if (!W) W = new Worker();
  return W;
}

static void fiber_entry()
{
  Worker *w = W;
  Fiber *f = w->running;
  w->job->tramp(w->job->closure);
  f->state = DONE;
  simt_switch(&f->sp, w->sched_sp);
  __builtin_trap();
}

static void park(int state, uint64_t pub, const char *what, const char *file, int line)
{
  Worker *w = W;
  Fiber *f = w->running;
  f->pub = pub; f->what = what; f->file = file; f->line = line; f->state = state; f->depth = f->L.depth;
  simt_switch(&f->sp, w->sched_sp);
}
const Xch &exchange(uint64_t mine, const char *what, const char *file, int line)
{
  park(WAVE_WAIT, mine, what, file, line);
  return W->xch[W->running->L.wave];
}
int barrier(int pred, const char *file, int line)
{
  park(BAR_WAIT, (uint64_t)pred, "syncthreads", file, line);
  return W->bar_or;
}
void spin() { park(SPIN, 0, "spin", "", 0); }

static bool same_site(const Fiber &a, const Fiber &b)
{
  return a.line == b.line && a.depth == b.depth && a.what == b.what && (a.file == b.file || !strcmp(a.file, b.file));
}
static void dump(const Worker *w, int n, const char *why)
{
  const Job &j = *w->job;
  fprintf(stderr, "simt: %s in %s, workgroup (%u,%u,%u) of (%u,%u,%u), %d threads\n", why, j.name, w->fibers[0].L.bid.x, w->fibers[0].L.bid.y,
          w->fibers[0].L.bid.z, j.grid.x, j.grid.y, j.grid.z, n);
  static const char *names[] = {"runnable", "cross-lane", "barrier", "spin", "done"};
  for (int t = 0; t < n;) {          // runs of lanes in the same state at the same place
    int u = t + 1;
    while (u < n && (u & 63) && w->fibers[u].state == w->fibers[t].state && (w->fibers[t].state == DONE || w->fibers[t].state == RUNNABLE || same_site(w->fibers[u], w->fibers[t]))) u++;
    const Fiber &f = w->fibers[t];
    if (f.state == DONE || f.state == RUNNABLE) fprintf(stderr, "  threads %d..%d: %s\n", t, u - 1, names[f.state]);
    else fprintf(stderr, "  threads %d..%d: %s (%s) at %s:%d\n", t, u - 1, names[f.state], f.what, f.file, f.line);
    t = u;
  }
}

// two groups of lanes of one wave wait at different operations at the same depth of MJH_DIVERGENT_SCOPE: either the two arms
// of an if / else (any order is right) or a divergent region that lacks its annotation (then the first group may be the
// reconvergence point, taken too early with part of the wave).  Reported once per pair of sites; SIMT_STRICT=1 aborts.
static void ambiguous(const Worker *w, int n, const Fiber &a, const Fiber &b)
{
  static std::mutex m;
  static std::vector<std::pair<int, int>> seen;
  static const bool strict = getenv("SIMT_STRICT") && atoi(getenv("SIMT_STRICT"));
  std::lock_guard<std::mutex> g(m);
  for (auto &p : seen) if (p.first == a.line && p.second == b.line) return;
  seen.emplace_back(a.line, b.line);
  fprintf(stderr, "simt: %s: lanes of one wave wait at %s:%d (%s, taken first) and at %s:%d (%s) -- a divergent region without MJH_DIVERGENT_SCOPE?\n",
          w->job->name, a.file, a.line, a.what, b.file, b.line, b.what);
  if (strict) { dump(w, n, "ambiguous order of cross-lane operations"); abort(); }
}

static void run_group(Worker *w, const Job &job, unsigned bx, unsigned by, unsigned bz)
{
  const int n = (int)(job.block.x * job.block.y * job.block.z), nw = (n + 63) / 64;
  if ((int)w->fibers.size() < n) w->fibers.resize(n);
  while ((int)w->stacks.size() < n) {
    char *s = (char *)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (s == MAP_FAILED) { perror("simt: mmap of a lane stack"); abort(); }
    mprotect(s, 4096, PROT_NONE);           // a lane that overruns its stack faults
    w->stacks.push_back(s);
  }
  if ((int)w->xch.size() < nw) w->xch.resize(nw);
  if (w->dyn.size() < job.dyn_lds) w->dyn.resize(job.dyn_lds);
  w->job = &job;
  for (int t = 0; t < n; t++) {
    Fiber &f = w->fibers[t];
    f.L.tid = dim3(t % job.block.x, (t / job.block.x) % job.block.y, t / (job.block.x * job.block.y));
    f.L.bid = dim3(bx, by, bz); f.L.bdim = job.block; f.L.gdim = job.grid;
    f.L.lane = t & 63; f.L.wave = t >> 6; f.L.flat = t; f.L.depth = 0; f.L.width = 64; f.L.dyn_lds = w->dyn.data();
    f.state = RUNNABLE;
    void **top = (void **)(w->stacks[t] + STACK - 16);      // the return address; 16-byte aligned, so rsp % 16 == 8 at fiber_entry
    *top = (void *)&fiber_entry;
    f.sp = (void *)(top - 6);
    memset(f.sp, 0, 6 * sizeof(void *));
  }
  int done = 0;
  long idle_spins = 0;
  // SIMT_ORDER: the order in which the waves of a workgroup get their turn (the hardware promises none: a result that changes
  // with it is a race between waves, i.e. a missing barrier).  0 ascending, 1 descending, 2 a different shuffle every round;
  // +4: the lanes of a wave run from 63 down to 0 between two rendezvous (a difference there = lock-step order the kernel
  // relies on without MJH_WAVE_SYNC: an emulator artefact, not a device bug).
  static const int order_mode = getenv("SIMT_ORDER") ? atoi(getenv("SIMT_ORDER")) : 0;
  unsigned rng = 12345u + bx * 7919u + by * 104729u + bz * 1299709u;
  std::vector<int> &worder = w->worder;
  worder.resize(nw);
  for (int i = 0; i < nw; i++) worder[i] = (order_mode & 3) == 1 ? nw - 1 - i : i;
  while (done < n) {
    bool progress = false, any_spin = false;
    if ((order_mode & 3) == 2)
      for (int i = nw - 1; i > 0; i--) { rng = rng * 1664525u + 1013904223u; std::swap(worder[i], worder[(rng >> 8) % (unsigned)(i + 1)]); }
    for (int wi = 0; wi < nw; wi++) {
      const int wv = worder[wi];
      const int t0 = wv * 64, t1 = std::min(n, t0 + 64);
      for (;;) {
        bool ran = false;
        for (int tt = t0; tt < t1; tt++) {
          const int t = (order_mode & 4) ? t0 + t1 - 1 - tt : tt;
          Fiber &f = w->fibers[t];
          if (f.state != RUNNABLE) continue;
          w->running = &f; cur = &f.L;
          simt_switch(&w->sched_sp, f.sp);
          ran = true;
          if (f.state == DONE) done++;
        }
        progress |= ran;
        // every lane of the wave is parked now; one rendezvous per independent group of lanes (the whole wave unless the kernel
        // declared rows of 16, MJH_WAVE_GROUPS)
        bool released = false;
        Xch &x = w->xch[wv];
        for (int g0 = t0; g0 < t1;) {
          int wd = w->fibers[g0].L.width;
          if (wd != 16 && wd != 32) wd = 64;
          const int g1 = std::min(t1, (g0 & ~(wd - 1)) + wd);
          int nwait = 0, nspin = 0, first = -1;
          for (int t = g0; t < g1; t++) {
            const Fiber &f = w->fibers[t];
            if (f.state == WAVE_WAIT) { if (first < 0 || f.depth > w->fibers[first].depth) first = t; nwait++; }   // the innermost divergent region goes first
            else if (f.state == SPIN) nspin++;
          }
          if (nspin) {            // a polling lane: come back to this wave after the others had their turn
            for (int t = g0; t < g1; t++) if (w->fibers[t].state == SPIN) w->fibers[t].state = RUNNABLE;
            any_spin = true;
          } else if (nwait) {
            uint64_t mask = 0;
            for (int t = g0; t < g1; t++) {
              Fiber &f = w->fibers[t];
              if (f.state != WAVE_WAIT) continue;
              if (!same_site(f, w->fibers[first])) {
                if (f.depth == w->fibers[first].depth) ambiguous(w, n, w->fibers[first], f);
                continue;           // waits at another operation: not part of this one's exec mask
              }
              x.v[t - t0] = f.pub; mask |= 1ull << (t - t0);
              f.state = RUNNABLE;
            }
            const uint64_t span = (wd == 64 ? ~0ull : ((1ull << wd) - 1)) << (g0 - t0);
            x.mask = (x.mask & ~span) | mask;
            released = true;
          }
          g0 = g1;
        }
        if (!released) break;
        progress = true;
      }
    }
    int nbar = 0;
    for (int t = 0; t < n; t++) nbar += w->fibers[t].state == BAR_WAIT;
    if (nbar && nbar + done == n) {
      w->bar_or = 0;
      for (int t = 0; t < n; t++) if (w->fibers[t].state == BAR_WAIT) { w->bar_or |= w->fibers[t].pub != 0; w->fibers[t].state = RUNNABLE; }
      progress = true;
    }
    if (!progress) {
      if (!any_spin || ++idle_spins > 100000000L) { dump(w, n, any_spin ? "polling loop that never ends" : "deadlock"); abort(); }
    } else idle_spins = 0;
  }
  cur = nullptr; w->running = nullptr;
}

// ------------------------------------------------------------------------------------------------ worker pool
struct Pool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  const Job *job = nullptr;
  std::atomic<unsigned long long> next{0};
  unsigned long long total = 0;
  int busy = 0;
  unsigned long long generation = 0;
  int nthreads = 1;
  Pool()
  {
    const char *e = getenv("SIMT_THREADS");
    nthreads = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
  }
  void drain(const Job &j)
  {
    Worker *w = worker();
    const unsigned long long per_z = (unsigned long long)j.grid.x * j.grid.y;
    for (;;) {
      const unsigned long long g = next.fetch_add(1);
      if (g >= total) break;
      run_group(w, j, (unsigned)(g % j.grid.x), (unsigned)((g / j.grid.x) % j.grid.y), (unsigned)(g / per_z));
    }
  }
  void loop()
  {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_work.wait(lk, [&] { return generation != seen; });
      seen = generation;
      const Job *j = job;
      lk.unlock();
      drain(*j);
      lk.lock();
      if (--busy == 0) cv_done.notify_all();
    }
  }
  void run(const Job &j)
  {
    total = (unsigned long long)j.grid.x * j.grid.y * j.grid.z;
    if (!total) return;
    next = 0;
    if (nthreads == 1 || total == 1) { drain(j); return; }
    std::unique_lock<std::mutex> lk(m);
    if (threads.empty())
      for (int i = 0; i < nthreads - 1; i++) { threads.emplace_back([this] { loop(); }); threads.back().detach(); }
    job = &j; busy = (int)threads.size(); generation++;
    cv_work.notify_all();
    lk.unlock();
    drain(j);
    lk.lock();
    cv_done.wait(lk, [&] { return busy == 0; });
  }
};
static Pool *pool() { static Pool *p = new Pool(); return p; }     // leaked on purpose (detached threads)
static std::mutex launch_mutex;                                   // one launch at a time (host threads of the batcher)
static std::atomic<unsigned long long> n_launches{0};

void launch(const char *name, dim3 grid, dim3 block, size_t dyn_lds, void (*tramp)(void *), void *closure)
{
  static const bool trace = getenv("SIMT_TRACE") != nullptr;
  if (cur) { fprintf(stderr, "simt: kernel launch from inside a kernel\n"); abort(); }
  // what the device would refuse (hipErrorInvalidConfiguration; the sources launch without looking at the return value, so on
  // the chip such a launch simply does not happen): a dimension of zero, more than 1024 threads per workgroup, grid.y / grid.z
  // beyond 65535, 2^32 or more threads along x
  {
    const unsigned long long tx = (unsigned long long)grid.x * block.x, nthr = (unsigned long long)block.x * block.y * block.z;
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || nthr == 0 || nthr > 1024 || grid.y > 65535u || grid.z > 65535u || tx >= (1ull << 32)) {
      fprintf(stderr, "simt: launch of %s with grid (%u,%u,%u) block (%u,%u,%u): the device refuses this configuration\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z);
      abort();
    }
  }
  std::lock_guard<std::mutex> g(launch_mutex);
  Job j{name, grid, block, dyn_lds, tramp, closure};
  n_launches++;
  timespec a, b;
  if (trace) clock_gettime(CLOCK_MONOTONIC, &a);
  pool()->run(j);
  if (trace) {
    clock_gettime(CLOCK_MONOTONIC, &b);
    fprintf(stderr, "simt: %-60.60s grid (%u,%u,%u) x %u  %.1f ms\n", name, grid.x, grid.y, grid.z, block.x * block.y * block.z,
            (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6);
  }
}
}   // namespace simt

unsigned long long wall_clock64()
{
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (unsigned long long)t.tv_sec * 100000000ull + (unsigned long long)t.tv_nsec / 10ull;
}
int simt_readlane_missing(int lane, const char *file, int line)
{
  fprintf(stderr, "simt: readlane of lane %d, which does not take part, at %s:%d (the hardware would return that lane's register; the emulator has no value for it)\n", lane, file, line);
  abort();
}

// ------------------------------------------------------------------------------------------------ memory
namespace {
struct Alloc { char *base; size_t total; size_t bytes; bool host; };
std::mutex amx;
std::map<uintptr_t, Alloc> allocs;      // by user pointer
const size_t PG = 4096;

hipError_t fenced_alloc(void **p, size_t bytes, bool host)
{
  static const bool poison = !getenv("SIMT_POISON") || atoi(getenv("SIMT_POISON"));
  const size_t body = (bytes + 15) & ~(size_t)15, span = (body + PG - 1) / PG * PG, total = span + PG;
  char *base = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (base == MAP_FAILED) return hipErrorOutOfMemory;
  mprotect(base + span, PG, PROT_NONE);            // the page behind the buffer: an overrun faults
  char *user = base + span - body;
  if (poison && bytes) memset(user, 0xA5, body);
  std::lock_guard<std::mutex> g(amx);
  allocs[(uintptr_t)user] = Alloc{base, total, bytes, host};
  *p = user;
  return hipSuccess;
}
hipError_t fenced_free(void *p)
{
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> g(amx);
  auto it = allocs.find((uintptr_t)p);
  if (it == allocs.end()) return hipErrorInvalidValue;
  munmap(it->second.base, it->second.total);
  allocs.erase(it);
  return hipSuccess;
}
}   // namespace

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMaxSharedMemoryPerBlock ? 163840 : 0; return hipSuccess; }   // gfx950: 160 KB of LDS per workgroup
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorNotSupported ? "not supported by the emulator" : "error (simt emulator)"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { return fenced_alloc(p, bytes, false); }
hipError_t hipFree(void *p) { return fenced_free(p); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return fenced_alloc(p, bytes, true); }
hipError_t hipHostFree(void *p) { return fenced_free(p); }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
  std::lock_guard<std::mutex> g(amx);
  auto it = allocs.upper_bound((uintptr_t)p);
  if (it == allocs.begin()) return hipErrorInvalidValue;
  --it;
  if ((uintptr_t)p >= it->first + std::max<size_t>(it->second.bytes, 1)) return hipErrorInvalidValue;
  a->type = it->second.host ? hipMemoryTypeHost : hipMemoryTypeDevice;
  a->device = 0; a->devicePointer = (void *)p; a->hostPointer = it->second.host ? (void *)p : nullptr;
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{
  for (size_t y = 0; y < h; y++) memmove((char *)d + y * dp, (const char *)s + y * sp, w);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
struct simt_stream { int id; };
struct simt_event { double t; };
hipError_t hipStreamCreate(hipStream_t *s) { *s = new simt_stream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new simt_event{0.0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  e->t = t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipMemGetAllocationGranularity(size_t *, const hipMemAllocationProp *, int) { return hipErrorNotSupported; }
hipError_t hipMemAddressReserve(void **, size_t, size_t, void *, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *, size_t, const hipMemAllocationProp *, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemMap(void *, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipErrorNotSupported; }
hipError_t hipMemUnmap(void *, size_t) { return hipErrorNotSupported; }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipErrorNotSupported; }

extern "C" unsigned long long simt_launch_count() { return simt::n_launches.load(); }
