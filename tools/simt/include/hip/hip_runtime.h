// tools/simt: a lock-step wave64 emulator that runs the product's HIP kernel SOURCES on the host.
//
// DEVELOPMENT / TEST INFRASTRUCTURE ONLY.  Nothing under mozjpeg_amd/ loads, links or falls back to what is built from this
// directory; the product library is libmozjpeg_hip.so (gfx950 code objects) and fails loudly without a GPU.  The emulator
// exists so that a kernel edit can be parity-checked against the oracle in the build container (which has no GPU) before
// GPU minutes are spent on it: tools/simt/build_simt.py compiles mozjpeg_amd/csrc/*.hip|*.cpp as plain C++ against THIS
// header (it stands in for <hip/hip_runtime.h>) into tools/simt/_build/libmozjpeg_hip_simt.so.
//
// Execution model (simt.cpp): one fiber per lane, all fibers of a workgroup on one OS thread.  A lane runs until it reaches a
// cross-lane operation (__shfl*, __ballot, readlane, DPP, ds_bpermute), a __syncthreads() or the end of the kernel; when every
// lane of a wave is parked, the lanes waiting at the SAME call site exchange their values (that set is the operation's exec
// mask).  Lanes of one wave parked at two different cross-lane sites at once = a cross-lane operation in divergent control
// flow whose order the emulator cannot know: reported and aborted (the kernels keep such operations in wave-uniform code).
// Memory: hipMalloc'd buffers end at an unmapped page, so an overrun faults here instead of passing silently.
#pragma once
#define MJH_SIMT_HOST 1
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>

// ------------------------------------------------------------------------------------------------ qualifiers
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static const

// ------------------------------------------------------------------------------------------------ vector types
struct dim3 { unsigned x, y, z; constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
#define SIMT_VEC(T, N)                                                                                               \
  struct N##2 { T x, y; }; struct N##3 { T x, y, z; }; struct N##4 { T x, y, z, w; };                                \
  static inline N##2 make_##N##2(T x, T y) { return N##2{x, y}; }                                                    \
  static inline N##3 make_##N##3(T x, T y, T z) { return N##3{x, y, z}; }                                            \
  static inline N##4 make_##N##4(T x, T y, T z, T w) { return N##4{x, y, z, w}; }
SIMT_VEC(int, int) SIMT_VEC(unsigned, uint) SIMT_VEC(float, float) SIMT_VEC(short, short) SIMT_VEC(unsigned short, ushort)
SIMT_VEC(unsigned char, uchar) SIMT_VEC(signed char, char) SIMT_VEC(long long, longlong) SIMT_VEC(unsigned long long, ulonglong)
#undef SIMT_VEC

// ------------------------------------------------------------------------------------------------ the lane that is running
namespace simt {
struct Lane {
  dim3 tid, bid, bdim, gdim;
  int lane, wave, flat;
  int depth;               // how many MJH_DIVERGENT_SCOPEs the lane is inside (see exchange())
  int width;               // lanes per independent group of the wave (64 unless the kernel says MJH_WAVE_GROUPS(16))
  void *dyn_lds;
};
// A cross-lane operation inside a divergent region (`if (act) { ... ballot ... }`): the lanes that skipped the region wait at
// the NEXT operation (the point where the wave reconverges) while the others wait inside.  The emulator cannot derive which
// of the two sites is the inner one, so the source says it: MJH_DIVERGENT_SCOPE at the top of such a region (nothing on the
// device).  Lanes at the greatest depth go first.
struct DivergentScope { DivergentScope(); ~DivergentScope(); };
extern thread_local Lane *cur;
struct Xch { uint64_t v[64]; uint64_t mask; };
const Xch &exchange(uint64_t mine, const char *what, const char *file, int line);   // rendezvous of the wave's lanes parked at this site
int barrier(int pred, const char *file, int line);                                  // __syncthreads[_or]: the OR of pred over the lanes that arrive
void spin();                                                                        // inside a polling loop: let the other lanes run
void launch(const char *name, dim3 grid, dim3 block, size_t dyn_lds, void (*tramp)(void *), void *closure);
template <class F> static void tramp_of(void *p) { (*static_cast<F *>(p))(); }
template <class T> static inline uint64_t to64(T v) { static_assert(sizeof(T) <= 8, ""); uint64_t m = 0; memcpy(&m, &v, sizeof(T)); return m; }
template <class T> static inline T from64(uint64_t m) { T v; memcpy(&v, &m, sizeof(T)); return v; }
inline DivergentScope::DivergentScope() { cur->depth++; }
inline DivergentScope::~DivergentScope() { cur->depth--; }
}
#define MJH_DIVERGENT_SCOPE simt::DivergentScope simt_divergent_scope_
// A kernel whose wave holds independent groups of 16 lanes (one chain per DPP row: cross-lane operations never leave the row,
// the rows follow their own control flow) says so at its top; the emulator then lets every row rendezvous by itself.
#define MJH_WAVE_GROUPS(n) (simt::cur->width = (n))
// lock-step order inside a wave that the device gets for free (all lanes read an LDS word, then one lane overwrites it)
#define MJH_WAVE_SYNC() ((void)simt::exchange(0, "wave_sync", __FILE__, __LINE__))
#define MJH_SCHED_BARRIER() ((void)0)
#define threadIdx (simt::cur->tid)
#define blockIdx (simt::cur->bid)
#define blockDim (simt::cur->bdim)
#define gridDim (simt::cur->gdim)
#define warpSize 64
#define SIMT_HERE const char *file_ = __builtin_FILE(), int line_ = __builtin_LINE()

static inline void __syncthreads(SIMT_HERE) { (void)simt::barrier(0, file_, line_); }
static inline int __syncthreads_or(int pred, SIMT_HERE) { return simt::barrier(pred != 0, file_, line_); }
// dynamic LDS as HIP's own macro spells it (extern __shared__ T name[] on the device)
#define HIP_DYNAMIC_SHARED(type, name) type *name = static_cast<type *>(simt::cur->dyn_lds)

// ------------------------------------------------------------------------------------------------ cross-lane operations
// a source lane that does not take part (switched off or gone): ds_bpermute / __shfl return 0 for it on the hardware
template <class T> static inline T __shfl(T v, int src, int width = 64, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "shfl", file_, line_);
  const int self = simt::cur->lane, s = (self & ~(width - 1)) | (src & (width - 1));
  return simt::from64<T>(((x.mask >> s) & 1) ? x.v[s] : 0);
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "shfl_up", file_, line_);
  const int self = simt::cur->lane;
  if ((unsigned)(self & (width - 1)) < delta) return v;
  const int s = self - (int)delta;
  return simt::from64<T>(((x.mask >> s) & 1) ? x.v[s] : 0);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "shfl_down", file_, line_);
  const int self = simt::cur->lane;
  if ((unsigned)(self & (width - 1)) + delta >= (unsigned)width) return v;
  const int s = self + (int)delta;
  return simt::from64<T>(((x.mask >> s) & 1) ? x.v[s] : 0);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "shfl_xor", file_, line_);
  const int self = simt::cur->lane, s = self ^ m;
  if ((s & ~(width - 1)) != (self & ~(width - 1))) return v;
  return simt::from64<T>(((x.mask >> s) & 1) ? x.v[s] : 0);
}
static inline unsigned long long simt_ballot(bool p, const char *file_, int line_) {
  const simt::Xch &x = simt::exchange(p ? 1u : 0u, "ballot", file_, line_);
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (((x.mask >> i) & 1) && x.v[i]) r |= 1ull << i;
  return r;
}
static inline unsigned long long __ballot(int p, SIMT_HERE) { return simt_ballot(p != 0, file_, line_); }
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p, SIMT_HERE) { return simt_ballot(p, file_, line_); }
// v_readlane reads the register of ANY lane, exec or not; the emulator only has the values of the lanes that take part
int simt_readlane_missing(int lane, const char *file, int line);
static inline int __builtin_amdgcn_readlane(int v, int lane, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "readlane", file_, line_);
  lane &= 63;
  if (!((x.mask >> lane) & 1)) return simt_readlane_missing(lane, file_, line_);
  return simt::from64<int>(x.v[lane]);
}
static inline int __builtin_amdgcn_readfirstlane(int v, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "readfirstlane", file_, line_);
  return simt::from64<int>(x.v[__builtin_ctzll(x.mask)]);
}
static inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(v), "ds_bpermute", file_, line_);
  const int s = (byte_addr >> 2) & 63;
  return ((x.mask >> s) & 1) ? simt::from64<int>(x.v[s]) : 0;
}
// DPP (the controls the kernels use; semantics as verified on the hardware by tools/probes/dpp_probe.hip): a lane whose source
// is outside its row of 16 or switched off keeps `old` (bound_ctrl: takes 0); row_mask / bank_mask switch the write off per
// row of 16 / bank of 4
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, SIMT_HERE) {
  const simt::Xch &x = simt::exchange(simt::to64(src), "dpp", file_, line_);
  const int self = simt::cur->lane, row = self >> 4, inrow = self & 15;
  if (!((row_mask >> (row & 3)) & 1) || !((bank_mask >> ((inrow >> 2) & 3)) & 1)) return old;
  int s = -1;                                   // source lane, -1: out of bounds
  if (ctrl >= 0x000 && ctrl <= 0x0FF) s = (self & ~3) | ((ctrl >> (2 * (self & 3))) & 3);        // quad_perm
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int t = inrow + (ctrl & 15); if (t < 16) s = (row << 4) | t; }   // row_shl:n  (lane i reads i+n)
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int t = inrow - (ctrl & 15); if (t >= 0) s = (row << 4) | t; }   // row_shr:n  (lane i reads i-n)
  else if (ctrl >= 0x121 && ctrl <= 0x12F) s = (row << 4) | ((inrow - (ctrl & 15)) & 15);                           // row_ror:n
  else if (ctrl == 0x140) s = (row << 4) | (15 - inrow);                                                            // row_mirror
  else if (ctrl == 0x141) s = (self & ~7) | (7 - (self & 7));                                                       // row_half_mirror
  else if (ctrl >= 0x150 && ctrl <= 0x15F) s = (row << 4) | (ctrl & 15);                                            // row_newbcast:n
  else { abort(); }
  if (s < 0 || !((x.mask >> s) & 1)) return bound_ctrl ? 0 : old;
  return simt::from64<int>(x.v[s]);
}
static inline void __builtin_amdgcn_wave_barrier(SIMT_HERE) { (void)simt::exchange(0, "wave_barrier", file_, line_); }
// fences order one lane's LDS / memory traffic for the rest of its wave or workgroup: a rendezvous of the wave here
#define __builtin_amdgcn_fence(order, scope) ((void)simt::exchange(0, "fence", __FILE__, __LINE__))
static inline void __builtin_amdgcn_s_sleep(int) { simt::spin(); }
// v_perm_b32: byte i of the result = byte sel[i] of the 8 bytes { b (0-3), a (4-7) }; 0x0c = 0x00, >= 0x0d = 0xff (8-11: sign bytes, unused here)
static inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {
  const unsigned long long both = ((unsigned long long)a << 32) | b;
  unsigned r = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned s = (sel >> (8 * i)) & 0xFFu;
    unsigned v;
    if (s < 8) v = (unsigned)(both >> (8 * s)) & 0xFFu;
    else if (s < 12) v = ((both >> (16 * (s - 8) + 15)) & 1ull) ? 0xFFu : 0u;
    else v = s == 12 ? 0u : 0xFFu;
    r |= v << (8 * i);
  }
  return r;
}
unsigned long long wall_clock64();            // the 100 MHz constant clock
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))

// ------------------------------------------------------------------------------------------------ atomics and bit tricks
template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned *p, int v) { return __atomic_fetch_add(p, (unsigned)v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __mulhi(int a, int b) { return (int)(((long long)a * b) >> 32); }
static inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline float __uint_as_float(unsigned v) { return simt::from64<float>(v); }
static inline float __int_as_float(int v) { return simt::from64<float>((unsigned)v); }
static inline unsigned __float_as_uint(float v) { return (unsigned)simt::to64(v); }
static inline int __float_as_int(float v) { return (int)(unsigned)simt::to64(v); }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
template <class T> static inline T __ldg(const T *p) { return *p; }
using std::min;
using std::max;
static inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
static inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline size_t min(size_t a, int b) { return a < (size_t)b ? a : (size_t)b; }

// ------------------------------------------------------------------------------------------------ runtime API (simt.cpp)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801, hipErrorUnknown = 999 };
typedef struct simt_stream *hipStream_t;
typedef struct simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocNumaUser = 0x20000000 };
static inline hipError_t hipDeviceGetPCIBusId(char *, int, int) { return hipErrorNotSupported; }   // (no placement in the emulator: mjh_numa.cpp)
enum { hipHostRegisterDefault = 0 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer; void *hostPointer; };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
// the virtual-memory API (mjh_guard.cpp's page fences): not emulated, every call reports "not supported"
typedef void *hipMemGenericAllocationHandle_t;
enum { hipMemAllocationTypePinned = 1, hipMemLocationTypeDevice = 1, hipMemAccessFlagsProtReadWrite = 3, hipMemAllocationGranularityMinimum = 0 };
struct hipMemLocation { int type; int id; };
struct hipMemAllocationProp { int type; int requestedHandleTypes; hipMemLocation location; void *win32HandleMetaData; struct { unsigned char compressionType, gpuDirectRDMACapable; unsigned short usage; } allocFlags; };
struct hipMemAccessDesc { hipMemLocation location; int flags; };
typedef void *hipDeviceptr_t;

extern "C++" {
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock = 74 };
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int dev);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi);
hipError_t hipMalloc(void **p, size_t bytes);
template <class T> static inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc((void **)p, bytes); }
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags = 0);
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void **)p, bytes, flags); }
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned flags);
template <class T> static inline hipError_t hipHostGetDevicePointer(T **d, void *h, unsigned flags) { return hipHostGetDevicePointer((void **)d, h, flags); }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k);
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // (the emulator runs every launch to completion)
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
hipError_t hipMemGetAllocationGranularity(size_t *g, const hipMemAllocationProp *p, int opt);
hipError_t hipMemAddressReserve(void **p, size_t n, size_t align, void *addr, unsigned long long flags);
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t n, const hipMemAllocationProp *p, unsigned long long flags);
hipError_t hipMemMap(void *p, size_t n, size_t off, hipMemGenericAllocationHandle_t h, unsigned long long flags);
hipError_t hipMemSetAccess(void *p, size_t n, const hipMemAccessDesc *d, size_t cnt);
hipError_t hipMemUnmap(void *p, size_t n);
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h);
}

// kernel<<<...>>> as the sources spell it: every lane of the grid calls the kernel with the same (by-value) arguments
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                        \
  do {                                                                                                 \
    auto simt_body_ = [=]() { kern(__VA_ARGS__); };                                                    \
    simt::launch(#kern, (grid), (block), (size_t)(lds), &simt::tramp_of<decltype(simt_body_)>, &simt_body_); \
  } while (0)
