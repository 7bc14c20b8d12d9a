// mjh_guard.cpp -- the encoder's device allocator and its checking modes (see mjh_guard.h).
// Mode 2/3 is an "electric fence" for device memory: hipMemAddressReserve + hipMemCreate + hipMemMap place a buffer so
// that the page next to it is a hole in the address space; a kernel that strays there dies with a memory access fault
// whose address this file's table (MJH_GUARD_LOG) resolves to "N bytes past the end of <buffer>".
#include "mjh_guard.h"
#include "mjh_numa.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace {
enum { CANARY = 4096, CANARY_BYTE = 0xC5, POISON_BYTE = 0xA5 };
struct Rec {
  void *user = nullptr;
  uint8_t *base = nullptr;        // start of the hipMalloc block (mode 1) / of the reserved range (modes 2, 3)
  size_t bytes = 0, reserved = 0, mapped = 0;
  uint8_t *map = nullptr;         // start of the mapping (modes 2, 3)
  hipMemGenericAllocationHandle_t handle{};
  int mode = 0, device = 0;
  bool host = false;
  char name[48] = "";
};
std::mutex g_m;
std::vector<Rec> g_recs;
FILE *g_log = nullptr;

int read_mode()
{
  const char *v = getenv("MJH_GUARD");
  const int m = v ? atoi(v) : 0;
  if (const char *l = getenv("MJH_GUARD_LOG")) if (m && *l) g_log = fopen(l, "a");
  return m < 0 || m > 3 ? 0 : m;
}

size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

void log_alloc(const Rec &r)
{
  if (!g_log) return;
  fprintf(g_log, "alloc %-24s user %p..%p (%zu bytes) mapping %p..%p mode %d\n", r.name, r.user, (void *)((uint8_t *)r.user + r.bytes), r.bytes,
          (void *)(r.map ? r.map : r.base), (void *)((r.map ? r.map + r.mapped : r.base + r.bytes + 2 * CANARY)), r.mode);
  fflush(g_log);
}
}  // namespace

int mjh_guard_mode()
{
  static const int m = read_mode();
  return m;
}

bool mjh_guard_serial() { return mjh_guard_mode() != 0 && g_log != nullptr; }

void mjh_guard_note(const char *what)
{
  if (!g_log) return;
  fprintf(g_log, "step %s\n", what ? what : "(end)");
  fflush(g_log);
}

hipError_t mjh_guard_alloc(void **p, size_t bytes, const char *name, int device)
{
  const int mode = mjh_guard_mode();
  *p = nullptr;
  if (!bytes) return hipSuccess;
  if (mode == 0) {
    const hipError_t rc = hipMalloc(p, bytes);
    if (rc != hipSuccess) return rc;
    // zero-filled: no kernel may depend on it (modes 1-3 poison instead and the parity tests stay green), but what a first
    // call reads of padding entries it never wrote is then the same in every process
    const hipError_t rm = hipMemsetAsync(*p, 0, bytes, 0);
    return rm != hipSuccess ? rm : hipStreamSynchronize(0);   // (the encoder's streams do not wait for the null stream)
  }
  Rec r;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  r.bytes = bytes; r.mode = mode; r.device = device;
  {   // "(void **)&e->d_planes" -> "d_planes"
    const char *nm = name ? name : "?";
    if (const char *a = strrchr(nm, '>')) nm = a + 1;
    else if (const char *b = strrchr(nm, '&')) nm = b + 1;
    snprintf(r.name, sizeof(r.name), "%s", nm);
  }
  hipError_t rc;
  if (mode == 1) {
    rc = hipMalloc((void **)&r.base, bytes + 2 * CANARY);
    if (rc != hipSuccess) return rc;
    r.user = r.base + CANARY;
    if ((rc = hipMemsetAsync(r.base, CANARY_BYTE, bytes + 2 * CANARY, 0)) != hipSuccess) return rc;
  } else {
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if ((rc = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return rc;
    if (gran < 4096) gran = 4096;
    const size_t padded = round_up(bytes, 16);      // the buffer keeps 16-byte alignment; the padding holds canary bytes
    r.mapped = round_up(padded, gran);
    r.reserved = r.mapped + 2 * gran;               // one unmapped granule on either side
    void *va = nullptr;
    if ((rc = hipMemAddressReserve(&va, r.reserved, gran, nullptr, 0)) != hipSuccess) return rc;
    r.base = (uint8_t *)va;
    r.map = r.base + gran;
    if ((rc = hipMemCreate(&r.handle, r.mapped, &prop, 0)) != hipSuccess) return rc;
    if ((rc = hipMemMap(r.map, r.mapped, 0, r.handle, 0)) != hipSuccess) return rc;
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((rc = hipMemSetAccess(r.map, r.mapped, &acc, 1)) != hipSuccess) return rc;
    r.user = mode == 2 ? r.map + r.mapped - padded : r.map;
    if ((rc = hipMemsetAsync(r.map, CANARY_BYTE, r.mapped, 0)) != hipSuccess) return rc;
  }
  if ((rc = hipMemsetAsync(r.user, POISON_BYTE, bytes, 0)) != hipSuccess) return rc;
  if ((rc = hipStreamSynchronize(0)) != hipSuccess) return rc;
  *p = r.user;
  std::lock_guard<std::mutex> lk(g_m);
  g_recs.push_back(r);
  log_alloc(r);
  return hipSuccess;
}

hipError_t mjh_guard_free(void *p)
{
  if (!p) return hipSuccess;
  if (mjh_guard_mode() == 0) return hipFree(p);
  // (under the lock to the end: a canary check of another thread's encoder walks every live buffer)
  std::lock_guard<std::mutex> lk(g_m);
  size_t i = 0;
  while (i < g_recs.size() && g_recs[i].user != p) i++;
  if (i == g_recs.size()) return hipErrorInvalidValue;
  const Rec r = g_recs[i];
  g_recs.erase(g_recs.begin() + i);
  if (r.mode == 1) return hipFree(r.base);
  (void)hipDeviceSynchronize();
  hipError_t rc = hipMemUnmap(r.map, r.mapped);
  if (rc == hipSuccess) rc = hipMemRelease(r.handle);
  // the address range stays reserved for the life of the process: a later buffer never lands on addresses an earlier one
  // had (a re-used range gave stale reads on this runtime, see guard_input in mjh_encoder.cpp; the address space is 2^47)
  return rc;
}

hipError_t mjh_guard_host_alloc(void **p, size_t bytes, unsigned flags, const char *name)
{
  if (mjh_guard_mode() == 0) {   // the product path: pinned on the NUMA node of the current device (mjh_numa.cpp)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return mjh_numa_host_alloc(p, bytes, flags, dev);
  }
  Rec r;
  r.bytes = bytes; r.mode = 1; r.host = true;
  snprintf(r.name, sizeof(r.name), "%s", name ? name : "?");
  const hipError_t rc = hipHostMalloc((void **)&r.base, bytes + 2 * CANARY, flags);
  if (rc != hipSuccess) return rc;
  memset(r.base, CANARY_BYTE, bytes + 2 * CANARY);
  r.user = r.base + CANARY;
  memset(r.user, POISON_BYTE, bytes);
  *p = r.user;
  std::lock_guard<std::mutex> lk(g_m);
  g_recs.push_back(r);
  log_alloc(r);
  return hipSuccess;
}

hipError_t mjh_guard_host_free(void *p)
{
  if (!p) return hipSuccess;
  if (mjh_guard_mode() == 0) return hipHostFree(p);
  std::lock_guard<std::mutex> lk(g_m);
  size_t i = 0;
  while (i < g_recs.size() && g_recs[i].user != p) i++;
  if (i == g_recs.size()) return hipErrorInvalidValue;
  uint8_t *base = g_recs[i].base;
  g_recs.erase(g_recs.begin() + i);
  return hipHostFree(base);
}

int mjh_guard_check(char *msg, size_t cap)
{
  if (msg && cap) msg[0] = 0;
  if (mjh_guard_mode() == 0) return 0;
  std::lock_guard<std::mutex> lk(g_m);
  const std::vector<Rec> &recs = g_recs;
  int bad = 0;
  size_t used = 0;
  std::vector<uint8_t> h(CANARY);
  for (const Rec &r : recs) {
    // the canary bytes next to the buffer on either side (as many as there are, at most 4 KB each)
    const uint8_t *lo_end = (const uint8_t *)r.user, *hi_begin = lo_end + r.bytes;
    const uint8_t *lo_begin = r.mode == 1 ? r.base : r.map, *hi_end = r.mode == 1 ? r.base + r.bytes + 2 * CANARY : r.map + r.mapped;
    if ((size_t)(lo_end - lo_begin) > CANARY) lo_begin = lo_end - CANARY;
    if ((size_t)(hi_end - hi_begin) > CANARY) hi_end = hi_begin + CANARY;
    for (int side = 0; side < 2; side++) {
      const uint8_t *b = side ? hi_begin : lo_begin, *e = side ? hi_end : lo_end;
      if (e <= b) continue;
      if (r.host) memcpy(h.data(), b, (size_t)(e - b));
      else {
        (void)hipSetDevice(r.device);
        if (hipMemcpy(h.data(), b, (size_t)(e - b), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); continue; }
      }
      size_t first = (size_t)-1, last = 0, cnt = 0;
      for (size_t i = 0; i < (size_t)(e - b); i++)
        if (h[i] != CANARY_BYTE) { if (first == (size_t)-1) first = i; last = i; cnt++; }
      if (!cnt) continue;
      bad++;
      const long d0 = side ? (long)first : (long)first - (long)(e - b), d1 = side ? (long)last : (long)last - (long)(e - b);
      if (msg && used < cap)
        used += (size_t)snprintf(msg + used, cap - used, "%s%s: %zu bytes written %s the buffer (%zu bytes), offsets %+ld..%+ld relative to its %s",
                                 used ? "; " : "", r.name, cnt, side ? "behind" : "in front of", r.bytes, d0, d1, side ? "end" : "start");
      if (g_log) { fprintf(g_log, "DAMAGE %s side %d count %zu\n", r.name, side, cnt); fflush(g_log); }
      // repaired, so that the next check reports new damage only
      if (r.host) memset((void *)b, CANARY_BYTE, (size_t)(e - b));
      else (void)hipMemset((void *)b, CANARY_BYTE, (size_t)(e - b));
    }
  }
  return bad;
}

// Self-test of the tool: one byte read (write = 0) or written at `offset` bytes relative to the END of a 1000-byte buffer
// (negative = relative to its start).  Mode 1 must report a damaged canary for writes within 4 KB, mode 2 must fault for
// offsets >= 8 (1000 is padded to 1008), mode 3 for offsets in front of the start.  Returns what the kernel read, or -1.
__global__ void k_guard_probe(volatile uint8_t *p, long idx, int write, int *out)
{
  if (write) p[idx] = 0x11;
  else *out = p[idx];
}

extern "C" int mjh_debug_guard_selftest(long offset, int write)
{
  uint8_t *buf = nullptr;
  int *out = nullptr, h = -1;
  if (mjh_guard_alloc((void **)&buf, 1000, "selftest", -1) != hipSuccess) return -1;
  if (hipMalloc((void **)&out, 4) != hipSuccess) return -1;
  const long idx = offset >= 0 ? 1000 + offset : offset;
  hipLaunchKernelGGL(k_guard_probe, dim3(1), dim3(1), 0, 0, buf, idx, write, out);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (!write) (void)hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
  else h = 0;
  (void)hipFree(out);
  // (the buffer stays allocated so that a canary check afterwards sees what the probe did)
  return h;
}
