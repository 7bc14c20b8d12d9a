// mjh_numa.cpp -- where the HOST side of one device lives (a batch of independent images is dealt over the GPUs
// of one node; the reference's contract is one compress object per thread, libjpeg.txt:2198-2200).  At 8 GPUs the host -> host
// path reads 8 x ~54 GB/s of pinned pixels: a two-socket host only delivers that when every device's staging buffers lie
// in the memory of the socket its PCIe root hangs off, and the threads that fill them run there.  Nothing here touches
// the device; everything degrades to "no placement" when the sysfs files or the system calls are not available (containers).
#include "mjh_numa.h"

#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <mutex>

namespace {
constexpr int MAXDEV = 64;
struct Place { bool known = false; int node = -1; cpu_set_t cpus; int ncpus = 0; };
Place g_place[MAXDEV];
std::mutex g_mu;

bool read_line(const char *path, char *buf, size_t n)
{
  FILE *f = fopen(path, "r");
  if (!f) return false;
  const bool ok = fgets(buf, (int)n, f) != nullptr;
  fclose(f);
  return ok;
}

const Place &place_of(int dev)
{
  static Place none;
  if (dev < 0 || dev >= MAXDEV) return none;
  std::lock_guard<std::mutex> lk(g_mu);
  Place &p = g_place[dev];
  if (p.known) return p;
  p.known = true;
  CPU_ZERO(&p.cpus);
  if (const char *v = getenv("MJH_NUMA")) if (atoi(v) == 0 && *v == '0') return p;   // MJH_NUMA=0: no placement
  char bus[64] = { 0 }, path[160], line[4096];
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) { (void)hipGetLastError(); return p; }
  for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');   // sysfs spells the address in lower case
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  if (!read_line(path, line, sizeof(line))) return p;
  const int node = atoi(line);
  if (node < 0) return p;                                   // -1: the platform reports no affinity
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  if (!read_line(path, line, sizeof(line))) return p;
  cpu_set_t want, allowed;
  if (mjh_numa_parse_cpulist(line, &want) <= 0) return p;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return p;
  CPU_AND(&p.cpus, &want, &allowed);                        // never widen what the process was given (cgroups, taskset)
  p.ncpus = CPU_COUNT(&p.cpus);
  if (p.ncpus > 0) p.node = node;
  return p;
}

long set_mempolicy_preferred(int node)   // MPOL_PREFERRED = 1, MPOL_DEFAULT = 0 (linux/mempolicy.h; no libnuma in the image)
{
#ifdef SYS_set_mempolicy
  if (node < 0) return syscall(SYS_set_mempolicy, 0, nullptr, 0);
  unsigned long mask[16] = { 0 };
  if (node >= (int)(sizeof(mask) * 8)) return -1;
  mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
  return syscall(SYS_set_mempolicy, 1, mask, sizeof(mask) * 8);
#else
  (void)node;
  return -1;
#endif
}
}  // namespace

extern "C" int mjh_numa_parse_cpulist(const char *s, cpu_set_t *out)
{
  CPU_ZERO(out);
  int n = 0;
  while (s && *s) {
    while (*s == ' ' || *s == ',' || *s == '\n' || *s == '\t') s++;
    if (*s < '0' || *s > '9') break;
    char *end;
    long a = strtol(s, &end, 10), b = a;
    if (*end == '-') { s = end + 1; if (*s < '0' || *s > '9') return -1; b = strtol(s, &end, 10); }
    if (b < a || b >= CPU_SETSIZE) return -1;
    for (long c = a; c <= b; c++) if (!CPU_ISSET(c, out)) { CPU_SET(c, out); n++; }
    s = end;
  }
  return n;
}

extern "C" int mjh_numa_node_of_device(int dev) { return place_of(dev).node; }

extern "C" int mjh_numa_bind_thread(int dev)
{
  const Place &p = place_of(dev);
  if (p.node < 0) return -1;
  return pthread_setaffinity_np(pthread_self(), sizeof(p.cpus), &p.cpus) == 0 ? p.node : -1;
}

// Pinned host memory for device `dev`'s staging: allocated under a "prefer the device's node" policy of the calling thread with
// hipHostMallocNumaUser (the runtime then leaves the placement to that policy); without a known node, or when the policy
// call is refused, the runtime's own choice (hipHostMallocDefault places near the current device).
extern "C" hipError_t mjh_numa_host_alloc(void **ptr, size_t bytes, unsigned flags, int dev)
{
  const Place &p = place_of(dev);
  // the calling thread may be the application's (mjh_host_alloc, the first mjh_encode_host): its own policy (numactl --interleave,
  // set_mempolicy of its own) is read first and put back afterwards; if it cannot be read the placement is skipped
  int old_mode = 0;
  unsigned long old_mask[16] = { 0 };
#ifdef SYS_get_mempolicy
  const bool have_old = syscall(SYS_get_mempolicy, &old_mode, old_mask, sizeof(old_mask) * 8, nullptr, 0) == 0;
#else
  const bool have_old = false;
#endif
  if (p.node >= 0 && have_old && set_mempolicy_preferred(p.node) == 0) {
    const hipError_t rc = hipHostMalloc(ptr, bytes, flags | hipHostMallocNumaUser);
    if (rc == hipSuccess) {
      // first touch under the policy: one write per page (pinning faults the pages in, this makes it explicit)
      volatile char *c = (volatile char *)*ptr;
      const long pg = sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096;
      for (size_t o = 0; o < bytes; o += (size_t)pg) c[o] = 0;
    }
#ifdef SYS_set_mempolicy
    if (old_mode == 0) (void)syscall(SYS_set_mempolicy, 0, nullptr, 0);
    else (void)syscall(SYS_set_mempolicy, old_mode, old_mask, sizeof(old_mask) * 8);
#endif
    if (rc == hipSuccess) return rc;
    (void)hipGetLastError();
  }
  return hipHostMalloc(ptr, bytes, flags);
}

extern "C" int mjh_numa_describe(int dev, char *buf, size_t n)
{
  const Place &p = place_of(dev);
  if (p.node < 0) return snprintf(buf, n, "device %d: no NUMA placement (node unknown or MJH_NUMA=0)", dev);
  return snprintf(buf, n, "device %d: NUMA node %d, %d usable CPUs; staging pinned under MPOL_PREFERRED(node %d), host threads bound to the node",
                  dev, p.node, p.ncpus, p.node);
}
