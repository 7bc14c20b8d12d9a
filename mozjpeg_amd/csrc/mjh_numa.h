// mjh_numa.h -- NUMA placement of a device's host side (mjh_numa.cpp)
#ifndef MJH_NUMA_H
#define MJH_NUMA_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
#include <stddef.h>
#include <hip/hip_runtime.h>
#ifdef __cplusplus
extern "C" {
#endif
int mjh_numa_parse_cpulist(const char *s, cpu_set_t *out);   // "0-15,64-79" -> set; returns the number of CPUs or -1
int mjh_numa_node_of_device(int dev);                        // -1: unknown / placement off (MJH_NUMA=0)
int mjh_numa_bind_thread(int dev);                           // calling thread -> the CPUs of the device's node; returns the node or -1
hipError_t mjh_numa_host_alloc(void **ptr, size_t bytes, unsigned flags, int dev);
int mjh_numa_describe(int dev, char *buf, size_t n);
#ifdef __cplusplus
}
#endif
#endif
