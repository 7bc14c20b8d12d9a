// mjh_guard.h -- device allocations of the encoder and their checking modes (MJH_GUARD, read once per process):
//   0  plain: hipMalloc of exactly the bytes asked for, zero-filled
//   1  canaries: 4 KB of a known pattern in front of and behind every buffer, the buffer itself poisoned (0xA5);
//      mjh_guard_check() compares the patterns (stray WRITES show up as damaged canaries, reads of memory nobody
//      wrote show up as output that differs from the oracle's)
//   2  fence behind: every buffer is mapped through the virtual-memory API so that its last byte is the last byte
//      of its mapping and the page behind it is not mapped: a read or write past the end FAULTS in the kernel
//      that does it (the buffer is poisoned as in mode 1)
//   3  fence in front: the same with the buffer's first byte at the start of its mapping (underruns)
// In every mode but 0 the schedule can log its steps and synchronise after each (MJH_GUARD_LOG=file), so that the
// last line of the log names the step a fault happened in.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

int mjh_guard_mode();
hipError_t mjh_guard_alloc(void **p, size_t bytes, const char *name, int device);
hipError_t mjh_guard_free(void *p);
// pinned host memory the DEVICE writes (result arenas): modes 1-3 put 4 KB canaries around it, compared by mjh_guard_check
hipError_t mjh_guard_host_alloc(void **p, size_t bytes, unsigned flags, const char *name);
hipError_t mjh_guard_host_free(void *p);
// canary comparison of every live allocation (modes 1-3: modes 2/3 keep canaries in the alignment padding);
// returns the number of damaged allocations, a description of the first few in msg
int mjh_guard_check(char *msg, size_t cap);
// step log (no-op without MJH_GUARD_LOG): `what` is written and flushed BEFORE the step is queued
bool mjh_guard_serial();
void mjh_guard_note(const char *what);
