// tools/simt/selftest.cpp -- the emulator's cross-lane operations against their documented results (tests/test_simt_kernels.py).
// `selftest overrun` reads one element past a device buffer and must die at the unmapped page behind it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

__global__ void k_shuffles(int *out)
{
  const int lane = threadIdx.x;
  out[0 * 64 + lane] = __shfl(lane * 3, 5, 64);
  out[1 * 64 + lane] = __shfl_up(lane, 2u, 64);
  out[2 * 64 + lane] = __shfl_xor(lane, 1, 64);
  out[3 * 64 + lane] = __shfl(lane, 1, 16);                      // width 16: source 1 of the lane's own group
  out[4 * 64 + lane] = (int)__popcll(__ballot(lane & 1));
  out[5 * 64 + lane] = __builtin_amdgcn_readlane(lane * 7, 9);
  out[6 * 64 + lane] = __builtin_amdgcn_readfirstlane(lane + 100);
  out[7 * 64 + lane] = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, lane);
}

__global__ void k_dpp(int *out)
{
  const int lane = threadIdx.x;
  out[0 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xF, 0xF, false);   // row_shr:1, no source: old
  out[1 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xF, 0xF, true);    // ... bound_ctrl: 0
  out[2 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x103, 0xF, 0xF, false);   // row_shl:3: lane i reads i + 3
  out[3 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x155, 0xF, 0xF, false);   // row_newbcast:5
  out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x114, 0x5, 0xF, true);    // row_shr:4, rows 0 and 2 only
}

// a ballot inside a divergent region: the lanes inside go first, the others already wait at the loop head
__global__ void k_divergent(int *out)
{
  const int lane = threadIdx.x;
  int rounds = lane & 7, seen = 0;
  bool act = rounds > 0;
  while (__builtin_amdgcn_ballot_w64(act) != 0ull) {
    if (act) {
      MJH_DIVERGENT_SCOPE;
      seen += (int)__popcll(__builtin_amdgcn_ballot_w64(true));   // how many lanes are still inside
      if (--rounds == 0) act = false;
    }
  }
  out[lane] = seen;
}

// four independent rows of 16 with their own trip counts
__global__ void k_rows(int *out)
{
  MJH_WAVE_GROUPS(16);
  const int lane = threadIdx.x, row = lane >> 4;
  int acc = 0;
  for (int i = 0; i <= row * 3; i++) acc += __builtin_amdgcn_update_dpp(0, lane + i, 0x150, 0xF, 0xF, true);   // lane 0 of the row
  out[lane] = acc;
}

__global__ void k_block(int *out, unsigned *counter)
{
  __shared__ int sh[256];
  const int t = threadIdx.x;
  sh[t] = t * 2;
  __syncthreads();
  const int other = sh[255 - t];
  const int any = __syncthreads_or(t == 77 && blockIdx.x == 1);
  out[blockIdx.x * 256 + t] = other + (any ? 1000 : 0);
  if (t == 0) atomicAdd(counter, blockIdx.x + 1u);
  // lock-step order inside a wave: everybody reads, then one lane overwrites
  __shared__ int word[4];
  if ((t & 63) == 0) word[t >> 6] = 5;
  __syncthreads();
  const int w = word[t >> 6];
  MJH_WAVE_SYNC();
  if ((t & 63) == 0) word[t >> 6] = 9;
  out[blockIdx.x * 256 + t] += w * 10000;
}

__global__ void k_overrun(const int *in, int *out, int n) { out[threadIdx.x] = in[n + threadIdx.x]; }

int main(int argc, char **argv)
{
  int *d;
  unsigned *cnt;
  hipMalloc(&d, 8 * 64 * sizeof(int));
  hipMalloc(&cnt, sizeof(unsigned));
  if (argc > 1 && !strcmp(argv[1], "overrun")) {
    int *small;
    hipMalloc(&small, 100 * sizeof(int));
    hipLaunchKernelGGL(k_overrun, dim3(1), dim3(64), 0, 0, small, d, 100);      // small[100..163]: behind the buffer
    printf("the overrun went unnoticed\n");
    return 0;
  }
  int h[8 * 64];
  hipLaunchKernelGGL(k_shuffles, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) {
    CHECK(h[l] == 15, "shfl lane %d: %d", l, h[l]);
    CHECK(h[64 + l] == (l < 2 ? l : l - 2), "shfl_up lane %d: %d", l, h[64 + l]);
    CHECK(h[128 + l] == (l ^ 1), "shfl_xor lane %d: %d", l, h[128 + l]);
    CHECK(h[192 + l] == ((l & ~15) | 1), "shfl width 16 lane %d: %d", l, h[192 + l]);
    CHECK(h[256 + l] == 32, "ballot lane %d: %d", l, h[256 + l]);
    CHECK(h[320 + l] == 63, "readlane lane %d: %d", l, h[320 + l]);
    CHECK(h[384 + l] == 100, "readfirstlane lane %d: %d", l, h[384 + l]);
    CHECK(h[448 + l] == ((l + 1) & 63), "ds_bpermute lane %d: %d", l, h[448 + l]);
  }
  hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) {
    const int in = l & 15, row = l >> 4;
    CHECK(h[l] == (in == 0 ? -1 : l - 1), "row_shr:1 lane %d: %d", l, h[l]);
    CHECK(h[64 + l] == (in == 0 ? 0 : l - 1), "row_shr:1 bound_ctrl lane %d: %d", l, h[64 + l]);
    CHECK(h[128 + l] == (in + 3 < 16 ? l + 3 : -1), "row_shl:3 lane %d: %d", l, h[128 + l]);
    CHECK(h[192 + l] == (row * 16 + 5), "row_newbcast:5 lane %d: %d", l, h[192 + l]);
    CHECK(h[256 + l] == ((row & 1) ? -1 : (in < 4 ? 0 : l - 4)), "row_shr:4 row_mask lane %d: %d", l, h[256 + l]);
  }
  hipLaunchKernelGGL(k_divergent, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 64 * sizeof(int), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) {
    int want = 0;                          // round r (1-based): the lanes with (lane & 7) >= r are inside
    for (int r = 1; r <= (l & 7); r++) want += 8 * (8 - r);
    CHECK(h[l] == want, "divergent ballot lane %d: %d (want %d)", l, h[l], want);
  }
  hipLaunchKernelGGL(k_rows, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 64 * sizeof(int), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) {
    const int row = l >> 4;
    int want = 0;
    for (int i = 0; i <= row * 3; i++) want += row * 16 + i;
    CHECK(h[l] == want, "independent rows lane %d: %d (want %d)", l, h[l], want);
  }
  int *big;
  hipMalloc(&big, 3 * 256 * sizeof(int));
  hipMemset(cnt, 0, sizeof(unsigned));
  hipLaunchKernelGGL(k_block, dim3(3), dim3(256), 0, 0, big, cnt);
  int hb[3 * 256];
  unsigned hc;
  hipMemcpy(hb, big, sizeof(hb), hipMemcpyDeviceToHost);
  hipMemcpy(&hc, cnt, sizeof(hc), hipMemcpyDeviceToHost);
  for (int b = 0; b < 3; b++)
    for (int t = 0; t < 256; t++)
      CHECK(hb[b * 256 + t] == (255 - t) * 2 + (b == 1 ? 1000 : 0) + 50000, "block %d thread %d: %d", b, t, hb[b * 256 + t]);
  CHECK(hc == 6u, "atomics across workgroups: %u", hc);
  hipFree(big); hipFree(d); hipFree(cnt);
  if (!failures) printf("all ok\n");
  return failures ? 1 : 0;
}
