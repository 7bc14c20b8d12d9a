// probe of DPP controls on gfx950: prints what each lane reads (hipcc --offload-arch=gfx950 dpp_probe.hip -o dpp_probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __global__ void k(int *out) { const int v = threadIdx.x; out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> void run(const char *name)
{
  int *d, h[64];
  hipMalloc(&d, 256);
  hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  printf("%-16s", name);
  for (int i = 0; i < 36; i++) printf(" %d", h[i]);
  printf("\n");
  hipFree(d);
}
int main()
{
  run<0x150>("row_newbcast:0"); run<0x153>("row_newbcast:3"); run<0x15F>("row_newbcast:15");
  run<0x121>("row_ror:1"); run<0x12F>("row_ror:15"); run<0x101>("row_shl:1"); run<0x111>("row_shr:1");
  return 0;
}
