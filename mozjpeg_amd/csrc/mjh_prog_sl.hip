// mjh_prog_sl.hip -- opt-in variants of two kernels of the progressive parallel chain (MJH_PP_SKIPLOW=1)
//
// The first-pass AC scans of the scan search walk a block's compact record (its non-zeros in position order) once per
// candidate scan.  For the UPPER band of a frequency split (Ss = 3 / 6 / 9 / 13 / 19 ... 63: half of the candidates) most of
// the block's non-zeros lie below Ss; the default walk (pp_band_nonzeros) visits and drops them one by one -- about a third
// of all visits of the search at q85.  The SKIPLOW form takes them out of the mask with one popcount, tests the rest with one
// unsigned compare per value, and does not load bursts no lane of the wave needs.  Same calls of the visitor in the same
// order, so statistics, sizes and bits are those of the default kernels (jcphuff.c:648-764).
//
// A translation unit of its own, like mjh_sorted.hip and for the same reason: a second kernel with k_pp_emit's LDS variables
// in mjh_prog.hip changed k_pp_emit's machine code (two instructions of the workgroup scan's addressing), and the default
// kernels are to stay, instruction for instruction, what was validated and profiled on the chip (tools/kernel_isa.py).
// mjh_prog.hip, included with MJH_TU_PROG_SL, leaves its device functions and kernel templates only.
// Never timed on the chip (round 4 ended without GPU minutes); bit-exact in the emulator (tools/simt).
#define MJH_TU_PROG_SL 1
#include "mjh_prog.hip"

__global__ void __launch_bounds__(256)
k_pp_emit_sl(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl,
             const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
             unsigned *__restrict__ pool, size_t pool_words_per_image, MjhProgPE pe)
{
  constexpr bool SKIPLOW = true;
#include "mjh_pp_emit.inc"
}

void mjh_launch_pp_stats_sl(const MjhConst &C, const void *scans, const int *list, const void *ctl, const void *q, const unsigned long long *nzmask,
                            MjhHuffTable *tabs, int spi, const MjhProgPE &pe, int nacf, int n, hipStream_t s)
{
  hipLaunchKernelGGL((k_pp_stats<true, 3>), dim3(pe.chunks_per_scan, nacf, n), dim3(256), 0, s, C, (const MjhProgScan *)scans, list, (const MjhProgCtl *)ctl,
                     (const int16_t *)q, nzmask, tabs, spi, pe, 0);
}

void mjh_launch_pp_emit_sl(const MjhConst &C, const void *scans, const int *par_list, const void *ctl, const void *q, const unsigned long long *nzmask,
                           const MjhHuffTable *tabs, int spi, unsigned *pool, size_t pool_words, const MjhProgPE &pe, int nacf, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_pp_emit_sl, dim3(pe.chunks_per_scan, nacf, n), dim3(256), 0, s, C, (const MjhProgScan *)scans, par_list, (const MjhProgCtl *)ctl,
                     (const int16_t *)q, nzmask, tabs, spi, pool, pool_words, pe);
}
