// mjh_sorted.hip -- the tile-sorted coefficient planes ("tile-sorted planes"; opt-in: MJH_SORTED_UQ=1)
//
// A translation unit of its own on purpose.  The kernels here are variants of three kernels of mjh_kernels.hip and share
// every device function with them (this file includes mjh_kernels.hip with MJH_TU_SORTED defined, which leaves only those
// functions), but they are compiled separately: the kernels of the default path then keep, instruction for instruction,
// the machine code they were validated and profiled with (tools/kernel_isa.py compares it per kernel; merely instantiating
// the variants in the same translation unit changed the register allocation of k_trellis_ac_qd).
//
//  * k_dct_quant_sorted: the FDCT kernel as a workgroup of FOUR waves = the four 64-block lines of one trellis tile; every
//    lane still owns one block.  The waves exchange only the blocks' sort keys through LDS and store planes 1..63 of coef_uq
//    at the block's place in the tile's descending-key order, plus the permutation (perm16).
//  * k_trellis_ac_v3s: the tile-sorted AC trellis without its own sort; pass p reads line p of every plane, each line once.
//  * k_trellis_ac_qds: the general tiers behind it; a deferred block without a dense copy is named by its place in the
//    sorted planes and found through the permutation.
// Reference behaviour: the same as the kernels they vary (jcdctmgr.c:646-678, jfdctint.c:142-286, jcdctmgr.c:1120-1222).
#define MJH_TU_SORTED 1
#include "mjh_kernels.hip"
#include "mjh_launch.h"

template <bool STATS, bool FD>
__global__ void __launch_bounds__(256)
k_dct_quant_sorted(MjhConst C, const MjhQuant *__restrict__ Q, const uint8_t *__restrict__ planes,
                   int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out,
                   MjhHuffTable *__restrict__ stat_tabs, int slots_per_image, int4 stat_slot_of_comp, uint8_t *__restrict__ nq8_out,
                   uint16_t *__restrict__ perm_out)
{
  dct_quant_body<uint8_t, STATS, FD, true>(C, Q, planes, coef_uq, coef_q, lambda_out, stat_tabs, slots_per_image, stat_slot_of_comp, nq8_out, perm_out);
}

// the same passes over tile-sorted coefficient planes (k_dct_quant_sorted wrote them and perm16): four passes, fast division
template <int QN, bool FST>
__global__ void __launch_bounds__(64)
k_trellis_ac_v3s(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q,
                 const MjhHuffTable *__restrict__ tabs, int slots_per_image, int4 ac_slot_of_comp, int4 tile0_of_comp,
                 const float *__restrict__ lambda_in, uint8_t *__restrict__ nq8, unsigned *__restrict__ worklist,
                 int16_t *__restrict__ dense, unsigned dense_cap, unsigned long long *__restrict__ nzmask,
                 MjhHuffTable *__restrict__ stat_tabs, int4 stat_slot_of_comp, const uint16_t *__restrict__ perm16)
{
  constexpr bool SORTED = true, FD = true;
  constexpr int NPASS = 4;
#include "mjh_trellis_v3.inc"
}

template <int QN2>   // the general tiers behind k_trellis_ac_v3s (plain compact pass, no fused statistics)
__global__ void __launch_bounds__(64)
k_trellis_ac_qds(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
                 int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
                 int4 ac_slot_of_comp, const float *__restrict__ lambda_in, const unsigned *__restrict__ worklist,
                 unsigned *__restrict__ worklist_next, const int16_t *__restrict__ dense, unsigned dense_cap,
                 MjhHuffTable *__restrict__ stat_tabs, int4 stat_slot_of_comp, MjhTrellisExt ext, const uint16_t *__restrict__ perm16)
{
  constexpr bool PERM = true, FSTATS = false, EXT = false, COMPACT = true;
#include "mjh_trellis_qd.inc"
}


static int sorted_max_nblk(const MjhConst &C) { int m = 0; for (int i = 0; i < C.ncomp; i++) m = C.c[i].nblk > m ? C.c[i].nblk : m; return m; }

void mjh_launch_dct_sorted(const MjhConst &C, const MjhQuant *Q, const void *planes, void *uq, void *q, float *lambda,
                           MjhHuffTable *stat_tabs, int spi, const int stat_slot[4], uint8_t *nq8, int n, hipStream_t s, uint16_t *perm16)
{
  // one workgroup of four waves per 256-block tile
  const int4 sl = stat_tabs ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  dim3 gridt((sorted_max_nblk(C) + 255) / 256, C.ncomp, n);
  if (stat_tabs) hipLaunchKernelGGL((k_dct_quant_sorted<true, true>), gridt, dim3(256), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8, perm16);
  else hipLaunchKernelGGL((k_dct_quant_sorted<false, true>), gridt, dim3(256), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8, perm16);
}

// first tier (four passes per tile of 256 blocks, whatever else the caller would choose) + the general tiers; the caller has
// zeroed the work-list counters and counts the deferred blocks' statistics afterwards (mjh_launch_trellis_ac)
void mjh_launch_trellis_ac_sorted(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, MjhHuffTable *tabs, int spi, const int ac_slot[4], const float *lambda,
                                  unsigned *worklist, unsigned *worklist2, void *dense, unsigned dense_cap, const int *stat_slot, int variant,
                                  unsigned long long *nzmask, int n, hipStream_t s, uint8_t *nq8, const uint16_t *perm16)
{
  MjhTrellisExt ext;
  ext.Ss = 1; ext.Se = 63; ext.eob_cost = nullptr; ext.eob_has = nullptr; ext.nzmask = nzmask; ext.qstride = 0;
  const int4 sl = make_int4(ac_slot[0], ac_slot[1], ac_slot[2], ac_slot[3]);
  MjhHuffTable *st = stat_slot ? tabs : nullptr;
  const int4 ss = stat_slot ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  int t0[5] = { 0, 0, 0, 0, 0 };
  for (int i = 0; i < 4; i++) t0[i + 1] = t0[i] + (i < C.ncomp ? (C.c[i].nblk + 255) / 256 : 0);
  dim3 gridt(t0[C.ncomp], n);
  for (int i = C.ncomp; i < 4; i++) t0[i] = 0x7FFFFFFF;   // components that do not exist never match
  const int4 tv = make_int4(t0[0], t0[1], t0[2], t0[3]);
#define LV3S(QN, FSV) hipLaunchKernelGGL((k_trellis_ac_v3s<QN, FSV>), gridt, dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, tv, lambda, nq8, worklist, (int16_t *)dense, dense_cap, nzmask, st, ss, perm16)
  if (variant >= 3 && !st) { if (variant == 3) LV3S(32, false); else LV3S(48, false); }
  else if (variant > 0) { if (st) LV3S(24, true); else LV3S(24, false); }
  else if (st) LV3S(16, true);
  else LV3S(16, false);
#undef LV3S
#define LDS_(QN, GRID, WL, WLN) hipLaunchKernelGGL((k_trellis_ac_qds<QN>), dim3(GRID), dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, lambda, \
                                                   (const unsigned *)WL, WLN, (const int16_t *)dense, dense_cap, (MjhHuffTable *)nullptr, ss, ext, perm16)
  if (variant >= 3 && !st) LDS_(63, 2048, worklist, (unsigned *)nullptr);   // what is left has more than 32 records or a magnitude >= 16
  else { LDS_(32, 2048, worklist, worklist2); LDS_(63, 1024, worklist2, (unsigned *)nullptr); }
#undef LDS_
}
