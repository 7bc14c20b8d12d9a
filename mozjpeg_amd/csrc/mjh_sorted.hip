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

template <bool STATS, int NW>   // NW waves = one trellis tile of 64 * NW blocks per workgroup
__global__ void __launch_bounds__(64 * NW)
k_dct_quant_sorted(MjhConst C, const MjhQuant *__restrict__ Q, const uint8_t *__restrict__ planes,
                   int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out,
                   MjhHuffTable *__restrict__ stat_tabs, int slots_per_image, int4 stat_slot_of_comp, uint8_t *__restrict__ nq8_out,
                   uint16_t *__restrict__ perm_out)
{
  dct_quant_body<uint8_t, STATS, true, NW>(C, Q, planes, coef_uq, coef_q, lambda_out, stat_tabs, slots_per_image, stat_slot_of_comp, nq8_out, perm_out);
}

// the same passes over tile-sorted coefficient planes (k_dct_quant_sorted wrote them and perm16): NPASS passes per tile of
// 64 * NPASS blocks, fast division
template <int QN, bool FST, int NPASS>
__global__ void __launch_bounds__(64)
k_trellis_ac_v3s(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q,
                 const MjhHuffTable *__restrict__ tabs, int slots_per_image, int4 ac_slot_of_comp, int4 tile0_of_comp,
                 const float *__restrict__ lambda_in, uint8_t *__restrict__ nq8, unsigned *__restrict__ worklist,
                 int16_t *__restrict__ dense, unsigned dense_cap, unsigned long long *__restrict__ nzmask,
                 MjhHuffTable *__restrict__ stat_tabs, int4 stat_slot_of_comp, const uint16_t *__restrict__ perm16)
{
  constexpr bool SORTED = true, FD = true, RECORDS = false;
  const uint2 *const rec_in = nullptr; const float *const azd_in = nullptr; constexpr size_t rec_stride = 0;
#include "mjh_trellis_v3.inc"
}

// ---- queue records from the FDCT kernel (MJH_TRELLIS_REC=1; natural block order, the trellis kernel sorts its tiles itself) ----
template <bool STATS>
__global__ void __launch_bounds__(64)
k_dct_quant_rec(MjhConst C, const MjhQuant *__restrict__ Q, const uint8_t *__restrict__ planes,
                int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out,
                MjhHuffTable *__restrict__ stat_tabs, int slots_per_image, int4 stat_slot_of_comp, uint8_t *__restrict__ nq8_out, MjhRecOut rec)
{
  dct_quant_body<uint8_t, STATS, true, 0, true>(C, Q, planes, coef_uq, coef_q, lambda_out, stat_tabs, slots_per_image, stat_slot_of_comp, nq8_out, nullptr, &rec);
}

template <int QN, bool FST, int NPASS>
__global__ void __launch_bounds__(64)
k_trellis_ac_v3r(MjhConst C, const MjhQuant *__restrict__ Q, int16_t *__restrict__ coef_q,
                 const MjhHuffTable *__restrict__ tabs, int slots_per_image, int4 ac_slot_of_comp, int4 tile0_of_comp,
                 const float *__restrict__ lambda_in, uint8_t *__restrict__ nq8, unsigned long long *__restrict__ nzmask,
                 MjhHuffTable *__restrict__ stat_tabs, int4 stat_slot_of_comp,
                 const uint2 *__restrict__ rec_in, size_t rec_stride, const float *__restrict__ azd_in)
{
  constexpr bool SORTED = false, FD = true, RECORDS = true;
  const uint16_t *const perm16 = nullptr;
  const int16_t *const coef_uq = nullptr; unsigned *const worklist = nullptr; int16_t *const dense = nullptr; constexpr unsigned dense_cap = 0u;   // (phase 1's inputs and outputs: not used)
#include "mjh_trellis_v3.inc"
}

template <int QN2>   // the general tiers behind k_trellis_ac_v3s (plain compact pass, no fused statistics); perm_tile: blocks per sorted tile
__global__ void __launch_bounds__(64)
k_trellis_ac_qds(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
                 int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
                 int4 ac_slot_of_comp, const float *__restrict__ lambda_in, const unsigned *__restrict__ worklist,
                 unsigned *__restrict__ worklist_next, const int16_t *__restrict__ dense, unsigned dense_cap,
                 MjhHuffTable *__restrict__ stat_tabs, int4 stat_slot_of_comp, MjhTrellisExt ext, const uint16_t *__restrict__ perm16, int perm_tile)
{
  constexpr bool PERM = true, FSTATS = false, EXT = false, COMPACT = true;
#include "mjh_trellis_qd.inc"
}

static int sorted_max_nblk(const MjhConst &C) { int m = 0; for (int i = 0; i < C.ncomp; i++) m = C.c[i].nblk > m ? C.c[i].nblk : m; return m; }
static void bad_tile(int t) { fprintf(stderr, "mjh_sorted: tiles of %d blocks do not exist (128, 256, 512)\n", t); abort(); }

void mjh_launch_dct_sorted(const MjhConst &C, const MjhQuant *Q, const void *planes, void *uq, void *q, float *lambda,
                           MjhHuffTable *stat_tabs, int spi, const int stat_slot[4], uint8_t *nq8, int n, hipStream_t s, uint16_t *perm16, int sorted_tile)
{
  // one workgroup of sorted_tile / 64 waves per tile
  const int4 sl = stat_tabs ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  dim3 gridt((sorted_max_nblk(C) + sorted_tile - 1) / sorted_tile, C.ncomp, n);
#define LDCT(ST, NW) hipLaunchKernelGGL((k_dct_quant_sorted<ST, NW>), gridt, dim3(64 * NW), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8, perm16)
  if (sorted_tile == 512) {   // 8 waves x 8 KB + the sort's counters: 65792 bytes of LDS per workgroup (gfx950: up to 160 KB)
    static int lds_ok = -1;
    if (lds_ok < 0) {
      int dev = 0, lim = 0;
      (void)hipGetDevice(&dev);
      lds_ok = hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && lim >= 65792;
      if (!lds_ok) fprintf(stderr, "mjh_sorted: tiles of 512 blocks need 65792 bytes of LDS per workgroup, the device allows %d\n", lim);
    }
    if (!lds_ok) abort();
  }
  switch (sorted_tile) {
    case 128: if (stat_tabs) LDCT(true, 2); else LDCT(false, 2); break;
    case 256: if (stat_tabs) LDCT(true, 4); else LDCT(false, 4); break;
    case 512: if (stat_tabs) LDCT(true, 8); else LDCT(false, 8); break;
    default: bad_tile(sorted_tile);
  }
#undef LDCT
}

// first tier (sorted_tile / 64 passes per tile, whatever else the caller would choose) + the general tiers; the caller has
// zeroed the work-list counters and counts the deferred blocks' statistics afterwards (mjh_launch_trellis_ac)
void mjh_launch_trellis_ac_sorted(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, MjhHuffTable *tabs, int spi, const int ac_slot[4], const float *lambda,
                                  unsigned *worklist, unsigned *worklist2, void *dense, unsigned dense_cap, const int *stat_slot, int variant,
                                  unsigned long long *nzmask, int n, hipStream_t s, uint8_t *nq8, const uint16_t *perm16, int sorted_tile)
{
  MjhTrellisExt ext;
  ext.Ss = 1; ext.Se = 63; ext.eob_cost = nullptr; ext.eob_has = nullptr; ext.nzmask = nzmask; ext.qstride = 0;
  const int4 sl = make_int4(ac_slot[0], ac_slot[1], ac_slot[2], ac_slot[3]);
  MjhHuffTable *st = stat_slot ? tabs : nullptr;
  const int4 ss = stat_slot ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  if (sorted_tile != 128 && sorted_tile != 256 && sorted_tile != 512) bad_tile(sorted_tile);
  int t0[5] = { 0, 0, 0, 0, 0 };
  for (int i = 0; i < 4; i++) t0[i + 1] = t0[i] + (i < C.ncomp ? (C.c[i].nblk + sorted_tile - 1) / sorted_tile : 0);
  dim3 gridt(t0[C.ncomp], n);
  for (int i = C.ncomp; i < 4; i++) t0[i] = 0x7FFFFFFF;   // components that do not exist never match
  const int4 tv = make_int4(t0[0], t0[1], t0[2], t0[3]);
#define LV3S(QN, FSV, NP) hipLaunchKernelGGL((k_trellis_ac_v3s<QN, FSV, NP>), gridt, dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, tv, lambda, nq8, worklist, (int16_t *)dense, dense_cap, nzmask, st, ss, perm16)
#define LV3T(QN, FSV) do { if (sorted_tile == 128) LV3S(QN, FSV, 2); else if (sorted_tile == 512) LV3S(QN, FSV, 8); else LV3S(QN, FSV, 4); } while (0)
  if (variant >= 3 && !st) { if (variant == 3) LV3T(32, false); else LV3T(48, false); }
  else if (variant > 0) { if (st) LV3T(24, true); else LV3T(24, false); }
  else if (st) LV3T(16, true);
  else LV3T(16, false);
#undef LV3T
#undef LV3S
#define LDS_(QN, GRID, WL, WLN) hipLaunchKernelGGL((k_trellis_ac_qds<QN>), dim3(GRID), dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, lambda, \
                                                   (const unsigned *)WL, WLN, (const int16_t *)dense, dense_cap, (MjhHuffTable *)nullptr, ss, ext, perm16, sorted_tile)
  if (variant >= 3 && !st) LDS_(63, 2048, worklist, (unsigned *)nullptr);   // what is left has more than 32 records or a magnitude >= 16
  else { LDS_(32, 2048, worklist, worklist2); LDS_(63, 1024, worklist2, (unsigned *)nullptr); }
#undef LDS_
}

void mjh_launch_dct_rec(const MjhConst &C, const MjhQuant *Q, const void *planes, void *uq, void *q, float *lambda,
                        MjhHuffTable *stat_tabs, int spi, const int stat_slot[4], uint8_t *nq8, int n, hipStream_t s, const MjhRecOut &rec)
{
  const int4 sl = stat_tabs ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  dim3 grid((sorted_max_nblk(C) + 63) / 64, C.ncomp, n);
  if (stat_tabs) hipLaunchKernelGGL(k_dct_quant_rec<true>, grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8, rec);
  else hipLaunchKernelGGL(k_dct_quant_rec<false>, grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8, rec);
}

// the first tier over the FDCT kernel's records (npass passes per tile of 64 * npass blocks, sorted here); the general tiers and
// the deferred blocks' statistics stay with the caller (mjh_launch_trellis_ac: natural order, nothing special about them)
void mjh_launch_trellis_ac_rec(const MjhConst &C, const MjhQuant *Q, void *q, MjhHuffTable *tabs, int spi, const int ac_slot[4], const float *lambda,
                               const int *stat_slot, unsigned long long *nzmask, int n, hipStream_t s, uint8_t *nq8, const MjhRecOut &rec, int npass)
{
  const int4 sl = make_int4(ac_slot[0], ac_slot[1], ac_slot[2], ac_slot[3]);
  MjhHuffTable *st = stat_slot ? tabs : nullptr;
  const int4 ss = stat_slot ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  const int tile = 64 * npass;
  int t0[5] = { 0, 0, 0, 0, 0 };
  for (int i = 0; i < 4; i++) t0[i + 1] = t0[i] + (i < C.ncomp ? (C.c[i].nblk + tile - 1) / tile : 0);
  dim3 gridt(t0[C.ncomp], n);
  for (int i = C.ncomp; i < 4; i++) t0[i] = 0x7FFFFFFF;   // components that do not exist never match
  const int4 tv = make_int4(t0[0], t0[1], t0[2], t0[3]);
#define LV3R(QN, FSV, NP) hipLaunchKernelGGL((k_trellis_ac_v3r<QN, FSV, NP>), gridt, dim3(64), 0, s, C, Q, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, tv, lambda, nq8, nzmask, st, ss, \
                                             (const uint2 *)rec.records, rec.row_stride, (const float *)rec.azd)
  bool ok = true;
  if (rec.qn == 32 || rec.qn == 48) {       // q90 and up (20 / 30 KB of LDS per wave): four passes, no fused statistics
    if (st || npass != 4) ok = false;
    else if (rec.qn == 32) LV3R(32, false, 4);
    else LV3R(48, false, 4);
  } else
  if (rec.qn == 24) {
    if (st) { if (npass == 4) LV3R(24, true, 4); else ok = false; }
    else if (npass == 4) LV3R(24, false, 4);
    else if (npass == 1) LV3R(24, false, 1);
    else ok = false;
  } else if (rec.qn == 16) {
    if (st) { if (npass == 4) LV3R(16, true, 4); else ok = false; }
    else switch (npass) { case 8: LV3R(16, false, 8); break; case 4: LV3R(16, false, 4); break; case 2: LV3R(16, false, 2); break; case 1: LV3R(16, false, 1); break; default: ok = false; }
  } else ok = false;
#undef LV3R
  if (!ok) { fprintf(stderr, "mjh_sorted: no record-reading first tier for %d records, %d passes%s\n", rec.qn, npass, st ? ", fused statistics" : ""); abort(); }
}
