// mjh_arith_coder.h -- the QM coder (ITU-T T.81 Annex D; jcarith.c:229-320) and the decision sequences of the JPEG
// statistics models (F.1.4 / G.1.3; jcarith.c:402-687), written for ONE wave running as a scalar machine.
//
// Every lane of the coding wave executes this code with the same (wave-uniform) values, so the compiler keeps the coder's
// registers in SGPRs and its branches are scalar jumps.  The coder's MEMORY lives in vector registers used as 64-entry RAMs:
// entry i = lane i, read with v_readlane_b32, written with v_writelane_b32.  A lone wave pays for every instruction with
// 4-5 cycles and for every TAKEN branch with several issue slots, which shapes the layout:
//   * one statistics bin per lane, and the bin's word carries the Qe of its state next to the state byte (Qe << 16 | MPS << 7
//     | index): the common decision -- the more probable symbol, no renormalisation -- is one register read, a subtraction
//     and two compares; the probability table (T.81 Table D.3) is only consulted when a bin changes state;
//   * the AC bins of a table are four registers by KIND, each indexed by what the call site already has in a register: the
//     end-of-block bins and the zero bins by position (jcarith.c's st = 3 (k - 1) + 0 / 1), the first magnitude bin by
//     position (+ 2), the magnitude-category and magnitude-bit bins by their offset from 189 (k <= Kx: 0..27, above: 28..55);
//   * the tables of the block being coded are BOUND to fixed registers around every block (ari_run), the coder itself never
//     chooses between registers;
//   * both halves of the probability table are read and one is selected, the more / less probable paths share their
//     arithmetic through selects, and the renormalisation shifts by the whole distance at once (count-leading-zeros; the
//     reference's bit-by-bit loop only looks at the registers at byte boundaries).
// The file compiles for the host as well (MJH_ARI_HOST: a register is an array of 64 ints): tests/native/arith_coder_check.cpp
// runs it against a plain restatement of jcarith.c on random blocks, which is how a change here is checked before it goes
// near a GPU.
#ifndef MJH_ARITH_CODER_H
#define MJH_ARITH_CODER_H
#include <stdint.h>
#include "mjh_arith_table.h"

// conditioning (cinfo->arith_dc_L / arith_dc_U / arith_ac_K, defaults 0 / 1 / 5: jcparam.c:417-419; the DAC marker carries
// them): thresholds of the bound tables live in the model (AriModel.dc_lo / dc_hi / ac_k)

#ifdef MJH_ARI_HOST
#define ARI_FN static inline
#define ARI_MFN inline
struct ari_reg { int v[64]; };
static inline int rl(const ari_reg &r, int lane) { return r.v[lane & 63]; }
static inline ari_reg &wl(int val, int lane, ari_reg &old) { old.v[lane & 63] = val; return old; }
#else
#define ARI_FN __device__ __forceinline__
#define ARI_MFN __device__ __forceinline__
typedef int ari_reg;
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
#ifdef MJH_SIMT_HOST   // (tools/simt: v_writelane spelled out)
__device__ __forceinline__ int wl(int val, int lane, int old)
{
  const int v = __builtin_amdgcn_readfirstlane(val), l = __builtin_amdgcn_readfirstlane(lane) & 63;
  return simt::cur->lane == l ? v : old;
}
#else
__device__ __forceinline__ int wl(int val, int lane, int old)
{ // (this compiler has no writelane builtin; the s_nop covers the lane-select hazard the hazard recognizer cannot see inside asm)
  // (lane select through M0: a VALU instruction may read one SGPR over the constant bus, M0 does not count)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(__builtin_amdgcn_readfirstlane(val)), "s"(__builtin_amdgcn_readfirstlane(lane)) : "m0");
  return old;
}
#endif
#endif

// a bin's word when its state is 0 with MPS 0 (what a reset leaves behind): Qe of state 0 above the state byte
#define ARI_BIN_RESET ((int)0x5a1d0000)

enum { ARI_E = 0, ARI_Z = 1, ARI_M = 2, ARI_X = 3, ARI_D = 4, ARI_F = 5 };   // kinds of bins: AC end-of-block / zero / first magnitude / magnitude tail; DC; the fixed 0.5 bin

struct AriModel {      // vector registers as RAM
  ari_reg ac[2][4];    // AC statistics of table 0 / 1 by kind (ARI_E .. ARI_X), one bin per lane
  ari_reg cur[4];      // ... of the table bound to the block being coded
  ari_reg dc[2];       // DC statistics of table 0 / 1: lane = jcarith.c's bin index (< 49)
  ari_reg dcur;        // ... of the bound table
  ari_reg tab[2];      // T.81 Table D.3, entries 0..63 / 64..113: Qe << 16 | next state after an MPS << 8 | after an LPS (bit 7: MPS flips)
  ari_reg coef;        // the block being coded: lane k = coefficient k (zig-zag)
  // conditioning of the bound tables (wave-uniform): the DC category thresholds (1 << L) >> 1 and (1 << U) >> 1
  // (jcarith.c:442-445), the AC position Kx up to which the low magnitude bins are used (:533)
  int dc_lo, dc_hi, ac_k;
};

struct AriCoder {      // jcarith.c:28-52, all wave-uniform
  unsigned c, a;
  int sc, zc, ct, buffer;
  uint8_t *out;        // nullptr: only sizes
  unsigned pos, cap;
  bool lane0;
  ARI_MFN void byte(int v)
  {
    if (lane0 && out && pos < cap) out[pos] = (uint8_t)v;
    pos++;
  }
  ARI_MFN void zeros() { while (zc) { byte(0x00); zc--; } }
  ARI_MFN void reset() { c = 0; a = 0x10000u; sc = 0; zc = 0; ct = 11; buffer = -1; }
  // a byte leaves the code register (D.1.6) or the register is flushed (D.1.8): jcarith.c:278-316 / :160-190
  ARI_MFN void shift_out(unsigned temp, bool final)
  {
    if (final ? (c & 0xF8000000u) != 0u : temp > 0xFFu) {
      if (buffer >= 0) {
        zeros();
        byte(buffer + 1);
        if (buffer + 1 == 0xFF) byte(0x00);
      }
      zc += sc;
      sc = 0;
      if (!final) buffer = (int)(temp & 0xFFu);
    } else if (!final && temp == 0xFFu) {
      sc++;
    } else {
      if (buffer == 0) zc++;
      else if (buffer >= 0) { zeros(); byte(buffer); }
      if (sc) {
        zeros();
        do { byte(0xFF); byte(0x00); } while (--sc);
      }
      if (!final) buffer = (int)(temp & 0xFFu);
    }
  }
  ARI_MFN void finish()      // finish_pass jcarith.c:142-203
  {
    const unsigned temp = (a - 1u + c) & 0xFFFF0000u;
    c = temp < c ? temp + 0x8000u : temp;
    c <<= ct;
    shift_out(0u, true);
    if (c & 0x7FFF800u) {
      zeros();
      byte((int)((c >> 19) & 0xFFu));
      if (((c >> 19) & 0xFFu) == 0xFFu) byte(0x00);
      if (c & 0x7F800u) {
        byte((int)((c >> 11) & 0xFFu));
        if (((c >> 11) & 0xFFu) == 0xFFu) byte(0x00);
      }
    }
  }
  // arith_encode jcarith.c:229-320 on bin `lane` of kind KIND (known at every call site)
  template <int KIND>
  ARI_MFN void encode(AriModel &M, int lane, int val)
  {
    unsigned w;
    if (KIND == ARI_F) w = ((unsigned)0x5a1d << 16) | 113u;      // state 113 never adapts: no RAM access at all
    else if (KIND == ARI_D) w = (unsigned)rl(M.dcur, lane);
    else w = (unsigned)rl(M.cur[KIND], lane);
    const unsigned qe = w >> 16, sv = w & 0xFFu;
    const bool lps = (unsigned)val != (sv >> 7);
    a -= qe;
    if (!lps && a >= 0x8000u) return;
    const bool exchange = lps ? a >= qe : a < qe;      // conditional exchange (D.1.4 / D.1.5)
    c += exchange ? a : 0u;
    a = exchange ? qe : a;
    if (KIND != ARI_F) {
      const int s = (int)(sv & 0x7Fu);
      const unsigned lo = (unsigned)rl(M.tab[0], s & 63), hi = (unsigned)rl(M.tab[1], s & 63);
      const unsigned t = s < 64 ? lo : hi;
      const unsigned ns = (sv & 0x80u) ^ ((lps ? t : t >> 8) & 0xFFu);
      const int s2 = (int)(ns & 0x7Fu);
      const unsigned lo2 = (unsigned)rl(M.tab[0], s2 & 63), hi2 = (unsigned)rl(M.tab[1], s2 & 63);
      const int nw = (int)(((s2 < 64 ? lo2 : hi2) & 0xFFFF0000u) | ns);
      if (KIND == ARI_D) M.dcur = wl(nw, lane, M.dcur);
      else M.cur[KIND == ARI_D || KIND == ARI_F ? 0 : KIND] = wl(nw, lane, M.cur[KIND == ARI_D || KIND == ARI_F ? 0 : KIND]);
    }
    // renormalisation (D.1.6): a is in [1, 0x7FFF] here, n >= 1 shifts bring it back to [0x8000, 0xFFFF]
    int n = __builtin_clz(a) - 16;
    while (n >= ct) {
      a <<= ct; c <<= ct; n -= ct;
      shift_out(c >> 19, false);
      c &= 0x7FFFFu;
      ct = 8;
    }
    a <<= n; c <<= n; ct -= n;
  }
};

ARI_FN int ari_coef(const AriModel &M, int k) { return (int)(short)rl(M.coef, k); }

// Figures F.8 / F.9 for an AC coefficient of magnitude v >= 1 at position k (jcarith.c:514-546): p = k - 1 is the lane of its
// first magnitude bin; the category bins continue at offset 0 (k <= Kx) / 28 of the tail register, the bit bins 14 further
ARI_FN void ari_ac_magnitude(AriCoder &A, AriModel &M, int p, int v, int k)
{
  if (v -= 1) {
    A.encode<ARI_M>(M, p, 1);
    int v2 = v;
    if (v2 >>= 1) {
      A.encode<ARI_M>(M, p, 1);
      int m = 2, x = k <= M.ac_k ? 0 : 28;
      while (v2 >>= 1) { A.encode<ARI_X>(M, x, 1); m <<= 1; x++; }
      A.encode<ARI_X>(M, x, 0);
      x += 14;
      for (int mm = m >> 1; mm; mm >>= 1) A.encode<ARI_X>(M, x, (mm & v) ? 1 : 0);
      return;
    }
  }
  A.encode<ARI_M>(M, p, 0);      // (magnitude 1 or 2: no magnitude bits follow)
}

// ... and for a DC difference of magnitude v >= 1 whose first magnitude bin is st (jcarith.c:427-446); returns the category mask
ARI_FN int ari_dc_magnitude(AriCoder &A, AriModel &M, int st, int v)
{
  int m = 0;
  if (v -= 1) {
    A.encode<ARI_D>(M, st, 1);
    m = 1;
    int v2 = v;
    st = 20;
    while (v2 >>= 1) { A.encode<ARI_D>(M, st, 1); m <<= 1; st++; }
  }
  A.encode<ARI_D>(M, st, 0);
  st += 14;
  for (int mm = m >> 1; mm; mm >>= 1) A.encode<ARI_D>(M, st, (mm & v) ? 1 : 0);
  return m;
}

// Encode_DC_DIFF (jcarith.c:402-448 / :715-762) with the bound DC table
ARI_FN void ari_dc(AriCoder &A, AriModel &M, int &last_dc, int &ctx, int value)
{
  int st = ctx;
  int v = value - last_dc;
  if (v == 0) { A.encode<ARI_D>(M, st, 0); ctx = 0; return; }
  last_dc = value;
  A.encode<ARI_D>(M, st, 1);
  if (v > 0) { A.encode<ARI_D>(M, st + 1, 0); st += 2; ctx = 4; }
  else { v = -v; A.encode<ARI_D>(M, st + 1, 1); st += 3; ctx = 8; }
  const int m = ari_dc_magnitude(A, M, st, v);
  if (m < M.dc_lo) ctx = 0;
  else if (m > M.dc_hi) ctx += 8;
}

// Encode_AC_Coefficients: encode_mcu_AC_first jcarith.c:456-552; with Ss = 1, Se = 63, Al = 0 the AC part of encode_mcu :764-817.
// ke = the block's end-of-block index for this scan (jcarith.c:484-496), found by the lane that loaded the block
ARI_FN void ari_ac_first(AriCoder &A, AriModel &M, int Ss, int Se, int Al, int ke)
{
  int k, v;
  for (k = Ss; k <= ke; k++) {
    int p = k - 1;
    int neg;
    A.encode<ARI_E>(M, p, 0);
    for (;;) {
      v = ari_coef(M, k);
      neg = v < 0;
      if (neg) v = -v;
      v >>= Al;
      if (v) break;
      A.encode<ARI_Z>(M, p, 0);
      p++;
      k++;
    }
    A.encode<ARI_Z>(M, p, 1);
    A.encode<ARI_F>(M, 0, neg);
    ari_ac_magnitude(A, M, p, v, k);
  }
  if (k <= Se) A.encode<ARI_E>(M, k - 1, 1);
}

// encode_mcu_AC_refine jcarith.c:596-687
ARI_FN void ari_ac_refine(AriCoder &A, AriModel &M, int Ss, int Se, int Ah, int Al, int ke, int kex)
{
  int k, v;
  (void)Ah;
  for (k = Ss; k <= ke; k++) {
    int p = k - 1;
    if (k > kex) A.encode<ARI_E>(M, p, 0);
    for (;;) {
      v = ari_coef(M, k);
      const int neg = v < 0;
      if (neg) v = -v;
      v >>= Al;
      if (v) {
        if (v >> 1) A.encode<ARI_M>(M, p, v & 1);
        else { A.encode<ARI_Z>(M, p, 1); A.encode<ARI_F>(M, 0, neg); }
        break;
      }
      A.encode<ARI_Z>(M, p, 0);
      p++;
      k++;
    }
  }
  if (k <= Se) A.encode<ARI_E>(M, k - 1, 1);
}
#endif
