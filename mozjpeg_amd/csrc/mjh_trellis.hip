// mjh_trellis.hip -- the AC trellis kernels (AC part; 8f row 4: its options) and their launch wrappers.
//
// A translation unit of its own since round 6 because of its compile flags (mozjpeg_amd/build.py: EXTRA_FLAGS): the walks of
// k_trellis_ac_v3 / k_trellis_ac_qd are chains of dependent LDS look-ups and float adds, which the max-ILP scheduling strategy
// orders better (trellis interval 1.53 -> 1.48 ms per 64 4K frames, profiles/r06q_sched.md), while the same strategy costs the
// bit writers and the FDCT kernel of mjh_kernels.hip more than that.  Everything else is as in mjh_kernels.hip: one block per
// lane, the reference's float recipe operation by operation, -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "mjh_internal.h"
#include "mjh_device.h"
#include "mjh_launch.h"

// =============================================================================================
// K5  AC trellis quantization (row a9): quantize_trellis jcdctmgr.c:1120-1222 (+ norm/lambda
// :1011-1037).  One lane = one block.  The rate-distortion DP only ever looks back at
// positions whose chosen coefficient is non-zero, so each lane keeps a compact list of "live"
// predecessors {accumulated zero distortion, accumulated cost} + {back pointer, value} in LDS columns
// ([entries][64 lanes], bank = lane => conflict-free) and a 64-bit mask of their positions (bit 0 = the virtual
// start) in registers.
// Float recipe T5 of SURVEY 9 is followed operation by operation; first-minimum ties resolve
// in (predecessor, candidate) lexicographic order exactly as the reference's strict '<' scan.
// T6: a position whose candidates all lack a Huffman code keeps a stale value in the
// reference; such a position can never win (its cost is >= 1e38) and is zeroed by the
// back-track, so it is simply not appended here.
//  * predecessors are walked NEWEST FIRST, two per step.  The reference scans them oldest first and keeps the first
//    minimum (strict '<'), i.e. on equal cost the OLDER predecessor (and for one predecessor the smaller candidate)
//    wins -- reproduced by the explicit tie rule.  cost = (rate + dist) + rhs >= rhs >= gap in float arithmetic (adding
//    a non-negative term never rounds below the other operand), so once the gap (azd difference) of the oldest entry
//    looked at exceeds the best cost every older predecessor is out; evaluating a predecessor that could have been
//    pruned changes nothing, which is why pairs can be evaluated without branches.
//  * AC code lengths: one 16-symbol row (= one zero-run length) pre-converted to floats with the magnitude bits
//    folded in (rate_rows in LDS); all sums stay exact small integers = the reference's (float)(size + nbits + zrl).
//  * candidates 0..3 (|q| < 16) are unrolled, instantiated with 1 / 2 / 4 candidates by a wave-uniform test;
//    larger magnitudes take a rare rolled loop.
// =============================================================================================
__device__ __forceinline__ int row_byte(const uint4 &r, int b)
{
  const unsigned w = b < 4 ? r.x : (b < 8 ? r.y : (b < 12 ? r.z : r.w));
  return (int)((w >> (8 * (b & 3))) & 0xFFu);
}

// code lengths of one zero-run row as floats with the magnitude bits folded in: element k =
// (float)(size(run, k+1) + k + 1), or 3e38 where the table has no code (such a candidate can never win)
__device__ __forceinline__ float4 rate_row(const uint4 &r)
{
  const int b1 = (int)((r.x >> 8) & 0xFFu), b2 = (int)((r.x >> 16) & 0xFFu), b3 = (int)(r.x >> 24), b4 = (int)(r.y & 0xFFu);
  return make_float4(b1 ? (float)(b1 + 1) : 3e38f, b2 ? (float)(b2 + 2) : 3e38f, b3 ? (float)(b3 + 3) : 3e38f,
                     b4 ? (float)(b4 + 4) : 3e38f);
}

// cost of ONE predecessor for every candidate of the current position: local minimum lb, its candidate index lk
template <int NC, bool LDS_ROWS>
__device__ __forceinline__ void pred_cost(const uint4 *si_rows, const float4 &rr, int zero_run, float rhs, const float *dist,
                                          int ncd, int x, int dq, int qval, float lambda, float lti, int si_f0, float f0f,
                                          float &lb, int &lk)
{
  const int hi = zero_run >> 4;
  const float rb = (float)hi * f0f;
  lk = 0;
  if (NC == 1) {
    lb = (rr.x + rb) + dist[0];
    lb = lb + rhs;
  } else if (NC == 2) {
    float c0 = (rr.x + rb) + dist[0];
    float c1 = (rr.y + rb) + dist[NC > 1 ? 1 : 0];
    c0 = c0 + rhs; c1 = c1 + rhs;
    lb = c0;
    if (c1 < lb) { lb = c1; lk = 1; }
  } else {
    float c0 = (rr.x + rb) + dist[0];
    float c1 = (rr.y + rb) + dist[NC > 1 ? 1 : 0];
    float c2 = (rr.z + rb) + dist[NC > 2 ? 2 : 0];
    float c3 = (rr.w + rb) + dist[NC > 3 ? 3 : 0];
    c0 = c0 + rhs; c1 = c1 + rhs; c2 = c2 + rhs; c3 = c3 + rhs;
    lb = c0;
    if (c1 < lb) { lb = c1; lk = 1; }
    if (c2 < lb) { lb = c2; lk = 2; }
    if (c3 < lb) { lb = c3; lk = 3; }
    if (ncd > 4 && !(hi && si_f0 == 0)) {        // |q| >= 16: rare
      const uint4 row = si_rows[zero_run & 15];
      const int rbase = hi * si_f0;
#pragma nounroll
      for (int k = 4; k < ncd; k++) {
        const int cb = row_byte(row, k + 1);
        if (cb != 0) {
          const int cand = (k < ncd - 1) ? (2 << k) - 1 : qval;
          const int delta = mul24(cand, dq) - x;
          float d = (float)mul24(delta, delta) * lambda;
          d = d * lti;
          float cost = (float)(cb + (k + 1) + rbase) + d;
          cost = cost + rhs;
          if (cost < lb) { lb = cost; lk = k; }
        }
      }
    }
  }
}

template <int NC, bool LDS_ROWS>
__device__ __forceinline__ float4 load_rate(const uint4 *si_rows, const float4 *rate_rows, int zero_run)
{
  if (LDS_ROWS) {
    if (NC == 1) return make_float4(rate_rows[zero_run & 15].x, 0.f, 0.f, 0.f);
    if (NC == 2) { const float2 t = *reinterpret_cast<const float2 *>(&rate_rows[zero_run & 15]); return make_float4(t.x, t.y, 0.f, 0.f); }
    return rate_rows[zero_run & 15];
  }
  return rate_row(si_rows[zero_run & 15]);
}

// =============================================================================================
// LANE-AUTONOMOUS walk.  A wave that moves over the 63 positions in lockstep (round 1's kernel) visits a position if
// ANY of its 64 blocks quantizes it to non-zero and walks predecessors as long as ANY lane still has one to look at:
// measured, 17 of 64 lanes were active per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU) with the VALU
// pipe ~75 % busy.  Here the work is split in two:
//   phase 1 (uniform, no divergence, all 63 plane loads in flight at once): the accumulated zero distortion of
//     every position in float, in position order, and one queue record {position, sign, quantized value, |x|,
//     azd before the position} per position with a non-zero quantized value, pushed to the lane's own LDS column;
//   phase 2: every lane pops ITS OWN records and walks ITS OWN live predecessors, two per step (the two newest come
//     from registers); a wave iterates until its busiest lane is done, i.e. max over lanes of the lane's own work
//     instead of the sum over positions of the busiest lane per position.
// Live entry e (created while record r >= e-1 is being consumed) overwrites queue slot e-1, so queue and live list
// share one LDS column of QN 8-byte slots; entry 0 (the virtual start, {0, 0}) is not stored.  Blocks with more
// than QN non-zero positions go to the work list (their raw coefficients are still in registers for the dense copy).
// =============================================================================================
template <int QN>
__device__ __forceinline__ float2 q_entry(const uint2 (*col)[64], int lane, int e)
{   // {azd, acc} of live entry e; entry 0 is the virtual start
  const uint2 v = col[e > 0 ? e - 1 : 0][lane];
  return e > 0 ? make_float2(__uint_as_float(v.x), __uint_as_float(v.y)) : make_float2(0.0f, 0.0f);
}

template <int NC, bool LDS_ROWS>
__device__ __forceinline__ void q_pair_step(const uint4 *si_rows, const float4 *rate_rows, const uint2 (*col)[64], int lane,
                                            unsigned long long &m, int &e, bool &first, float2 n0, float2 n1, float azd_prev, int i,
                                            int x, int dq, int qval, int ncd, float lambda, float lti, int si_f0, float f0f,
                                            float &best, int &bestp, int &bestk, bool &fin)
{
  const int p0 = 63 - __builtin_clzll(m);
  m &= ~(1ull << p0);
  const bool has1 = m != 0ull;
  const int p1 = has1 ? 63 - __builtin_clzll(m) : p0;
  if (has1) m &= ~(1ull << p1);
  const int zr0 = i - 1 - p0, zr1 = i - 1 - p1;
  // every load of the step is issued before the first use
  const float4 r0 = load_rate<NC, LDS_ROWS>(si_rows, rate_rows, zr0);
  const float4 r1 = load_rate<NC, LDS_ROWS>(si_rows, rate_rows, zr1);
  float2 a0 = n0, a1 = n1;
  if (!first) {
    a0 = q_entry<0>(col, lane, e - 1);
    a1 = q_entry<0>(col, lane, has1 ? e - 2 : e - 1);
  }
  first = false;
  e -= 2;
  float dist[NC];
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int cand = (k < ncd - 1) ? (2 << k) - 1 : qval;
    const int delta = mul24(cand, dq) - x;
    float d = (float)mul24(delta, delta) * lambda;
    dist[k] = k < ncd ? d * lti : 3e38f;
  }
  const float gap0 = azd_prev - a0.x, gap1 = azd_prev - a1.x;
  const float rhs0 = gap0 + a0.y, rhs1 = gap1 + a1.y;
  float lb0, lb1;
  int lk0, lk1;
  pred_cost<NC, LDS_ROWS>(si_rows, r0, zr0, rhs0, dist, ncd, x, dq, qval, lambda, lti, si_f0, f0f, lb0, lk0);
  pred_cost<NC, LDS_ROWS>(si_rows, r1, zr1, rhs1, dist, ncd, x, dq, qval, lambda, lti, si_f0, f0f, lb1, lk1);
  if (lb0 < best || (lb0 == best && bestp >= 0)) { best = lb0; bestp = p0; bestk = lk0; }
  if (has1 && (lb1 < best || (lb1 == best && bestp >= 0))) { best = lb1; bestp = p1; bestk = lk1; }
  fin = m == 0ull || gap1 > best;
}

// The DP of one block per lane, lane-autonomous.  xs[1..63] = the lane's raw coefficients (zig-zag order), dq8/rcp/lt =
// the component's quantizer constants in GLOBAL memory (wave-uniform: scalar loads) when UNIFORM_Q, else per lane;
// dqT/ltT = the same constants in LDS for the per-lane lookups of phase 2 (row qrow).  Returns false, with nothing
// written, when the block has more than QN non-zero positions; `active` false = the lane has no block (it still has to
// take part in the wave-level loop).
// phase 1 of the lane-autonomous DP: zero-distortion prefix + queue of the positions with a non-zero quantized value.
// Returns the number of such positions (> QN: the block has to be deferred; the queue then holds the first QN).
// EXT (use_scans_in_trellis / trellis_eob_opt): only the positions Ss..Se take part (the zero-distortion
// prefix starts at Ss, azd63 is its value at Se); Ss/Se are ignored otherwise.
template <int QN, bool EXT = false>
__device__ __forceinline__ int trellis_q_phase1(const short (&xs)[64], const int *__restrict__ dq8, const float *__restrict__ rcp,
                                                const float *__restrict__ lt, float lambda, uint2 (*col)[64], int lane, float &azd63,
                                                int Ss = 1, int Se = 63)
{
  int nq = 0;
  {
    float azd = 0.0f;
#pragma unroll
    for (int k = 1; k < 64; k++) {
      if (EXT && (k < Ss || k > Se)) continue;   // wave-uniform
      const int xsg = xs[k];
      const int x = xsg < 0 ? -xsg : xsg;
      const int dq = dq8[k];
      float t = (float)mul24(x, x) * lambda;
      t = t * lt[k];
      const float azd_cur = t + azd;
      if (x + (dq >> 1) >= dq) {
        int qval = udiv_exact(x + (dq >> 1), dq, rcp[k]);
        if (qval >= 1024) qval = 1023;
        if (nq < QN) col[nq][lane] = make_uint2((unsigned)k | (xsg < 0 ? 64u : 0u) | ((unsigned)qval << 7) | ((unsigned)x << 17), __float_as_uint(azd));
        nq++;
      }
      azd = azd_cur;
    }
    azd63 = azd;
  }
  return nq;
}

// phases 2.. : nq_in = what phase 1 returned.  `active` false or nq_in > QN: the lane has nothing to do but still takes
// part in the wave-level loop.
// EXT: band Ss..Se (the virtual start sits at position Ss-1, positions outside the band are neither read nor written)
// and, when eob_out is not null, the three per-block results the end-of-band-run optimisation needs (jcdctmgr.c:1187-1209):
// eob_out[0] = cost of the all-zero band, eob_out[1] = cost of the chosen path without its EOB, eob_has = 0/1/2.
// COMPACT (sequential mode, plain 1..63 pass): instead of 63 position planes the block's result is written as a RECORD --
// *nz_out = 64-bit mask of its non-zero positions, plane i+1 of the block = its i-th non-zero value in position order --
// which is all the statistics / bit-length / bit-writing passes behind the trellis need: they then touch as many
// planes as the busiest block of a wave has non-zero coefficients (~20 at q75) instead of 63.
template <int QN, bool LDS_ROWS, bool EXT = false, bool COMPACT = false>
__device__ __forceinline__ void trellis_q_walk(const uint4 *si_rows, const float4 *rate_rows, const int (*dqT)[64], const float (*ltT)[64],
                                               int qrow, int nq_in, float azd63, float lambda, bool active, int16_t *__restrict__ qo,
                                               int kstride, uint2 (*col)[64], unsigned short (*e_pk)[64], int lane,
                                               int Ss = 1, int Se = 63, float2 *__restrict__ eob_out = nullptr, int *__restrict__ eob_has = nullptr,
                                               unsigned long long *__restrict__ nz_out = nullptr,
                                               const int *__restrict__ dq_lane = nullptr, const float *__restrict__ lt_lane = nullptr)
{
  // dq_lane / lt_lane (EXT only): this lane's own quantizer rows in global memory instead of the LDS copies (per-image tables)
  static_assert(!(COMPACT && EXT), "compact records exist for the plain pass");
  const int vstart = EXT ? Ss - 1 : 0;          // position of the virtual start entry
  const int si_f0 = (int)(si_rows[15].x & 0xFFu), si_eob = (int)(si_rows[0].x & 0xFFu);
  const float f0f = si_f0 ? (float)si_f0 : 3e38f;
  const bool over_q = active && nq_in > QN;
  int nq = (!active || over_q) ? 0 : nq_in;     // nothing to walk (the lane stays for the wave-level loop)

  // ---- phase 2: every lane consumes its own queue; the NEXT record is always one load ahead ----
  unsigned long long live = 1ull << vstart, neg = 0ull;
  int nlive = 1;
  float2 n0 = make_float2(0.0f, 0.0f), n1 = n0;
  int qi = 0;
  bool done = nq == 0;
  int i = 0, x = 0, dq = 1, qval = 0, ncd = 0, sgn = 0, e = 0, bestp = -1, bestk = 0;
  float lti = 0.0f, azd_prev = 0.0f, azd_cur = 0.0f, best = 1e38f;
  unsigned long long m = 0ull;
  bool first = true;
  uint2 rec_n = col[0][lane];
  // one ROUND per queue record, as in k_trellis_ac_v3: every working lane sets its record up, the wave scans (pair steps)
  // until the last lane's scan has ended, every working lane commits -- each part runs with all of the round's lanes instead
  // of a handful per iteration (tools/model_sched.py).  The same operations per lane in the same order.
  while (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
    if (!done) {
      const uint2 rec = rec_n;
      qi++;
      rec_n = col[qi < QN ? qi : QN - 1][lane];          // unconsumed slots are never overwritten (entry e lives in slot e-1 <= qi-1)
      i = (int)(rec.x & 63u); sgn = (int)((rec.x >> 6) & 1u); qval = (int)((rec.x >> 7) & 1023u); x = (int)(rec.x >> 17);
      azd_prev = __uint_as_float(rec.y);
      if (EXT && dq_lane) { dq = dq_lane[i]; lti = lt_lane[i]; }
      else { dq = dqT[qrow][i]; lti = ltT[qrow][i]; }
      float t = (float)mul24(x, x) * lambda;
      t = t * lti;
      azd_cur = t + azd_prev;
      ncd = bitlen((unsigned)qval);
      best = 1e38f; bestp = -1; bestk = 0;
      m = live; e = nlive; first = true;
    }
    const bool any2 = __builtin_amdgcn_ballot_w64(!done && ncd > 1) != 0ull, any4 = __builtin_amdgcn_ballot_w64(!done && ncd > 2) != 0ull;
    if (!done) {      // (a plain divergent loop, as in k_trellis_ac_v3: lanes whose scan has ended wait masked for the last one)
      bool fin = false;
      do {
        if (!any2)
          q_pair_step<1, LDS_ROWS>(si_rows, rate_rows, col, lane, m, e, first, n0, n1, azd_prev, i, x, dq, qval, ncd, lambda, lti, si_f0, f0f, best, bestp, bestk, fin);
        else if (!any4)
          q_pair_step<2, LDS_ROWS>(si_rows, rate_rows, col, lane, m, e, first, n0, n1, azd_prev, i, x, dq, qval, ncd, lambda, lti, si_f0, f0f, best, bestp, bestk, fin);
        else
          q_pair_step<4, LDS_ROWS>(si_rows, rate_rows, col, lane, m, e, first, n0, n1, azd_prev, i, x, dq, qval, ncd, lambda, lti, si_f0, f0f, best, bestp, bestk, fin);
      } while (!fin);
    }
    if (!done) {
      if (bestp >= 0) {
        const int mag = (bestk < ncd - 1) ? (2 << bestk) - 1 : qval;
        n1 = n0;
        n0 = make_float2(azd_cur, best);
        col[nlive - 1][lane] = make_uint2(__float_as_uint(azd_cur), __float_as_uint(best));   // live entry nlive
        e_pk[nlive][lane] = (unsigned short)(bestp | (mag << 6));
        live |= 1ull << i;
        if (sgn) neg |= 1ull << i;
        nlive++;
      }
      done = qi >= nq;
    }
  }
  const bool work = active && !over_q;
  unsigned pk[QN <= 24 ? QN + 1 : 1];
  if (QN <= 24) {
#pragma unroll
    for (int e2 = 1; e2 <= QN; e2++) pk[e2] = e_pk[e2][lane];
  }
  if (!work) return;

  // ---- end-of-block choice (jcdctmgr.c:1187-1207): independent loads of every live entry, then the scan in position order
  float best_cost = azd63 + (float)si_eob;
  float best_skip = azd63;                       // EXT: best_cost_skip (:1190, :1203)
  const int last_pos = EXT ? Se : 63;
  int last = vstart;
  {
    unsigned long long mm = live & ~(1ull << vstart);
    if (QN <= 24) {
      uint2 ent[QN];
#pragma unroll
      for (int s2 = 0; s2 < QN; s2++) ent[s2] = col[s2][lane];
#pragma unroll
      for (int s2 = 0; s2 < QN; s2++) {
        if (s2 + 1 < nlive) {
          const int p = __builtin_ctzll(mm);
          mm &= mm - 1;
          float cost = __uint_as_float(ent[s2].y) + azd63;
          cost = cost - __uint_as_float(ent[s2].x);
          const float wo = cost;
          if (p < last_pos) cost = cost + (float)si_eob;
          if (cost < best_cost) { best_cost = cost; last = p; if (EXT) best_skip = wo; }
        }
      }
    } else {
      for (int s2 = 0; s2 + 1 < nlive; s2++) {
        const uint2 en = col[s2][lane];
        const int p = __builtin_ctzll(mm);
        mm &= mm - 1;
        float cost = __uint_as_float(en.y) + azd63;
        cost = cost - __uint_as_float(en.x);
        const float wo = cost;
        if (p < last_pos) cost = cost + (float)si_eob;
        if (cost < best_cost) { best_cost = cost; last = p; if (EXT) best_skip = wo; }
      }
    }
  }
  if (EXT && eob_out) {
    *eob_out = make_float2(azd63, best_skip);
    *eob_has = (last < Se ? 1 : 0) + (last == vstart ? 1 : 0);   // :1209
  }
  // ---- back-track (jcdctmgr.c:1211-1222): the path is followed newest entry first, the values travel through the
  // lane's LDS column (64 int16 = 16 slots; the queue is dead by now) so that the 63 plane stores use static registers
  {
    typedef unsigned short __attribute__((may_alias)) us_alias;
    typedef uint2 __attribute__((may_alias)) u2_alias;
    u2_alias *colw = reinterpret_cast<u2_alias *>(&col[0][0]);
    us_alias *colh = reinterpret_cast<us_alias *>(&col[0][0]);   // value of position k: row k>>2, half-word k&3
    unsigned long long mm = live & ~(1ull << vstart);
    int p = last;
    unsigned long long pmask = 0ull;   // COMPACT: positions on the path; their values go to slots 0,1,.. in visiting (descending) order
    int cnt = 0;
    if (QN <= 24) {
      if (!COMPACT) {
#pragma unroll
        for (int r = 0; r < 16; r++) colw[r * 64 + lane] = make_uint2(0u, 0u);
      }
#pragma unroll
      for (int e2 = QN; e2 >= 1; e2--) {
        if (e2 < nlive) {
          const int pos = 63 - __builtin_clzll(mm);
          mm &= ~(1ull << pos);
          if (pos == p) {
            const int mag = (int)(pk[e2] >> 6);
            const int v = ((neg >> pos) & 1ull) ? -mag : mag;
            const int slot = COMPACT ? cnt : pos;
            colh[((slot >> 2) * 64 + lane) * 4 + (slot & 3)] = (unsigned short)v;
            if (COMPACT) { pmask |= 1ull << pos; cnt++; }
            p = (int)(pk[e2] & 63u);
          }
        }
      }
    } else {
      if (!COMPACT) {
#pragma unroll
        for (int r = 0; r < 16; r++) colw[r * 64 + lane] = make_uint2(0u, 0u);
      }
      for (int e2 = nlive - 1; e2 >= 1; e2--) {
        const int pos = 63 - __builtin_clzll(mm);
        mm &= ~(1ull << pos);
        if (pos == p) {
          const unsigned pkv = e_pk[e2][lane];
          const int mag = (int)(pkv >> 6);
          const int v = ((neg >> pos) & 1ull) ? -mag : mag;
          const int slot = COMPACT ? cnt : pos;
          colh[((slot >> 2) * 64 + lane) * 4 + (slot & 3)] = (unsigned short)v;
          if (COMPACT) { pmask |= 1ull << pos; cnt++; }
          p = (int)(pkv & 63u);
        }
      }
    }
    if (COMPACT) {
      *nz_out = pmask;
      // plane i+1 <- the i-th non-zero in position order = slot cnt-1-i (per-lane LDS address); a plane is stored only
      // while some block of the wave still has a value for it
#pragma unroll
      for (int i = 0; i < (QN < 63 ? QN : 63); i++) {
        if (__builtin_amdgcn_ballot_w64(i < cnt) == 0ull) break;
        if (i < cnt) {
          const int slot = cnt - 1 - i;
          qo[(size_t)(i + 1) * kstride] = (int16_t)colh[((slot >> 2) * 64 + lane) * 4 + (slot & 3)];
        }
      }
      return;
    }
    uint2 vals[16];
#pragma unroll
    for (int r = 0; r < 16; r++) vals[r] = colw[r * 64 + lane];
#pragma unroll
    for (int k = 1; k < 64; k++) {
      if (EXT && (k < Ss || k > Se)) continue;   // positions outside the band keep what they hold
      const unsigned w = (k & 2) ? vals[k >> 2].y : vals[k >> 2].x;
      qo[(size_t)k * kstride] = (int16_t)((k & 1) ? (w >> 16) : (w & 0xFFFFu));
    }
  }
}

// =============================================================================================
// trellis_eob_opt: jcdctmgr.c:1224-1297.  After the per-block DP of a band the reference walks
// every block row once more: the cheapest way to reach block bi through runs of blocks whose band is all zero (coded as
// EOBRUN symbols in progressive mode), then the blocks inside the chosen runs lose their band.  The recursion is
// sequential in float along the row (abc[bi+1] needs abc[0..bi]) and quadratic (every earlier block is a candidate
// start), so: one wave per (image, component, block row); for each bi the candidates i = lane, lane + 64, ... are
// evaluated in parallel and reduced to the FIRST minimum (the reference's strict '<' in increasing i).
// LDS: azbc / abc (float), requires_eob (byte), block_run_start (16 bit), zero flags: 13 bytes per block.
// =============================================================================================
__device__ __forceinline__ void wave_first_min(float &c, int &idx)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float oc = __shfl_xor(c, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    if (oc < c || (oc == c && oi < idx)) { c = oc; idx = oi; }
  }
}

__global__ void __launch_bounds__(64)
k_trellis_eob_chain(MjhConst C, int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
                    int4 ac_slot_of_comp, int4 row0_of_comp, const float2 *__restrict__ eob_cost, const int *__restrict__ eob_has, int Ss, int Se)
{
  HIP_DYNAMIC_SHARED(unsigned char, dyn_lds);
  const int img = blockIdx.y, row = blockIdx.x, lane = threadIdx.x;
  const int comp = row >= row0_of_comp.w ? 3 : row >= row0_of_comp.z ? 2 : row >= row0_of_comp.y ? 1 : 0;
  const int r0 = comp == 0 ? 0 : comp == 1 ? row0_of_comp.y : comp == 2 ? row0_of_comp.z : row0_of_comp.w;
  const MjhComp cc = C.c[comp];
  const int br = row - r0, n = cc.wib;
  const int slot = comp == 0 ? ac_slot_of_comp.x : comp == 1 ? ac_slot_of_comp.y : comp == 2 ? ac_slot_of_comp.z : ac_slot_of_comp.w;
  const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  float *azbc = reinterpret_cast<float *>(dyn_lds);                  // [n + 1]
  float *abc = azbc + (n + 1);                                        // [n + 1]
  float *skip = abc + (n + 1);                                        // [n]     best_cost_skip of every block
  unsigned short *run_start = reinterpret_cast<unsigned short *>(skip + n);   // [n]
  unsigned char *req = reinterpret_cast<unsigned char *>(run_start + n);      // [n + 1]
  unsigned char *zero = req + (n + 1);                                // [n]
  __shared__ float eobrun_cost[16];                                   // (float)(size[16 * nb] + nb)
  if (lane < 16) eobrun_cost[lane] = (float)((int)T->ehufsi[16 * lane] + lane);
  const size_t g0 = (size_t)img * C.total_real_blocks + cc.blk_off + (size_t)br * n;
  for (int b = lane; b < n; b += 64) {
    const float2 c2 = eob_cost[g0 + b];
    abc[b + 1] = c2.x;            // parked here until the prefix below has consumed it
    skip[b] = c2.y;
    req[b + 1] = (unsigned char)eob_has[g0 + b];
    zero[b] = 0;
  }
  __syncthreads();
  if (lane == 0) {   // azbc[bi+1] = azbc[bi] + cost_all_zeros(bi): a float sum in block order (:1226-1227)
    float a = 0.0f;
    azbc[0] = 0.0f; abc[0] = 0.0f; req[0] = 0;
    for (int b = 0; b < n; b++) { a = a + abc[b + 1]; azbc[b + 1] = a; }
  }
  __syncthreads();
  for (int bi = 0; bi < n; bi++) {
    if (req[bi + 1] != 2) {        // wave-uniform
      const float sk = skip[bi], ab = azbc[bi];
      float best = 1e38f;
      int bidx = 0x7FFFFFFF;
      for (int i = lane; i <= bi; i += 64) {
        const int rq = req[i];
        if (rq == 2) continue;
        float cost = sk;
        cost = cost + ab;
        cost = cost - azbc[i];
        cost = cost + abc[i];
        cost = cost + eobrun_cost[bitlen((unsigned)(bi - i + rq))];
        if (cost < best) { best = cost; bidx = i; }
      }
      wave_first_min(best, bidx);
      if (lane == 0) { abc[bi + 1] = best; run_start[bi] = (unsigned short)bidx; }
    }
    __syncthreads();
  }
  // end of the last run (:1259-1276; NB the reference leaves abc out of this one), then the back-track (:1278-1293)
  {
    float best = 1e38f;
    int bidx = 0x7FFFFFFF;
    const float an = azbc[n];
    for (int i = lane; i <= n; i += 64) {
      const int rq = req[i];
      if (rq == 2) continue;
      float cost = 0.0f;
      cost = cost + an;
      cost = cost - azbc[i];
      cost = cost + eobrun_cost[bitlen((unsigned)(n - i + rq))];
      if (cost < best) { best = cost; bidx = i; }
    }
    wave_first_min(best, bidx);
    if (lane == 0) {
      int last_block = bidx - 1, bi = n - 1;
      while (bi >= 0) {
        while (bi > last_block) { zero[bi] = 1; bi--; }
        if (bi < 0) break;
        last_block = (int)run_start[bi] - 1;
        bi--;
      }
    }
  }
  __syncthreads();
  int16_t *qo = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + (size_t)br * n;
  for (int b = lane; b < n; b += 64)
    if (zero[b])
      for (int k = Ss; k <= Se; k++) qo[(size_t)k * cc.kstride + b] = 0;
}

// =============================================================================================
// trellis_q_opt: jcdctmgr.c:1299-1306 sums, per quantization table and coefficient, raw * quantized and
// 8 * quantized^2 over every block of every trellis pass; jcmaster.c:1014-1030 turns them into new table entries
// q = clamp((int)(sum_src / sum_coef + 0.5), 1, 254) for the coefficients with a non-zero denominator.  Every term is an
// integer and the totals stay far below 2^53, so the reference's double sums are exact and equal these 64-bit integer
// sums whatever the order; the division is one IEEE double division, the same on both sides.
// sums[image][table][64][2] (signed 64 bit).
// =============================================================================================
__global__ void __launch_bounds__(256)
k_qopt_accumulate(MjhConst C, const int16_t *__restrict__ coef_uq, const int16_t *__restrict__ coef_q, long long *__restrict__ sums)
{
  const int k = blockIdx.x + 1, comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int16_t *u = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off + (size_t)k * cc.kstride;
  const int16_t *q = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + (size_t)k * cc.kstride;
  long long a = 0, b = 0;
  for (int i = threadIdx.x; i < cc.nblk; i += 256) {
    const int x = u[i], v = q[i];
    a += (long long)(x * v);
    b += (long long)(8 * v * v);
  }
  __shared__ long long sa[256], sb[256];
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) { sa[threadIdx.x] += sa[threadIdx.x + off]; sb[threadIdx.x] += sb[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    long long *d = sums + (((size_t)img * 4 + cc.qtbl) * 64 + k) * 2;
    atomicAdd(reinterpret_cast<unsigned long long *>(d), (unsigned long long)sa[0]);
    atomicAdd(reinterpret_cast<unsigned long long *>(d + 1), (unsigned long long)sb[0]);
  }
}

// end of a group of num_components passes (finish_pass_master jcmaster.c:1014-1030): new entries of every table with a
// non-zero denominator into the image's own MjhQuant (all four derived rows); the sums start over (prepare_for_pass :687-698)
__global__ void __launch_bounds__(64)
k_qopt_update(long long *__restrict__ sums, MjhQuant *__restrict__ Q)
{
  const int img = blockIdx.x, t = blockIdx.y, k = threadIdx.x;
  long long *d = sums + (((size_t)img * 4 + t) * 64 + k) * 2;
  const long long a = d[0], b = d[1];
  d[0] = 0; d[1] = 0;
  if (k == 0 || b == 0) return;
  int q = (int)((double)a / (double)b + 0.5);
  if (q > 254) q = 254;
  if (q < 1) q = 1;
  MjhQuant *Qi = Q + img;
  Qi->q[t][k] = (uint16_t)q;
  Qi->dq8[t][k] = 8 * q;
  Qi->rcp8q[t][k] = 1.0f / (float)(8 * q);
  Qi->thr8[t][k] = (float)(8 * q - ((8 * q) >> 1));
  Qi->dqc8[t][k] = 8 * q;                       // (q <= 254: nothing wraps)
  Qi->rcpc8q[t][k] = 1.0f / (float)(8 * q);
  Qi->lambda_tbl[t][k] = (float)(1.0 / (double)(q * q));
  {   // (q <= 254: the multiply-high constants exist)
    const unsigned dd = 8u * (unsigned)q;
    const int kk = min(32, 22 + bitlen(dd));
    Qi->mdiv[t][k] = (uint32_t)((1ull << kk) / dd + 1ull);
    Qi->sdiv[t][k] = 32 - kk;
  }
}

// trellis_q_opt, end of the encode: the image's FINAL tables -> the DQT segment(s) of its finished file, rebuilt at the
// precision the final values ask for (emit_multi_dqt / emit_dqt jcmarker.c:140-254 look at quantval > 255 when they WRITE the
// marker, i.e. after finish_pass_master jcmaster.c:1014-1030 has replaced the estimated entries by values <= 254: a table that
// started with 16-bit entries keeps those the estimate never touched -- the DC entry, coefficients no block quantizes to
// non-zero -- and turns into an 8-bit table only if none of them is left).  The file was assembled with the DQT layout of the
// ORIGINAL tables; if a table shrinks, everything behind the segment moves up and a baseline-capable frame gets SOF0
// instead of SOF1 (write_frame_header :699-734).  One workgroup per image.
struct MjhDqtLayout {
  int dqt_start, sof_off;     // [dqt_start, sof_off): the DQT segment(s) as first written; the SOF marker follows
  int ntab, tab[4];           // tables in marker order
  int multi;                  // one DQT marker with all tables (max-compression profile) / one marker per table
  int baseline_capable;       // sequential Huffman, 8-bit samples, table numbers <= 1: SOF0 unless a table is 16-bit
};

__global__ void __launch_bounds__(256)
k_qopt_fix(const MjhQuant *__restrict__ Q, uint8_t *__restrict__ out, size_t out_stride, unsigned *__restrict__ sizes, MjhDqtLayout L)
{
  __shared__ uint8_t nd[4 * (4 + 1 + 128) + 16];
  __shared__ int s_len, s_any16;
  const int img = blockIdx.x, tid = threadIdx.x;
  const unsigned size = sizes[img];
  if (size == 0) return;
  uint8_t *f = out + (size_t)img * out_stride;
  const MjhQuant *Qi = Q + img;
  if (tid == 0) {
    int prec[4], any = 0, n = 0;
    for (int i = 0; i < L.ntab; i++) {
      prec[i] = 0;
      for (int k = 0; k < 64; k++) if (Qi->q[L.tab[i]][k] > 255) prec[i] = 1;
      any |= prec[i];
    }
    if (L.multi) {
      int sz = 2;
      for (int i = 0; i < L.ntab; i++) sz += 64 * (prec[i] + 1) + 1;
      nd[n++] = 0xFF; nd[n++] = 0xDB; nd[n++] = (uint8_t)(sz >> 8); nd[n++] = (uint8_t)sz;
    }
    for (int i = 0; i < L.ntab; i++) {
      if (!L.multi) {
        const int sz = 64 * (prec[i] + 1) + 1 + 2;
        nd[n++] = 0xFF; nd[n++] = 0xDB; nd[n++] = (uint8_t)(sz >> 8); nd[n++] = (uint8_t)sz;
      }
      nd[n++] = (uint8_t)(L.tab[i] + (prec[i] << 4));
      for (int k = 0; k < 64; k++) {          // MjhQuant.q is in zig-zag order, like the marker
        const unsigned qv = Qi->q[L.tab[i]][k];
        if (prec[i]) nd[n++] = (uint8_t)(qv >> 8);
        nd[n++] = (uint8_t)(qv & 0xFF);
      }
    }
    s_len = n; s_any16 = any;
  }
  __syncthreads();
  const int old_len = L.sof_off - L.dqt_start, new_len = s_len, delta = old_len - new_len;   // (>= 0: entries only ever shrink)
  if (delta > 0) {
    // move [sof_off, size) up by delta: ascending chunks, every chunk read completely before it is written
    for (unsigned base = (unsigned)L.sof_off; base < size; base += 256u * 16u) {
      uint8_t v[16];
      const unsigned at = base + (unsigned)tid * 16u;
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = at + j < size ? f[at + j] : (uint8_t)0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; j++) if (at + j < size) f[at + j - delta] = v[j];
      __syncthreads();
    }
  }
  for (int i = tid; i < new_len; i += 256) f[L.dqt_start + i] = nd[i];
  if (tid == 0) {
    if (L.baseline_capable) f[L.sof_off - delta + 1] = s_any16 ? 0xC1 : 0xC0;
    sizes[img] = size - (unsigned)delta;
  }
}

// the 16-byte headers of up to four work-list pairs (one pair per image range of a chunked AC trellis), `step` words apart
__global__ void k_zero_counters(unsigned *__restrict__ a, unsigned *__restrict__ b, int4 off)
{
  const int t = threadIdx.x & 3, c = threadIdx.x >> 2;
  const int o = c == 0 ? off.x : c == 1 ? off.y : c == 2 ? off.z : off.w;
  if (threadIdx.x < 16 && o >= 0) { a[o + t] = 0; b[o + t] = 0; }
}

// work-list entries: 3 words per deferred block at [4 + 3i]: image, component << 28 | block, slot of its dense copy
__device__ __forceinline__ void defer_blocks(bool mine, unsigned *__restrict__ list, unsigned img, unsigned compblk, unsigned dense_slot_in,
                                             const short (&xs)[64], int16_t *__restrict__ dense, unsigned dense_cap, bool make_copy, int lane)
{
  const unsigned long long over = __ballot(mine);
  if (over == 0ull) return;
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(&list[0], (unsigned)__popcll(over));   // one atomic per wave, consecutive slots
  base = __shfl(base, 0, 64);
  if (!mine) return;
  const unsigned idx = base + (unsigned)__popcll(over & ((1ull << lane) - 1ull));
  list[4 + 3 * (size_t)idx] = img;
  list[5 + 3 * (size_t)idx] = compblk;
  list[6 + 3 * (size_t)idx] = make_copy ? idx : dense_slot_in;
  if (make_copy && idx < dense_cap) {
    // the raw coefficients are still in registers: one 128-byte line per block for the next kernel, instead of 63
    // different DRAM sectors in the coefficient-major planes
    uint4 *d = reinterpret_cast<uint4 *>(dense + (size_t)idx * 64);
#pragma unroll
    for (int v = 0; v < 8; v++) {
      unsigned w[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = 8 * v + 2 * j;
        w[j] = (k == 0 ? 0u : (unsigned)(unsigned short)xs[k]) | ((unsigned)(unsigned short)xs[k + 1] << 16);
      }
      d[v] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// What the host sizes the first tier's queue capacity by (mjh_encoder.cpp, run_pipeline): list[1..3] = blocks of the batch with
// more than 16 / 24 / 32 queue records, whatever the capacity of the kernel that counts them -- the SAME quantity at every
// capacity, so the choice made from it is a fixed point for a steady workload.  Three atomics per wave and pass.
__device__ __forceinline__ void count_heavy(unsigned *__restrict__ list, bool inside, int nq, int lane)
{
  const unsigned long long b16 = __ballot(inside && nq > 16);
  if (b16 == 0ull) return;
  const unsigned long long b24 = __ballot(inside && nq > 24), b32 = __ballot(inside && nq > 32);
  if (lane == 0) {
    atomicAdd(&list[1], (unsigned)__popcll(b16));
    if (b24) atomicAdd(&list[2], (unsigned)__popcll(b24));
    if (b32) atomicAdd(&list[3], (unsigned)__popcll(b32));
  }
}

// band limits + the per-block outputs of trellis_eob_opt ([image][real blocks of all components]); EXT kernels only
struct MjhTrellisExt {
  int Ss, Se;
  float2 *eob_cost;   // {cost of the all-zero band, cost of the chosen path without its EOB}; null: trellis_eob_opt off
  int *eob_has;       // has_eob 0 / 1 / 2 (jcdctmgr.c:1209)
  unsigned long long *nzmask;   // COMPACT instantiations: non-zero position mask per block, [image][real blocks of all components]
  int qstride;        // EXT instantiations: 1 = one MjhQuant per image (trellis_q_opt re-estimates the tables between passes), 0 = shared
};

template <int QN, bool EXT = false, bool COMPACT = false>
__global__ void __launch_bounds__(64)
k_trellis_ac_q(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
               int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
               int4 ac_slot_of_comp, int4 wave0_of_comp, const float *__restrict__ lambda_in, unsigned *__restrict__ worklist,
               int16_t *__restrict__ dense, unsigned dense_cap, MjhTrellisExt ext)
{
  static_assert(QN >= 16 && QN <= 63, "queue capacity");
  __shared__ uint2 col[QN][64];                  // queue records, then live entries {azd, acc}, then the value column
  __shared__ unsigned short e_pk[QN + 1][64];    // back position | magnitude << 6 of live entry e
  __shared__ uint4 si_rows[16];
  __shared__ float4 rate_rows[16];
  __shared__ int dqT[1][64];
  __shared__ float ltT[1][64];
  // flattened grid: blockIdx.x counts the waves of all components of one image (wave0_of_comp = first wave of each)
  const int img = blockIdx.y;
  const int wv = blockIdx.x;
  const int comp = wv >= wave0_of_comp.w ? 3 : wv >= wave0_of_comp.z ? 2 : wv >= wave0_of_comp.y ? 1 : 0;
  const int w0 = comp == 0 ? 0 : comp == 1 ? wave0_of_comp.y : comp == 2 ? wave0_of_comp.z : wave0_of_comp.w;
  const MjhComp cc = C.c[comp];
  const int lane = threadIdx.x;
  const int slot = comp == 0 ? ac_slot_of_comp.x : comp == 1 ? ac_slot_of_comp.y : comp == 2 ? ac_slot_of_comp.z : ac_slot_of_comp.w;
  const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  const int blk = (wv - w0) * 64 + lane;
  const bool inside = blk < cc.nblk;
  const float lambda = lambda_in[(size_t)img * C.total_real_blocks + cc.blk_off + (inside ? blk : cc.nblk - 1)];
  if (lane < 16) {
    const uint4 r = reinterpret_cast<const uint4 *>(T->ehufsi)[lane];
    si_rows[lane] = r;
    rate_rows[lane] = rate_row(r);
  }
  if (EXT) Q += (size_t)img * ext.qstride;
  dqT[0][lane] = Q->dq8[cc.qtbl][lane];
  ltT[0][lane] = Q->lambda_tbl[cc.qtbl][lane];
  int nq;
  float azd63;
  {
    // all 63 raw coefficients at once (coalesced lines, one burst); they are dead after phase 1 / the dense copy
    const int16_t *uq = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off + (inside ? blk : cc.nblk - 1);
    short xs[64];
#pragma unroll
    for (int k = 1; k < 64; k++) xs[k] = uq[(size_t)k * cc.kstride];
    nq = trellis_q_phase1<QN, EXT>(xs, Q->dq8[cc.qtbl], Q->rcp8q[cc.qtbl], Q->lambda_tbl[cc.qtbl], lambda, col, lane, azd63, ext.Ss, ext.Se);
    defer_blocks(inside && nq > QN, worklist, (unsigned)img, ((unsigned)comp << 28) | (unsigned)blk, 0u, xs, dense, dense_cap, true, lane);
    if (!EXT) count_heavy(worklist, inside, nq, lane);
  }
  __syncthreads();
  int16_t *qo = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
  const size_t gblk = (size_t)img * C.total_real_blocks + cc.blk_off + (inside ? blk : 0);
  trellis_q_walk<QN, true, EXT, COMPACT>(si_rows, rate_rows, dqT, ltT, 0, nq, azd63, lambda, inside, qo, cc.kstride, col, e_pk, lane,
                                         ext.Ss, ext.Se, EXT && ext.eob_cost ? ext.eob_cost + gblk : nullptr, EXT && ext.eob_cost ? ext.eob_has + gblk : nullptr,
                                         COMPACT ? ext.nzmask + gblk : nullptr);
}

// Deferred blocks (any image / component per lane), same walk with a longer queue: raw coefficients come from the dense
// copies (one line per block), code lengths from the image's table in global memory (L2-resident), quantizer constants
// of all four tables from LDS.  Blocks beyond QN2 non-zero positions go to the next list (QN2 = 63 takes everything).
template <int QN2, bool EXT = false, bool COMPACT = false>
__global__ void __launch_bounds__(64)
k_trellis_ac_qd(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
                int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
                int4 ac_slot_of_comp, const float *__restrict__ lambda_in, const unsigned *__restrict__ worklist,
                unsigned *__restrict__ worklist_next, const int16_t *__restrict__ dense, unsigned dense_cap, MjhTrellisExt ext)
{
  __shared__ uint2 col[QN2][64];
  __shared__ unsigned short e_pk[QN2 + 1][64];
  __shared__ int dqT[4][64];
  __shared__ float ltT[4][64];
  const int lane = threadIdx.x;
#pragma unroll
  for (int t = 0; t < 4; t++) { dqT[t][lane] = Q->dq8[t][lane]; ltT[t][lane] = Q->lambda_tbl[t][lane]; }
  __syncthreads();
  const unsigned count = worklist[0];
  for (unsigned base = blockIdx.x * 64; base < count; base += gridDim.x * 64) {   // wave-uniform trip count
    const unsigned it = base + lane;
    const bool active = it < count;
    const unsigned ii = active ? it : count - 1;
    const int img = (int)worklist[4 + 3 * (size_t)ii];
    const unsigned w = worklist[5 + 3 * (size_t)ii];
    const unsigned ds = worklist[6 + 3 * (size_t)ii];
    const int comp = (int)(w >> 28);
    const MjhComp cc = C.c[comp];
    const int blk = (int)(w & 0x0FFFFFFFu);
    const int slot = comp == 0 ? ac_slot_of_comp.x : comp == 1 ? ac_slot_of_comp.y : comp == 2 ? ac_slot_of_comp.z : ac_slot_of_comp.w;
    const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
    const float lambda = lambda_in[(size_t)img * C.total_real_blocks + cc.blk_off + blk];
    int nq;
    float azd63;
    {
      short xs[64];
      if (ds < dense_cap) {
        const uint4 *d = reinterpret_cast<const uint4 *>(dense + (size_t)ds * 64);
#pragma unroll
        for (int v = 0; v < 8; v++) {
          const uint4 q4 = d[v];
          const unsigned ww[4] = { q4.x, q4.y, q4.z, q4.w };
#pragma unroll
          for (int j = 0; j < 4; j++) { xs[8 * v + 2 * j] = (short)(ww[j] & 0xFFFFu); xs[8 * v + 2 * j + 1] = (short)(ww[j] >> 16); }
        }
      } else {
        const int16_t *uq = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
#pragma unroll
        for (int k = 1; k < 64; k++) xs[k] = uq[(size_t)k * cc.kstride];
      }
      if (EXT && ext.qstride) {
        const MjhQuant *Qi = Q + (size_t)img * ext.qstride;   // per-lane image: vector loads of its own rows
        nq = trellis_q_phase1<QN2, EXT>(xs, Qi->dq8[cc.qtbl], Qi->rcp8q[cc.qtbl], Qi->lambda_tbl[cc.qtbl], lambda, col, lane, azd63, ext.Ss, ext.Se);
      } else
      nq = trellis_q_phase1<QN2, EXT>(xs, dqT[cc.qtbl], Q->rcp8q[cc.qtbl], ltT[cc.qtbl], lambda, col, lane, azd63, ext.Ss, ext.Se);
      if (QN2 < 63) defer_blocks(active && nq > QN2, worklist_next, (unsigned)img, w, ds, xs, nullptr, 0u, false, lane);
    }
    int16_t *qo = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
    const size_t gblk = (size_t)img * C.total_real_blocks + cc.blk_off + blk;
    trellis_q_walk<QN2, false, EXT, COMPACT>(reinterpret_cast<const uint4 *>(T->ehufsi), nullptr, dqT, ltT, cc.qtbl, nq, azd63, lambda, active, qo, cc.kstride,
                                                              col, e_pk, lane, ext.Ss, ext.Se, EXT && ext.eob_cost ? ext.eob_cost + gblk : nullptr,
                                                              EXT && ext.eob_cost ? ext.eob_has + gblk : nullptr, COMPACT ? ext.nzmask + gblk : nullptr,
                                                              EXT && ext.qstride ? (Q + (size_t)img * ext.qstride)->dq8[cc.qtbl] : nullptr,
                                                              EXT && ext.qstride ? (Q + (size_t)img * ext.qstride)->lambda_tbl[cc.qtbl] : nullptr);
    __syncthreads();   // the LDS columns are reused by the next round
  }
}

// =============================================================================================
// K5 v3  AC trellis, tile-sorted passes + lean walk (the plain 1..63 pass with compact record output: the metric's
// configuration).  Same DP, same float recipe, same outputs as k_trellis_ac_q<.., COMPACT>; what changed is the shape:
//  * TILE-SORTED PASSES.  A wave's walk lasts as long as its busiest lane (measured before: 29.5 of 64 lanes active per
//    VALU instruction, profiles/r03a_pmc_sq_batch64.json; tools/model_trellis.py: 0.43 of the lane-steps useful).  One
//    workgroup (= one wave) now owns a tile of 64*NPASS consecutive blocks, counting-sorts them by the number of non-zero
//    conventionally quantized AC coefficients (one byte per block, written by the FDCT kernel) and runs NPASS passes,
//    heaviest blocks first, so that a pass carries blocks of similar weight (model: 0.60 at 4 passes, 0.67 at 8).  Any
//    assignment of blocks to lanes gives the same files; the tile's coefficients stay within the same cache lines.
//    A pass whose blocks all have key 0 has nothing to decide (every AC coefficient stays zero): it only writes empty masks.
//  * LEAN WALK.  Live entries carry their own position (info word: position | back ENTRY index | magnitude | sign), so a
//    step is two entry loads instead of 64-bit mask arithmetic (a 16-bit word: only magnitudes below 16 get here); the candidate distortions are computed once per
//    position, not once per step; the end-of-block choice (jcdctmgr.c:1187-1207) is folded into the entry creation (entries
//    are created in position order, strict '<' keeps the first minimum); the back-track follows entry indices.
//    Blocks with a quantized magnitude >= 16 (more than 4 candidates) or more than QN queue records go to the work list
//    of the general kernels above.
// LDS per wave: (QN + 1) * (8 + 2) * 64 + 256 bytes; up to 16 records: (QN + 1) * (8 + 1) * 64 + 256 = 10 048 bytes, 16 waves per CU (SLIM below); quantizer rows travel through ds_bpermute.
// =============================================================================================
template <int NC>
__device__ __forceinline__ void v3_eval(const float4 &rr, float rb, float rhs, float d0, float d1, float d2, float d3, float &lb, int &lk)
{
  float c0 = (rr.x + rb) + d0;
  c0 = c0 + rhs;
  lb = c0; lk = 0;
  if (NC >= 2) {
    float c1 = (rr.y + rb) + d1;
    c1 = c1 + rhs;
    if (c1 < lb) { lb = c1; lk = 1; }
  }
  if (NC >= 3) {
    float c2 = (rr.z + rb) + d2;
    float c3 = (rr.w + rb) + d3;
    c2 = c2 + rhs; c3 = c3 + rhs;
    if (c2 < lb) { lb = c2; lk = 2; }
    if (c3 < lb) { lb = c3; lk = 3; }
  }
}

template <int NC>
__device__ __forceinline__ float4 v3_rate(const float4 *rate_rows, int run)
{
  if (NC <= 2) { const float2 t = *reinterpret_cast<const float2 *>(&rate_rows[run & 15]); return make_float4(t.x, t.y, 0.f, 0.f); }
  return rate_rows[run & 15];
}

// The scan of one record: pair steps over the live entries, newest first -- two entries (e-1, e-2) per step.  Entry e lives in
// slot e; entry 0, the virtual start (position 0, no distortion, no cost), is a slot like the others, written before the walk
// -- a step has no special case for it (until round 5 it was not stored and every step selected around it: ~10 of its 58
// instructions).
// SOFTWARE PIPELINE (round 6).  The kernel turned out to be as sensitive to occupancy as a latency-bound one (14 / 10 / 6 waves
// per CU: 1.47 / 1.85 / 2.46 ms, profiles/r06g_occupancy.md), and a step used to be two LDS round trips one behind the other:
// entries + info words, then -- addressed by the run lengths the info words give -- the rate rows.  The entries and info
// words of the NEXT step are now requested at the top of a step, so that a step waits for its rate rows only.  Same loads,
// same operations on the same values in the same order per lane: the files do not change.  (Indices past the oldest entry are
// clamped to 0: a harmless repeat, read only if the scan goes on.)
template <int QN, int NC, typename IT>
__device__ __forceinline__ void v3_scan(const uint2 (*col)[64], const IT (*info)[64], const float4 *rate_rows, int lane, int e, int im1,
                                        float azd_prev, float f0f, float d0, float d1, float d2, float d3,
                                        float &best, int &beste, int &bestk)
{
  int ea = e - 1, eb = e >= 2 ? e - 2 : 0;        // (e == 1: b repeats a, the same entry: harmless)
  uint2 va = col[ea][lane], vb = col[eb][lane];
  unsigned ia = info[ea][lane], ib = info[eb][lane];
  float gap_old;
  do {
    // this step's rate rows first (their addresses come from info words that are in registers), then the next step's entries:
    // LDS answers in order, so the step waits for the rate rows while the prefetch is still on its way
    // (a one-byte info word IS the position: nothing to mask)
    const int run_a = im1 - (int)(sizeof(IT) == 1 ? ia : ia & 63u), run_b = im1 - (int)(sizeof(IT) == 1 ? ib : ib & 63u);
    const float4 ra = v3_rate<NC>(rate_rows, run_a), rb4 = v3_rate<NC>(rate_rows, run_b);
    const int ea_n = e >= 3 ? e - 3 : 0, eb_n = e >= 4 ? e - 4 : 0;
    const uint2 va_n = col[ea_n][lane], vb_n = col[eb_n][lane];
    const unsigned ia_n = info[ea_n][lane], ib_n = info[eb_n][lane];
    MJH_SCHED_BARRIER();
    const float azd_a = __uint_as_float(va.x), acc_a = __uint_as_float(va.y);
    const float azd_b = __uint_as_float(vb.x), acc_b = __uint_as_float(vb.y);
    const float rba = (float)(run_a >> 4) * f0f, rbb = (float)(run_b >> 4) * f0f;
    const float gap_a = azd_prev - azd_a, gap_b = azd_prev - azd_b;
    const float rhs_a = gap_a + acc_a, rhs_b = gap_b + acc_b;
    float lba, lbb;
    int lka, lkb;
    v3_eval<NC>(ra, rba, rhs_a, d0, d1, d2, d3, lba, lka);
    v3_eval<NC>(rb4, rbb, rhs_b, d0, d1, d2, d3, lbb, lkb);
    // newest first, '<=': on equal cost the OLDER predecessor wins, as in the reference's oldest-first strict '<' scan (a cost
    // without a Huffman code is >= 3e38 and never reaches the initial 1e38)
    if (lba <= best) { best = lba; beste = ea; bestk = lka; }
    if (lbb <= best) { best = lbb; beste = eb; bestk = lkb; }
    gap_old = gap_b;
    e -= 2;
    ea = ea_n; eb = eb_n; va = va_n; vb = vb_n; ia = ia_n; ib = ib_n;
    // cost >= rhs >= gap in float arithmetic, and the gap only grows towards older entries: once it exceeds the best cost
    // no older predecessor can win or tie
  } while (e > 0 && !(gap_old > best));
}

// (Round 3 also counted the AC symbol statistics of the final coefficients in this kernel's back-track, MJH_FUSE bit 4: the
// trellis paid 0.4 ms for the 0.33 ms of the separate pass over the compact records; removed in round 5.)
template <int QN, int NPASS, bool FD>
__global__ void __launch_bounds__(64)
k_trellis_ac_v3(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q,
                const MjhHuffTable *__restrict__ tabs, int slots_per_image, int4 ac_slot_of_comp, int4 tile0_of_comp,
                const float *__restrict__ lambda_in, uint8_t *__restrict__ nq8, unsigned *__restrict__ worklist,
                int16_t *__restrict__ dense, unsigned dense_cap, unsigned long long *__restrict__ nzmask, int img0, unsigned *__restrict__ counts, int count_mask)
{
  static_assert(QN >= 16 && QN <= 63 && NPASS >= 1 && NPASS <= 8, "queue capacity / passes");
  constexpr int TILE = 64 * NPASS;
  __shared__ uint2 col[QN + 1][64];      // tile sort scratch; per pass: queue records (record r in slot r) -> live entries {azd, acc} (entry e in slot e; 0 = the virtual start) -> value column
  // live entry e at [e]: position | back entry << 6 | magnitude (< 16) << 12 (the signs: one bit per position in a register).
  // SLIM (up to 16 records): the word keeps the position only -- ONE byte -- and the back entry (< 16) and the magnitude (< 16) of
  // entry e are nibble e - 1 of two 64-bit registers.  17 slots of 9 instead of 10 bytes per lane + the rate rows = 10 048 bytes:
  // 16 waves per CU instead of 14 (the kernel follows its occupancy, profiles/r06g_occupancy.md: T(w) ~ 0.52 + 13.3 / w), and the
  // scan loop loses the two masks of its position fields.  Same operations on the same values per lane: the files do not change.
  constexpr bool SLIM = QN <= 16;
  typedef typename std::conditional<SLIM, unsigned char, unsigned short>::type info_t;
  __shared__ info_t info[QN + 1][64];
  __shared__ float4 rate_rows[16];
  typedef unsigned __attribute__((may_alias)) u_alias;
  typedef unsigned short __attribute__((may_alias)) us_alias;
  const int img = blockIdx.y + img0, tl = blockIdx.x, lane = threadIdx.x;      // (img0: first image of this launch's range of the batch)
  const int comp = tl >= tile0_of_comp.w ? 3 : tl >= tile0_of_comp.z ? 2 : tl >= tile0_of_comp.y ? 1 : 0;
  const int t0 = comp == 0 ? 0 : comp == 1 ? tile0_of_comp.y : comp == 2 ? tile0_of_comp.z : tile0_of_comp.w;
  const MjhComp cc = C.c[comp];
  const int tile_base = (tl - t0) * TILE;
  const int slot = comp == 0 ? ac_slot_of_comp.x : comp == 1 ? ac_slot_of_comp.y : comp == 2 ? ac_slot_of_comp.z : ac_slot_of_comp.w;
  const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  const size_t gblk0 = (size_t)img * C.total_real_blocks + cc.blk_off;
  if (lane < 16) rate_rows[lane] = rate_row(reinterpret_cast<const uint4 *>(T->ehufsi)[lane]);
  const int si_f0 = (int)T->ehufsi[0xF0], si_eob = (int)T->ehufsi[0];
  const float f0f = si_f0 ? (float)si_f0 : 3e38f, eobf = (float)si_eob;
  const int dq_lane = Q->dq8[cc.qtbl][lane];                    // lane k holds the row entry of position k (ds_bpermute lookups)
  const float lt_lane = Q->lambda_tbl[cc.qtbl][lane];

  // ---- tile sort: descending key; perm entry = index in tile | key << 9 ----
  unsigned long long mine0 = 0ull, mine1 = 0ull;
  {
    u_alias *hist = reinterpret_cast<u_alias *>(&col[0][0]);              // [64]
    us_alias *perm = reinterpret_cast<us_alias *>(&col[0][0]) + 128;      // [TILE], behind the histogram
    hist[lane] = 0u;
    __syncthreads();
    unsigned key[NPASS], rank[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; j++) {
      const int b = tile_base + j * 64 + lane;
      unsigned k = b < cc.nblk ? (unsigned)nq8[gblk0 + b] : 0u;
      key[j] = k > 63u ? 63u : k;
      rank[j] = atomicAdd(&hist[key[j]], 1u);
    }
    __syncthreads();
    const unsigned h = hist[63 - lane];     // lane L: blocks with key 63-L; blocks with a larger key come first
    unsigned inc = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned n = __shfl_up(inc, o, 64);
      if (lane >= o) inc += n;
    }
    __syncthreads();
    hist[63 - lane] = inc - h;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPASS; j++) perm[hist[key[j]] + rank[j]] = (unsigned short)((unsigned)(j * 64 + lane) | (key[j] << 9));
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPASS; j++) {
      const unsigned long long v = perm[j * 64 + lane];
      if (j < 4) mine0 |= v << (16 * j); else mine1 |= v << (16 * (j - 4));
    }
    __syncthreads();
  }

  us_alias *colh = reinterpret_cast<us_alias *>(&col[0][0]);   // value of slot s: row s>>2, half-word s&3 of the lane's uint2
#pragma unroll 1
  for (int pass = 0; pass < NPASS; pass++) {
    const unsigned pe = (unsigned)(((pass < 4 ? mine0 : mine1) >> (16 * (pass & 3))) & 0xFFFFull);
    const int blk = tile_base + (int)(pe & 511u);
    const bool inside = blk < cc.nblk;
    const size_t gblk = gblk0 + (inside ? blk : 0);
    if (__builtin_amdgcn_ballot_w64(inside && (pe >> 9) != 0u) == 0ull) {   // nothing quantizes to non-zero: all-zero blocks
      if (inside) nzmask[gblk] = 0ull;
      continue;
    }
    const float lambda = lambda_in[gblk0 + (inside ? blk : cc.nblk - 1)];
    int nq = 0, qmax = 0;
    float azd63;
    {
      const int16_t *uq = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off + (inside ? blk : cc.nblk - 1);
      short xs[64];
#pragma unroll
      for (int k = 1; k < 64; k++) xs[k] = uq[(size_t)k * cc.kstride];
      // (the rows' wave-uniform constants: fetched for eight positions at a time, outside the per-position branch -- one scalar-memory
      // round trip per eight positions instead of two per position; see dct_quant_body)
      constexpr int QCH = 8;
      int dq_c[QCH], sdiv_c[QCH];
      unsigned mdiv_c[QCH];
      float lt_c[QCH], rcp_c[QCH], thr_c[QCH];
      float azd = 0.0f;
#pragma unroll
      for (int k = 1; k < 64; k++) {
        if (k == 1 || (k % QCH) == 0) {
#pragma unroll
          for (int j = 0; j < QCH; j++) {
            const int kk = (k / QCH) * QCH + j;
            dq_c[j] = Q->dq8[cc.qtbl][kk];
            lt_c[j] = Q->lambda_tbl[cc.qtbl][kk];
            thr_c[j] = Q->thr8[cc.qtbl][kk];
            if (FD) { sdiv_c[j] = Q->sdiv[cc.qtbl][kk]; mdiv_c[j] = Q->mdiv[cc.qtbl][kk]; }
            else rcp_c[j] = Q->rcp8q[cc.qtbl][kk];
          }
        }
        // The position's share of the all-zero distortion needs x^2 only: from the SIGNED coefficient converted to float -- xf * xf is
        // the correctly rounded x^2, which is what (float)(x * x) is (one rounding of the same exact integer either way) -- and
        // "quantizes to non-zero" is a compare of |xf| (a source modifier) with the table's float threshold: six VALU instructions per
        // position instead of nine; |x| as an integer exists only inside the branch few positions take.
        const int xsg = xs[k];
        const float xf = (float)xsg;
        float t = (xf * xf) * lambda;
        t = t * lt_c[k % QCH];
        const float azd_cur = t + azd;
        if (__builtin_fabsf(xf) >= thr_c[k % QCH]) {
          const int x = (int)__builtin_fabsf(xf);      // (one conversion with a source modifier)
          const int dq = dq_c[k % QCH];
          int qval = FD ? udiv_mh(x + (dq >> 1), sdiv_c[k % QCH], mdiv_c[k % QCH]) : udiv_exact(x + (dq >> 1), dq, rcp_c[k % QCH]);
          if (qval >= 1024) qval = 1023;
          qmax = qval > qmax ? qval : qmax;
          // (a block with more than QN records is deferred: what its surplus records overwrite in the last slot is never read)
          col[nq < QN ? nq : QN - 1][lane] = make_uint2((unsigned)k | ((__float_as_uint(xf) >> 25) & 64u) | ((unsigned)qval << 7) | ((unsigned)x << 17), __float_as_uint(azd));      // (the sign: the float's)
          nq++;
        }
        azd = azd_cur;
      }
      azd63 = azd;
      defer_blocks(inside && (nq > QN || qmax >= 16), worklist, (unsigned)img, ((unsigned)comp << 28) | (unsigned)blk, 0u, xs, dense, dense_cap, true, lane);
      // (the whole batch's counts: the first range's header.  A large batch counts in every (count_mask + 1)-th tile only -- the host
      // scales: three more atomics per pass on the line of the work-list counter, which one word's ~88 operations per microsecond make
      // ~1.3 % of the kernel, profiles/r06g_occupancy.md)
      if ((tl & count_mask) == 0) count_heavy(counts, inside, nq, lane);
    }
    const bool work = inside && nq <= QN && qmax < 16;

    // ---- the walk: every lane consumes its own records, one record per round; the next record is always one load ahead ----
    int nlive = 1, qi = 0, last = 0;
    unsigned long long neg = 0ull;          // positions whose coefficient is negative (of the entries created so far)
    bool act = work && nq > 0;
    int i = 0, x = 0, qval = 0, ncd = 0, sgn = 0, e = 0, beste = -1, bestk = 0;
    float azd_prev = 0.0f, azd_cur = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f, best = 1e38f;
    float end_best = azd63 + eobf;
    uint2 rec_n = col[0][lane];
    col[0][lane] = make_uint2(0u, 0u);     // record 0 is in registers: its slot becomes entry 0, the virtual start (position 0, azd 0, cost 0)
    info[0][lane] = (info_t)0;
    unsigned long long back4 = 0ull, mag4 = 0ull;      // SLIM: nibble e - 1 = back entry / magnitude of live entry e
    // quantizer constants of the NEXT record's position: lane k holds entry k of the component's rows, fetched with ds_bpermute
    // at a point where every lane of the wave is enabled (a disabled source lane would read as 0)
    int dq_n = 1;
    float lt_n = 0.0f;
    auto lookup = [&]() {
      const int a = (int)(rec_n.x & 63u) << 2;
      dq_n = __builtin_amdgcn_ds_bpermute(a, dq_lane) & 0x3FFFF;      // 8q <= 8 * 32767: tells the compiler the 24-bit multiplies are exact
      lt_n = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(lt_lane)));
    };
    lookup();
    // `wide`: some lane of the round has a record with more than two candidates (quantized magnitude >= 4); without one, the
    // round's steps evaluate two candidates and the distortions of the other two are not computed
    auto setup = [&](bool wide) {
      const uint2 rec = rec_n;
      qi++;
      rec_n = col[qi][lane];      // qi <= nq <= QN: slot QN exists (a slot is overwritten only after its record was consumed: entry e lives in slot e <= qi - 1 when it is written, and record qi - 1 is in registers by then)
      i = (int)(rec.x & 63u); sgn = (int)((rec.x >> 6) & 1u); qval = (int)((rec.x >> 7) & 1023u); x = (int)(rec.x >> 17);
      azd_prev = __uint_as_float(rec.y);
      const int dq = dq_n;
      const float lti = lt_n;
      float t = squaref(x) * lambda;      // ((float)x * (float)x == (float)(x * x): one rounding of the same exact integer)
      t = t * lti;
      azd_cur = t + azd_prev;
      ncd = bitlen((unsigned)qval);
      float dd[4] = { 3e38f, 3e38f, 3e38f, 3e38f };
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k >= 2 && !wide) break;      // uniform
        const int cand = (k < ncd - 1) ? (2 << k) - 1 : qval;
        const int delta = mul24(cand, dq) - x;
        const float d = squaref(delta) * lambda;      // (|delta| <= x < 2^15)
        dd[k] = k < ncd ? d * lti : 3e38f;
      }
      d0 = dd[0]; d1 = dd[1]; d2 = dd[2]; d3 = dd[3];
      e = nlive; best = 1e38f; beste = -1; bestk = 0;
    };
    auto next_wide = [&]() { return __builtin_amdgcn_ballot_w64(act && qi < nq && ((rec_n.x >> 7) & 1023u) >= 4u) != 0ull; };   // of the records about to be set up
    bool wide = next_wide();
    if (act) setup(wide);
    // One ROUND per queue record.  A round is the scan of the lane's live entries for its current record (pair steps, run
    // until the last lane's scan has ended) and then, at a point where the wave is whole again, the commit of the new entry and
    // the setup of the next record for every working lane at once.  (Until round 5 a lane committed and set up as soon as its own
    // scan ended: with 64 lanes some lane nearly always did, so those ~100 instructions were issued in almost every iteration
    // for a handful of lanes -- tools/model_sched.py: 0.74 of the issued instructions this way.)  The same operations per lane
    // in the same order: the files do not change.  Every working lane is at record `qi` of its queue in the same round.
    while (__builtin_amdgcn_ballot_w64(act) != 0ull) {
      if (act) {
        // (a plain divergent loop: lanes whose scan has ended wait masked until the last one is done.  Written as
        // `while (ballot(scan)) if (scan) {..}` until late in round 5, the loop carried its state through a bypass block: eight
        // register copies plus a flag materialised and re-tested per pair step, 10 of its ~75 instructions)
        if (!wide) v3_scan<QN, 2, info_t>(col, info, rate_rows, lane, e, i - 1, azd_prev, f0f, d0, d1, d2, d3, best, beste, bestk);
        else v3_scan<QN, 4, info_t>(col, info, rate_rows, lane, e, i - 1, azd_prev, f0f, d0, d1, d2, d3, best, beste, bestk);
      }
      lookup();
      const bool wide_n = next_wide();
      if (act) {
        if (beste >= 0) {
          const int mag = (bestk < ncd - 1) ? (2 << bestk) - 1 : qval;
          col[nlive][lane] = make_uint2(__float_as_uint(azd_cur), __float_as_uint(best));     // live entry nlive
          if (SLIM) {
            info[nlive][lane] = (info_t)i;
            const int sh = 4 * (nlive - 1);      // (nlive >= 1; beste < nlive <= 16)
            back4 |= (unsigned long long)(unsigned)beste << sh;
            mag4 |= (unsigned long long)(unsigned)mag << sh;
          } else
          info[nlive][lane] = (info_t)((unsigned)i | ((unsigned)beste << 6) | ((unsigned)mag << 12));
          neg |= (unsigned long long)sgn << i;
          // end-of-block choice (jcdctmgr.c:1187-1207): entries appear in position order, strict '<' keeps the first minimum
          float c = best + azd63;
          c = c - azd_cur;
          if (i < 63) c = c + eobf;
          if (c < end_best) { end_best = c; last = nlive; }
          nlive++;
        }
        if (qi >= nq) act = false;
        else setup(wide_n);
      }
      wide = wide_n;
    }

    // ---- back-track (jcdctmgr.c:1211-1222) along the entry indices; values in visiting (descending position) order ----
    unsigned long long pmask = 0ull;
    int cnt = 0, e2 = work ? last : 0;
    while (__builtin_amdgcn_ballot_w64(e2 > 0) != 0ull) {
      if (e2 > 0) {
        const unsigned inf = info[e2][lane];
        const int sh = 4 * (e2 - 1);
        const int mag = SLIM ? (int)((mag4 >> sh) & 15ull) : (int)(inf >> 12), pos = SLIM ? (int)inf : (int)(inf & 63u);
        const int v = ((neg >> pos) & 1ull) ? -mag : mag;
        colh[((cnt >> 2) * 64 + lane) * 4 + (cnt & 3)] = (unsigned short)v;
        pmask |= 1ull << pos;
        cnt++;
        e2 = SLIM ? (int)((back4 >> sh) & 15ull) : (int)((inf >> 6) & 63u);
      }
    }
    if (work) nzmask[gblk] = pmask;
    {
      int16_t *qo = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
      // plane i+1 <- the i-th non-zero in position order = slot cnt-1-i; a plane is stored only while some block has a value for it
#pragma unroll
      for (int i2 = 0; i2 < QN; i2++) {
        if (__builtin_amdgcn_ballot_w64(i2 < cnt) == 0ull) break;
        if (i2 < cnt) {
          const int s2 = cnt - 1 - i2;
          qo[(size_t)(i2 + 1) * cc.kstride] = (int16_t)colh[((s2 >> 2) * 64 + lane) * 4 + (s2 & 3)];
        }
      }
    }
    __syncthreads();
  }
}

// =============================================================================================
// host-side launch wrappers
// =============================================================================================
void mjh_launch_trellis_ac(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, MjhHuffTable *tabs, int spi, const int ac_slot[4], const float *lambda,
                           unsigned *worklist, unsigned *worklist2, void *dense, unsigned dense_cap, int variant,
                           int Ss, int Se, void *eob_cost, int *eob_has, unsigned long long *nzmask, int qstride, int n, hipStream_t s,
                           uint8_t *nq8, int v3_passes, int fastdiv, hipEvent_t after_first_tier, hipEvent_t after_first_tier2,
                           int chunks, hipStream_t side, hipEvent_t *ev_chunk, int count_mask)
{
  // band-limited pass (use_scans_in_trellis), the per-block outputs of trellis_eob_opt, per-image tables (trellis_q_opt):
  // the EXT instantiations
  const bool extended = Ss != 1 || Se != 63 || eob_cost != nullptr || qstride != 0;
  MjhTrellisExt ext;
  ext.Ss = Ss; ext.Se = Se; ext.eob_cost = (float2 *)eob_cost; ext.eob_has = eob_has; ext.nzmask = nzmask; ext.qstride = qstride;
  const int4 sl = make_int4(ac_slot[0], ac_slot[1], ac_slot[2], ac_slot[3]);
  const bool sorted = !extended && nzmask && nq8 && v3_passes > 0 && variant <= 4;
  // Image ranges of the tile-sorted tier (see below): range c owns the work-list pair at word offset wo[c] (its own 16-byte
  // header; room for every block of its images) and its share of the dense copies
  const int nch = sorted && chunks > 1 && chunks <= 4 && side && ev_chunk && n >= 2 * chunks ? chunks : 1;
  int n0[5], wo[4] = { 0, -1, -1, -1 };
  for (int c = 0; c <= nch; c++) n0[c] = (int)((long long)c * n / nch);
  // two ranges: the first one larger -- its general tiers hide under the second range's first tier either way, and the second
  // range's, which nothing hides, shrink with it (MJH_TRELLIS_FRONT: per mille of the images in the first range)
  if (nch == 2) {
    const char *fv = getenv("MJH_TRELLIS_FRONT");
    const int front = fv ? atoi(fv) : 625;      // (metric: interval 1.59 -> 1.55 ms against an even split, step -0.8 %; 750 the same, profiles/r06f_chunks.md)
    n0[1] = (int)((long long)n * front / 1000);
    if (n0[1] < 1) n0[1] = 1;
    if (n0[1] > n - 1) n0[1] = n - 1;
  }
  for (int c = 0; c < nch; c++) wo[c] = 4 * c + 3 * n0[c] * C.total_real_blocks;
  hipLaunchKernelGGL(k_zero_counters, dim3(1), dim3(64), 0, s, worklist, worklist2, make_int4(wo[0], wo[1], wo[2], wo[3]));   // (a 16-byte hipMemsetAsync costs ~80 us of stream time)
  int w0[5] = { 0, 0, 0, 0, 0 };
  for (int i = 0; i < 4; i++) w0[i + 1] = w0[i] + (i < C.ncomp ? (C.c[i].nblk + 63) / 64 : 0);
  dim3 gridq(w0[C.ncomp], n);
  for (int i = C.ncomp; i < 4; i++) w0[i] = 0x7FFFFFFF;   // components that do not exist never match
  const int4 wv = make_int4(w0[0], w0[1], w0[2], w0[3]);
  // the general tiers behind a first tier: blocks with more than its capacity (then 32) queue records, from their dense copies
#define LQ(QN, EXTV, CMP) hipLaunchKernelGGL((k_trellis_ac_q<QN, EXTV, CMP>), gridq, dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, wv, lambda, worklist, (int16_t *)dense, dense_cap, ext)
#define LDX(QN, EXTV, CMP, GRID, WL, WLN, ST, DN, DCAP) hipLaunchKernelGGL((k_trellis_ac_qd<QN, EXTV, CMP>), dim3(GRID), dim3(64), 0, ST, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, lambda, (const unsigned *)WL, WLN, (const int16_t *)DN, DCAP, ext)
#define LD(QN, EXTV, CMP, GRID, WL, WLN) LDX(QN, EXTV, CMP, GRID, WL, WLN, s, dense, dense_cap)
  if (extended) {   // the rarely used options take one fixed tiering (16 -> 32 -> 63)
    LQ(16, true, false);
    LD(32, true, false, 2048, worklist, worklist2);
    LD(63, true, false, 1024, worklist2, (unsigned *)nullptr);
  } else if (sorted) {
    // MJH_TRELLIS_VARIANT: queue capacity of the first tier: 0 = 16, 1 / 2 = 24, 3 = 32, 4 = 48 (all bit-identical).
    // The tile-sorted kernel: first tier of the plain compact pass; its work list (more records than its capacity, or a
    // magnitude >= 16) goes through the general tiers below.
    // One or two frames (the caller asks for one pass per tile then): occupancy is no concern on an empty chip, and with 24 records
    // next to nothing is left for the general tiers, whose fixed latency (~80 us) would sit on the critical path
    const bool small24 = v3_passes == 1 && variant <= 2 && fastdiv;
    if (small24) variant = 2;
    const int np = small24 ? 1 : (!fastdiv || variant > 0) ? 4 : v3_passes >= 8 ? 8 : v3_passes >= 4 ? 4 : v3_passes >= 2 ? 2 : 1;
    int t0[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) t0[i + 1] = t0[i] + (i < C.ncomp ? (C.c[i].nblk + 64 * np - 1) / (64 * np) : 0);
    const int ntiles = t0[C.ncomp];
    for (int i = C.ncomp; i < 4; i++) t0[i] = 0x7FFFFFFF;
    const int4 tv = make_int4(t0[0], t0[1], t0[2], t0[3]);
    // IMAGE RANGES (round 6).  The general tiers are bound by latency (7 waves per CU at 32 records: 0.25 ms for 0.04 ms of
    // instructions per 64 4K frames) and used to start when the whole first tier had finished.  With `chunks` > 1 the first tier
    // runs as that many launches over consecutive image ranges, each with its own work list; the general tiers of range c go to
    // the side stream and run next to the first tier of range c + 1 (disjoint blocks, disjoint lists); only the last range's
    // general tiers stay behind the first tier on `s`, which then joins the side stream (ev_chunk[chunks - 1]).
    const unsigned capc = dense_cap / (unsigned)nch;
    const int qd_grid = 2048;     // (1024 ... 8192 workgroups: no difference beyond noise, gpurun_out/r5j)
    for (int c = 0; c < nch; c++) {
      const dim3 gridt(ntiles, n0[c + 1] - n0[c]);
      unsigned *wl = worklist + wo[c], *wl2 = worklist2 + wo[c];
      int16_t *dn = (int16_t *)dense + (size_t)c * capc * 64;
      const int img0 = n0[c];
#define LV3(QN, NP, FDV) hipLaunchKernelGGL((k_trellis_ac_v3<QN, NP, FDV>), gridt, dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, (const MjhHuffTable *)tabs, spi, sl, tv, lambda, nq8, wl, dn, capc, nzmask, img0, worklist, count_mask)
      if (small24) LV3(24, 1, true);
      else if (variant >= 4) { if (fastdiv) LV3(48, 4, true); else LV3(48, 4, false); }      // q90 and up: 32 / 48 records (21 / 31 KB of LDS per wave)
      else if (variant == 3) { if (fastdiv) LV3(32, 4, true); else LV3(32, 4, false); }
      else if (variant > 0) { if (fastdiv) LV3(24, 4, true); else LV3(24, 4, false); }       // more records per block (higher qualities)
      else if (!fastdiv) LV3(16, 4, false);
      else switch (np) { case 8: LV3(16, 8, true); break; case 4: LV3(16, 4, true); break; case 2: LV3(16, 2, true); break; default: LV3(16, 1, true); break; }
#undef LV3
      const bool last = c == nch - 1;
      hipStream_t st = last ? s : side;
      if (last) {
        if (after_first_tier) (void)hipEventRecord(after_first_tier, s);      // (what only waits for the big kernel starts here, next to the general tiers)
        if (after_first_tier2) (void)hipEventRecord(after_first_tier2, s);    // (the other buffer set of an encoder with two batches in flight)
      } else {
        (void)hipEventRecord(ev_chunk[c], s);
        (void)hipStreamWaitEvent(side, ev_chunk[c], 0);
      }
      if (variant >= 3) LDX(63, false, true, qd_grid, wl, (unsigned *)nullptr, st, dn, capc);   // what is left has more than 32 records or a magnitude >= 16: one general tier that takes everything
      else { LDX(32, false, true, qd_grid, wl, wl2, st, dn, capc); LDX(63, false, true, 1024, wl2, (unsigned *)nullptr, st, dn, capc); }
    }
    if (nch > 1) {
      (void)hipEventRecord(ev_chunk[nch - 1], side);
      (void)hipStreamWaitEvent(s, ev_chunk[nch - 1], 0);
    }
  } else if (nzmask) {   // compact records out of the general first tier (the caller guarantees: plain pass)
    if (variant > 3) variant = 3;   // (the general first tier stops at 32 records)
    switch (variant) { case 1: LQ(20, false, true); break; case 2: LQ(24, false, true); break; case 3: LQ(32, false, true); break; default: LQ(16, false, true); break; }
    if (variant == 3) LD(63, false, true, 2048, worklist, (unsigned *)nullptr);
    else { LD(32, false, true, 2048, worklist, worklist2); LD(63, false, true, 1024, worklist2, (unsigned *)nullptr); }
  } else {               // one plane per position (trellis loops, progressive scans with restart intervals ...)
    if (variant > 3) variant = 3;
    switch (variant) { case 1: LQ(20, false, false); break; case 2: LQ(24, false, false); break; case 3: LQ(32, false, false); break; default: LQ(16, false, false); break; }
    if (variant == 3) LD(63, false, false, 2048, worklist, (unsigned *)nullptr);
    else { LD(32, false, false, 2048, worklist, worklist2); LD(63, false, false, 1024, worklist2, (unsigned *)nullptr); }
  }
#undef LQ
#undef LD
#undef LDX
}

void mjh_launch_trellis_eob_chain(const MjhConst &C, void *q, const MjhHuffTable *tabs, int spi, const int ac_slot[4], const void *eob_cost, const int *eob_has,
                                  int Ss, int Se, int n, hipStream_t s)
{
  int r0[5] = { 0, 0, 0, 0, 0 }, maxw = 0;
  for (int i = 0; i < 4; i++) { r0[i + 1] = r0[i] + (i < C.ncomp ? C.c[i].hib : 0); if (i < C.ncomp && C.c[i].wib > maxw) maxw = C.c[i].wib; }
  const int rows = r0[C.ncomp];
  for (int i = C.ncomp; i < 4; i++) r0[i] = 0x7FFFFFFF;
  const size_t lds = (size_t)(maxw + 1) * 8 + (size_t)maxw * 4 + (size_t)maxw * 2 + (size_t)(maxw + 1) + (size_t)maxw + 16;
  static bool raised = false;
  if (lds > 48 * 1024 && !raised) {   // very wide images only: allow the dynamic allocation beyond the default limit
    raised = hipFuncSetAttribute(reinterpret_cast<const void *>(k_trellis_eob_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
  }
  hipLaunchKernelGGL(k_trellis_eob_chain, dim3(rows, n), dim3(64), lds, s, C, (int16_t *)q, tabs, spi, make_int4(ac_slot[0], ac_slot[1], ac_slot[2], ac_slot[3]),
                     make_int4(r0[0], r0[1], r0[2], r0[3]), (const float2 *)eob_cost, eob_has, Ss, Se);
}

void mjh_launch_qopt_accumulate(const MjhConst &C, const void *uq, const void *q, void *sums, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_qopt_accumulate, dim3(63, C.ncomp, n), dim3(256), 0, s, C, (const int16_t *)uq, (const int16_t *)q, (long long *)sums);
}

void mjh_launch_qopt_update(void *sums, MjhQuant *Q, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_qopt_update, dim3(n, 4), dim3(64), 0, s, (long long *)sums, Q);
}

void mjh_launch_qopt_fix(const MjhQuant *Q, void *out, size_t out_stride, unsigned *sizes, int dqt_start, int sof_off, const int *tabs, int ntab,
                         int multi, int baseline_capable, int n, hipStream_t s)
{
  MjhDqtLayout L;
  L.dqt_start = dqt_start; L.sof_off = sof_off; L.ntab = ntab; L.multi = multi; L.baseline_capable = baseline_capable;
  for (int i = 0; i < 4; i++) L.tab[i] = i < ntab ? tabs[i] : 0;
  hipLaunchKernelGGL(k_qopt_fix, dim3(n), dim3(256), 0, s, Q, (uint8_t *)out, out_stride, sizes, L);
}
