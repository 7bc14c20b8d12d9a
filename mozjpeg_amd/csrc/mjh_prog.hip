// mjh_prog.hip -- progressive-JPEG entropy coding on gfx950.
//
// Reference behaviour: jcphuff.c (encode_mcu_DC_first :468, encode_mcu_AC_first :648,
// encode_mcu_DC_refine :746, encode_mcu_AC_refine :918, emit_eobrun :409, flush_bits :362),
// per-scan optimal tables (finish_pass_gather_phuff :1055), scan buffers with their own headers
// (jcmaster.c:671-684) and the scan search (select_scans jcmaster.c:773-962).
//
// Parallelisation.  Progressive AC coding carries state across blocks (EOBRUN, and for refinement
// scans up to ~1000 buffered correction bits with a data-dependent forced-flush rule), so a scan
// is a sequence in general.  What IS independent: scans (up to 64 candidates in the scan search),
// images, and -- inside a scan -- the per-block symbol work.  Hence, for DC scans, refinement scans and
// scans with restart intervals (k_prog_scan):
//   one WORKGROUP (16 waves) per (scan, image); wave w takes the 64-block steps w, w+16, ...:
//     phase A  every lane analyses its block in parallel (63 coalesced plane loads in one burst):
//              own symbol bits, "non-empty" / "contributes to EOBRUN" flags; refinement scans reduce a
//              block to four 64-bit masks (new / already-nonzero / correction bit / sign);
//     phase B  the state {EOBRUN, BE, bit offset} passes from step to step under a token (LDS word).  Inside a
//              run EOBRUN and BE only grow, so if the step cannot reach a forced flush (EOBRUN < 0x7FFF
//              and BE <= 937 at its end) every flush is the natural one in front of a non-empty lane and
//              all offsets follow in closed form from ballots and wave prefix sums.  B1 (before the token,
//              concurrent) does all of that except the first non-empty lane's flush, the only quantity that
//              depends on the carried-in state; B2 (under the token) is a handful of scalar operations plus,
//              in refinement scans, the accesses to the shared LDS buffer of pending correction bits; B3
//              (token already passed on) places the flush symbols and correction bits.
//              Otherwise the reference's state machine is run lane by lane (ordered path), with the
//              pending correction bits in the LDS bit buffer that is copied out at each flush;
//     phase C  lanes write their own symbols (and their trailing correction bits) at their offsets
//              (atomicOr into the zeroed pool), overlapping the next steps' phase A/B of other waves.
// The same kernel in statistics mode feeds the on-device Huffman table builder.  All candidate
// scans of a search phase run concurrently; the host never sees a symbol.
// AC-FIRST scans without restart intervals need none of this: a block's bits depend on the blocks before it
// only through the LENGTH of the EOB run in front of it, so their statistics and their encoding run in
// parallel over the whole component (k_prog_stats_acfirst/_resolve, k_pe_len/_resolve/_write/_finish below).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mjh_internal.h"
#include "mjh_device.h"
#include "mjh_launch.h"
#include <stdlib.h>

__device__ __forceinline__ unsigned wave_excl_scan(unsigned v, int lane, unsigned *total)
{
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  *total = __shfl(inc, 63, 64);
  return inc - v;
}

__device__ __forceinline__ void put_long(BitWriter &bw, unsigned v, int n)   // n <= 32
{
  if (n > 16) { bw.put(v >> 16, n - 16); bw.put(v & 0xFFFFu, 16); }
  else if (n > 0) bw.put(v, n);
}

// EOBRUN symbol: (nbits-1) << 4, followed by nbits-1 extra bits (emit_eobrun jcphuff.c:409-431)
__device__ __forceinline__ int eobrun_symbol(unsigned eobrun, int *nextra)
{
  const int nb = bitlen(eobrun) - 1;
  *nextra = nb;
  return nb << 4;
}

// waves per (scan, image) workgroup: the walk is latency-bound (one workgroup per CU, dependent LDS look-ups
// and 63 plane loads per step), so as many waves as a workgroup can have: 16 = 4 per SIMD = 128 VGPRs each
// (statistics 96; encode exactly 128 with 3 spilled registers -- still 10 % faster than 12 waves of 154)
// scan search: a candidate of luma level k+1 is coded only for the images whose search still improved at level k
__device__ __forceinline__ bool prog_skip(const MjhProgScan &sc, const MjhProgCtl *ct) { return sc.cond > 0 && ct->al_continue < sc.cond; }

#define PROG_WAVES(ENCODE) 16
template <int ENCODE>
__global__ void __launch_bounds__(64 * PROG_WAVES(ENCODE))
k_prog_scan(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list,
            MjhProgCtl *__restrict__ ctl, const int16_t *__restrict__ coef_q, MjhHuffTable *__restrict__ tabs,
            int slots_per_image, unsigned *__restrict__ pool, size_t pool_words_per_image,
            unsigned *__restrict__ mpos_pool, int mpos_per_image, const MjhProgPair *__restrict__ run_if)
{
  // fallback of the parallel encode: only the pairs it flagged are walked sequentially
  if (run_if && !run_if[(size_t)blockIdx.x * gridDim.y + blockIdx.y].fallback) return;
  __shared__ unsigned hist[2][256];
  __shared__ unsigned s_tab[2][256];   // size << 16 | code
  __shared__ unsigned pend[36];        // pending correction bits, MSB first
  __shared__ unsigned st_turn, st_eobrun, st_be, st_cur;   // the token: which 64-block step may run phase B, and its state
  // image index fastest: the workgroups of the long scans (the luma scans lead every script) of ALL images are
  // dispatched first and the short chroma scans fill in behind them
  const int img = blockIdx.x;
  const int sidx = scan_list[blockIdx.y];
  const MjhProgScan &sc = scans[sidx];      // (a reference: a local copy whose arrays are indexed dynamically lives in scratch)
  MjhProgCtl *ct = ctl + img;
  if (prog_skip(sc, ct)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Al = sc.al_sel == 1 ? ct->best_Al_luma : (sc.al_sel == 2 ? ct->best_Al_chroma : sc.Al);
  const int16_t *qimg = coef_q + (size_t)img * C.coefs_per_image;
  const bool has0 = sc.slot[0] >= 0, has1 = sc.slot[1] >= 0;   // DC scans: table number 0 / 1; AC scans: slot[0]
  MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + (has0 ? sc.slot[0] : 0);
  MjhHuffTable *T1 = tabs + (size_t)img * slots_per_image + (has1 ? sc.slot[1] : 0);
  unsigned *stream = pool + (size_t)img * pool_words_per_image;
  unsigned *mp = mpos_pool + (size_t)img * mpos_per_image + sc.mpos_off;   // byte positions of this scan's RSTn markers
  const int ri = sc.ri;
  if (ENCODE && ct->error) return;
  const unsigned long long t_start = wall_clock64();   // 100 MHz constant clock

  for (int i = threadIdx.x; i < 256; i += 64 * PROG_WAVES(ENCODE)) {
    hist[0][i] = 0; hist[1][i] = 0;
    if (ENCODE) {
      s_tab[0][i] = has0 ? ((unsigned)T0->ehufsi[i] << 16) | T0->ehufco[i] : 0u;
      s_tab[1][i] = has1 ? ((unsigned)T1->ehufsi[i] << 16) | T1->ehufco[i] : 0u;
    }
  }
  if (threadIdx.x < 36) pend[threadIdx.x] = 0;
  if (threadIdx.x == 0) { st_turn = 0; st_eobrun = 0; st_be = 0; st_cur = ENCODE ? ct->scan_words_off[sidx] * 32u : 0u; }
  __syncthreads();

  unsigned cur = ENCODE ? ct->scan_words_off[sidx] * 32u : 0u;   // running bit offset in the pool
  const unsigned start_bits = cur;
  unsigned corr_total = 0;   // statistics mode: total correction bits of a refinement scan

  if (sc.Ss == 0) {
    // ------------------------------------------------------------------ DC scans (first / refine)
    // No state crosses blocks except the bit offset (the predictor is simply the previous block's value), so
    // the 64-unit steps are spread over all waves; in encode mode the offset passes along under the token.
    const bool inter = sc.ncomp > 1;
    const MjhComp c0 = C.c[sc.comp[0]];
    const int nunits = inter ? C.mcus_per_row * C.mcu_rows : c0.nblk;
    const int nsteps = (nunits + 63) >> 6;
    for (int step = wave; step < nsteps; step += PROG_WAVES(ENCODE)) {
      const int u = step * 64 + lane;
      const bool valid = u < nunits;
      unsigned mybits = 0;
      // emit_restart jcphuff.c:438-465 in front of unit u: the DC predictions restart at 0, the bit stream is padded
      // to a byte and RSTn follows
      const bool rst_here = ri && u > 0 && (u % ri) == 0;
      const int last_u = min(step * 64 + 63, nunits - 1), mrst = ri ? (last_u / ri) * ri : 0;
      const bool step_has_rst = ri && mrst > 0 && mrst >= step * 64;
      // pass 0: bits (or statistics); pass 1 (encode only): write
      for (int pass = 0; pass < (ENCODE ? 2 : 1); pass++) {
        BitWriter bw;
        if (pass == 1) {
          unsigned tot;
          unsigned off = wave_excl_scan(mybits, lane, &tot);
          while (__hip_atomic_load(&st_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)step) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          cur = st_cur;
          MJH_WAVE_SYNC();          // (lane 0 overwrites st_cur below)
          unsigned cur_out = cur + tot;
          if (step_has_rst) {       // offsets depend on the byte alignment at every marker: walk the units in order
            unsigned c2 = cur;
            off = 0;
            for (int j = 0; j <= last_u - step * 64; j++) {
              const int uj = step * 64 + j;
              if (uj > 0 && (uj % ri) == 0) {
                const unsigned pad = (8u - (c2 & 7u)) & 7u;
                const int idx = uj / ri - 1;
                if (lane == 0) {
                  BitWriter mw;
                  mw.init(stream, c2);
                  if (pad) mw.put((1u << pad) - 1u, (int)pad);
                  mw.put(0xFFD0u + (unsigned)(idx & 7), 16);
                  mw.flush();
                  mp[idx] = (c2 + pad - start_bits) >> 3;
                }
                c2 += pad + 16u;
              }
              if (lane == j) off = c2 - cur;
              c2 += (unsigned)__builtin_amdgcn_readlane((int)mybits, j);
            }
            cur_out = c2;
          }
          if (lane == 0) st_cur = cur_out;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(&st_turn, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          bw.init(stream, cur + off);
        }
        if (valid) {
          for (int ci = 0; ci < sc.ncomp; ci++) {
            const MjhComp cc = C.c[scans[sidx].comp[ci]];
            const int16_t *q0 = qimg + cc.coef_off;
            const int tb = (cc.dctbl >> 8) & 1;      // the DC table's class (MjhComp)
            const int mh = inter ? cc.v : 1, mw = inter ? cc.h : 1;
            for (int yi = 0; yi < mh; yi++)
              for (int xi = 0; xi < mw; xi++) {
                int dc, pred = 0;
                if (inter) {
                  const int my = u / C.mcus_per_row, mx = u - my * C.mcus_per_row;
                  const int r = my * cc.v + yi, c = mx * cc.h + xi;
                  dc = q0[dc_source_block(cc, r, c)];
                  int pr, pc;
                if (sc.Ah == 0 && mcu_prev_block(C, cc, r, c, pr, pc) && !(rst_here && yi == 0 && xi == 0))
                    pred = q0[dc_source_block(cc, pr, pc)] >> Al;
                } else {
                  dc = q0[u];
                  if (sc.Ah == 0 && u > 0 && !rst_here) pred = q0[u - 1] >> Al;
                }
                if (sc.Ah == 0) {                       // encode_mcu_DC_first
                  const int v = dc >> Al;               // arithmetic shift = point transform
                  const int df = v - pred;
                  const int a = df < 0 ? -df : df;
                  const int nb = bitlen((unsigned)a);
                  if (!ENCODE) atomicAdd(&hist[tb][nb], 1u);
                  else if (pass == 0) mybits += (s_tab[tb][nb] >> 16) + nb;
                  else {
                    const unsigned e = s_tab[tb][nb];
                    bw.put(e & 0xFFFF, (int)(e >> 16));
                    if (nb) bw.put((unsigned)(df < 0 ? df - 1 : df), nb);
                  }
                } else {                                // encode_mcu_DC_refine: the Al'th bit
                  if (ENCODE) { if (pass == 0) mybits += 1; else bw.put((unsigned)(dc >> Al) & 1u, 1); }
                }
              }
          }
        }
        if (pass == 1) bw.flush();
      }
    }
    __syncthreads();
    cur = st_cur;
  } else {
    // ------------------------------------------------------------------ AC scans (first / refine)
    const MjhComp cc = C.c[sc.comp[0]];
    const int16_t *qc = qimg + cc.coef_off;
    const int Ss = sc.Ss, Se = sc.Se;
    const bool refine = sc.Ah != 0;
    unsigned EOBRUN = 0, BE = 0;

    // flush the pending EOB run (+ buffered correction bits) at bit offset `cur`
    auto flush_run = [&]() {
      int nextra;
      const int sym = eobrun_symbol(EOBRUN, &nextra);
      if (!ENCODE) { if (lane == 0) atomicAdd(&hist[0][sym], 1u); }
      else {
        const unsigned e = s_tab[0][sym];
        const int len = (int)(e >> 16);
        if (lane == 0) {
          BitWriter bw;
          bw.init(stream, cur);
          bw.put(e & 0xFFFF, len);
          if (nextra) bw.put(EOBRUN & ((1u << nextra) - 1u), nextra);
          bw.flush();
        }
        cur += (unsigned)(len + nextra);
        if (BE) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // lane 0's LDS appends -> visible to the wave
          const int nw = (int)((BE + 31) >> 5);
          if (lane < nw) {
            const int nb = (int)min(32u, BE - 32u * lane);
            BitWriter bw;
            bw.init(stream, cur + 32u * lane);
            put_long(bw, pend[lane] >> (32 - nb), nb);
            bw.flush();
            pend[lane] = 0;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          cur += BE;
        }
      }
      EOBRUN = 0;
      BE = 0;
    };

    // emit_restart jcphuff.c:438-465 in front of block bj: pending EOB run out, pad to a byte, RSTn, run state cleared
    auto do_restart = [&](int idx) {
      if (EOBRUN > 0) flush_run();
      if (ENCODE) {
        const unsigned pad = (8u - (cur & 7u)) & 7u;
        if (lane == 0) {
          BitWriter mw;
          mw.init(stream, cur);
          if (pad) mw.put((1u << pad) - 1u, (int)pad);
          mw.put(0xFFD0u + (unsigned)(idx & 7), 16);
          mw.flush();
          mp[idx] = (cur + pad - start_bits) >> 3;
        }
        cur += pad + 16u;
      }
    };

    const int nsteps = (cc.nblk + 63) >> 6;
    for (int step = wave; step < nsteps; step += PROG_WAVES(ENCODE)) {
      const int base = step * 64;
      const int b = base + lane;
      const bool valid = b < cc.nblk;
      const int nvalid = min(64, cc.nblk - base);
      // ---- phase A: per-lane block analysis.  The band is fetched with a fully unrolled, statically
      // indexed loop: all (Se-Ss+1) coalesced plane loads are in flight together and the values stay
      // in registers for phase C (a dependent load per coefficient would cost one HBM latency each).
      bool ne = false, E = false;
      unsigned own_bits = 0;
      unsigned long long newm = 0, nzm = 0, corrm = 0, posm = 0, tailm = 0;
      int tail_cnt = 0;
      int x[64];
      {
        // unconditional, in-bounds loads (invalid lanes re-read the last block): straight-line code, so the
        // compiler issues all 63 before the first use instead of one waitcnt per predicated load
        const int16_t *qs = qc + (valid ? b : cc.nblk - 1);
#pragma unroll
        for (int k = 1; k < 64; k++) x[k] = (int)qs[(size_t)k * cc.kstride];
      }
      if (valid) {
        if (!refine) {
          int r = 0;
#pragma unroll
          for (int k = 1; k < 64; k++) {
            if (k >= Ss && k <= Se) {
              const int v = x[k];
              const int a = (v < 0 ? -v : v) >> Al;
              if (a == 0) r++;
              else {
                ne = true;
                const int nz16 = r >> 4;
                r &= 15;
                const int nb = bitlen((unsigned)a);
                const int sym = (r << 4) + nb;
                if (!ENCODE) { if (nz16) atomicAdd(&hist[0][0xF0], (unsigned)nz16); atomicAdd(&hist[0][sym], 1u); }
                else own_bits += (unsigned)nz16 * (s_tab[0][0xF0] >> 16) + (s_tab[0][sym] >> 16) + nb;
                r = 0;
              }
            }
          }
          E = r > 0;
        } else {
#pragma unroll
          for (int k = 1; k < 64; k++) {
            if (k >= Ss && k <= Se) {
              const int v = x[k];
              const int a = (v < 0 ? -v : v) >> Al;
              if (a == 1) { newm |= 1ull << k; if (v >= 0) posm |= 1ull << k; }
              else if (a > 1) { nzm |= 1ull << k; if (a & 1) corrm |= 1ull << k; }
            }
          }
          ne = newm != 0;
          const int EOBk = ne ? 63 - __builtin_clzll(newm) : -1;
          int r = 0, prev = Ss - 1, BR = 0, flushed_below = Ss;
          unsigned long long mm = newm | nzm;
          while (mm) {
            const int k = __builtin_ctzll(mm);
            mm &= mm - 1;
            r += k - prev - 1;
            prev = k;
            while (r > 15 && k <= EOBk) {
              if (!ENCODE) atomicAdd(&hist[0][0xF0], 1u);
              else own_bits += (s_tab[0][0xF0] >> 16) + BR;
              BR = 0; flushed_below = k; r -= 16;
            }
            if ((nzm >> k) & 1ull) { BR++; continue; }
            const int sym = (r << 4) + 1;
            if (!ENCODE) atomicAdd(&hist[0][sym], 1u);
            else own_bits += (s_tab[0][sym] >> 16) + 1 + BR;
            BR = 0; flushed_below = k + 1; r = 0;
          }
          r += Se - prev;
          tail_cnt = BR;
          E = (r > 0) || (BR > 0);
          tailm = flushed_below < 64 ? (nzm & ~((1ull << flushed_below) - 1ull)) : 0ull;
          if (!ENCODE) corr_total += (unsigned)__popcll(nzm);
        }
      }
      // trailing correction bits as an MSB-first string
      unsigned long long tail_bits = 0;
      {
        unsigned long long tm = tailm;
        while (tm) { const int k = __builtin_ctzll(tm); tm &= tm - 1; tail_bits = (tail_bits << 1) | ((corrm >> k) & 1ull); }
      }
      // ---- phase B1: everything that does not depend on the state carried in from earlier steps -- runs
      // concurrently in all waves.  Inside a run EOBRUN and BE only grow, so if the step cannot reach a forced
      // flush (checked under the token) every flush is the natural one in front of a non-empty lane and its
      // pending state follows from the position of the previous non-empty lane: closed form, all lanes in
      // parallel.  Only the FIRST non-empty lane's flush depends on the carried-in (EOBRUN, BE).
      const unsigned long long ne_mask = __ballot(ne), E_mask = __ballot(E);
      unsigned tsum = 0, texcl = 0;
      if (refine) texcl = wave_excl_scan((unsigned)tail_cnt, lane, &tsum);
      const unsigned long long below = ne_mask & ((1ull << lane) - 1ull);
      const int pl = below ? 63 - __builtin_clzll(below) : -1;          // previous non-empty lane
      const unsigned texcl_pl = refine ? (unsigned)__shfl((int)texcl, pl < 0 ? 0 : pl, 64) : 0u;
      const int f0 = ne_mask ? __builtin_ctzll(ne_mask) : 0;            // first non-empty lane
      const unsigned texcl_f0 = refine ? (unsigned)__shfl((int)texcl, f0, 64) : 0u;
      unsigned l_flush = 0, l_cnt = 0, l_be = 0;
      int l_sym = 0, l_extra = 0;
      if (ne && pl >= 0) {                                  // the run = pl's EOB + the empties between
        l_cnt = (unsigned)((E_mask >> pl) & 1ull) + (unsigned)(lane - pl - 1);
        l_be = texcl - texcl_pl;
        if (l_cnt > 0) {
          l_sym = eobrun_symbol(l_cnt, &l_extra);
          if (ENCODE) l_flush = (s_tab[0][l_sym] >> 16) + (unsigned)l_extra + l_be;
        }
      }
      unsigned l_tot;
      const unsigned l_ex = wave_excl_scan(ne ? l_flush + own_bits : 0u, lane, &l_tot);
      unsigned run_out = 0, be_out = 0;                     // run state behind this step's last non-empty lane
      if (ne_mask) {
        const int last = 63 - __builtin_clzll(ne_mask);
        run_out = (unsigned)((E_mask >> last) & 1ull) + (unsigned)(nvalid - 1 - last);
        if (refine) be_out = tsum - (unsigned)__shfl((int)texcl, last, 64);
      }
      const unsigned long long above = lane < 63 ? (ne_mask & ~((2ull << lane) - 1ull)) : 0ull;
      const int rs = ne ? lane : pl;                        // first lane of my run (-1: carried in)
      const unsigned rs_excl = ne ? texcl : texcl_pl;
      // ---- phase B2 (in step order under the token): a few scalar operations in the common case
      while (__hip_atomic_load(&st_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)step) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      EOBRUN = st_eobrun; BE = st_be; cur = st_cur;
      MJH_WAVE_SYNC();              // (lane 0 overwrites the three below)
      unsigned out_off = 0;
      const int mrst = ri ? ((base + nvalid - 1) / ri) * ri : 0;          // a restart boundary inside this step?
      const bool step_has_rst = ri && mrst > 0 && mrst >= base;
      const bool fast = EOBRUN + 64u < 0x7FFFu && (!refine || BE + tsum <= 937u) && !step_has_rst;
      const unsigned cur_in = cur, BE_in = BE;
      unsigned fb0 = 0, be0 = 0, cnt0 = 0;
      int sym0 = 0, extra0 = 0;
      if (fast) {
        if (ne_mask) {                                      // run carried in from earlier steps + leading empties
          cnt0 = EOBRUN + (unsigned)f0;
          be0 = BE + texcl_f0;
          if (cnt0 > 0) {
            sym0 = eobrun_symbol(cnt0, &extra0);
            if (ENCODE) fb0 = (s_tab[0][sym0] >> 16) + (unsigned)extra0 + be0;
          }
        }
        if (ENCODE && refine) {
          // the LDS buffer of pending correction bits is shared by the steps: touched only under the token
          // (1) correction bits carried in from earlier steps: flushed behind the first non-empty lane's EOBRUN symbol
          if (ne_mask && BE_in) {
            const unsigned ca0 = cur_in + fb0 - be0;
            const int nw = (int)((BE_in + 31) >> 5);
            if (lane < nw) {
              const int nb = (int)min(32u, BE_in - 32u * lane);
              BitWriter bw;
              bw.init(stream, ca0 + 32u * lane);
              put_long(bw, pend[lane] >> (32 - nb), nb);
              bw.flush();
              pend[lane] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          }
          // (2) trailing correction bits of a run that is still pending at the end of this step
          if (tail_cnt > 0 && !above) {
            unsigned pos = rs >= 0 ? texcl - rs_excl : BE_in + texcl;
            int rem = tail_cnt;
            while (rem > 0) {
              const int w = (int)(pos >> 5), o = (int)(pos & 31);
              const int take = min(32 - o, rem);
              const unsigned chunk = (unsigned)((tail_bits >> (rem - take)) & ((1ull << take) - 1ull));
              atomicOr(&pend[w], chunk << (32 - o - take));
              pos += take; rem -= take;
            }
          }
        }
        cur = cur_in + l_tot + fb0;
        if (ne_mask) { EOBRUN = run_out; BE = be_out; }
        else { EOBRUN += (unsigned)nvalid; BE += tsum; }
      } else {
      // ordered path: the reference's state machine over the 64 lane summaries (forced flushes at
      // EOBRUN == 0x7FFF and BE > 937, pending correction bits buffered in LDS)
#pragma nounroll
      for (int j = 0; j < nvalid; j++) {
        const bool ne_j = (ne_mask >> j) & 1ull, E_j = (E_mask >> j) & 1ull;
        if (step_has_rst && base + j > 0 && ((base + j) % ri) == 0) do_restart((base + j) / ri - 1);
        if (!ne_j && !E_j) continue;
        const unsigned own_j = (unsigned)__builtin_amdgcn_readlane((int)own_bits, j);
        if (ne_j) {
          if (EOBRUN > 0) flush_run();
          if (lane == j) out_off = cur;
          cur += own_j;
        }
        if (E_j) {
          if (refine) {
            const int tc = __builtin_amdgcn_readlane(tail_cnt, j);
            if (ENCODE && tc > 0) {
              const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)tail_bits, j);
              const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(tail_bits >> 32), j);
              const unsigned long long tb = ((unsigned long long)hi << 32) | lo;
              if (lane == 0) {
                unsigned pos = BE;
                int rem = tc;
                while (rem > 0) {
                  const int w = (int)(pos >> 5), o = (int)(pos & 31);
                  const int take = min(32 - o, rem);
                  const unsigned chunk = (unsigned)((tb >> (rem - take)) & ((1ull << take) - 1ull));
                  pend[w] |= chunk << (32 - o - take);
                  pos += take; rem -= take;
                }
              }
            }
            BE += (unsigned)tc;
          }
          EOBRUN++;
          if (EOBRUN == 0x7FFF || BE > 1000u - 64u + 1u) flush_run();   // jcphuff.c:719,:998-1000
        }
      }
      }
      if (lane == 0) { st_eobrun = EOBRUN; st_be = BE; st_cur = cur; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&st_turn, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // ---- phase B3 (token passed on): offsets and the writes that belong to the flushes
      if (fast) {
        const bool first = ne_mask && lane == f0;
        if (!ENCODE) { if (ne && (first ? cnt0 : l_cnt) > 0) atomicAdd(&hist[0][first ? sym0 : l_sym], 1u); }
        else {
          const unsigned my_flush = first ? fb0 : l_flush, my_be = first ? be0 : l_be, my_cnt = first ? cnt0 : l_cnt;
          const int my_sym = first ? sym0 : l_sym, my_extra = first ? extra0 : l_extra;
          const unsigned pos = cur_in + ((ne_mask && lane > f0) ? fb0 : 0u) + l_ex;   // start of this lane's flush + symbols
          const unsigned ca = pos + (my_flush - my_be);     // correction area of a flushing lane = right behind its EOBRUN symbol
          if (ne) {
            if (my_cnt > 0) {      // the EOBRUN symbol of the pending run goes in front of the lane's own symbols
              const unsigned e = s_tab[0][my_sym];
              BitWriter bw;
              bw.init(stream, pos);
              bw.put(e & 0xFFFF, (int)(e >> 16));
              if (my_extra) bw.put(my_cnt & ((1u << my_extra) - 1u), my_extra);
              bw.flush();
            }
            out_off = pos + my_flush;
          }
          if (refine) {
            // every lane's trailing correction bits go to the correction area of the next non-empty lane
            const int fl = above ? __builtin_ctzll(above) : 0;
            const unsigned ca_f = (unsigned)__shfl((int)ca, fl, 64);
            const unsigned offs = rs >= 0 ? texcl - rs_excl : BE_in + texcl;
            if (tail_cnt > 0 && above) {
              BitWriter bw;
              bw.init(stream, ca_f + offs);
              put_long(bw, (unsigned)(tail_bits >> 32), tail_cnt > 32 ? tail_cnt - 32 : 0);
              put_long(bw, (unsigned)tail_bits, tail_cnt > 32 ? 32 : tail_cnt);
              bw.flush();
            }
          }
        }
      }
      // ---- phase C: lanes write their own symbols
      if (ENCODE && ne) {
        BitWriter bw;
        bw.init(stream, out_off);
        if (!refine) {
          int r = 0;
#pragma unroll
          for (int k = 1; k < 64; k++) {
            if (k >= Ss && k <= Se) {
              const int v = x[k];
              const int a = (v < 0 ? -v : v) >> Al;
              if (a == 0) r++;
              else {
                while (r > 15) { const unsigned e = s_tab[0][0xF0]; bw.put(e & 0xFFFF, (int)(e >> 16)); r -= 16; }
                const int nb = bitlen((unsigned)a);
                const unsigned e = s_tab[0][(r << 4) + nb];
                bw.put(e & 0xFFFF, (int)(e >> 16));
                bw.put((unsigned)(v < 0 ? ~a : a), nb);
                r = 0;
              }
            }
          }
        } else {
          const int EOBk = 63 - __builtin_clzll(newm);
          int r = 0, prev = Ss - 1, fb = Ss;
          unsigned long long mm = newm | nzm;
          auto put_corr = [&](int lo, int hi) {   // correction bits of already-nonzero positions in [lo, hi)
            unsigned long long m = nzm & ~((1ull << lo) - 1ull);
            if (hi < 64) m &= (1ull << hi) - 1ull;
            while (m) { const int k = __builtin_ctzll(m); m &= m - 1; bw.put((unsigned)((corrm >> k) & 1ull), 1); }
          };
          while (mm) {
            const int k = __builtin_ctzll(mm);
            mm &= mm - 1;
            r += k - prev - 1;
            prev = k;
            while (r > 15 && k <= EOBk) {
              const unsigned e = s_tab[0][0xF0];
              bw.put(e & 0xFFFF, (int)(e >> 16));
              put_corr(fb, k);
              fb = k; r -= 16;
            }
            if ((nzm >> k) & 1ull) continue;
            const unsigned e = s_tab[0][(r << 4) + 1];
            bw.put(e & 0xFFFF, (int)(e >> 16));
            bw.put((unsigned)((posm >> k) & 1ull), 1);
            put_corr(fb, k);
            fb = k + 1; r = 0;
          }
        }
        bw.flush();
      }
    }
    __syncthreads();
    if (wave == 0) {
      EOBRUN = st_eobrun; BE = st_be; cur = st_cur;
      if (EOBRUN > 0) flush_run();   // finish_pass_phuff / finish_pass_gather_phuff
    }
  }

  __syncthreads();
  if (!ENCODE) {
    // statistics -> table slots (+ the trellis-pass seeding of jcphuff.c:257-264)
    for (int i = threadIdx.x; i < 256; i += 64 * PROG_WAVES(ENCODE)) {
      unsigned s0 = hist[0][i];
      if (sc.seed && (i & 15) < 12) s0 += 1;
      if (has0) T0->counts[i] = s0;
      if (has1) T1->counts[i] = hist[1][i];
    }
    // total correction bits of this scan (needed to size its bit stream): wave sum -> counts[258]
    unsigned t = corr_total;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x == 0) st_be = 0;               // st_be is free again: sum of the waves' correction-bit counts
    __syncthreads();
    if (lane == 0 && t) atomicAdd(&st_be, t);
    __syncthreads();
    if (threadIdx.x == 0 && has0) { T0->counts[256] = 0; T0->counts[257] = 0; T0->counts[258] = st_be; }
  } else {
    // flush_bits jcphuff.c:362-367: pad the last byte with 1-bits
    const unsigned tb = cur - start_bits;
    if (threadIdx.x == 0) {
      if (tb & 7u) {
        const unsigned padbits = 8u - (tb & 7u), bitpos = cur & 31u;
        atomicOr(&stream[cur >> 5], __builtin_bswap32(((1u << padbits) - 1u) << (32u - bitpos - padbits)));
      }
      // the size predicted from the statistics must be exact; with restart markers the pads (0..7 bits each) are
      // only bounded, so the prediction is an upper bound there
      if (sc.nrst ? tb > ct->scan_bits[sidx] : ct->scan_bits[sidx] != tb) ct->error = 2;
      ct->scan_bits[sidx] = tb;
    }
  }
  if (threadIdx.x == 0) ct->scan_us[ENCODE][sidx] = (unsigned)((wall_clock64() - t_start) / 100ull);
}

// =============================================================================================
// Scans WITHOUT restart intervals, in parallel over the whole component ("parallel chain").
//
// A progressive scan looks like a sequence (EOB runs across blocks, buffered correction bits), but what one block
// contributes depends on the blocks before it only through two numbers:
//   AC first   : the LENGTH of the EOB run in front of a non-empty block (encode_mcu_AC_first jcphuff.c:648-745,
//                emit_eobrun :409): run(j) = E(p) + (j - p - 1), p = previous non-empty block, E = "ends in zeros";
//   AC refine  : that run length AND the number of correction bits buffered along the run (encode_mcu_AC_refine
//                :918-1000): with T = exclusive prefix sum of the per-block trailing-correction-bit counts,
//                be(j) = T[j] - T[p].  Block j writes [EOBRUN symbol][be(j) correction bits][own symbols]; a block i of the
//                run writes its trailing bits at off(j) + size(symbol) + T[i] - T[p]   (tools/prototype_refine_parallel.py
//                checks this against the sequential state machine, forced flushes included);
//   DC first / refine : nothing at all (the predictor is the previous block's value, no coder state).
// The forced flushes of the reference -- after a block that brings the pending run to 0x7FFF blocks (jcphuff.c:719) or
// the buffered correction bits beyond 937 (:998) -- fit in as well: a forced flush after block i writes exactly what a
// flush in front of a non-empty block i+1 would write, and leaves the coder in the state an empty block i+1 expects.  So
// block i+1 is simply MARKED as a flush point (a non-empty block without symbols of its own): k_pp_cuts finds the marks of
// every gap between two non-empty blocks by the greedy rule of the reference (binary search in T, one thread per gap).
//
// Work units are 2048-block chunks (DC scans: 2048 units = MCUs of an interleaved scan, blocks otherwise).
//   statistics:  k_pp_stats   per block: symbols -> LDS histogram, non-empty / ends-in-zeros bitmaps, trailing-bit counts
//                (prefix sum of the trailing-bit counts: the sequential coder's scan kernels)
//                k_pp_carry   per pair: last non-empty block in front of every chunk; forced flushes of the final gap
//                k_pp_cuts    per gap: forced-flush marks
//                k_pp_runs    per flush point: run(j), be(j), EOBRUN symbol statistics
//                k_pp_resolve per pair: runs that cross chunk borders, the run pending at the end
//   encode:      k_pp_len     per block: bits = [EOBRUN symbol + be] + own symbols
//                (prefix sum of the lengths)
//                k_pp_write   per block: the bits at their offsets
//                k_pp_finish  per pair: pending run, pad, size check
// The statistics kernels leave bitmaps, runs, trailing counts and T behind: the encode kernels of the same phase reuse them.
// =============================================================================================
enum { PP_DC_FIRST = 0, PP_DC_REFINE = 1, PP_AC_FIRST = 2, PP_AC_REFINE = 3 };
__device__ __forceinline__ int pp_kind(const MjhProgScan &sc) { return sc.Ss == 0 ? (sc.Ah == 0 ? PP_DC_FIRST : PP_DC_REFINE) : (sc.Ah == 0 ? PP_AC_FIRST : PP_AC_REFINE); }

// units of a scan: MCUs of an interleaved DC scan, blocks of the component otherwise
__device__ __forceinline__ int pp_units(const MjhConst &C, const MjhProgScan &sc)
{
  return sc.Ss == 0 && sc.ncomp > 1 ? C.mcus_per_row * C.mcu_rows : C.c[sc.comp[0]].nblk;
}

// the band of one block, all loads in flight together (only the 8-coefficient groups that overlap the band)
#define PP_LOAD_BAND(x, qs, kstride, Ss, Se)                                                                         \
  _Pragma("unroll") for (int g_ = 0; g_ < 8; g_++) {                                                                 \
    if (8 * g_ + 7 >= (Ss) && 8 * g_ <= (Se)) {                                                                       \
      _Pragma("unroll") for (int kk_ = 0; kk_ < 8; kk_++) { const int k_ = 8 * g_ + kk_; if (k_ >= 1) x[k_] = (int)(qs)[(size_t)k_ * (kstride)]; } \
    } else {                                                                                                           \
      _Pragma("unroll") for (int kk_ = 0; kk_ < 8; kk_++) x[8 * g_ + kk_] = 0;                                        \
    }                                                                                                                  \
  }

// refinement-scan analysis of one block (encode_mcu_AC_refine :918-1000): masks of newly non-zero / already non-zero
// positions, their sign / correction bits; own = bits of the block's own symbols incl. the correction bits flushed inside;
// tail = correction bits still buffered behind its last symbol (positions tailm); SIZES: 0 = count symbols into hist
struct PPRefine { unsigned long long newm, nzm, corrm, posm, tailm; int tail_cnt; bool ne, E; unsigned own; };

// COMPACT coefficient records (written by the compact AC trellis, mjh_kernels.hip: nzmask = non-zero positions of the block,
// plane i+1 = its i-th non-zero value in position order): f(position, value) for the non-zero coefficients at positions
// Ss..Se in position order.  Every lane of a wave reads the same plane at a time (coalesced), bursts of 8, up to the
// rank of the last position <= Se the busiest block of the wave has; positions below Ss are read and dropped.
// SKIPLOW (the first-pass AC scans of the parallel chain, k_pp_stats<true, 1> and k_pp_emit; Ss >= 1): the non-zeros below Ss -- for the
// upper band of a frequency split most of a block's non-zeros, about a third of all visits of the scan search at q85 -- are taken
// out of the mask up front instead of being visited and dropped one by one, and bursts in which no lane of the wave has anything
// to visit are not even loaded; f sees the same sequence (C3 on the chip: 24.79 -> 23.82 ms per 32 frames, profiles/r05a_optin_kernels_ab.md).
template <bool SKIPLOW = false, class F>
__device__ __forceinline__ void pp_band_nonzeros(const int16_t *__restrict__ qb, size_t kstride, unsigned long long mask, int Ss, int Se, bool active, F &&f)
{
  if (Se < 63) mask &= (2ull << Se) - 1ull;
  const int n = active ? __popcll(mask) : 0;
  if (__builtin_amdgcn_ballot_w64(0 < n) == 0ull) return;
  int lo = 0, base0 = 0;      // SKIPLOW: the lane's non-zeros below Ss (values 0..lo-1 of its record); first burst any lane needs
  if (SKIPLOW) {
    const unsigned long long lowm = mask & ((1ull << Ss) - 1ull);     // (1 <= Ss <= 63)
    lo = active ? __popcll(lowm) : 0;
    mask ^= lowm;
    if (__builtin_amdgcn_ballot_w64(lo < n) == 0ull) return;          // nothing at or above Ss in the whole wave
    while (__builtin_amdgcn_ballot_w64(lo < n && lo < base0 + 8) == 0ull) base0 += 8;   // (ends: some lane has lo < n <= 63)
  }
  // double-buffered bursts: the loads of burst b+1 are in flight while burst b is consumed (these kernels are bound by
  // the latency of their dependent loads, not by bandwidth or issue)
  int v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = (base0 + j < n) ? (int)qb[(size_t)(base0 + j + 1) * kstride] : 0;
#pragma unroll 1
  for (int base = base0; base < 63; base += 8) {
    const bool more = __builtin_amdgcn_ballot_w64(base + 8 < n) != 0ull;
    int w[8];
    if (more) {
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = (base + 8 + j < n && base + 8 + j < 63) ? (int)qb[(size_t)(base + 8 + j + 1) * kstride] : 0;
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (SKIPLOW ? (unsigned)(base + j - lo) < (unsigned)(n - lo) : base + j < n) {
        const int pos = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        if (SKIPLOW || pos >= Ss) f(pos, v[j]);
      }
    if (!more) break;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = w[j];
  }
}

template <bool COUNT>
__device__ __forceinline__ PPRefine pp_refine_finish(PPRefine R, int Ss, int Se, const unsigned char *s_size, unsigned *hist);

// the four masks of a refinement block from a compact record, then the common part
template <bool COUNT>
__device__ __forceinline__ PPRefine pp_refine_block_compact(const int16_t *__restrict__ qb, size_t kstride, unsigned long long mask, bool active,
                                                            int Ss, int Se, int Al, const unsigned char *s_size, unsigned *hist)
{
  PPRefine R;
  R.newm = R.nzm = R.corrm = R.posm = R.tailm = 0; R.own = 0;
  pp_band_nonzeros(qb, kstride, mask, Ss, Se, active, [&](int k, int v) {
    const int a = (v < 0 ? -v : v) >> Al;
    // (selects, not branches: `if (a == 1) R.newm |= .. else if (a > 1) R.nzm |= ..` became ONE read-modify-write through a selected
    // ADDRESS -- the two masks as an array in scratch memory, a scratch load + store per visited coefficient, found in the third
    // session of round 6 by the kernels' .private_segment_fixed_size)
    const unsigned long long bit = 1ull << k;
    const bool one = a == 1, big = a > 1;
    R.newm |= one ? bit : 0ull;
    R.posm |= (one && v >= 0) ? bit : 0ull;
    R.nzm |= big ? bit : 0ull;
    R.corrm |= (big && (a & 1)) ? bit : 0ull;
  });
  return pp_refine_finish<COUNT>(R, Ss, Se, s_size, hist);
}

// ... or from the masks the statistics pass of the same scan kept (pe.rmask: three arrays of nblk_pad entries per pair)
template <bool COUNT>
__device__ __forceinline__ PPRefine pp_refine_block_cached(const unsigned long long *__restrict__ rm, size_t stride, bool active,
                                                           int Ss, int Se, const unsigned char *s_size, unsigned *hist)
{
  PPRefine R;
  const unsigned long long nw = active ? rm[0] : 0ull, nz = active ? rm[stride] : 0ull, sc = active ? rm[2 * stride] : 0ull;
  R.newm = nw; R.nzm = nz; R.posm = sc & nw; R.corrm = sc & nz; R.tailm = 0; R.own = 0;
  return pp_refine_finish<COUNT>(R, Ss, Se, s_size, hist);
}

template <bool COUNT>
__device__ __forceinline__ PPRefine pp_refine_block(const int (&x)[64], int Ss, int Se, int Al, const unsigned char *s_size, unsigned *hist)
{
  PPRefine R;
  R.newm = R.nzm = R.corrm = R.posm = R.tailm = 0; R.own = 0;
#pragma unroll
  for (int g = 0; g < 8; g++) if (8 * g + 7 >= Ss && 8 * g <= Se) {   // wave-uniform: only the groups of 8 that overlap the band
#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    const int k = 8 * g + kk;
    if (k >= 1 && k >= Ss && k <= Se) {
      const int v = x[k];
      const int a = (v < 0 ? -v : v) >> Al;
      if (a == 1) { R.newm |= 1ull << k; if (v >= 0) R.posm |= 1ull << k; }
      else if (a > 1) { R.nzm |= 1ull << k; if (a & 1) R.corrm |= 1ull << k; }
    }
  } }
  return pp_refine_finish<COUNT>(R, Ss, Se, s_size, hist);
}

template <bool COUNT>
__device__ __forceinline__ PPRefine pp_refine_finish(PPRefine R, int Ss, int Se, const unsigned char *s_size, unsigned *hist)
{
  R.ne = R.newm != 0;
  const int EOBk = R.ne ? 63 - __builtin_clzll(R.newm) : -1;
  int r = 0, prev = Ss - 1, BR = 0, flushed_below = Ss;
  unsigned long long mm = R.newm | R.nzm;
  while (mm) {
    const int k = __builtin_ctzll(mm);
    mm &= mm - 1;
    r += k - prev - 1;
    prev = k;
    while (r > 15 && k <= EOBk) {
      if (COUNT) atomicAdd(&hist[0xF0], 1u); else R.own += s_size[0xF0] + (unsigned)BR;
      BR = 0; flushed_below = k; r -= 16;
    }
    if ((R.nzm >> k) & 1ull) { BR++; continue; }
    const int sym = (r << 4) + 1;
    if (COUNT) atomicAdd(&hist[sym], 1u); else R.own += s_size[sym] + 1u + (unsigned)BR;
    BR = 0; flushed_below = k + 1; r = 0;
  }
  r += Se - prev;
  R.tail_cnt = BR;
  R.E = (r > 0) || (BR > 0);
  R.tailm = flushed_below < 64 ? (R.nzm & ~((1ull << flushed_below) - 1ull)) : 0ull;
  return R;
}

// previous non-empty block of block j inside a chunk bitmap (-1: none)
__device__ __forceinline__ int pp_prev_ne(const unsigned long long *ne_bits, int j)
{
  int w = j >> 6;
  unsigned long long m = ne_bits[w] & ((1ull << (j & 63)) - 1ull);
  while (!m && w > 0) { w--; m = ne_bits[w]; }
  return m ? w * 64 + 63 - __builtin_clzll(m) : -1;
}
__device__ __forceinline__ int pp_next_ne(const unsigned long long *ne_bits, int j)
{
  int w = j >> 6;
  unsigned long long m = (j & 63) == 63 ? 0ull : (ne_bits[w] & ~((2ull << (j & 63)) - 1ull));
  while (!m && w < MJH_PSTAT_BLOCKS / 64 - 1) { w++; m = ne_bits[w]; }
  return m ? w * 64 + __builtin_ctzll(m) : -1;
}

__global__ void __launch_bounds__(64)
k_pp_init(MjhProgPE pe, int npairs)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < npairs) { pe.info[i].final_run = 0; pe.info[i].final_be = 0; pe.info[i].fallback = 0; pe.info[i].corr_total = 0; }
}


template <bool COMPACT, int SELX>   // SELX: SEL as in k_pp_len
__global__ void __launch_bounds__(256)
k_pp_stats(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list,
           const MjhProgCtl *__restrict__ ctl, const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask,
           MjhHuffTable *__restrict__ tabs, int slots_per_image, MjhProgPE pe, int li0)
{
  constexpr int SEL = SELX;
  constexpr bool SKIPLOW = SELX == 1;     // first-pass AC scans: the non-zeros below the band leave the mask up front (pp_band_nonzeros)
  constexpr int NH = SEL == 1 ? 16 : 4;   // (first-pass AC scans: a handful of symbols takes most of the counts)
  __shared__ unsigned hist[NH][256];   // DC scans: [table 0 / 1]; AC scans: NH interleaved copies (the hot symbols serialise the LDS atomics)
  __shared__ unsigned long long ne_bits[MJH_PSTAT_BLOCKS / 64], e_bits[MJH_PSTAT_BLOCKS / 64];
  __shared__ unsigned s_corr;
  const int img = blockIdx.z, li = li0 + blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;   // scans li0.. of the list
  const size_t pair = (size_t)li * gridDim.z + img;   // (the pairs of a scan are contiguous: image index fastest)
  const int sidx = scan_list[li];
  const MjhProgScan &sc = scans[sidx];
  const int kind = pp_kind(sc);
  if (SEL == 1 && kind != PP_AC_FIRST) return;
  if (SEL == 2 && kind == PP_AC_FIRST) return;
  const int nunits = pp_units(C, sc);
  const int cb = chunk * MJH_PSTAT_BLOCKS;
  if (cb >= nunits || kind == PP_DC_REFINE) return;
  const int nb = min(MJH_PSTAT_BLOCKS, nunits - cb);
  const MjhProgCtl *ct = ctl + img;
  if (prog_skip(sc, ct)) return;
  const int Al = sc.al_sel == 1 ? ct->best_Al_luma : (sc.al_sel == 2 ? ct->best_Al_chroma : sc.Al);
  const int16_t *qimg = coef_q + (size_t)img * C.coefs_per_image;
#pragma unroll
  for (int c = 0; c < NH; c++) hist[c][tid] = 0;
  if (tid < MJH_PSTAT_BLOCKS / 64) { ne_bits[tid] = 0; e_bits[tid] = 0; }
  if (tid == 0) s_corr = 0;
  __syncthreads();

  if (SEL != 1 && kind == PP_DC_FIRST) {
    // encode_mcu_DC_first jcphuff.c:468-555: category of the difference to the previous block of the component in scan order
    const bool inter = sc.ncomp > 1;
    for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
      const int u = cb + i * 256 + tid;
      if (u >= nunits) break;
      for (int ci = 0; ci < sc.ncomp; ci++) {
        const MjhComp cc = C.c[sc.comp[ci]];
        const int16_t *q0 = qimg + cc.coef_off;
        const int tb = (cc.dctbl >> 8) & 1;      // the DC table's class (MjhComp)
        const int mh = inter ? cc.v : 1, mw = inter ? cc.h : 1;
        for (int yi = 0; yi < mh; yi++)
          for (int xi = 0; xi < mw; xi++) {
            int dc, pred = 0;
            if (inter) {
              const int my = u / C.mcus_per_row, mx = u - my * C.mcus_per_row;
              const int r = my * cc.v + yi, c = mx * cc.h + xi;
              dc = q0[dc_source_block(cc, r, c)];
              int pr = 0, pc = 0;
              bool has = true;    // no restart intervals here: the predictor only starts at 0 in the first MCU
              if (xi > 0) { pr = r; pc = c - 1; }
              else if (yi > 0) { pr = r - 1; pc = c + cc.h - 1; }
              else if (u == 0) has = false;
              else { const int pm = u - 1, pmy = pm / C.mcus_per_row, pmx = pm - pmy * C.mcus_per_row; pr = pmy * cc.v + cc.v - 1; pc = pmx * cc.h + cc.h - 1; }
              if (has) pred = q0[dc_source_block(cc, pr, pc)] >> Al;
            } else {
              dc = q0[u];
              if (u > 0) pred = q0[u - 1] >> Al;
            }
            const int df = (dc >> Al) - pred;
            atomicAdd(&hist[tb][bitlen((unsigned)(df < 0 ? -df : df))], 1u);
          }
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int t = tid >> 4, sym = tid & 15;
      if (sc.slot[t] >= 0 && hist[t][sym]) atomicAdd(&tabs[(size_t)img * slots_per_image + sc.slot[t]].counts[sym], hist[t][sym]);
    }
    return;
  }

  const MjhComp cc = C.c[sc.comp[0]];
  const int Ss = sc.Ss, Se = sc.Se;
  const bool refine = SEL == 1 ? false : (SEL == 2 ? true : kind == PP_AC_REFINE);
  const int16_t *qc = qimg + cc.coef_off;
  uint16_t *tail = pe.tail16 + pair * pe.nblk_pad + cb;
  unsigned corr = 0;
#pragma unroll 1
  for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
    const int j = i * 256 + tid;
    if (i * 256 >= nb) { if (refine) tail[j] = 0; continue; }   // uniform
    const int16_t *qs = qc + cb + (j < nb ? j : nb - 1);
    bool ne = false, E = false;
    int tc = 0;
    if (COMPACT) {
      const unsigned long long m = nzmask[(size_t)img * C.total_real_blocks + cc.blk_off + cb + (j < nb ? j : nb - 1)];
      if (!refine) {
        int prev = Ss - 1;
        unsigned *hh = hist[tid & (NH - 1)];
        pp_band_nonzeros<SKIPLOW>(qs, (size_t)cc.kstride, m, Ss, Se, j < nb, [&](int k, int v) {
          const int a = (v < 0 ? -v : v) >> Al;
          if (a == 0) return;
          int r = k - prev - 1;
          prev = k;
          if (r > 15) { atomicAdd(&hh[0xF0], (unsigned)(r >> 4)); r &= 15; }
          atomicAdd(&hh[(r << 4) + bitlen((unsigned)a)], 1u);
        });
        ne = prev != Ss - 1;
        E = prev < Se;
      } else {
        const PPRefine R = pp_refine_block_compact<true>(qs, (size_t)cc.kstride, m, j < nb, Ss, Se, Al, nullptr, hist[tid & 3]);
        ne = R.ne; E = R.E; tc = R.tail_cnt;
        if (SEL == 2 && pe.rmask && j < nb) {      // the sizes and the bits of this scan are made from these masks, not from the records again
          unsigned long long *rm = pe.rmask + (pair - (size_t)li0 * gridDim.z) * 3 * pe.nblk_pad + cb + j;
          rm[0] = R.newm; rm[pe.nblk_pad] = R.nzm; rm[2 * (size_t)pe.nblk_pad] = R.posm | R.corrm;
        }
        corr += (unsigned)__popcll(R.nzm);
      }
      if (j < nb) {
        if (ne) atomicOr(&ne_bits[j >> 6], 1ull << (j & 63));
        if (E) atomicOr(&e_bits[j >> 6], 1ull << (j & 63));
      }
      if (refine) tail[j] = (uint16_t)(j < nb ? tc : 0);
      continue;
    }
    int x[64];
    PP_LOAD_BAND(x, qs, cc.kstride, Ss, Se)
    if (j < nb) {
      if (!refine) {
        int r = 0;
#pragma unroll
        for (int g = 0; g < 8; g++) if (8 * g + 7 >= Ss && 8 * g <= Se) {   // wave-uniform: only the groups of 8 that overlap the band
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const int k = 8 * g + kk;
          if (k >= 1 && k >= Ss && k <= Se) {
            const int v = x[k];
            const int a = (v < 0 ? -v : v) >> Al;
            if (a == 0) r++;
            else {
              ne = true;
              const int nz16 = r >> 4;
              r &= 15;
              if (nz16) atomicAdd(&hist[tid & 3][0xF0], (unsigned)nz16);
              atomicAdd(&hist[tid & 3][(r << 4) + bitlen((unsigned)a)], 1u);
              r = 0;
            }
          }
        } }
        E = r > 0;
      } else {
        const PPRefine R = pp_refine_block<true>(x, Ss, Se, Al, nullptr, hist[tid & 3]);
        ne = R.ne; E = R.E; tc = R.tail_cnt;
        corr += (unsigned)__popcll(R.nzm);
      }
      if (ne) atomicOr(&ne_bits[j >> 6], 1ull << (j & 63));
      if (E) atomicOr(&e_bits[j >> 6], 1ull << (j & 63));
    }
    if (refine) tail[j] = (uint16_t)tc;
  }
  if (refine && corr) atomicAdd(&s_corr, corr);
  __syncthreads();
  if (tid < MJH_PSTAT_BLOCKS / 64) {
    pe.ne_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid] = ne_bits[tid];
    pe.ne2_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid] = ne_bits[tid];   // + forced-flush marks (k_pp_cuts)
    pe.e_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid] = e_bits[tid];
  }
  if (tid == 0) {
    MjhProgChunk ch;
    ch.first_ne = -1; ch.last_ne = -1; ch.e_last = 0; ch.nblk = nb; ch.carry_p = -1; ch.carry_e = 0; ch.next_after = -1; ch.pad = 0;
    for (int w = 0; w < MJH_PSTAT_BLOCKS / 64; w++)
      if (ne_bits[w]) { ch.first_ne = w * 64 + __builtin_ctzll(ne_bits[w]); break; }
    for (int w = MJH_PSTAT_BLOCKS / 64 - 1; w >= 0; w--)
      if (ne_bits[w]) { ch.last_ne = w * 64 + 63 - __builtin_clzll(ne_bits[w]); break; }
    if (ch.last_ne >= 0) ch.e_last = (int)((e_bits[ch.last_ne >> 6] >> (ch.last_ne & 63)) & 1ull);
    pe.chunks[pair * pe.chunks_per_scan + chunk] = ch;
    if (refine && s_corr) atomicAdd(&pe.info[pair].corr_total, s_corr);
  }
  MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
  unsigned hsum = 0;
#pragma unroll
  for (int c = 0; c < NH; c++) hsum += hist[c][tid];
  if (hsum) atomicAdd(&T0->counts[tid], hsum);
  if (SEL == 1) pe.chist[(pair * pe.chunks_per_scan + chunk) * 256 + tid] = hsum;   // k_pp_runs / k_pp_resolve add the EOBRUN symbols
}

// T at index i of a pair (exclusive prefix sum of the trailing-bit counts); the entry behind the last unit is the total
__device__ __forceinline__ unsigned pp_T(const MjhProgPE &pe, size_t pair, int i, bool refine)
{
  if (!refine) return 0u;
  return i < pe.nblk_pad ? pe.T32[pair * pe.nblk_pad + i] : pe.ttotals[pair];
}

// forced flushes of the gap that starts accumulating at block a and ends in front of flush point j (j = number of blocks:
// the end of the scan): the reference flushes after block i as soon as the run has 0x7FFF blocks or more than 937 buffered
// bits (jcphuff.c:719,:998-1000); block i+1 becomes a flush point.  Greedy, as in the reference.
__device__ __forceinline__ void pp_mark_cuts(const MjhProgPE &pe, size_t pair, int a, int j, bool refine)
{
  unsigned long long *ne2 = pe.ne2_bits + pair * pe.chunks_per_scan * (MJH_PSTAT_BLOCKS / 64);
  while (a < j) {
    int cut = a + 0x7FFF - 1;                       // the block that makes the run 0x7FFF long
    if (refine) {
      const unsigned Ta = pp_T(pe, pair, a, true);
      if (pp_T(pe, pair, j, true) - Ta > 937u) {    // smallest i in [a, j-1] with T[i+1] - T[a] > 937
        int lo = a, hi = j - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (pp_T(pe, pair, mid + 1, true) - Ta > 937u) hi = mid; else lo = mid + 1; }
        cut = lo < cut ? lo : cut;
      }
    }
    if (cut + 1 >= j) break;                        // a flush right in front of j happens anyway
    a = cut + 1;
    atomicOr(&ne2[a >> 6], 1ull << (a & 63));       // (word index over the whole pair: chunks are 32 consecutive words)
  }
}

// per (scan, image) pair, before the marks: the last real non-empty block in front of every chunk; marks of the final gap
__global__ void __launch_bounds__(64)
k_pp_carry(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl, MjhProgPE pe)
{
  const int img = blockIdx.y, li = blockIdx.x;
  if (threadIdx.x != 0) return;
  const size_t pair = (size_t)li * gridDim.y + img;
  const MjhProgScan &sc = scans[scan_list[li]];
  if (prog_skip(sc, ctl + img)) return;
  const int kind = pp_kind(sc);
  if (kind == PP_DC_FIRST || kind == PP_DC_REFINE) return;
  const MjhComp cc = C.c[sc.comp[0]];
  const int nchunks = (cc.nblk + MJH_PSTAT_BLOCKS - 1) / MJH_PSTAT_BLOCKS;
  MjhProgChunk *chs = pe.chunks + pair * pe.chunks_per_scan;
  int last = -1, last_e = 0;
  for (int c = 0; c < nchunks; c++) {
    chs[c].carry_p = last; chs[c].carry_e = last_e;
    if (chs[c].first_ne >= 0) { last = c * MJH_PSTAT_BLOCKS + chs[c].last_ne; last_e = chs[c].e_last; }
  }
  pp_mark_cuts(pe, pair, last < 0 ? 0 : (last_e ? last : last + 1), cc.nblk, kind == PP_AC_REFINE);
}

// one thread per real non-empty block: the gap in front of it
__global__ void __launch_bounds__(256)
k_pp_cuts(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl, MjhProgPE pe)
{
  __shared__ unsigned long long ne_bits[MJH_PSTAT_BLOCKS / 64], e_bits[MJH_PSTAT_BLOCKS / 64];
  const int img = blockIdx.z, li = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const size_t pair = (size_t)li * gridDim.z + img;
  const MjhProgScan &sc = scans[scan_list[li]];
  if (prog_skip(sc, ctl + img)) return;
  const int kind = pp_kind(sc);
  if (kind == PP_DC_FIRST || kind == PP_DC_REFINE) return;
  const MjhComp cc = C.c[sc.comp[0]];
  const int cb = chunk * MJH_PSTAT_BLOCKS;
  if (cb >= cc.nblk) return;
  const MjhProgChunk ch = pe.chunks[pair * pe.chunks_per_scan + chunk];
  if (ch.first_ne < 0) return;      // uniform: no real non-empty block here
  if (tid < MJH_PSTAT_BLOCKS / 64) {
    ne_bits[tid] = pe.ne_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
    e_bits[tid] = pe.e_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
  }
  __syncthreads();
  for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
    const int j = i * 256 + tid;
    if (!((ne_bits[j >> 6] >> (j & 63)) & 1ull)) continue;
    const int p = pp_prev_ne(ne_bits, j);
    int a;
    if (p >= 0) a = cb + (((e_bits[p >> 6] >> (p & 63)) & 1ull) ? p : p + 1);
    else a = ch.carry_p < 0 ? 0 : (ch.carry_e ? ch.carry_p : ch.carry_p + 1);
    pp_mark_cuts(pe, pair, a, cb + j, kind == PP_AC_REFINE);
  }
}

// per (scan, image) pair, after the marks: what crosses chunk borders.  One lane walks the chunks (at most a few dozen).
__global__ void __launch_bounds__(64)
k_pp_resolve(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl,
             MjhHuffTable *__restrict__ tabs, int slots_per_image, MjhProgPE pe)
{
  const int img = blockIdx.y, li = blockIdx.x, lane = threadIdx.x;
  const size_t pair = (size_t)li * gridDim.y + img;
  const MjhProgScan &sc = scans[scan_list[li]];
  if (prog_skip(sc, ctl + img)) return;
  const int kind = pp_kind(sc);
  if (kind == PP_DC_FIRST || kind == PP_DC_REFINE) return;
  const bool refine = kind == PP_AC_REFINE;
  const MjhComp cc = C.c[sc.comp[0]];
  MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
  if (lane == 0) {
    const int nchunks = (cc.nblk + MJH_PSTAT_BLOCKS - 1) / MJH_PSTAT_BLOCKS;
    MjhProgChunk *chs = pe.chunks + pair * pe.chunks_per_scan;
    uint16_t *run = pe.run16 + pair * pe.nblk_pad;
    uint16_t *be16 = pe.be16 + pair * pe.nblk_pad;
    unsigned pending = 0;
    int last = -1, last_e = 0, nextra;
    for (int c = 0; c < nchunks; c++) {
      MjhProgChunk ch = chs[c];          // first_ne / last_ne / e_last: flush points of the chunk (k_pp_runs)
      ch.carry_p = last; ch.carry_e = last_e;
      if (ch.first_ne >= 0) {
        const int b = c * MJH_PSTAT_BLOCKS + ch.first_ne;
        const unsigned r = pending + (unsigned)ch.first_ne;
        run[b] = (uint16_t)r;
        if (r) {
          const int sym = eobrun_symbol(r, &nextra);
          T0->counts[sym] += 1;
          if (!refine && pe.chist) pe.chist[(pair * pe.chunks_per_scan + c) * 256 + sym] += 1;
        }
        if (refine) be16[b] = (uint16_t)(pp_T(pe, pair, b, true) - (last >= 0 ? pp_T(pe, pair, last, true) : 0u));
        last = c * MJH_PSTAT_BLOCKS + ch.last_ne; last_e = ch.e_last;
        pending = (unsigned)ch.e_last + (unsigned)(ch.nblk - 1 - ch.last_ne);
      } else
        pending += (unsigned)ch.nblk;
      chs[c] = ch;
    }
    if (pending) T0->counts[eobrun_symbol(pending, &nextra)] += 1;
    int nxt = -1;
    for (int c = nchunks - 1; c >= 0; c--) {
      chs[c].next_after = nxt;
      if (chs[c].first_ne >= 0) nxt = c * MJH_PSTAT_BLOCKS + chs[c].first_ne;
    }
    pe.info[pair].final_run = pending;
    pe.info[pair].final_be = refine ? pp_T(pe, pair, cc.nblk, true) - (last >= 0 ? pp_T(pe, pair, last, true) : 0u) : 0u;
    T0->counts[256] = 0; T0->counts[257] = 0;
    T0->counts[258] = refine ? pe.info[pair].corr_total : 0u;   // correction bits of the scan: needed to size its bit stream
  }
  __syncthreads();
  if (sc.seed)   // trellis passes: every (run, size < 12) count starts at 1 (jcphuff.c:257-264)
    for (int i = lane; i < 256; i += 64)
      if ((i & 15) < 12) T0->counts[i] += 1;
}

// every flush point that is not the first of its chunk: the run and the correction bits in front of it, EOBRUN statistics
__global__ void __launch_bounds__(256)
k_pp_runs(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl, MjhHuffTable *__restrict__ tabs,
          int slots_per_image, MjhProgPE pe)
{
  __shared__ unsigned long long ne_bits[MJH_PSTAT_BLOCKS / 64], e_bits[MJH_PSTAT_BLOCKS / 64];
  __shared__ unsigned hist[16];       // EOBRUN symbols: (nbits - 1) << 4, nbits - 1 = 0..14
  const int img = blockIdx.z, li = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const size_t pair = (size_t)li * gridDim.z + img;
  const MjhProgScan &sc = scans[scan_list[li]];
  if (prog_skip(sc, ctl + img)) return;
  const int kind = pp_kind(sc);
  if (kind == PP_DC_FIRST || kind == PP_DC_REFINE) return;
  const bool refine = kind == PP_AC_REFINE;
  const MjhComp cc = C.c[sc.comp[0]];
  const int cb = chunk * MJH_PSTAT_BLOCKS;
  if (cb >= cc.nblk) return;
  if (tid < MJH_PSTAT_BLOCKS / 64) {
    ne_bits[tid] = pe.ne2_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
    e_bits[tid] = pe.e_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
  }
  if (tid < 16) hist[tid] = 0;
  __syncthreads();
  const unsigned *T = pe.T32 + pair * pe.nblk_pad + cb;
  uint16_t *run = pe.run16 + pair * pe.nblk_pad + cb;
  uint16_t *be16 = pe.be16 + pair * pe.nblk_pad + cb;
  for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
    const int j = i * 256 + tid;
    if (!((ne_bits[j >> 6] >> (j & 63)) & 1ull)) continue;
    const int p = pp_prev_ne(ne_bits, j);
    if (p < 0) continue;                 // first of the chunk: k_pp_resolve
    const unsigned cnt = (unsigned)((e_bits[p >> 6] >> (p & 63)) & 1ull) + (unsigned)(j - p - 1);
    run[j] = (uint16_t)cnt;
    if (refine) be16[j] = (uint16_t)(T[j] - T[p]);
    if (cnt) { int nextra; atomicAdd(&hist[eobrun_symbol(cnt, &nextra) >> 4], 1u); }
  }
  __syncthreads();
  if (tid < 16 && hist[tid]) {
    atomicAdd(&tabs[(size_t)img * slots_per_image + sc.slot[0]].counts[tid << 4], hist[tid]);
    if (!refine && pe.chist) pe.chist[(pair * pe.chunks_per_scan + chunk) * 256 + (tid << 4)] += hist[tid];   // (own symbols never have size 0)
  }
  if (tid == 0) {   // first / last flush point of the chunk (real non-empty blocks and forced-flush marks) for k_pp_resolve
    MjhProgChunk *ch = pe.chunks + pair * pe.chunks_per_scan + chunk;
    int first = -1, last = -1;
    for (int w = 0; w < MJH_PSTAT_BLOCKS / 64; w++) if (ne_bits[w]) { first = w * 64 + __builtin_ctzll(ne_bits[w]); break; }
    for (int w = MJH_PSTAT_BLOCKS / 64 - 1; w >= 0; w--) if (ne_bits[w]) { last = w * 64 + 63 - __builtin_clzll(ne_bits[w]); break; }
    ch->first_ne = first; ch->last_ne = last;
    ch->e_last = last >= 0 ? (int)((e_bits[last >> 6] >> (last & 63)) & 1ull) : 0;
  }
}


// ---- encode -------------------------------------------------------------------------------------------------------
// DC units: bits of one unit (all its blocks) / writing them
template <bool WRITE>
__device__ __forceinline__ unsigned pp_dc_unit(const MjhConst &C, const MjhProgScan &sc, const int16_t *qimg, int u, int Al,
                                               const unsigned (*s_tab)[16], BitWriter *bw)
{
  const bool inter = sc.ncomp > 1;
  unsigned bits = 0;
  for (int ci = 0; ci < sc.ncomp; ci++) {
    const MjhComp cc = C.c[sc.comp[ci]];
    const int16_t *q0 = qimg + cc.coef_off;
    const int tb = (cc.dctbl >> 8) & 1;      // the DC table's class (MjhComp)
    const int mh = inter ? cc.v : 1, mw = inter ? cc.h : 1;
    for (int yi = 0; yi < mh; yi++)
      for (int xi = 0; xi < mw; xi++) {
        int dc, pred = 0;
        if (inter) {
          const int my = u / C.mcus_per_row, mx = u - my * C.mcus_per_row;
          const int r = my * cc.v + yi, c = mx * cc.h + xi;
          dc = q0[dc_source_block(cc, r, c)];
          if (sc.Ah == 0) {
            int pr = 0, pc = 0;
            bool has = true;
            if (xi > 0) { pr = r; pc = c - 1; }
            else if (yi > 0) { pr = r - 1; pc = c + cc.h - 1; }
            else if (u == 0) has = false;
            else { const int pm = u - 1, pmy = pm / C.mcus_per_row, pmx = pm - pmy * C.mcus_per_row; pr = pmy * cc.v + cc.v - 1; pc = pmx * cc.h + cc.h - 1; }
            if (has) pred = q0[dc_source_block(cc, pr, pc)] >> Al;
          }
        } else {
          dc = q0[u];
          if (sc.Ah == 0 && u > 0) pred = q0[u - 1] >> Al;
        }
        if (sc.Ah == 0) {
          const int df = (dc >> Al) - pred;
          const int nb = bitlen((unsigned)(df < 0 ? -df : df));
          const unsigned e = s_tab[tb][nb];
          if (!WRITE) bits += (e >> 16) + (unsigned)nb;
          else { bw->put(e & 0xFFFF, (int)(e >> 16)); if (nb) bw->put((unsigned)(df < 0 ? df - 1 : df), nb); }
        } else {
          if (!WRITE) bits += 1; else bw->put((unsigned)(dc >> Al) & 1u, 1);
        }
      }
  }
  return bits;
}

// SEL: 0 = every kind of scan; 1 = only the first-pass AC scans (most scans of a search; its own kernel needs half the
// registers and no scratch, so twice the waves hide the dependent loads); 2 = the other kinds
template <bool COMPACT, int SEL>
__global__ void __launch_bounds__(256)
k_pp_len(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl,
         const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, const MjhHuffTable *__restrict__ tabs, int slots_per_image, MjhProgPE pe, int li0)
{
  __shared__ unsigned char s_size[256];
  __shared__ unsigned s_dc[2][16];
  const int img = blockIdx.z, li = li0 + blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;   // scans li0.. of the list
  const size_t pair = (size_t)li * gridDim.z + img;   // (the pairs of a scan are contiguous: image index fastest)
  const int sidx = scan_list[li];
  const MjhProgScan &sc = scans[sidx];
  const int kind = pp_kind(sc);
  if (SEL == 1 && kind != PP_AC_FIRST) return;
  if (SEL == 2 && kind == PP_AC_FIRST) return;
  const int nunits = pp_units(C, sc);
  const int cb = chunk * MJH_PSTAT_BLOCKS;
  uint16_t *len = pe.len16 + pair * pe.nblk_pad + cb;
  if (cb >= nunits) {         // padding of the length array behind the scan's units: contributes nothing to the prefix sum
    for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) len[i * 256 + tid] = 0;
    return;
  }
  const int nb = min(MJH_PSTAT_BLOCKS, nunits - cb);
  const MjhProgCtl *ct = ctl + img;
  if (prog_skip(sc, ct)) return;
  const int Al = sc.al_sel == 1 ? ct->best_Al_luma : (sc.al_sel == 2 ? ct->best_Al_chroma : sc.Al);
  const int16_t *qimg = coef_q + (size_t)img * C.coefs_per_image;
  if (SEL != 1 && (kind == PP_DC_FIRST || kind == PP_DC_REFINE)) {
    if (tid < 32) {
      const int t = tid >> 4, sym = tid & 15;
      const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + (sc.slot[t] >= 0 ? sc.slot[t] : 0);
      s_dc[t][sym] = sc.slot[t] >= 0 && kind == PP_DC_FIRST ? ((unsigned)T->ehufsi[sym] << 16) | T->ehufco[sym] : 0u;
    }
    __syncthreads();
    for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
      const int j = i * 256 + tid;
      len[j] = j < nb ? (uint16_t)pp_dc_unit<false>(C, sc, qimg, cb + j, Al, s_dc, nullptr) : (uint16_t)0;
    }
    return;
  }
  const MjhComp cc = C.c[sc.comp[0]];
  const int Ss = sc.Ss, Se = sc.Se;
  const bool refine = SEL == 1 ? false : (SEL == 2 ? true : kind == PP_AC_REFINE);
  const MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
  const int16_t *qc = qimg + cc.coef_off;
  const uint16_t *run = pe.run16 + pair * pe.nblk_pad + cb;
  const uint16_t *be16 = pe.be16 + pair * pe.nblk_pad + cb;
  __shared__ unsigned long long fp_bits[MJH_PSTAT_BLOCKS / 64];   // flush points: real non-empty blocks + forced-flush marks
  __shared__ unsigned long long rn_bits[MJH_PSTAT_BLOCKS / 64];   // real non-empty blocks (from the statistics pass of this phase)
  s_size[tid] = T0->ehufsi[tid];
  if (tid < MJH_PSTAT_BLOCKS / 64) {
    fp_bits[tid] = pe.ne2_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
    rn_bits[tid] = pe.ne_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
  }
  __syncthreads();
#pragma unroll 1
  for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
    const int j = i * 256 + tid;
    if (i * 256 >= nb) { len[j] = 0; continue; }   // uniform
    // a wave whose 64 blocks are all empty (sparse bands, refinement scans) has no symbols to size: skip its plane loads
    const bool has_own = j < nb && ((rn_bits[j >> 6] >> (j & 63)) & 1ull);
    unsigned own = 0;
    bool ne = false;
    if (COMPACT) {
      if (__builtin_amdgcn_ballot_w64(has_own) != 0ull) {
        const int jb = cb + (j < nb ? j : nb - 1);
        const bool cached = SEL == 2 && pe.rmask != nullptr;
        const unsigned long long m = cached ? 0ull : nzmask[(size_t)img * C.total_real_blocks + cc.blk_off + jb];
        if (!refine) {
          int prev = Ss - 1;
          const unsigned zrl = s_size[0xF0];
          pp_band_nonzeros(qc + jb, (size_t)cc.kstride, m, Ss, Se, has_own, [&](int k, int v) {
            const int a = (v < 0 ? -v : v) >> Al;
            if (a == 0) return;
            const int r = k - prev - 1;
            prev = k;
            const int nbits = bitlen((unsigned)a);
            own += (unsigned)(r >> 4) * zrl + s_size[((r & 15) << 4) + nbits] + (unsigned)nbits;
          });
          ne = prev != Ss - 1;
        } else {
          const PPRefine R = cached ? pp_refine_block_cached<false>(pe.rmask + (pair - (size_t)li0 * gridDim.z) * 3 * pe.nblk_pad + jb, (size_t)pe.nblk_pad, has_own, Ss, Se, s_size, nullptr)
                                    : pp_refine_block_compact<false>(qc + jb, (size_t)cc.kstride, m, has_own, Ss, Se, Al, s_size, nullptr);
          ne = R.ne; own = R.own;
        }
        if (!ne || !has_own) own = 0;
      }
      if (j < nb && ((fp_bits[j >> 6] >> (j & 63)) & 1ull)) {
        const unsigned cnt = run[j];
        if (cnt) { int nextra; const int sym = eobrun_symbol(cnt, &nextra); own += s_size[sym] + (unsigned)nextra + (refine ? be16[j] : 0u); }
      }
      len[j] = (uint16_t)own;
      continue;
    }
    int x[64];
    if (__builtin_amdgcn_ballot_w64(has_own) != 0ull) {
      const int16_t *qs = qc + cb + (j < nb ? j : nb - 1);
      PP_LOAD_BAND(x, qs, cc.kstride, Ss, Se)
    } else {
#pragma unroll
      for (int k = 0; k < 64; k++) x[k] = 0;
    }
    if (has_own) {
      if (!refine) {
        int r = 0;
#pragma unroll
        for (int g = 0; g < 8; g++) if (8 * g + 7 >= Ss && 8 * g <= Se) {   // wave-uniform: only the groups of 8 that overlap the band
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const int k = 8 * g + kk;
          if (k >= 1 && k >= Ss && k <= Se) {
            const int v = x[k];
            const int a = (v < 0 ? -v : v) >> Al;
            if (a == 0) r++;
            else {
              ne = true;
              const int nz16 = r >> 4;
              r &= 15;
              const int nbits = bitlen((unsigned)a);
              own += (unsigned)nz16 * s_size[0xF0] + s_size[(r << 4) + nbits] + (unsigned)nbits;
              r = 0;
            }
          }
        } }
      } else {
        const PPRefine R = pp_refine_block<false>(x, Ss, Se, Al, s_size, nullptr);
        ne = R.ne; own = R.own;
      }
      if (!ne) own = 0;
    }
    if (j < nb && ((fp_bits[j >> 6] >> (j & 63)) & 1ull)) {     // the pending run (+ its buffered correction bits) goes out in front of this block
      const unsigned cnt = run[j];
      if (cnt) { int nextra; const int sym = eobrun_symbol(cnt, &nextra); own += s_size[sym] + (unsigned)nextra + (refine ? be16[j] : 0u); }
    }
    len[j] = (uint16_t)own;
  }
}

template <bool COMPACT, int SEL>
__global__ void __launch_bounds__(256)
k_pp_write(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl,
           const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
           unsigned *__restrict__ pool, size_t pool_words_per_image, MjhProgPE pe, int li0)
{
  __shared__ unsigned s_tab[256];   // size << 16 | code
  __shared__ unsigned s_dc[2][16];
  __shared__ unsigned long long ne_bits[MJH_PSTAT_BLOCKS / 64];
  const int img = blockIdx.z, li = li0 + blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;   // scans li0.. of the list
  const size_t pair = (size_t)li * gridDim.z + img;   // (the pairs of a scan are contiguous: image index fastest)
  const int sidx = scan_list[li];
  const MjhProgScan &sc = scans[sidx];
  const int kind = pp_kind(sc);
  if (SEL == 1 && kind != PP_AC_FIRST) return;
  if (SEL == 2 && kind == PP_AC_FIRST) return;
  const int nunits = pp_units(C, sc);
  const int cb = chunk * MJH_PSTAT_BLOCKS;
  const MjhProgCtl *ct = ctl + img;
  if (prog_skip(sc, ct)) return;
  if (cb >= nunits || ct->error) return;
  const int nb = min(MJH_PSTAT_BLOCKS, nunits - cb);
  const int Al = sc.al_sel == 1 ? ct->best_Al_luma : (sc.al_sel == 2 ? ct->best_Al_chroma : sc.Al);
  const int16_t *qimg = coef_q + (size_t)img * C.coefs_per_image;
  unsigned *stream = pool + (size_t)img * pool_words_per_image;
  const unsigned base = ct->scan_words_off[sidx] * 32u;
  const uint16_t *len = pe.len16 + pair * pe.nblk_pad + cb;
  const unsigned *off = pe.off32 + pair * pe.nblk_pad + cb;
  if (SEL != 1 && (kind == PP_DC_FIRST || kind == PP_DC_REFINE)) {
    if (tid < 32) {
      const int t = tid >> 4, sym = tid & 15;
      const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + (sc.slot[t] >= 0 ? sc.slot[t] : 0);
      s_dc[t][sym] = sc.slot[t] >= 0 && kind == PP_DC_FIRST ? ((unsigned)T->ehufsi[sym] << 16) | T->ehufco[sym] : 0u;
    }
    __syncthreads();
    for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
      const int j = i * 256 + tid;
      if (j >= nb) break;
      BitWriter bw;
      bw.init(stream, base + off[j]);
      pp_dc_unit<true>(C, sc, qimg, cb + j, Al, s_dc, &bw);
      bw.flush();
    }
    return;
  }
  const MjhComp cc = C.c[sc.comp[0]];
  const int Ss = sc.Ss, Se = sc.Se;
  const bool refine = SEL == 1 ? false : (SEL == 2 ? true : kind == PP_AC_REFINE);
  const MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
  const int16_t *qc = qimg + cc.coef_off;
  const uint16_t *run = pe.run16 + pair * pe.nblk_pad + cb;
  const uint16_t *be16 = pe.be16 + pair * pe.nblk_pad + cb;
  const unsigned *T = pe.T32 + pair * pe.nblk_pad;          // pair-global index
  const unsigned *offg = pe.off32 + pair * pe.nblk_pad;
  const uint16_t *rung = pe.run16 + pair * pe.nblk_pad;
  const MjhProgChunk ch = pe.chunks[pair * pe.chunks_per_scan + chunk];
  s_tab[tid] = ((unsigned)T0->ehufsi[tid] << 16) | T0->ehufco[tid];
  if (tid < MJH_PSTAT_BLOCKS / 64) ne_bits[tid] = pe.ne2_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];   // flush points
  __syncthreads();
  // the record mask and the block's bit count of the NEXT round are fetched while this round is being written (the loop is
  // bound by the latency of its dependent loads: count -> mask -> values)
  const unsigned long long *nzm = COMPACT ? nzmask + (size_t)img * C.total_real_blocks + cc.blk_off + cb : nullptr;
  unsigned long long mask_next = COMPACT ? nzm[tid < nb ? tid : nb - 1] : 0ull;
  unsigned len_next = tid < nb ? len[tid] : 0u;
#pragma unroll 1
  for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
    const int j = i * 256 + tid;
    if (i * 256 >= nb) break;   // uniform
    const unsigned long long cmask = mask_next;
    const unsigned len_j = len_next;
    if ((i + 1) * 256 < nb) {
      const int jn = j + 256 < nb ? j + 256 : nb - 1;
      if (COMPACT) mask_next = nzm[jn];
      len_next = j + 256 < nb ? len[j + 256] : 0u;
    }
    // blocks without symbols or trailing correction bits of their own need no coefficients: a wave of such blocks skips the loads
    const bool has_data = j < nb && (len_j != 0 || (refine && pe.tail16[pair * pe.nblk_pad + cb + j] != 0));
    if (__builtin_amdgcn_ballot_w64(has_data) == 0ull) continue;
    const int jb = cb + (j < nb ? j : nb - 1);
    int x[64];
    if (!COMPACT) {
      const int16_t *qs = qc + jb;
      PP_LOAD_BAND(x, qs, cc.kstride, Ss, Se)
    }
    if (COMPACT && !refine) {
      // AC-first scan from compact records: every lane of the wave takes part in the (wave-uniform) plane loads
      const bool wr = has_data && len_j != 0;
      BitWriter bw;
      bw.init(stream, base + (wr ? off[j] : 0u));
      if (wr) {
        const unsigned cnt = run[j];
        if (cnt) {
          int nextra;
          const unsigned e = s_tab[eobrun_symbol(cnt, &nextra)];
          bw.put(e & 0xFFFF, (int)(e >> 16));
          if (nextra) bw.put(cnt & ((1u << nextra) - 1u), nextra);
        }
      }
      int prev = Ss - 1;
      pp_band_nonzeros(qc + jb, (size_t)cc.kstride, cmask, Ss, Se, wr, [&](int k, int v) {
        const int a = (v < 0 ? -v : v) >> Al;
        if (a == 0) return;
        int r = k - prev - 1;
        prev = k;
        while (r > 15) { const unsigned e = s_tab[0xF0]; bw.put(e & 0xFFFF, (int)(e >> 16)); r -= 16; }
        const int nbits = bitlen((unsigned)a);
        const unsigned e = s_tab[(r << 4) + nbits];
        bw.put_sym(e, (unsigned)(v < 0 ? ~a : a), nbits);
      });
      if (wr) bw.flush();
      continue;
    }
    PPRefine Rc;
    if (COMPACT) {
      if (SEL == 2 && pe.rmask) Rc = pp_refine_block_cached<false>(pe.rmask + (pair - (size_t)li0 * gridDim.z) * 3 * pe.nblk_pad + jb, (size_t)pe.nblk_pad, has_data, Ss, Se, reinterpret_cast<const unsigned char *>(s_tab), nullptr);
      else Rc = pp_refine_block_compact<false>(qc + jb, (size_t)cc.kstride, cmask, has_data, Ss, Se, Al, reinterpret_cast<const unsigned char *>(s_tab), nullptr);
    }
    if (!has_data) continue;
    const bool flush_point = (ne_bits[j >> 6] >> (j & 63)) & 1ull;
    if (!refine) {
      if (len_j == 0) continue;          // nothing to write (a non-empty block has at least one bit of its own)
      BitWriter bw;
      bw.init(stream, base + off[j]);
      const unsigned cnt = run[j];
      if (cnt) {                          // the pending EOB run goes out in front of the block (emit_eobrun jcphuff.c:409)
        int nextra;
        const unsigned e = s_tab[eobrun_symbol(cnt, &nextra)];
        bw.put(e & 0xFFFF, (int)(e >> 16));
        if (nextra) bw.put(cnt & ((1u << nextra) - 1u), nextra);
      }
      int r = 0;
#pragma unroll
      for (int g = 0; g < 8; g++) if (8 * g + 7 >= Ss && 8 * g <= Se) {   // wave-uniform: only the groups of 8 that overlap the band
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const int k = 8 * g + kk;
        if (k >= 1 && k >= Ss && k <= Se) {
          const int v = x[k];
          const int a = (v < 0 ? -v : v) >> Al;
          if (a == 0) r++;
          else {
            while (r > 15) { const unsigned e = s_tab[0xF0]; bw.put(e & 0xFFFF, (int)(e >> 16)); r -= 16; }
            const int nbits = bitlen((unsigned)a);
            const unsigned e = s_tab[(r << 4) + nbits];
            bw.put_sym(e, (unsigned)(v < 0 ? ~a : a), nbits);
            r = 0;
          }
        }
      } }
      bw.flush();
      continue;
    }
    // ---- refinement scan
    const PPRefine R = COMPACT ? Rc : pp_refine_block<false>(x, Ss, Se, Al, reinterpret_cast<const unsigned char *>(s_tab), nullptr);   // (own bits unused here)
    if (flush_point) {
      BitWriter bw;
      bw.init(stream, base + off[j]);
      const unsigned cnt = run[j];
      unsigned flush = 0;
      if (cnt) {
        int nextra;
        const unsigned e = s_tab[eobrun_symbol(cnt, &nextra)];
        bw.put(e & 0xFFFF, (int)(e >> 16));
        if (nextra) bw.put(cnt & ((1u << nextra) - 1u), nextra);
        bw.flush();
        flush = (e >> 16) + (unsigned)nextra + be16[j];
      }
      if (!R.ne) goto tails;                    // a forced-flush mark: no symbols of its own
      bw.init(stream, base + off[j] + flush);   // the buffered correction bits of the run lie in between (written by their blocks)
      const int EOBk = 63 - __builtin_clzll(R.newm);
      int r = 0, prev = Ss - 1, fb = Ss;
      unsigned long long mm = R.newm | R.nzm;
      auto put_corr = [&](int lo, int hi) {   // correction bits of already-nonzero positions in [lo, hi)
        unsigned long long m = R.nzm & ~((1ull << lo) - 1ull);
        if (hi < 64) m &= (1ull << hi) - 1ull;
        while (m) { const int k = __builtin_ctzll(m); m &= m - 1; bw.put((unsigned)((R.corrm >> k) & 1ull), 1); }
      };
      while (mm) {
        const int k = __builtin_ctzll(mm);
        mm &= mm - 1;
        r += k - prev - 1;
        prev = k;
        while (r > 15 && k <= EOBk) {
          const unsigned e = s_tab[0xF0];
          bw.put(e & 0xFFFF, (int)(e >> 16));
          put_corr(fb, k);
          fb = k; r -= 16;
        }
        if ((R.nzm >> k) & 1ull) continue;
        const unsigned e = s_tab[(r << 4) + 1];
        bw.put_sym(e, (unsigned)((R.posm >> k) & 1ull), 1);
        put_corr(fb, k);
        fb = k + 1; r = 0;
      }
      bw.flush();
    }
  tails:
    if (R.tail_cnt > 0) {
      // my trailing correction bits belong to the run that the NEXT non-empty block (or the end of the scan) flushes:
      // they go behind that flush's EOBRUN symbol, at the distance T[me] - T[first block of the run]
      const int gj = cb + j;
      int start = flush_point ? gj : -2;
      if (start == -2) { const int p = pp_prev_ne(ne_bits, j); start = p >= 0 ? cb + p : ch.carry_p; }
      int nxt = pp_next_ne(ne_bits, j);
      nxt = nxt >= 0 ? cb + nxt : ch.next_after;
      const unsigned rel = T[gj] - (start >= 0 ? T[start] : 0u);
      unsigned where;
      int nextra;
      if (nxt >= 0) {
        const unsigned e = s_tab[eobrun_symbol(rung[nxt], &nextra)];
        where = base + offg[nxt] + (e >> 16) + (unsigned)nextra + rel;
      } else {
        const unsigned e = s_tab[eobrun_symbol(pe.info[pair].final_run, &nextra)];
        where = base + pe.totals[pair] + (e >> 16) + (unsigned)nextra + rel;
      }
      unsigned long long tail_bits = 0;
      {
        unsigned long long tm = R.tailm;
        while (tm) { const int k = __builtin_ctzll(tm); tm &= tm - 1; tail_bits = (tail_bits << 1) | ((R.corrm >> k) & 1ull); }
      }
      BitWriter bw;
      bw.init(stream, where);
      put_long(bw, (unsigned)(tail_bits >> 32), R.tail_cnt > 32 ? R.tail_cnt - 32 : 0);
      put_long(bw, (unsigned)tail_bits, R.tail_cnt > 32 ? 32 : R.tail_cnt);
      bw.flush();
    }
  }
}

// ---- first-pass AC scans from compact records: sizes and bits in ONE kernel ------------------------------------------
// The bits of a chunk follow from its symbol counts (k_pp_stats<.,1> keeps them per chunk, k_pp_runs / k_pp_resolve add the
// EOBRUN symbols of its flush points) and the code lengths, so the place of every chunk in the scan's stream is known
// without a pass over the blocks: k_pp_chunk_bits.  k_pp_emit then sizes the blocks of a chunk, scans the sizes inside the
// workgroup and writes the bits into a window of the stream kept in LDS -- no per-block length / offset arrays, no
// device-wide prefix sums, and the stream receives whole words (only the two boundary words of a chunk are ORed).
__global__ void __launch_bounds__(256)
k_pp_chunk_bits(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl,
                const MjhHuffTable *__restrict__ tabs, int slots_per_image, MjhProgPE pe)
{
  __shared__ unsigned s_bits[256];
  __shared__ unsigned sh[4];
  const int img = blockIdx.y, li = blockIdx.x, tid = threadIdx.x;
  const size_t pair = (size_t)li * gridDim.y + img;
  const MjhProgScan &sc = scans[scan_list[li]];
  if (prog_skip(sc, ctl + img)) return;
  const MjhComp cc = C.c[sc.comp[0]];
  const int nchunks = (cc.nblk + MJH_PSTAT_BLOCKS - 1) / MJH_PSTAT_BLOCKS;
  const MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
  // bits of one symbol tid: its code + the value bits (size nibble) / the run bits of an EOBRUN symbol (run nibble)
  const unsigned w = (unsigned)T0->ehufsi[tid] + (unsigned)((tid & 15) ? (tid & 15) : (tid == 0xF0 ? 0 : tid >> 4));
  const unsigned *ch = pe.chist + pair * pe.chunks_per_scan * 256;
  unsigned carry = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int nc = min(256, nchunks - c0);
    s_bits[tid] = 0;
    __syncthreads();
    for (int c = 0; c < nc; c++) {
      const unsigned v = ch[(size_t)(c0 + c) * 256 + tid];
      if (v) atomicAdd(&s_bits[c], v * w);
    }
    __syncthreads();
    unsigned tot;
    const unsigned ex = block_excl_scan_256(tid < nc ? s_bits[tid] : 0u, sh, &tot);
    if (tid < nc) pe.sums[pair * pe.chunks_per_scan + c0 + tid] = carry + ex;
    carry += tot;
  }
  if (tid == 0) pe.totals[pair] = carry;
}


#define PPE_WIN 4096          // words of the LDS window (16 KB): a chunk of 2048 blocks with up to 64 bits per block on average
template <bool LDSW, bool SKIPLOW = false>
__device__ __forceinline__ void pp_emit_rounds(unsigned *dst, unsigned bit0, unsigned cbase, int nb, int Ss, int Se, int Al, const int16_t *__restrict__ qc,
                                               size_t kstride, const unsigned long long *__restrict__ nzc, const uint16_t *__restrict__ run,
                                               const unsigned long long *fp_bits, const unsigned long long *rn_bits, const unsigned *s_tab, unsigned *sh)
{
  const int tid = threadIdx.x;
  const unsigned zrl = s_tab[0xF0];
  unsigned running = cbase;
#pragma unroll 1
  for (int i = 0; i < MJH_PSTAT_BLOCKS / 256; i++) {
    if (i * 256 >= nb) break;   // uniform
    const int j = i * 256 + tid;
    const bool has_own = j < nb && ((rn_bits[j >> 6] >> (j & 63)) & 1ull);
    const bool is_fp = j < nb && ((fp_bits[j >> 6] >> (j & 63)) & 1ull);
    // a wave of empty blocks (most waves of a sparse band) skips the sizing and takes part in the scan with zeros: EVERY wave
    // reaches the workgroup scan's barrier at the same call site
    const bool wave_empty = __builtin_amdgcn_ballot_w64(has_own || is_fp) == 0ull;
    const int jj = j < nb ? j : nb - 1;
    unsigned long long m = 0ull;
    unsigned own = 0, cnt = 0, fsym = 0;
    int nextra = 0;
    if (!wave_empty) {
      m = has_own ? nzc[jj] : 0ull;
      // the block's size: its own symbols ...
      int prev = Ss - 1;
      pp_band_nonzeros<SKIPLOW>(qc + jj, kstride, m, Ss, Se, has_own, [&](int k, int v) {
        const int a = (v < 0 ? -v : v) >> Al;
        if (a == 0) return;
        const int r = k - prev - 1;
        prev = k;
        const int nbits = bitlen((unsigned)a);
        own += (unsigned)(r >> 4) * (zrl >> 16) + (s_tab[((r & 15) << 4) + nbits] >> 16) + (unsigned)nbits;
      });
      // ... and the pending run that goes out in front of a flush point (emit_eobrun jcphuff.c:409)
      if (is_fp) {
        cnt = run[j];
        if (cnt) fsym = s_tab[eobrun_symbol(cnt, &nextra)];
      }
    }
    const unsigned blen = own + (cnt ? (fsym >> 16) + (unsigned)nextra : 0u);
    unsigned tot;
    const unsigned ex = block_excl_scan_256_1b(blen, sh, i & 1, &tot);
    if (wave_empty || __builtin_amdgcn_ballot_w64(blen != 0u) == 0ull) { running += tot; continue; }
    // the same walk again (the records are in the L1 now), this time with the place of every bit known
    BitSink<LDSW> bw;
    bw.init(dst, running + ex - bit0);
    if (cnt) {
      bw.put(fsym & 0xFFFF, (int)(fsym >> 16));
      if (nextra) bw.put(cnt & ((1u << nextra) - 1u), nextra);
    }
    {
      int prev = Ss - 1;
      pp_band_nonzeros<SKIPLOW>(qc + jj, kstride, m, Ss, Se, has_own && own != 0, [&](int k, int v) {
        const int a = (v < 0 ? -v : v) >> Al;
        if (a == 0) return;
        int r = k - prev - 1;
        prev = k;
        while (r > 15) { bw.put(zrl & 0xFFFF, (int)(zrl >> 16)); r -= 16; }
        const int nbits = bitlen((unsigned)a);
        const unsigned e = s_tab[(r << 4) + nbits];
        bw.put_sym(e, (unsigned)(v < 0 ? ~a : a), nbits);
      });
    }
    if (blen) bw.flush();
    running += tot;
  }
}

__global__ void __launch_bounds__(256)
k_pp_emit(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, const MjhProgCtl *__restrict__ ctl,
          const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
          unsigned *__restrict__ pool, size_t pool_words_per_image, MjhProgPE pe)
{
  constexpr bool SKIPLOW = true;
  __shared__ unsigned s_tab[256];   // size << 16 | code
  __shared__ unsigned s_win[PPE_WIN];
  __shared__ unsigned long long fp_bits[MJH_PSTAT_BLOCKS / 64];   // flush points: real non-empty blocks + forced-flush marks
  __shared__ unsigned long long rn_bits[MJH_PSTAT_BLOCKS / 64];   // real non-empty blocks
  __shared__ unsigned sh[8];
  const int img = blockIdx.z, li = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;   // the list starts with these scans
  const size_t pair = (size_t)li * gridDim.z + img;
  const int sidx = scan_list[li];
  const MjhProgScan &sc = scans[sidx];
  const MjhProgCtl *ct = ctl + img;
  if (prog_skip(sc, ct)) return;
  const MjhComp cc = C.c[sc.comp[0]];
  const int cb = chunk * MJH_PSTAT_BLOCKS;
  if (cb >= cc.nblk || ct->error) return;
  const int nb = min(MJH_PSTAT_BLOCKS, cc.nblk - cb);
  const bool last_chunk = cb + MJH_PSTAT_BLOCKS >= cc.nblk;
  const unsigned base = ct->scan_words_off[sidx] * 32u;
  const unsigned cbase = base + pe.sums[pair * pe.chunks_per_scan + chunk];
  const unsigned cend = base + (last_chunk ? pe.totals[pair] : pe.sums[pair * pe.chunks_per_scan + chunk + 1]);
  if (cend == cbase) return;        // uniform: the chunk has no bits
  const int Al = sc.al_sel == 1 ? ct->best_Al_luma : (sc.al_sel == 2 ? ct->best_Al_chroma : sc.Al);
  const MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
  const int16_t *qc = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + cb;
  const unsigned long long *nzc = nzmask + (size_t)img * C.total_real_blocks + cc.blk_off + cb;
  const uint16_t *run = pe.run16 + pair * pe.nblk_pad + cb;
  unsigned *stream = pool + (size_t)img * pool_words_per_image;
  const unsigned w0 = cbase >> 5, nw = ((cend - 1u) >> 5) - w0 + 1u;
  const bool window = nw <= PPE_WIN;
  s_tab[tid] = ((unsigned)T0->ehufsi[tid] << 16) | T0->ehufco[tid];
  if (tid < MJH_PSTAT_BLOCKS / 64) {
    fp_bits[tid] = pe.ne2_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
    rn_bits[tid] = pe.ne_bits[(pair * pe.chunks_per_scan + chunk) * (MJH_PSTAT_BLOCKS / 64) + tid];
  }
  if (window) for (unsigned w = tid; w < nw; w += 256) s_win[w] = 0u;
  __syncthreads();
  if (window) {
    pp_emit_rounds<true, SKIPLOW>(s_win, w0 << 5, cbase, nb, sc.Ss, sc.Se, Al, qc, (size_t)cc.kstride, nzc, run, fp_bits, rn_bits, s_tab, sh);
    __syncthreads();
    for (unsigned w = tid; w < nw; w += 256) {
      const unsigned v = s_win[w];
      if (!v) continue;                                  // (the pool is zeroed)
      if (w == 0 || w == nw - 1) atomicOr(&stream[w0 + w], v);   // shared with the neighbouring chunk / the end of the scan
      else stream[w0 + w] = v;
    }
  } else   // more than 64 bits per block on average: straight into the stream
    pp_emit_rounds<false, SKIPLOW>(stream, 0u, cbase, nb, sc.Ss, sc.Se, Al, qc, (size_t)cc.kstride, nzc, run, fp_bits, rn_bits, s_tab, sh);
}

__global__ void __launch_bounds__(64)
k_pp_finish(const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, MjhProgCtl *__restrict__ ctl,
            const MjhHuffTable *__restrict__ tabs, int slots_per_image, unsigned *__restrict__ pool, size_t pool_words_per_image,
            MjhProgPE pe)
{
  if (threadIdx.x != 0) return;
  const int img = blockIdx.y, li = blockIdx.x;
  const size_t pair = (size_t)li * gridDim.y + img;
  const int sidx = scan_list[li];
  const MjhProgScan &sc = scans[sidx];
  MjhProgCtl *ct = ctl + img;
  if (ct->error) return;
  if (prog_skip(sc, ct)) return;
  unsigned *stream = pool + (size_t)img * pool_words_per_image;
  const unsigned base = ct->scan_words_off[sidx] * 32u;
  unsigned cur = base + pe.totals[pair];
  if (sc.Ss != 0) {
    const MjhHuffTable *T0 = tabs + (size_t)img * slots_per_image + sc.slot[0];
    const unsigned fr = pe.info[pair].final_run;
    if (fr) {     // finish_pass_phuff: the run still pending at the end of the scan (+ its buffered correction bits, already in place)
      int nextra;
      const int sym = eobrun_symbol(fr, &nextra);
      BitWriter bw;
      bw.init(stream, cur);
      bw.put(T0->ehufco[sym], (int)T0->ehufsi[sym]);
      if (nextra) bw.put(fr & ((1u << nextra) - 1u), nextra);
      bw.flush();
      cur += (unsigned)T0->ehufsi[sym] + (unsigned)nextra + pe.info[pair].final_be;
    }
  }
  const unsigned tb = cur - base;
  if (tb & 7u) {   // flush_bits jcphuff.c:362-367: pad the last byte with 1-bits
    const unsigned padbits = 8u - (tb & 7u), bitpos = cur & 31u;
    atomicOr(&stream[cur >> 5], __builtin_bswap32(((1u << padbits) - 1u) << (32u - bitpos - padbits)));
  }
  if (ct->scan_bits[sidx] != tb) ct->error = 2;   // the size predicted from the statistics must be exact
  ct->scan_bits[sidx] = tb;
}

// exact size of every scan's bit stream from its statistics and code lengths, and its place in the
// pools.  One workgroup per image: wave w sizes the scans w, w+4, ... (lanes over the 256 symbols), then one
// thread hands out the pool space in list order.
__global__ void __launch_bounds__(256)
k_prog_alloc(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, int nlist,
             MjhProgCtl *__restrict__ ctl, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
             size_t pool_words_per_image, size_t out_bytes_per_image, int nimg)
{
  __shared__ unsigned long long s_bits[MJH_MAX_PROG_SCANS];
  const int img = blockIdx.x;
  if (img >= nimg) return;
  MjhProgCtl *ct = ctl + img;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int li = wave; li < nlist; li += 4) {
    const MjhProgScan &sc = scans[scan_list[li]];
    if (prog_skip(sc, ct)) { if (lane == 0) s_bits[li] = 0ull; continue; }   // not coded for this image: an empty stream
    unsigned long long bits = 0;
    if (sc.Ss == 0) {
      if (sc.Ah == 0) {
        for (int t = 0; t < 2; t++) {
          if (sc.slot[t] < 0) continue;
          const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + sc.slot[t];
          if (lane < 17) bits += (unsigned long long)T->counts[lane] * (T->ehufsi[lane] + lane);
        }
      } else if (lane == 0) {
        for (int ci = 0; ci < sc.ncomp; ci++) {
          const MjhComp &cc = C.c[sc.comp[ci]];
          bits += sc.ncomp > 1 ? (unsigned long long)cc.wpad * cc.hpad : (unsigned long long)cc.nblk;
        }
      }
    } else {
      const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + sc.slot[0];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int s = lane + 64 * j;
        const unsigned c = T->counts[s];
        const int extra = (s & 15) ? (s & 15) : (s == 0xF0 ? 0 : (s >> 4));
        bits += (unsigned long long)c * (T->ehufsi[s] + extra);
      }
      if (lane == 0) bits += T->counts[258];   // correction bits (refinement scans)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bits += __shfl_xor(bits, o, 64);
    if (lane == 0) s_bits[li] = bits + 23ull * (unsigned long long)sc.nrst;   // per RSTn: up to 7 pad bits + 16
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  ct->pool_zero_from = ct->pool_words_used;
  for (int li = 0; li < nlist; li++) {
    const int sidx = scan_list[li];
    const unsigned long long bits = s_bits[li];
    const unsigned words = (unsigned)((bits + 7) / 32) + 2;
    ct->scan_bits[sidx] = (unsigned)bits;
    ct->scan_words_off[sidx] = ct->pool_words_used;
    ct->scan_out_off[sidx] = ct->out_bytes_used;
    const unsigned long long pw = (unsigned long long)ct->pool_words_used + words;
    const unsigned long long ob = (unsigned long long)ct->out_bytes_used + 1280ull + 8ull * words;
    if (pw > pool_words_per_image || pw >= (1ull << 27) || ob > out_bytes_per_image || ob >= (1ull << 32)) { ct->error = 1; continue; }   // 32-bit bit / byte offsets
    ct->pool_words_used = (unsigned)pw;
    ct->out_bytes_used = (unsigned)ob;
  }
}

// the bit writers OR into their words: the part of the pool this phase handed out is zeroed (the pool is sized for the
// worst case, 8x the sequential stream with the scan search; what a phase uses is a few per cent of it)
__global__ void __launch_bounds__(256)
k_prog_zero_pool(const MjhProgCtl *__restrict__ ctl, unsigned *__restrict__ pool, size_t pool_words_per_image)
{
  const int img = blockIdx.y;
  const MjhProgCtl *ct = ctl + img;
  unsigned *p = pool + (size_t)img * pool_words_per_image;
  const unsigned lo = ct->pool_zero_from, hi = ct->pool_words_used;
  const unsigned lo4 = (lo + 3u) & ~3u, hi4 = hi & ~3u;
  if (blockIdx.x == 0 && threadIdx.x < 8) {   // unaligned ends
    const unsigned t = threadIdx.x;
    if (t < 4) { if (lo + t < (lo4 < hi ? lo4 : hi)) p[lo + t] = 0u; }
    else if (hi4 >= lo4 && hi4 + (t - 4) < hi) p[hi4 + (t - 4)] = 0u;
  }
  uint4 *p4 = reinterpret_cast<uint4 *>(p);
  for (unsigned i = lo4 / 4 + blockIdx.x * 256 + threadIdx.x; i < hi4 / 4; i += gridDim.x * 256) p4[i] = make_uint4(0u, 0u, 0u, 0u);
}

// scan header into the scan's buffer: [DQT + SOF for scan 0] DHT SOS (jcmaster.c:671-684,
// write_scan_header jcmarker.c:744-784)
__global__ void __launch_bounds__(64)
k_prog_header(const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, MjhProgCtl *__restrict__ ctl,
              const MjhHuffTable *__restrict__ tabs, int slots_per_image, const uint8_t *__restrict__ frame_hdr, int frame_hdr_len,
              int multi_dht, uint8_t *__restrict__ outpool, size_t out_bytes_per_image)
{
  const int img = blockIdx.y;
  const int sidx = scan_list[blockIdx.x];
  const MjhProgScan &sc = scans[sidx];
  MjhProgCtl *ct = ctl + img;
  const int lane = threadIdx.x;
  if (ct->error) return;
  if (prog_skip(sc, ct)) return;
  uint8_t *o = outpool + (size_t)img * out_bytes_per_image + ct->scan_out_off[sidx];
  int pos = 0;
  if (sc.frame_header) {
    for (int i = lane; i < frame_hdr_len; i += 64) o[i] = frame_hdr[i];
    pos = frame_hdr_len;
  }
  if (multi_dht) {     // emit_multi_dht jcmarker.c:293-401: one marker, possibly empty (DC refinement scans)
    int length = 2;
    for (int i = 0; i < sc.ndht; i++) length += (int)tabs[(size_t)img * slots_per_image + sc.dht_slot[i]].nsyms + 17;
    if (lane == 0) { o[pos] = 0xFF; o[pos + 1] = 0xC4; o[pos + 2] = (uint8_t)(length >> 8); o[pos + 3] = (uint8_t)length; }
    pos += 4;
  }
  for (int i = 0; i < sc.ndht; i++) {
    const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + sc.dht_slot[i];
    const int n = (int)T->nsyms;
    if (!multi_dht) {
      const int length = n + 2 + 1 + 16;
      if (lane == 0) { o[pos] = 0xFF; o[pos + 1] = 0xC4; o[pos + 2] = (uint8_t)(length >> 8); o[pos + 3] = (uint8_t)length; }
      pos += 4;
    }
    if (lane == 0) o[pos] = (uint8_t)sc.dht_id[i];
    if (lane < 16) o[pos + 1 + lane] = T->bits[lane + 1];
    for (int j = lane; j < n; j += 64) o[pos + 17 + j] = T->huffval[j];
    pos += 17 + n;
  }
  if (lane == 0) {     // emit_dri jcmarker.c:404-414 (only when the interval changes, :778-781), emit_sos :494-531
    const int Al = sc.al_sel == 1 ? ct->best_Al_luma : (sc.al_sel == 2 ? ct->best_Al_chroma : sc.Al);
    uint8_t *s = o + pos;
    int k = 0;
    if (sc.emit_dri) { s[k++] = 0xFF; s[k++] = 0xDD; s[k++] = 0; s[k++] = 4; s[k++] = (uint8_t)(sc.ri >> 8); s[k++] = (uint8_t)sc.ri; }
    s[k++] = 0xFF; s[k++] = 0xDA;
    const int len = 2 * sc.ncomp + 2 + 1 + 3;
    s[k++] = (uint8_t)(len >> 8); s[k++] = (uint8_t)len;
    s[k++] = (uint8_t)sc.ncomp;
    for (int i = 0; i < sc.ncomp; i++) { s[k++] = (uint8_t)sc.comp_id[i]; s[k++] = (uint8_t)((sc.td[i] << 4) + sc.ta[i]); }
    s[k++] = (uint8_t)sc.Ss; s[k++] = (uint8_t)sc.Se; s[k++] = (uint8_t)((sc.Ah << 4) + Al);
    ct->scan_hdr_len[sidx] = (unsigned)(pos + k);
  }
}

// byte stuffing of one scan's bit stream into its buffer: PROG_STUFF_SPLIT workgroups per (scan, image), each takes an
// equal share of the scan's words; k_prog_stuff_count leaves the number of stuffed zero bytes of every share behind so
// that k_prog_stuff knows where its share starts in the output
#define PROG_STUFF_SPLIT 8
#define PROG_CONCAT_PARTS 32
__device__ __forceinline__ bool prog_is_marker(const unsigned *mp, int nrst, unsigned pos)
{ // the 0xFF of an RSTn marker is not entropy-coded data: no zero byte behind it (binary search in the sorted positions)
  int lo = 0, hi = nrst - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const unsigned v = mp[mid];
    if (v == pos) return true;
    if (v < pos) lo = mid + 1; else hi = mid - 1;
  }
  return false;
}
__device__ __forceinline__ void prog_stuff_share(unsigned nwords, int part, unsigned &w0, unsigned &w1)
{
  const unsigned per = (((nwords + PROG_STUFF_SPLIT - 1) / PROG_STUFF_SPLIT) + 2047u) & ~2047u;   // whole 2048-word rounds
  w0 = min(nwords, per * (unsigned)part);
  w1 = min(nwords, per * (unsigned)(part + 1));
}

template <bool WRITE>
__global__ void __launch_bounds__(256)
k_prog_stuff(const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, MjhProgCtl *__restrict__ ctl,
             const unsigned *__restrict__ pool, size_t pool_words_per_image, uint8_t *__restrict__ outpool,
             size_t out_bytes_per_image, const unsigned *__restrict__ mpos_pool, int mpos_per_image, unsigned *__restrict__ ffsums)
{
  __shared__ unsigned sh[4];
  __shared__ unsigned s_out[WRITE ? STUFF_LDS_WORDS : 1];
  const int img = blockIdx.z, li = blockIdx.y, part = blockIdx.x;
  const int sidx = scan_list[li];
  MjhProgCtl *ct = ctl + img;
  const int nrst = scans[sidx].nrst;
  const unsigned *mp = mpos_pool + (size_t)img * mpos_per_image + scans[sidx].mpos_off;
  unsigned *fs = ffsums + ((size_t)img * gridDim.y + li) * PROG_STUFF_SPLIT;
  if (ct->error) return;
  if (prog_skip(scans[sidx], ct)) return;
  const unsigned nbytes = (ct->scan_bits[sidx] + 7) >> 3;
  const unsigned nwords = (nbytes + 3) >> 2;
  unsigned w0, w1;
  prog_stuff_share(nwords, part, w0, w1);
  const unsigned *p = pool + (size_t)img * pool_words_per_image + ct->scan_words_off[sidx];
  const unsigned hdr = ct->scan_hdr_len[sidx];
  uint8_t *o = outpool + (size_t)img * out_bytes_per_image + ct->scan_out_off[sidx] + hdr;
  unsigned carry = 0;
  if (WRITE) for (int q = 0; q < part; q++) carry += fs[q];
  for (unsigned cb = w0; cb < w1; cb += 2048) {
    const unsigned base = cb + threadIdx.x * 8;
    unsigned w[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      w[i] = base + i < w1 ? p[base + i] : 0u;
      s += ((w[i] & 0xFFu) == 0xFFu) + ((w[i] & 0xFF00u) == 0xFF00u) + ((w[i] & 0xFF0000u) == 0xFF0000u) + ((w[i] & 0xFF000000u) == 0xFF000000u);
      if (nrst) {
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (((w[i] >> (8 * b)) & 0xFFu) == 0xFFu && prog_is_marker(mp, nrst, (base + i) * 4 + b)) s--;
      }
    }
    unsigned tot;
    const unsigned ex = block_excl_scan_256(s, sh, &tot) + carry;
    if (WRITE) {
      unsigned keep = 0;
      if (nrst) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int b = 0; b < 4; b++)
            if (((w[i] >> (8 * b)) & 0xFFu) == 0xFFu && prog_is_marker(mp, nrst, (base + i) * 4 + b)) keep |= 1u << (4 * i + b);
      }
      const unsigned rend = min(min(w1, cb + 2048u) * 4u, nbytes);          // input bytes of the round: [cb * 4, rend)
      const int nvalid = base * 4u < rend ? (int)min(32u, rend - base * 4u) : 0;
      stuff_store_round(o, cb * 4u + carry, rend - cb * 4u + tot, base * 4u + ex, w, nvalid, keep, s_out);
    }
    carry += tot;
  }
  if (threadIdx.x == 0) {
    if (!WRITE) fs[part] = carry;
    else if (part == PROG_STUFF_SPLIT - 1) ct->scan_size[sidx] = hdr + nbytes + carry;
  }
}

// ---- scan search decisions: select_scans jcmaster.c:773-962 with the constants of
// jpeg_search_progression (jcparam.c:733-852).  Every candidate of a phase has been coded, but the
// decision consults sizes exactly as the reference would: it walks the candidates in order and
// stops where the reference would have stopped coding them.
__global__ void __launch_bounds__(64)
k_prog_select_al(MjhProgCtl *__restrict__ ctl, int ncomp, int nimg)
{
  const int img = blockIdx.x * 64 + threadIdx.x;
  if (img >= nimg) return;
  MjhProgCtl *ct = ctl + img;
  const unsigned *sz = ct->scan_size;
  // luma successive approximation: scans 1,2 (Al 0) then {3+3a, 4+3a, 5+3a} for a = 0..2
  unsigned long long best = 0;
  int bestAl = 0;
  for (int Al = 0; Al <= 3; Al++) {
    // next_scan_number = 3 + 3*Al: cost = the two band scans just coded + all refinements below
    unsigned long long cost = (unsigned long long)sz[3 * Al + 1] + sz[3 * Al + 2];
    for (int i = 0; i < Al; i++) cost += sz[3 + 3 * i];
    if (Al == 0 || cost < best) { best = cost; bestAl = Al; } else break;
  }
  ct->best_Al_luma = bestAl;
  ct->best_Al_chroma = 0;
  if (ncomp == 3) {
    const int base = 23 + 3;   // num_scans_luma + num_scans_chroma_dc
    best = 0; bestAl = 0;
    for (int Al = 0; Al <= 2; Al++) {
      // next_scan_number - base == 6*Al + 4: the four band scans just coded + refinements below
      unsigned long long cost = 0;
      for (int i = 0; i < 4; i++) cost += sz[base + 6 * Al + i];
      for (int i = 0; i < Al; i++) cost += (unsigned long long)sz[base + 4 + 6 * i] + sz[base + 5 + 6 * i];
      if (Al == 0 || cost < best) { best = cost; bestAl = Al; } else break;
    }
    ct->best_Al_chroma = bestAl;
  }
}

// between the sub-phases of phase A: does the luma successive-approximation search go on?  select_scans (jcmaster.c:799-818)
// compares, after the three scans of level Al, cost(Al) = the two band scans at Al + every refinement scan below with
// the best cost so far and stops coding candidates at the first level that is not cheaper; the candidates of level
// stage+1 (scans 3*stage+3 .. 3*stage+5, cond = stage) are therefore coded only where level `stage` improved.
__global__ void __launch_bounds__(64)
k_prog_select_stage(MjhProgCtl *__restrict__ ctl, int stage, int nimg)
{
  const int img = blockIdx.x * 64 + threadIdx.x;
  if (img >= nimg) return;
  MjhProgCtl *ct = ctl + img;
  if (ct->al_continue != stage - 1) return;          // the search stopped at an earlier level
  const unsigned *sz = ct->scan_size;
  unsigned long long best = (unsigned long long)sz[1] + sz[2];
  for (int Al = 1; Al <= stage; Al++) {
    unsigned long long cost = (unsigned long long)sz[3 * Al + 1] + sz[3 * Al + 2];
    for (int i = 0; i < Al; i++) cost += sz[3 + 3 * i];
    if (cost < best) best = cost; else return;       // (levels below `stage` improved, or al_continue would be smaller)
  }
  ct->al_continue = stage;
}

__global__ void __launch_bounds__(64)
k_prog_select_order(MjhProgCtl *__restrict__ ctl, int ncomp, int dc_scan_opt_mode, int nimg)
{
  const int img = blockIdx.x * 64 + threadIdx.x;
  if (img >= nimg) return;
  MjhProgCtl *ct = ctl + img;
  const unsigned *sz = ct->scan_size;
  const int lfs = 12;                    // luma_freq_split_scan_start
  const int nsl = 23;                    // num_scans_luma
  const int base = nsl + 3;              // + num_scans_chroma_dc
  const int cfs = nsl + 3 + (6 * 2 + 4); // chroma_freq_split_scan_start = 42
  // luma frequency split: scan 12 = full band, then pairs (13+2i, 14+2i), i = 0..4
  unsigned long long best = sz[lfs];
  int fsl = 0;
  for (int idx = 1; idx <= 5; idx++) {
    const unsigned long long cost = (unsigned long long)sz[lfs + 2 * idx - 1] + sz[lfs + 2 * idx];
    if (cost < best) { best = cost; fsl = idx; }
    if ((idx == 2 && fsl == 0) || (idx == 3 && fsl != 2) || (idx == 4 && fsl != 4)) break;
  }
  ct->best_fs_luma = fsl;
  int fsc = 0;
  if (ncomp == 3) {
    best = (unsigned long long)sz[cfs] + sz[cfs + 1];
    for (int idx = 1; idx <= 5; idx++) {
      unsigned long long cost = 0;
      for (int i = 0; i < 4; i++) cost += sz[cfs + 4 * idx - 2 + i];
      if (cost < best) { best = cost; fsc = idx; }
      if ((idx == 2 && fsc == 0) || (idx == 3 && fsc != 2) || (idx == 4 && fsc != 4)) break;
    }
  }
  ct->best_fs_chroma = fsc;
  // final order, jcmaster.c:898-956
  int n = 0;
  int *ord = ct->order;
  const int bl = ct->best_Al_luma, bc = ct->best_Al_chroma;
  const int min_Al = bl < bc ? bl : bc;
  ord[n++] = 0;
  if (ncomp == 3 && dc_scan_opt_mode != 0) {   // :836-838, :904-913: chroma DC interleaved (scan 23) or separate (24, 25)
    const bool interleave = (unsigned long long)sz[nsl] <= (unsigned long long)sz[nsl + 1] + sz[nsl + 2];
    if (interleave && dc_scan_opt_mode != 1) ord[n++] = nsl;
    else { ord[n++] = nsl + 1; ord[n++] = nsl + 2; }
  }
  if (fsl == 0) ord[n++] = lfs;
  else { ord[n++] = lfs + 2 * (fsl - 1) + 1; ord[n++] = lfs + 2 * (fsl - 1) + 2; }
  for (int Al = bl - 1; Al >= min_Al; Al--) ord[n++] = 3 + 3 * Al;
  if (ncomp == 3) {
    if (fsc == 0) { ord[n++] = cfs; ord[n++] = cfs + 1; }
    else for (int i = 2; i <= 5; i++) ord[n++] = cfs + 4 * (fsc - 1) + i;
    for (int Al = bc - 1; Al >= min_Al; Al--) { ord[n++] = base + 6 * Al + 4; ord[n++] = base + 6 * Al + 5; }
  }
  for (int Al = min_Al - 1; Al >= 0; Al--) {
    ord[n++] = 3 + 3 * Al;
    if (ncomp == 3) { ord[n++] = base + 6 * Al + 4; ord[n++] = base + 6 * Al + 5; }
  }
  ct->norder = n;
}

// final file: SOI (+APP0), the chosen scan buffers in order, EOI
__global__ void __launch_bounds__(256)
k_prog_concat(const MjhProgCtl *__restrict__ ctl, const uint8_t *__restrict__ file_hdr, int file_hdr_len,
              const uint8_t *__restrict__ outpool, size_t out_bytes_per_image, uint8_t *__restrict__ out, size_t out_stride,
              unsigned *__restrict__ sizes)
{
  // PROG_CONCAT_PARTS workgroups per image, each copies its share of the file's byte range (the layout is a walk over at
  // most MJH_MAX_PROG_SCANS sizes, repeated by every workgroup)
  const int img = blockIdx.y, part = blockIdx.x;
  const MjhProgCtl *ct = ctl + img;
  uint8_t *o = out + (size_t)img * out_stride;
  if (ct->error) { if (threadIdx.x == 0 && part == 0) sizes[img] = 0; return; }
  size_t total = file_hdr_len;
  for (int s = 0; s < ct->norder; s++) total += ct->scan_size[ct->order[s]];
  const size_t per = ((total + gridDim.x - 1) / gridDim.x + 255) & ~(size_t)255;
  const size_t lo = (size_t)part * per, hi = lo + per < total ? lo + per : total;
  if (part == 0) {
    for (int i = threadIdx.x; i < file_hdr_len; i += 256) o[i] = file_hdr[i];
    if (threadIdx.x == 0) { o[total] = 0xFF; o[total + 1] = 0xD9; sizes[img] = (unsigned)(total + 2); }
  }
  size_t pos = file_hdr_len;
  for (int s = 0; s < ct->norder && pos < hi; s++) {
    const int sidx = ct->order[s];
    const size_t n = ct->scan_size[sidx];
    if (pos + n > lo) {
      const uint8_t *src = outpool + (size_t)img * out_bytes_per_image + ct->scan_out_off[sidx];
      const size_t a = lo > pos ? lo - pos : 0, b = hi - pos < n ? hi - pos : n;
      for (size_t i = a + threadIdx.x; i < b; i += 256) o[pos + i] = src[i];
    }
    pos += n;
  }
}

__global__ void __launch_bounds__(64)
k_prog_reset(MjhProgCtl *__restrict__ ctl, int nscans, int nimg)
{
  const int img = blockIdx.x * 64 + threadIdx.x;
  if (img >= nimg) return;
  MjhProgCtl *ct = ctl + img;
  ct->best_Al_luma = ct->best_Al_chroma = ct->best_fs_luma = ct->best_fs_chroma = 0; ct->al_continue = 0;
  ct->pool_words_used = 0; ct->pool_zero_from = 0; ct->out_bytes_used = 0; ct->error = 0;
  ct->norder = nscans;
  for (int i = 0; i < nscans; i++) ct->order[i] = i;
  for (int i = 0; i < MJH_MAX_PROG_SCANS; i++) { ct->scan_us[0][i] = 0; ct->scan_us[1][i] = 0; }
}

// =============================================================================================
// launch wrappers
// =============================================================================================
void mjh_launch_prog_reset(void *ctl, int nscans, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_prog_reset, dim3((n + 63) / 64), dim3(64), 0, s, (MjhProgCtl *)ctl, nscans, n);
}

void mjh_launch_prog_stats(const MjhConst &C, const void *scans, const int *list, int nlist, void *ctl, const void *q,
                           MjhHuffTable *tabs, int spi, unsigned *mpos, int mpos_per_image, int n, hipStream_t s)
{
  hipLaunchKernelGGL((k_prog_scan<0>), dim3(n, nlist), dim3(64 * PROG_WAVES(0)), 0, s, C, (const MjhProgScan *)scans, list, (MjhProgCtl *)ctl,
                     (const int16_t *)q, tabs, spi, (unsigned *)nullptr, (size_t)0, mpos, mpos_per_image, (const MjhProgPair *)nullptr);
}

// statistics of the parallel chain (every scan of the list has no restart interval)
void mjh_launch_prog_stats_par(const MjhConst &C, const void *scans, const int *list, int nlist, void *ctl, const void *q,
                               MjhHuffTable *tabs, int spi, const MjhProgPE &pe, bool any_refine, const unsigned long long *nzmask, int n, hipStream_t s, int nacf)
{
  if (nlist <= 0) return;
  const dim3 gchunks(pe.chunks_per_scan, nlist, n), gpairs(nlist, n);
  hipLaunchKernelGGL(k_pp_init, dim3((nlist * n + 63) / 64), dim3(64), 0, s, pe, nlist * n);
  const void *sv = scans; const int16_t *qv = (const int16_t *)q;
  if (nzmask) {         // compact records: one kernel per kind of scan; the list starts with its nacf first-pass AC scans
    if (nacf > 0) hipLaunchKernelGGL((k_pp_stats<true, 1>), dim3(pe.chunks_per_scan, nacf, n), dim3(256), 0, s, C, (const MjhProgScan *)sv, list, (const MjhProgCtl *)ctl, qv,
                                     nzmask, tabs, spi, pe, 0);
    if (nlist > nacf) hipLaunchKernelGGL((k_pp_stats<true, 2>), dim3(pe.chunks_per_scan, nlist - nacf, n), dim3(256), 0, s, C, (const MjhProgScan *)sv, list, (const MjhProgCtl *)ctl, qv,
                                         nzmask, tabs, spi, pe, nacf);
  } else if (nzmask) hipLaunchKernelGGL((k_pp_stats<true, 0>), gchunks, dim3(256), 0, s, C, (const MjhProgScan *)sv, list, (const MjhProgCtl *)ctl, qv,
                                 nzmask, tabs, spi, pe, 0);
  else hipLaunchKernelGGL((k_pp_stats<false, 0>), gchunks, dim3(256), 0, s, C, (const MjhProgScan *)sv, list, (const MjhProgCtl *)ctl, qv,
                          nzmask, tabs, spi, pe, 0);
  if (any_refine) {   // prefix sums of the trailing correction bits: only refinement scans have any, and they sit behind the first-pass AC scans
    const size_t po = nzmask ? (size_t)nacf * n : 0;
    mjh_launch_scan16(pe.tail16 + po * pe.nblk_pad, pe.nblk_pad, pe.tsums + po * pe.chunks_per_scan, pe.chunks_per_scan, pe.ttotals + po,
                      pe.T32 + po * pe.nblk_pad, nlist * n - (int)po, s);
  }
  hipLaunchKernelGGL(k_pp_carry, gpairs, dim3(64), 0, s, C, (const MjhProgScan *)scans, list, (const MjhProgCtl *)ctl, pe);
  hipLaunchKernelGGL(k_pp_cuts, gchunks, dim3(256), 0, s, C, (const MjhProgScan *)scans, list, (const MjhProgCtl *)ctl, pe);
  hipLaunchKernelGGL(k_pp_runs, gchunks, dim3(256), 0, s, C, (const MjhProgScan *)scans, list, (const MjhProgCtl *)ctl, tabs, spi, pe);
  hipLaunchKernelGGL(k_pp_resolve, gpairs, dim3(64), 0, s, C, (const MjhProgScan *)scans, list, (const MjhProgCtl *)ctl, tabs, spi, pe);
}

void mjh_launch_prog_encode(const MjhConst &C, const void *scans, const int *list, int nlist, const int *seq_list, int nseq,
                            const int *par_list, int npar, const MjhProgPE &pe, void *ctl, const void *q,
                            MjhHuffTable *tabs, int spi, unsigned *pool, size_t pool_words, const void *frame_hdr, int frame_hdr_len,
                            int multi_dht, void *outpool, size_t out_bytes, unsigned *mpos, int mpos_per_image, unsigned *ffsums,
                            const unsigned long long *nzmask, int n, hipStream_t s,
                            hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, int nacf)
{
  hipLaunchKernelGGL(k_prog_alloc, dim3(n), dim3(256), 0, s, C, (const MjhProgScan *)scans, list, nlist, (MjhProgCtl *)ctl,
                     (const MjhHuffTable *)tabs, spi, pool_words, out_bytes, n);
  hipLaunchKernelGGL(k_prog_zero_pool, dim3(128, n), dim3(256), 0, s, (const MjhProgCtl *)ctl, pool, pool_words);
  hipLaunchKernelGGL(k_prog_header, dim3(nlist, n), dim3(64), 0, s, (const MjhProgScan *)scans, list, (MjhProgCtl *)ctl,
                     (const MjhHuffTable *)tabs, spi, (const uint8_t *)frame_hdr, frame_hdr_len, multi_dht, (uint8_t *)outpool, out_bytes);
  // the sequential walks (scans with restart intervals: a few long workgroups) and the parallel chain write disjoint scan
  // streams: the chain runs on the side stream underneath them
  const bool both = nseq > 0 && npar > 0;
  hipStream_t ps = s;
  if (both) {
    (void)hipEventRecord(ev_fork, s);
    (void)hipStreamWaitEvent(side, ev_fork, 0);
    ps = side;
  }
  if (npar > 0) {
    const dim3 gchunks(pe.chunks_per_scan, npar, n), gpairs(npar, n);
    const bool splitk = nzmask != nullptr;       // compact records: one kernel per kind of scan
    const MjhProgScan *sv = (const MjhProgScan *)scans; const MjhProgCtl *cv = (const MjhProgCtl *)ctl; const int16_t *qv = (const int16_t *)q;
    const MjhHuffTable *tv = (const MjhHuffTable *)tabs;
    const dim3 gacf(pe.chunks_per_scan, nacf, n), goth(pe.chunks_per_scan, npar - nacf, n);   // the list starts with its nacf first-pass AC scans
    if (splitk) {
      // first-pass AC scans (the list starts with them): k_pp_chunk_bits + k_pp_emit; the other scans: sizes, prefix sums, bits
      const size_t po = (size_t)nacf * n;      // their pairs come first
      if (nacf > 0) {
        hipLaunchKernelGGL(k_pp_chunk_bits, dim3(nacf, n), dim3(256), 0, ps, C, sv, par_list, cv, tv, spi, pe);
        hipLaunchKernelGGL(k_pp_emit, gacf, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pool, pool_words, pe);
      }
      if (npar > nacf) {
        hipLaunchKernelGGL((k_pp_len<true, 2>), goth, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pe, nacf);
        mjh_launch_scan16(pe.len16 + po * pe.nblk_pad, pe.nblk_pad, pe.sums + po * pe.chunks_per_scan, pe.chunks_per_scan, pe.totals + po,
                          pe.off32 + po * pe.nblk_pad, (npar - nacf) * n, ps);
        hipLaunchKernelGGL((k_pp_write<true, 2>), goth, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pool, pool_words, pe, nacf);
      }
    } else {
      if (nzmask) hipLaunchKernelGGL((k_pp_len<true, 0>), gchunks, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pe, 0);
      else hipLaunchKernelGGL((k_pp_len<false, 0>), gchunks, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pe, 0);
      mjh_launch_scan16(pe.len16, pe.nblk_pad, pe.sums, pe.chunks_per_scan, pe.totals, pe.off32, npar * n, ps);
      if (nzmask) hipLaunchKernelGGL((k_pp_write<true, 0>), gchunks, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pool, pool_words, pe, 0);
      else hipLaunchKernelGGL((k_pp_write<false, 0>), gchunks, dim3(256), 0, ps, C, sv, par_list, cv, qv, nzmask, tv, spi, pool, pool_words, pe, 0);
    }
    hipLaunchKernelGGL(k_pp_finish, gpairs, dim3(64), 0, ps, (const MjhProgScan *)scans, par_list, (MjhProgCtl *)ctl, (const MjhHuffTable *)tabs, spi,
                       pool, pool_words, pe);
  }
  if (both) (void)hipEventRecord(ev_join, side);
  if (nseq > 0)     // scans with restart intervals: the sequential walk
    hipLaunchKernelGGL((k_prog_scan<1>), dim3(n, nseq), dim3(64 * PROG_WAVES(1)), 0, s, C, (const MjhProgScan *)scans, seq_list, (MjhProgCtl *)ctl,
                       (const int16_t *)q, tabs, spi, pool, pool_words, mpos, mpos_per_image, (const MjhProgPair *)nullptr);
  if (both) (void)hipStreamWaitEvent(s, ev_join, 0);
  hipLaunchKernelGGL((k_prog_stuff<false>), dim3(PROG_STUFF_SPLIT, nlist, n), dim3(256), 0, s, (const MjhProgScan *)scans, list, (MjhProgCtl *)ctl, (const unsigned *)pool,
                     pool_words, (uint8_t *)outpool, out_bytes, (const unsigned *)mpos, mpos_per_image, ffsums);
  hipLaunchKernelGGL((k_prog_stuff<true>), dim3(PROG_STUFF_SPLIT, nlist, n), dim3(256), 0, s, (const MjhProgScan *)scans, list, (MjhProgCtl *)ctl, (const unsigned *)pool,
                     pool_words, (uint8_t *)outpool, out_bytes, (const unsigned *)mpos, mpos_per_image, ffsums);
}

void mjh_launch_prog_select(void *ctl, int ncomp, int phase, int dc_scan_opt_mode, int n, hipStream_t s)
{
  // phases of the scan search: 0, 1 = sub-phases A1, A2 (is the next luma level worth coding?), 2 = A3 (Al decisions), 3 = B (order)
  if (phase < 2) hipLaunchKernelGGL(k_prog_select_stage, dim3((n + 63) / 64), dim3(64), 0, s, (MjhProgCtl *)ctl, phase + 1, n);
  else if (phase == 2) hipLaunchKernelGGL(k_prog_select_al, dim3((n + 63) / 64), dim3(64), 0, s, (MjhProgCtl *)ctl, ncomp, n);
  else hipLaunchKernelGGL(k_prog_select_order, dim3((n + 63) / 64), dim3(64), 0, s, (MjhProgCtl *)ctl, ncomp, dc_scan_opt_mode, n);
}

void mjh_launch_prog_concat(const void *ctl, const void *file_hdr, int file_hdr_len, const void *outpool, size_t out_bytes,
                            void *out, size_t out_stride, unsigned *sizes, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_prog_concat, dim3(PROG_CONCAT_PARTS, n), dim3(256), 0, s, (const MjhProgCtl *)ctl, (const uint8_t *)file_hdr, file_hdr_len,
                     (const uint8_t *)outpool, out_bytes, (uint8_t *)out, out_stride, sizes);
}
