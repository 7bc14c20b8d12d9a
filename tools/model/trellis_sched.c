/* Scheduling model of the tile-sorted AC trellis walk (tools/model_sched.py drives it; no GPU, a WORK model in double
 * arithmetic like trellis_work.c): per block the pair-steps of EVERY queue record, then the instruction-issue cost of a wave
 * under different policies for the two phases of the walk loop (A = one pair step, B = commit the entry + set the next record
 * up).  Today every iteration runs A for all active lanes and then B for the lanes whose scan has ended -- with 64 lanes some
 * lane nearly always needs B, so B's ~95 instructions are issued almost every iteration for a handful of lanes. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int g_tight = 0;   /* 1: stop the scan once gap + (smallest possible rate + distortion of any candidate) exceeds the best cost */
static int bitlen(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

/* steps_out: [nblk][64] pair-steps of record r (0 = no such record); ncd4_out[nblk][64]: 1 when the record has more than two candidates */
void trellis_records(const int16_t *coef, int nblk, const uint16_t *q, const uint8_t *ehufsi, double s1, double s2, const int *zz,
                     int *nq_out, uint8_t *steps_out, uint8_t *ncd4_out, int *qmax_out)
{
  for (int b = 0; b < nblk; b++) {
    const int16_t *c = coef + (size_t)b * 64;
    double norm = 0;
    for (int i = 1; i < 64; i++) norm += (double)c[i] * c[i];
    norm /= 63.0;
    const double lambda = pow(2.0, s1) / (pow(2.0, s2) + norm);
    double azd[64], acc[64];
    int livepos[64], nlive = 1;
    livepos[0] = 0; azd[0] = 0; acc[0] = 0;
    double azd_run = 0;
    int nq = 0, qmax = 0;
    memset(steps_out + (size_t)b * 64, 0, 64); memset(ncd4_out + (size_t)b * 64, 0, 64);
    for (int k = 1; k < 64; k++) {
      const int x = abs(c[zz[k]]), qq = q[zz[k]], dq = 8 * qq;
      const double lt = 1.0 / ((double)qq * qq);
      const double azd_prev = azd_run;
      azd_run += (double)x * x * lambda * lt;
      if (x + dq / 2 < dq) continue;
      int qval = (x + dq / 2) / dq;
      if (qval > 1023) qval = 1023;
      if (qval > qmax) qmax = qval;
      const int ncd = bitlen(qval);
      double best = 1e38; int bestp = -1;
      int e = nlive - 1, st = 0;
      double lb = 0;
      if (g_tight) {
        lb = 1e38;
        for (int cd = 0; cd < ncd && cd < 4; cd++) {
          const int cand = cd < ncd - 1 ? (2 << cd) - 1 : qval;
          double rmin = 1e38;
          for (int run = 0; run < 16; run++) { const int sz = ehufsi[(run << 4) + cd + 1]; if (sz && sz + cd + 1 < rmin) rmin = sz + cd + 1; }
          const double d = (double)(cand * dq - x) * (cand * dq - x) * lambda * lt;
          if (rmin + d < lb) lb = rmin + d;
        }
      }
      while (e >= 0) {
        double gap_last = 0;
        for (int t = 0; t < 2 && e >= 0; t++, e--) {
          const int run = k - 1 - livepos[e];
          const double gap = azd_prev - azd[e];
          gap_last = gap;
          for (int cd = 0; cd < ncd && cd < 4; cd++) {
            const int cand = cd < ncd - 1 ? (2 << cd) - 1 : qval;
            const int sz = ehufsi[((run & 15) << 4) + cd + 1];
            if (!sz) continue;
            const double rate = sz + cd + 1 + (run >> 4) * ehufsi[0xF0];
            const double d = (double)(cand * dq - x) * (cand * dq - x) * lambda * lt;
            const double cost = rate + d + gap + acc[e];
            if (cost < best || (cost == best && bestp >= 0)) { best = cost; bestp = livepos[e]; }
          }
        }
        st++;
        if (gap_last + lb > best) break;
      }
      steps_out[(size_t)b * 64 + nq] = (uint8_t)st;
      ncd4_out[(size_t)b * 64 + nq] = (uint8_t)(ncd > 2);
      nq++;
      if (bestp >= 0) { livepos[nlive] = k; azd[nlive] = azd_run; acc[nlive] = best; nlive++; }
    }
    nq_out[b] = nq; qmax_out[b] = qmax;
  }
}

/* One wave of 64 lanes; lane l walks block idx[l] (-1: idle lane).  policy 0 = today (A for every active lane, then B for the lanes
 * that finished a record, every iteration); policy 1 = one phase per iteration: B when at least `thr` lanes wait for it or no lane
 * can run A, else A.  Costs in issued wave-instructions: cA2 / cA4 (pair step with <= 2 / > 2 candidates in the wave), cB, cTop
 * (loop overhead; cLook = the two ds_bpermute lookups, today in every iteration, with policy 1 only in B iterations).
 * Returns the issue cost; *lane_instr accumulates active-lane x instruction products (for the lanes-per-instruction figure). */
double wave_cost(const int *idx, const int *nq, const uint8_t *steps, const uint8_t *ncd4, int qn, int policy, int thr,
                 double cA2, double cA4, double cB, double cTop, double cLook, double *lane_instr)
{
  int rec[64], left[64], waitB[64], act[64];
  double cost = 0, li = 0;
  int nact = 0;
  for (int l = 0; l < 64; l++) {
    const int b = idx[l];
    act[l] = b >= 0 && nq[b] > 0 && nq[b] <= qn;
    rec[l] = 0; waitB[l] = 0;
    left[l] = act[l] ? steps[(size_t)b * 64] : 0;
    nact += act[l];
  }
  /* (the first record's setup runs once for all lanes in front of the loop: cB/2, uniform) */
  if (nact) { cost += cB * 0.6; li += cB * 0.6 * nact; }
  while (nact) {
    int nA = 0, nB = 0, any4 = 0;
    for (int l = 0; l < 64; l++) if (act[l]) { if (waitB[l]) nB++; else { nA++; any4 |= ncd4[(size_t)idx[l] * 64 + rec[l]]; } }
    const double cA = any4 ? cA4 : cA2;
    if (policy == 0) {
      /* A for all active lanes (none waits: B follows in the same iteration) */
      cost += cTop + cLook + cA; li += (cTop + cLook + cA) * nact;
      int nfin = 0;
      for (int l = 0; l < 64; l++) if (act[l]) { if (--left[l] <= 0) { waitB[l] = 1; nfin++; } }
      if (nfin) {
        cost += cB; li += cB * nfin;
        for (int l = 0; l < 64; l++) if (act[l] && waitB[l]) {
          waitB[l] = 0; rec[l]++;
          if (rec[l] >= nq[idx[l]]) { act[l] = 0; nact--; } else left[l] = steps[(size_t)idx[l] * 64 + rec[l]];
        }
      }
    } else {
      if (nA == 0 || (policy == 1 ? nB >= thr : nB * 100 >= thr * nA)) {   /* policy 2: B when nB / nA >= thr % */
        cost += cTop + cLook + cB; li += (cTop + cLook) * nact + cB * nB;
        for (int l = 0; l < 64; l++) if (act[l] && waitB[l]) {
          waitB[l] = 0; rec[l]++;
          if (rec[l] >= nq[idx[l]]) { act[l] = 0; nact--; } else left[l] = steps[(size_t)idx[l] * 64 + rec[l]];
        }
      } else {
        cost += cTop + cA; li += cTop * nact + cA * nA;
        for (int l = 0; l < 64; l++) if (act[l] && !waitB[l]) { if (--left[l] <= 0) waitB[l] = 1; }
      }
    }
  }
  *lane_instr += li;
  return cost;
}
