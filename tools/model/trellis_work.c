/* Work model of the lane-autonomous AC trellis walk (tools/model_trellis.py drives it): for every block the number of
 * queue records (positions with a non-zero conventionally quantized value) and the number of pair-steps the walk of
 * mjh_kernels.hip takes (newest-first, two predecessors per step, early exit on gap > best).  Double arithmetic: a
 * WORK model, not a parity tool. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int bitlen(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

/* coef: [nblk][64] natural order raw (x8) coefficients; q: natural-order table; zz: zig-zag->natural; ehufsi[256] */
void trellis_work(const int16_t *coef, int nblk, const uint16_t *q, const uint8_t *ehufsi, double s1, double s2, const int *zz,
                  int *nq_out, int *steps_out, int *evals_out, int *steps1_out, double *lambda_out)
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
    int nq = 0, steps = 0, evals = 0, steps1 = 0;
    for (int k = 1; k < 64; k++) {
      const int x = abs(c[zz[k]]), qq = q[zz[k]], dq = 8 * qq;
      const double lt = 1.0 / ((double)qq * qq);
      const double azd_prev = azd_run;
      azd_run += (double)x * x * lambda * lt;
      if (x + dq / 2 < dq) continue;
      int qval = (x + dq / 2) / dq;
      if (qval > 1023) qval = 1023;
      nq++;
      const int ncd = bitlen(qval);
      double best = 1e38; int bestp = -1;
      int e = nlive - 1, st = 0;
      while (e >= 0) {
        /* one pair step: entries e, e-1 */
        double gap_last = 0;
        for (int t = 0; t < 2 && e >= 0; t++, e--) {
          const int run = k - 1 - livepos[e];
          const double gap = azd_prev - azd[e];
          gap_last = gap;
          for (int cd = 0; cd < ncd; cd++) {
            const int cand = cd < ncd - 1 ? (2 << cd) - 1 : qval;
            const int sz = ehufsi[((run & 15) << 4) + cd + 1];
            if (!sz) continue;
            const double rate = sz + cd + 1 + (run >> 4) * ehufsi[0xF0];
            const double d = (double)(cand * dq - x) * (cand * dq - x) * lambda * lt;
            const double cost = rate + d + gap + acc[e];
            if (cost < best || (cost == best && bestp >= 0)) { best = cost; bestp = livepos[e]; }
          }
          evals++;
        }
        st++;
        if (gap_last > best) break;
      }
      steps += st;
      steps1 += (st * 2 > 0);
      if (bestp >= 0) { livepos[nlive] = k; azd[nlive] = azd_run; acc[nlive] = best; nlive++; }
    }
    lambda_out[b] = lambda; nq_out[b] = nq; steps_out[b] = steps; evals_out[b] = evals; steps1_out[b] = steps1;
  }
}
