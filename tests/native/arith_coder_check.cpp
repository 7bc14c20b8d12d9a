// TEST INFRASTRUCTURE.  The arithmetic coder of the product (mozjpeg_amd/csrc/mjh_arith_coder.h, compiled here for the host:
// a "vector register" is an array of 64 ints) against a plain restatement of jcarith.c with the reference's own bin
// numbering (flat arrays of state bytes, bit-by-bit renormalisation), on random blocks:
//   whole blocks (sequential files, jcarith.c:690-822), DC first / refinement scans (:355-455, :555-590), AC first /
//   refinement scans with random bands and point transforms (:456-552, :596-687), restarts (:321-352).
// Compared after every scan: the bytes, the coder's registers, and every statistics bin (state byte and the Qe cached
// next to it).  Exit status 0 = identical everywhere.  usage: arith_coder_check [scans] [seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define MJH_ARI_HOST 1
#include "../../mozjpeg_amd/csrc/mjh_arith_coder.h"

// conditioning of tables 0 / 1 (cinfo->arith_dc_L / arith_dc_U / arith_ac_K), redrawn for every scan of the check
static int CHK_DC_L[2] = { 0, 0 }, CHK_DC_U[2] = { 1, 1 }, CHK_AC_K[2] = { 5, 5 };

// ---- plain restatement ------------------------------------------------------------------------------------------------
struct Ref {
  unsigned char ac[2][256], dc[2][64], fixed_bin;
  long c, a;
  int sc, zc, ct, buffer;
  int last_dc[4], ctx[4];
  std::vector<unsigned char> out;
  void byte(int v) { out.push_back((unsigned char)v); }
  void reset_coder() { c = 0; a = 0x10000L; sc = 0; zc = 0; ct = 11; buffer = -1; }
  void encode(unsigned char *st, int val)      // arith_encode jcarith.c:229-320
  {
    int sv = *st;
    const long qe = mjh_ari_qe[sv & 0x7F];
    const int nl = mjh_ari_nlps[sv & 0x7F], nm = mjh_ari_nmps[sv & 0x7F];
    a -= qe;
    if (val != (sv >> 7)) {
      if (a >= qe) { c += a; a = qe; }
      *st = (unsigned char)((sv & 0x80) ^ nl);
    } else {
      if (a >= 0x8000L) return;
      if (a < qe) { c += a; a = qe; }
      *st = (unsigned char)((sv & 0x80) ^ nm);
    }
    do {
      a <<= 1;
      c <<= 1;
      if (--ct == 0) {
        long temp = c >> 19;
        if (temp > 0xFF) {
          if (buffer >= 0) {
            if (zc) do byte(0x00); while (--zc);
            byte(buffer + 1);
            if (buffer + 1 == 0xFF) byte(0x00);
          }
          zc += sc;
          sc = 0;
          buffer = (int)(temp & 0xFF);
        } else if (temp == 0xFF) {
          ++sc;
        } else {
          if (buffer == 0) ++zc;
          else if (buffer >= 0) {
            if (zc) do byte(0x00); while (--zc);
            byte(buffer);
          }
          if (sc) {
            if (zc) do byte(0x00); while (--zc);
            do { byte(0xFF); byte(0x00); } while (--sc);
          }
          buffer = (int)(temp & 0xFF);
        }
        c &= 0x7FFFFL;
        ct += 8;
      }
    } while (a < 0x8000L);
  }
  void finish()                                 // finish_pass jcarith.c:142-203
  {
    long temp;
    if ((temp = (a - 1 + c) & 0xFFFF0000L) < c) c = temp + 0x8000L; else c = temp;
    c <<= ct;
    if (c & 0xF8000000L) {
      if (buffer >= 0) {
        if (zc) do byte(0x00); while (--zc);
        byte(buffer + 1);
        if (buffer + 1 == 0xFF) byte(0x00);
      }
      zc += sc;
      sc = 0;
    } else {
      if (buffer == 0) ++zc;
      else if (buffer >= 0) {
        if (zc) do byte(0x00); while (--zc);
        byte(buffer);
      }
      if (sc) {
        if (zc) do byte(0x00); while (--zc);
        do { byte(0xFF); byte(0x00); } while (--sc);
      }
    }
    if (c & 0x7FFF800L) {
      if (zc) do byte(0x00); while (--zc);
      byte((int)((c >> 19) & 0xFF));
      if (((c >> 19) & 0xFF) == 0xFF) byte(0x00);
      if (c & 0x7F800L) {
        byte((int)((c >> 11) & 0xFF));
        if (((c >> 11) & 0xFF) == 0xFF) byte(0x00);
      }
    }
  }
  void dc_diff(int tbl, int ci, int value)       // jcarith.c:402-448
  {
    unsigned char *base = dc[tbl], *st = base + ctx[ci];
    int v = value - last_dc[ci], m, v2;
    if (v == 0) { encode(st, 0); ctx[ci] = 0; return; }
    last_dc[ci] = value;
    encode(st, 1);
    if (v > 0) { encode(st + 1, 0); st += 2; ctx[ci] = 4; }
    else { v = -v; encode(st + 1, 1); st += 3; ctx[ci] = 8; }
    m = 0;
    if (v -= 1) {
      encode(st, 1);
      m = 1;
      v2 = v;
      st = base + 20;
      while (v2 >>= 1) { encode(st, 1); m <<= 1; st += 1; }
    }
    encode(st, 0);
    if (m < (int)((1L << CHK_DC_L[tbl]) >> 1)) ctx[ci] = 0;
    else if (m > (int)((1L << CHK_DC_U[tbl]) >> 1)) ctx[ci] += 8;
    st += 14;
    while (m >>= 1) encode(st, (m & v) ? 1 : 0);
  }
  void ac_first(int tbl, const short *blk, int Ss, int Se, int Al)      // jcarith.c:456-552
  {
    unsigned char *base = ac[tbl], *st;
    int k, ke, v, v2, m;
    for (ke = Se; ke > 0; ke--) { v = blk[ke]; if (v < 0) v = -v; if (v >> Al) break; }
    for (k = Ss; k <= ke; k++) {
      st = base + 3 * (k - 1);
      encode(st, 0);
      int neg;
      for (;;) {
        v = blk[k];
        neg = v < 0;
        if (neg) v = -v;
        if (v >>= Al) break;
        encode(st + 1, 0);
        st += 3;
        k++;
      }
      encode(st + 1, 1);
      encode(&fixed_bin, neg);
      st += 2;
      m = 0;
      if (v -= 1) {
        encode(st, 1);
        m = 1;
        v2 = v;
        if (v2 >>= 1) {
          encode(st, 1);
          m <<= 1;
          st = base + (k <= CHK_AC_K[tbl] ? 189 : 217);
          while (v2 >>= 1) { encode(st, 1); m <<= 1; st += 1; }
        }
      }
      encode(st, 0);
      st += 14;
      while (m >>= 1) encode(st, (m & v) ? 1 : 0);
    }
    if (k <= Se) encode(base + 3 * (k - 1), 1);
  }
  void ac_refine(int tbl, const short *blk, int Ss, int Se, int Ah, int Al)   // jcarith.c:596-687
  {
    unsigned char *base = ac[tbl], *st;
    int k, ke, kex, v;
    for (ke = Se; ke > 0; ke--) { v = blk[ke]; if (v < 0) v = -v; if (v >> Al) break; }
    for (kex = ke; kex > 0; kex--) { v = blk[kex]; if (v < 0) v = -v; if (v >> Ah) break; }
    for (k = Ss; k <= ke; k++) {
      st = base + 3 * (k - 1);
      if (k > kex) encode(st, 0);
      for (;;) {
        v = blk[k];
        const int neg = v < 0;
        if (neg) v = -v;
        if (v >>= Al) {
          if (v >> 1) encode(st + 2, v & 1);
          else { encode(st + 1, 1); encode(&fixed_bin, neg); }
          break;
        }
        encode(st + 1, 0);
        st += 3;
        k++;
      }
    }
    if (k <= Se) encode(base + 3 * (k - 1), 1);
  }
};

// ---- the product's coder on the host ----------------------------------------------------------------------------------
static void fill(ari_reg &r, int v) { for (int i = 0; i < 64; i++) r.v[i] = v; }
struct Dev {
  AriModel M;
  AriCoder A;
  int last_dc[4], ctx[4];
  std::vector<unsigned char> buf;
  Dev() : buf(1 << 24)
  {
    for (int t = 0; t < 2; t++) { for (int r = 0; r < 4; r++) fill(M.ac[t][r], ARI_BIN_RESET); fill(M.dc[t], ARI_BIN_RESET); }
    for (int r = 0; r < 4; r++) fill(M.cur[r], ARI_BIN_RESET);
    fill(M.dcur, ARI_BIN_RESET); fill(M.coef, 0);
    for (int lane = 0; lane < 64; lane++) {
      M.tab[0].v[lane] = (int)(((unsigned)mjh_ari_qe[lane] << 16) | ((unsigned)mjh_ari_nmps[lane] << 8) | (unsigned)mjh_ari_nlps[lane]);
      const int j = lane + 64 < 114 ? lane + 64 : 113;
      M.tab[1].v[lane] = (int)(((unsigned)mjh_ari_qe[j] << 16) | ((unsigned)mjh_ari_nmps[j] << 8) | (unsigned)mjh_ari_nlps[j]);
    }
    A.lane0 = true; A.out = buf.data(); A.pos = 0; A.cap = (unsigned)buf.size();
    A.reset();
  }
  void bind(int ta, int td)
  {
    for (int r = 0; r < 4; r++) M.cur[r] = M.ac[ta][r];
    M.dcur = M.dc[td];
    M.dc_lo = (int)((1L << CHK_DC_L[td]) >> 1); M.dc_hi = (int)((1L << CHK_DC_U[td]) >> 1); M.ac_k = CHK_AC_K[ta];
  }
  void unbind(int ta, int td) { for (int r = 0; r < 4; r++) M.ac[ta][r] = M.cur[r]; M.dc[td] = M.dcur; }
  void load(const short *blk) { for (int k = 0; k < 64; k++) M.coef.v[k] = blk[k]; }
};

static unsigned rnd_state = 1;
static unsigned rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

static void make_block(short *blk, int dens, int amp, int dcamp)
{
  for (int k = 0; k < 64; k++) {
    int v = 0;
    if ((int)(rnd() % 100) < dens / (1 + k / 8)) {
      v = 1 + (int)(rnd() % (unsigned)amp);
      if (rnd() % 7 == 0) v = 1 + (int)(rnd() % 2);
      if (rnd() & 1) v = -v;
    }
    blk[k] = (short)v;
  }
  blk[0] = (short)((int)(rnd() % (unsigned)(2 * dcamp + 1)) - dcamp);
}

static long compare(const Ref &R, const Dev &D, const char *what, int scan)
{
  long bad = 0;
  if (R.out.size() != D.A.pos || memcmp(R.out.data(), D.buf.data(), R.out.size())) { bad++; printf("scan %d (%s): bytes differ (%zu vs %u)\n", scan, what, R.out.size(), D.A.pos); }
  if ((unsigned)R.c != D.A.c || (unsigned)R.a != D.A.a || R.sc != D.A.sc || R.zc != D.A.zc || R.ct != D.A.ct || R.buffer != D.A.buffer) { bad++; printf("scan %d (%s): registers differ\n", scan, what); }
  for (int t = 0; t < 2; t++) {
    for (int b = 0; b < 245; b++) {
      const int w = b < 189 ? D.M.ac[t][b % 3].v[b / 3] : D.M.ac[t][ARI_X].v[b - 189];
      if ((w & 0xFF) != R.ac[t][b] || (unsigned)(w >> 16) != mjh_ari_qe[w & 0x7F] || (w & 0xFF00)) { bad++; printf("scan %d (%s): AC bin %d of table %d differs\n", scan, what, b, t); break; }
    }
    for (int b = 0; b < 64; b++) {
      const int w = D.M.dc[t].v[b];
      if ((w & 0xFF) != R.dc[t][b] || (unsigned)(w >> 16) != mjh_ari_qe[w & 0x7F]) { bad++; printf("scan %d (%s): DC bin %d of table %d differs\n", scan, what, b, t); break; }
    }
  }
  return bad;
}

int main(int argc, char **argv)
{
  const int nscans = argc > 1 ? atoi(argv[1]) : 400;
  rnd_state = argc > 2 ? (unsigned)atoi(argv[2]) : 12345u;
  long bad = 0, blocks = 0;
  for (int scan = 0; scan < nscans && bad < 10; scan++) {
    Ref R;
    Dev D;
    memset(R.ac, 0, sizeof R.ac); memset(R.dc, 0, sizeof R.dc); R.fixed_bin = 113;
    for (int i = 0; i < 4; i++) { R.last_dc[i] = R.ctx[i] = D.last_dc[i] = D.ctx[i] = 0; }
    R.reset_coder();
    for (int t = 0; t < 2; t++) {           // every third scan with the defaults, the others with random legal values (0 <= L <= U <= 15, 1 <= K <= 63)
      const bool def = scan % 3 == 0;
      CHK_DC_L[t] = def ? 0 : (int)(rnd() % 6); CHK_DC_U[t] = def ? 1 : CHK_DC_L[t] + (int)(rnd() % (unsigned)(16 - CHK_DC_L[t])); CHK_AC_K[t] = def ? 5 : 1 + (int)(rnd() % 63);
    }
    const int mode = (int)(rnd() % 5);      // 0 whole blocks, 1 DC first, 2 DC refine, 3 AC first, 4 AC refine
    const int ncomp = mode <= 2 ? 1 + (int)(rnd() % 3) : 1;
    int ta[4], td[4];
    for (int i = 0; i < 4; i++) { ta[i] = (int)(rnd() & 1); td[i] = (int)(rnd() & 1); }
    int Ss = 1, Se = 63, Ah = 0, Al = 0;
    if (mode == 1) Al = (int)(rnd() % 3);
    if (mode == 2) { Ah = 1 + (int)(rnd() % 2); Al = Ah - 1; }
    if (mode >= 3) { Ss = 1 + (int)(rnd() % 40); Se = Ss + (int)(rnd() % (unsigned)(64 - Ss)); Al = (int)(rnd() % 3); }
    if (mode == 4) { Ah = Al + 1; }
    const int nblocks = 1 + (int)(rnd() % 600), dens = 5 + (int)(rnd() % 95), amp = 1 << (rnd() % 11), dcamp = 1 << (1 + rnd() % 11);
    const int ri = (rnd() % 3 == 0) ? 1 + (int)(rnd() % 40) : 0;
    int to_go = ri, next_rst = 0;
    const char *names[5] = { "whole blocks", "DC first", "DC refine", "AC first", "AC refine" };
    for (int b = 0; b < nblocks; b++) {
      short blk[64];
      make_block(blk, dens, amp, dcamp);
      const int ci = b % ncomp;
      if (ri && ci == 0) {                    // emit_restart jcarith.c:321-352 (statistics of the tables in use, predictions, coder)
        if (to_go == 0) {
          R.finish(); R.byte(0xFF); R.byte(0xD0 + next_rst);
          D.A.finish(); D.A.byte(0xFF); D.A.byte(0xD0 + next_rst);
          for (int i = 0; i < ncomp; i++) {
            if (mode == 0 || mode == 1) { memset(R.dc[td[i]], 0, 64); fill(D.M.dc[td[i]], ARI_BIN_RESET); R.last_dc[i] = R.ctx[i] = D.last_dc[i] = D.ctx[i] = 0; }
            if (mode == 0 || mode >= 3) { memset(R.ac[ta[i]], 0, 256); for (int r = 0; r < 4; r++) fill(D.M.ac[ta[i]][r], ARI_BIN_RESET); }
          }
          R.reset_coder(); D.A.reset();
          to_go = ri; next_rst = (next_rst + 1) & 7;
        }
        to_go--;
      }
      // the product's side: what ari_run does per block (mjh_arith.hip)
      int ke = 0, kex = 0;
      if (mode == 0 || mode >= 3) {
        const int al = mode == 0 ? 0 : Al, ah = mode == 0 ? 0 : Ah, se = mode == 0 ? 63 : Se;
        for (int k = 1; k <= se; k++) { const int av = blk[k] < 0 ? -blk[k] : blk[k]; if (av >> al) ke = k; }
        if (ah) for (int k = 1; k <= ke; k++) { const int av = blk[k] < 0 ? -blk[k] : blk[k]; if (av >> ah) kex = k; }
      }
      D.load(blk);
      D.bind(ta[ci], td[ci]);
      switch (mode) {
        case 0: R.dc_diff(td[ci], ci, blk[0]); R.ac_first(ta[ci], blk, 1, 63, 0);
                ari_dc(D.A, D.M, D.last_dc[ci], D.ctx[ci], blk[0]); ari_ac_first(D.A, D.M, 1, 63, 0, ke); break;
        case 1: R.dc_diff(td[ci], ci, blk[0] >> Al); ari_dc(D.A, D.M, D.last_dc[ci], D.ctx[ci], blk[0] >> Al); break;
        case 2: R.encode(&R.fixed_bin, (blk[0] >> Al) & 1); D.A.encode<ARI_F>(D.M, 0, (blk[0] >> Al) & 1); break;
        case 3: R.ac_first(ta[0], blk, Ss, Se, Al); ari_ac_first(D.A, D.M, Ss, Se, Al, ke); break;
        default: R.ac_refine(ta[0], blk, Ss, Se, Ah, Al); ari_ac_refine(D.A, D.M, Ss, Se, Ah, Al, ke, kex); break;
      }
      D.unbind(ta[ci], td[ci]);
      blocks++;
    }
    R.finish();
    D.A.finish();
    bad += compare(R, D, names[mode], scan);
  }
  printf("%ld blocks in %d scans: %s\n", blocks, nscans, bad ? "MISMATCH" : "identical");
  return bad ? 1 : 0;
}
