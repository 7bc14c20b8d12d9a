// mjh_arith.hip -- arithmetic entropy coding: jcarith.c (QM coder of ITU-T T.81 Annex D with the
// statistics models of F.1.4 / G.1.3) and the coder's own trellis rate model, quantize_trellis_arith jcdctmgr.c:1334-1667.
//
// The coder is adaptive: every binary decision changes the probability state the next decision is coded with, so a scan
// (between two restart markers) is ONE dependent chain -- there is no per-block parallelism to find, unlike the Huffman
// paths.  What is parallel: the scans of a script (64 candidates with the scan search), the images of a batch, and the
// loading of the coefficients.  Shape of every kernel here: one wave per chain; the 64 lanes fetch the next 64 blocks of
// the scan (coalesced plane loads) into LDS, lane 0 runs the coder over them.  This path exists for completeness (files
// identical to `cjpeg -arithmetic`); it is not a throughput path.
//
// With the scan search the candidates are only SIZED (the coder runs without writing); the scans the search keeps are then
// coded once more straight into the file, at offsets that follow from the sizes -- no scan pool, no concatenation pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mjh_internal.h"
#include "mjh_device.h"
#include "mjh_launch.h"

#include "mjh_arith_coder.h"      // the coder itself and the decision sequences (compiles for the host too: tests/native/arith_coder_check.cpp)

// zig-zag -> natural is not needed: the pipeline's coefficient planes are in zig-zag order already (plane k = position k)

// The scan being coded, in registers and wave-uniform FOR THE COMPILER (readfirstlane): a MjhProgScan copied as a struct lands
// in scratch memory as soon as one of its arrays is indexed dynamically, scratch loads count as divergent, and everything
// computed from them would run on the vector unit.  Arrays are only ever indexed statically here (pick4 for a run-time index).
struct AriScan {
  int ncomp, Ss, Se, Ah, ri, frame_header, emit_dri;
  int comp[4], td[4], ta[4], comp_id[4];
  int h[4], v[4];      // sampling factors of the scan's components
};
#define ARI_U(x) __builtin_amdgcn_readfirstlane((int)(x))
__device__ __forceinline__ int pick4(const int (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; }

__device__ __forceinline__ AriScan ari_load_scan(const MjhConst &C, const MjhProgScan *__restrict__ p)
{
  AriScan s;
  s.ncomp = ARI_U(p->ncomp); s.Ss = ARI_U(p->Ss); s.Se = ARI_U(p->Se); s.Ah = ARI_U(p->Ah); s.ri = ARI_U(p->ri);
  s.frame_header = ARI_U(p->frame_header); s.emit_dri = ARI_U(p->emit_dri);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s.comp[i] = ARI_U(p->comp[i]) & 3; s.td[i] = ARI_U(p->td[i]); s.ta[i] = ARI_U(p->ta[i]); s.comp_id[i] = ARI_U(p->comp_id[i]);
    s.h[i] = ARI_U(C.c[s.comp[i]].h); s.v[i] = ARI_U(C.c[s.comp[i]].v);
  }
  return s;
}

// the coefficient planes a lane has to fetch for one block of the scan, and where the block's DC comes from (dummy blocks of an
// interleaved scan repeat a neighbour's DC and have no AC: compress_first_pass jccoefct.c:312-345)
struct AriUnit { int comp_in_scan, comp, blk, dc_blk; bool dummy, mcu_start; };

__device__ __forceinline__ AriUnit ari_unit(const MjhConst &C, const AriScan &sc, int bpm, long long u, long long nunits)
{
  AriUnit r;
  if (u >= nunits) u = nunits - 1;
  if (sc.ncomp == 1) {
    r.comp_in_scan = 0; r.comp = sc.comp[0]; r.blk = (int)u; r.dc_blk = (int)u; r.dummy = false; r.mcu_start = true;
    return r;
  }
  const int m = (int)(u / bpm);
  int j = (int)(u - (long long)m * bpm), ci = 0;
  r.mcu_start = j == 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int n = sc.h[i] * sc.v[i];
    if (ci == i && i + 1 < sc.ncomp && j >= n) { j -= n; ci = i + 1; }
  }
  const int comp = pick4(sc.comp, ci), h = pick4(sc.h, ci), v = pick4(sc.v, ci);
  const MjhComp &cc = C.c[comp];
  const int yi = j / h, xi = j - yi * h;
  const int my = m / C.mcus_per_row, mx = m - my * C.mcus_per_row;
  const int row = my * v + yi, col = mx * h + xi;
  r.comp_in_scan = ci; r.comp = comp;
  r.dummy = row >= cc.hib || col >= cc.wib;
  r.dc_blk = dc_source_block(cc, row, col);
  r.blk = r.dummy ? r.dc_blk : row * cc.wib + col;
  return r;
}

struct AriChain {      // what the coder carries from block to block (wave-uniform)
  int last_dc[MJH_MAXC], ctx[MJH_MAXC];
  int to_go, next_rst;
};

__device__ __forceinline__ void ari_reset_stats(AriModel &M, AriChain &ch, const AriScan &sc, bool progressive, int lane)
{ // start_pass jcarith.c:845-875, emit_restart :328-342 (every bin of the tables in use: state 0, MPS 0)
  (void)lane;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i >= sc.ncomp) continue;
    if (!progressive || (sc.Ss == 0 && sc.Ah == 0)) {
      if (sc.td[i] & 1) M.dc[1] = ARI_BIN_RESET; else M.dc[0] = ARI_BIN_RESET;
      ch.last_dc[i] = 0;
      ch.ctx[i] = 0;
    }
    if (!progressive || sc.Se) {
#pragma unroll
      for (int r = 0; r < 4; r++) { if (sc.ta[i] & 1) M.ac[1][r] = ARI_BIN_RESET; else M.ac[0][r] = ARI_BIN_RESET; }
    }
  }
}

__device__ __forceinline__ void ari_init_model(AriModel &M, int lane)
{
#pragma unroll
  for (int r = 0; r < 4; r++) { M.ac[0][r] = M.ac[1][r] = ARI_BIN_RESET; M.cur[r] = ARI_BIN_RESET; }
  M.dc[0] = M.dc[1] = M.dcur = ARI_BIN_RESET;
  M.coef = 0;
  M.tab[0] = (int)(((unsigned)mjh_ari_qe[lane] << 16) | ((unsigned)mjh_ari_nmps[lane] << 8) | (unsigned)mjh_ari_nlps[lane]);
  const int j = lane + 64 < 114 ? lane + 64 : 113;
  M.tab[1] = (int)(((unsigned)mjh_ari_qe[j] << 16) | ((unsigned)mjh_ari_nmps[j] << 8) | (unsigned)mjh_ari_nlps[j]);
}

// the statistics of the block's tables are bound to fixed registers while it is coded, and put back behind it
__device__ __forceinline__ void ari_bind(AriModel &M, int ta, int td, const MjhConst &C)
{
  const int am = -(ta & 1), dm = -(td & 1);      // all ones: table 1
  M.dc_lo = (int)((1L << C.ari_L[td & 1]) >> 1); M.dc_hi = (int)((1L << C.ari_U[td & 1]) >> 1); M.ac_k = C.ari_K[ta & 1];
#pragma unroll
  for (int r = 0; r < 4; r++) M.cur[r] = (M.ac[1][r] & am) | (M.ac[0][r] & ~am);
  M.dcur = (M.dc[1] & dm) | (M.dc[0] & ~dm);
}
__device__ __forceinline__ void ari_unbind(AriModel &M, int ta, int td)
{
  const int am = -(ta & 1), dm = -(td & 1);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    M.ac[0][r] = (M.cur[r] & ~am) | (M.ac[0][r] & am);
    M.ac[1][r] = (M.cur[r] & am) | (M.ac[1][r] & ~am);
  }
  M.dc[0] = (M.dcur & ~dm) | (M.dc[0] & dm);
  M.dc[1] = (M.dcur & dm) | (M.dc[1] & ~dm);
}

// One chain: the units [u0, u1) of scan `sc` (an MCU = bpm units) run through the coder.  whole_blocks: DC + AC 1..63 of every
// block whatever the scan parameters say (sequential files; the state updates of the trellis passes, jcarith.c:824-826).
// dctbl / actbl of component i of the scan: sc.td[i] / sc.ta[i] (which hold the component's table numbers for whole_blocks).
// Called by every thread of the workgroup (the barriers); `coder`: this thread belongs to the coding wave (wave 0), which also
// loads.  s_blk: 64 x 64 int16 of LDS.
__device__ __forceinline__ void ari_run(const MjhConst &C, const AriScan &sc, int Al, bool whole_blocks, bool progressive,
                                        const int16_t *__restrict__ qimg, long long u0, long long u1, int bpm,
                                        AriCoder &A, AriModel &M, AriChain &ch, short *s_blk, int lane, bool coder)
{
  const int Ss = whole_blocks ? 0 : sc.Ss, Se = whole_blocks ? 63 : sc.Se;
  for (long long base = u0; base < u1; base += 64) {
    int kev = 0, kexv = 0, infov = 0;      // per lane: end-of-block indices and component / MCU-start flag of ITS block
    if (coder) {
      const AriUnit un = ari_unit(C, sc, bpm, base + lane, u1);
      // (the component's layout once per group of blocks, in registers: read through the reference inside the loop below it
      // was fetched again for every coefficient, one more memory round trip in front of each load)
      const int ks = C.c[un.comp].kstride, coff = C.c[un.comp].coef_off;
      const int16_t *q = qimg + coff;
      short *row = s_blk + lane * 64;
      if (Ss == 0) row[0] = q[un.dc_blk];
      if (Se > 0) {
        // (from position 1: the end-of-block searches of the AC scans look below Ss as well, jcarith.c:484-496); the last
        // position that is non-zero after the point transform by Al (ke), and by Ah below it (kex, refinement scans).
        // Eight loads in flight at a time, all unconditional (a dummy block's index is a real block's, positions behind Se
        // read Se again): a wave that waits for every load separately spends 63 memory round trips per group of blocks here.
        const int al = whole_blocks ? 0 : Al, ah = whole_blocks ? 0 : sc.Ah;
        const int16_t *qb = q + un.blk;
        for (int k0 = 1; k0 <= Se; k0 += 8) {
          short v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) { const int kk = k0 + j <= Se ? k0 + j : Se; v[j] = qb[(size_t)kk * ks]; }
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (k0 + j <= Se) {
              const short vv = un.dummy ? (short)0 : v[j];
              row[k0 + j] = vv;
              const int av = vv < 0 ? -(int)vv : (int)vv;
              if (av >> al) kev = k0 + j;
            }
          }
        }
        if (ah) { for (int k = 1; k <= kev; k++) { const int v = row[k], av = v < 0 ? -v : v; if (av >> ah) kexv = k; } }
      }
      infov = un.comp_in_scan | (un.mcu_start ? 16 : 0);
    }
    __syncthreads();
    if (coder) {
      const int nb = (int)(u1 - base < 64 ? u1 - base : 64);
      for (int b = 0; b < nb; b++) {
        M.coef = s_blk[b * 64 + lane];            // lane k: coefficient k of block b
        const int info = rl(infov, b), ke = rl(kev, b), kex = rl(kexv, b);
        const int ci = info & 15;
        if (sc.ri && (info & 16)) {               // first block of an MCU: restart bookkeeping (jcarith.c:371-379 and twins)
          if (ch.to_go == 0) {
            A.finish();
            A.byte(0xFF); A.byte(0xD0 + ch.next_rst);
            ari_reset_stats(M, ch, sc, progressive, lane);
            A.reset();
            ch.to_go = sc.ri;
            ch.next_rst = (ch.next_rst + 1) & 7;
          }
          ch.to_go--;
        }
        // (component ci of the scan: its tables, its DC prediction)
        const int td = pick4(sc.td, ci), ta = pick4(sc.ta, ci);
        int last = pick4(ch.last_dc, ci), ctx = pick4(ch.ctx, ci);
        const int dc = ari_coef(M, 0);
        ari_bind(M, ta, td, C);
        if (whole_blocks) {
          ari_dc(A, M, last, ctx, dc);
          ari_ac_first(A, M, 1, 63, 0, ke);
        } else if (sc.Ss == 0 && sc.Ah == 0) ari_dc(A, M, last, ctx, dc >> Al);
        else if (sc.Ss == 0) A.encode<ARI_F>(M, 0, (dc >> Al) & 1);                        // encode_mcu_DC_refine :560-590
        else if (sc.Ah == 0) ari_ac_first(A, M, sc.Ss, sc.Se, Al, ke);
        else ari_ac_refine(A, M, sc.Ss, sc.Se, sc.Ah, Al, ke, kex);
        ari_unbind(M, ta, td);
        if (ci == 0) { ch.last_dc[0] = last; ch.ctx[0] = ctx; } else if (ci == 1) { ch.last_dc[1] = last; ch.ctx[1] = ctx; }
        else if (ci == 2) { ch.last_dc[2] = last; ch.ctx[2] = ctx; } else { ch.last_dc[3] = last; ch.ctx[3] = ctx; }
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ long long ari_scan_units(const MjhConst &C, const AriScan &sc, int &bpm)
{
  if (sc.ncomp == 1) { bpm = 1; return C.c[sc.comp[0]].nblk; }
  bpm = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) if (i < sc.ncomp) bpm += sc.h[i] * sc.v[i];
  return (long long)C.mcus_per_row * C.mcu_rows * bpm;
}

// scan header: [DQT + SOF9/SOF10 for scan 0] DAC [DRI] SOS (write_scan_header jcmarker.c:744-784, emit_dac :404-448); o may be null (length only)
__device__ __forceinline__ unsigned ari_scan_header(const MjhConst &C, const AriScan &sc, int Al, const uint8_t *frame_hdr, int frame_hdr_len, uint8_t *o)
{
  unsigned n = 0;
  if (sc.frame_header) {
    if (o) for (int i = 0; i < frame_hdr_len; i++) o[i] = frame_hdr[i];
    n = (unsigned)frame_hdr_len;
  }
  auto put = [&](int b) { if (o) o[n] = (uint8_t)b; n++; };
  int dc_use0 = 0, dc_use1 = 0, ac_use0 = 0, ac_use1 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i >= sc.ncomp) continue;
    if (sc.Ss == 0 && sc.Ah == 0) { if (sc.td[i] & 1) dc_use1 = 1; else dc_use0 = 1; }
    if (sc.Se) { if (sc.ta[i] & 1) ac_use1 = 1; else ac_use0 = 1; }
  }
  const int ntab = dc_use0 + dc_use1 + ac_use0 + ac_use1;
  if (ntab) {
    put(0xFF); put(0xCC); put(0); put(ntab * 2 + 2);
    if (dc_use0) { put(0); put(C.ari_L[0] + (C.ari_U[0] << 4)); }
    if (ac_use0) { put(0x10); put(C.ari_K[0]); }
    if (dc_use1) { put(1); put(C.ari_L[1] + (C.ari_U[1] << 4)); }
    if (ac_use1) { put(0x11); put(C.ari_K[1]); }
  }
  if (sc.emit_dri) { put(0xFF); put(0xDD); put(0); put(4); put(sc.ri >> 8); put(sc.ri & 0xFF); }
  put(0xFF); put(0xDA);
  const int len = 2 * sc.ncomp + 2 + 1 + 3;
  put(len >> 8); put(len & 0xFF);
  put(sc.ncomp);
#pragma unroll
  for (int i = 0; i < 4; i++) if (i < sc.ncomp) { put(sc.comp_id[i]); put((sc.td[i] << 4) + sc.ta[i]); }
  put(sc.Ss); put(sc.Se); put((sc.Ah << 4) + Al);
  return n;
}


// One scan of one image.  WRITE = false: its size (header + entropy-coded bytes) goes to ctl.scan_size (what the scan search
// compares, jcmaster.c:773-962).  WRITE = true: grid.y walks the FINAL order (ctl.order); the scan is written at its place in
// the file, ctl.scan_out_off (k_arith_layout) -- or, `single_pass`, behind the file header with EOI and the file size (a script
// of one scan needs no sizing pass).  whole_blocks: a sequential file (SOF9).
template <bool WRITE>
__global__ void __launch_bounds__(64)
k_arith_scan(MjhConst C, const MjhProgScan *__restrict__ scans, const int *__restrict__ scan_list, MjhProgCtl *__restrict__ ctl,
             const int16_t *__restrict__ coef_q, const uint8_t *__restrict__ frame_hdr, int frame_hdr_len,
             const uint8_t *__restrict__ file_hdr, int file_hdr_len, uint8_t *__restrict__ out, size_t out_stride,
             unsigned *__restrict__ sizes, int whole_blocks, int single_pass)
{
  __shared__ short s_blk[64 * 64];
  // image index fastest: consecutive workgroups go to consecutive XCDs, so the chains of one scan (equally long, image by
  // image) spread evenly over the chip.  A coding wave keeps its SIMD's scalar issue slot busy by itself: where the long
  // chains of a batch crowd onto a few XCDs (scan index fastest did that), two share a SIMD and both run at half speed.
  const int img = blockIdx.x, li = blockIdx.y, lane = threadIdx.x;
  MjhProgCtl *ct = ctl + img;
  int sidx;
  if (WRITE && !single_pass) {
    if (li >= ARI_U(ct->norder)) return;
    sidx = ct->order[li];
  } else sidx = scan_list[li];
  sidx = ARI_U(sidx);
  const MjhProgScan *sp = scans + sidx;
  const int cond = ARI_U(sp->cond), al_sel = ARI_U(sp->al_sel);
  if (!WRITE && cond > 0 && ARI_U(ct->al_continue) < cond) { if (lane == 0) ct->scan_size[sidx] = 0; return; }   // the search stopped below this level
  if (WRITE && ARI_U(ct->error)) { if (lane == 0 && (single_pass || li == 0)) sizes[img] = 0; return; }
  const int Al = ARI_U(al_sel == 1 ? ct->best_Al_luma : (al_sel == 2 ? ct->best_Al_chroma : sp->Al));
  const AriScan sc = ari_load_scan(C, sp);
  int bpm;
  const long long nunits = ari_scan_units(C, sc, bpm);
  uint8_t *o = nullptr;
  unsigned hdr = 0, cap = 0;
  if (WRITE) {
    const size_t off = single_pass ? (size_t)file_hdr_len : (size_t)(unsigned)ARI_U(ct->scan_out_off[sidx]);
    o = out + (size_t)img * out_stride + off;
    cap = (unsigned)(out_stride - off > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : out_stride - off);
    if (single_pass) for (int i = lane; i < file_hdr_len; i += 64) out[(size_t)img * out_stride + i] = file_hdr[i];
  }
  AriCoder A;
  AriChain ch;
  AriModel M;
  ari_init_model(M, lane);
  hdr = ari_scan_header(C, sc, Al, frame_hdr, frame_hdr_len, WRITE && lane == 0 ? o : nullptr);   // (every lane: the length is wave-uniform)
  A.lane0 = lane == 0;
  A.pos = 0;
  A.out = WRITE ? o + hdr : nullptr;
  A.cap = WRITE ? (cap > hdr + 4 ? cap - hdr - 4 : 0) : 0;
  A.reset();
  for (int i = 0; i < MJH_MAXC; i++) { ch.last_dc[i] = 0; ch.ctx[i] = 0; }
  ari_reset_stats(M, ch, sc, !whole_blocks, lane);
  ch.to_go = sc.ri; ch.next_rst = 0;
  ari_run(C, sc, Al, whole_blocks != 0, !whole_blocks, coef_q + (size_t)img * C.coefs_per_image, 0, nunits, bpm, A, M, ch, s_blk, lane, true);
  A.finish();
  if (lane == 0) {
    const unsigned total = hdr + A.pos;
    if (!WRITE) ct->scan_size[sidx] = total;
    else {
      if (A.pos > A.cap) ct->error = 1;          // the file does not fit its buffer
      if (single_pass) {
        if (A.pos <= A.cap) { o[total] = 0xFF; o[total + 1] = 0xD9; }
        sizes[img] = A.pos <= A.cap ? (unsigned)file_hdr_len + total + 2u : 0u;
        ct->scan_size[sidx] = total;
      }
    }
  }
}

// where the chosen scans go in the file (sizes are exact: the same coder sized them), SOI/APP0 in front, EOI behind
__global__ void __launch_bounds__(64)
k_arith_layout(MjhProgCtl *__restrict__ ctl, const uint8_t *__restrict__ file_hdr, int file_hdr_len, uint8_t *__restrict__ out, size_t out_stride,
               unsigned *__restrict__ sizes, int nimg)
{
  const int img = blockIdx.x * 64 + threadIdx.x;
  if (img >= nimg) return;
  MjhProgCtl *ct = ctl + img;
  unsigned long long off = (unsigned long long)file_hdr_len;
  for (int s = 0; s < ct->norder; s++) {
    const int sidx = ct->order[s];
    ct->scan_out_off[sidx] = (unsigned)off;
    off += ct->scan_size[sidx];
  }
  uint8_t *o = out + (size_t)img * out_stride;
  if (off + 2 > out_stride || off + 2 >= (1ull << 32)) { ct->error = 1; sizes[img] = 0; return; }
  for (int i = 0; i < file_hdr_len; i++) o[i] = file_hdr[i];
  o[off] = 0xFF; o[off + 1] = 0xD9;
  sizes[img] = (unsigned)(off + 2);
}

// =============================================================================================
// quantize_trellis_arith (jcdctmgr.c:1334-1667) driven by compress_trellis_pass (jccoefct.c:356-486).
// The rate estimates are read from the CURRENT state of the adaptive coder once per iMCU row (jget_arith_rates
// jcarith.c:944-976); the row is quantized with them and then run through the coder (output discarded), which moves
// the state the next row reads.  So a component is one chain of (rates, quantize a row group, code it) steps: one
// workgroup of 256 threads per image --
//   * rates: 320 bins looked up in a 256-entry table the host computed with its libm (the estimate depends on the bin's
//     state byte only; -log(p)/log(2) in double like the reference);
//   * AC: one thread per block of the row group: the reference's DP (two candidates per coefficient, `int rate`) with the
//     same float operations in the same order, walked over the live predecessors only (quadratic instead of cubic);
//   * DC: lanes 0..8 of wave 0 = the candidates of a block, chained along each block row;
//   * state update: wave 0 codes the row group's blocks (whole blocks, raster order) with lane 0.
// Only component 0 is ever selected for these passes when arithmetic coding is on (jcmaster.c: prepare_for_pass's
// trellis_pass case does not re-select the scan and no statistics pass sits between them, :686-702, :1001-1005), each pass
// restarts from a zeroed state and the unquantized coefficients: one pass reproduces them all.
// =============================================================================================
struct MjhArithRates { float r[256][2]; };    // [state byte][decision]: -log2 of the decision's probability estimate

__device__ __forceinline__ float ari_dc_bits(const float (*rdc)[2], int st, int dc_delta, int &upd, int dc_lo, int dc_hi)
{ // the DC difference's estimated bits from context st (jcdctmgr.c:1464-1497); upd = the context it leaves behind
  float bits = rdc[st][dc_delta != 0];
  upd = 0;
  if (dc_delta != 0) {
    bits += rdc[st + 1][dc_delta < 0];
    st += 2 + (dc_delta < 0);
    upd = dc_delta < 0 ? 8 : 4;
    if (dc_delta < 0) dc_delta = -dc_delta;
    int m = 0;
    if (dc_delta -= 1) {
      bits += rdc[st][1];
      st = 20;
      m = 1;
      int v2 = dc_delta;
      while (v2 >>= 1) { bits += rdc[st][1]; m <<= 1; st++; }
    }
    bits += rdc[st][0];
    if (m < dc_lo) upd = 0;
    else if (m > dc_hi) upd += 8;
    st += 14;
    while (m >>= 1) bits += rdc[st][(m & dc_delta) ? 1 : 0];
  }
  return bits;
}

__global__ void __launch_bounds__(256)
k_trellis_arith(MjhConst C, const MjhQuant *__restrict__ Q, int qstride, const int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q,
                const float *__restrict__ lambda_in, const MjhArithRates *__restrict__ rate_tab, uint8_t *__restrict__ back,
                int Ss, int Se, int quant_dc, float delta_dc_weight, int restart_blocks, int prog_file)
{
  __shared__ short s_blk[64 * 64];
  __shared__ unsigned char s_state[64 + 256];     // the coder's statistics bins, dumped by the coding wave once per iMCU row
  __shared__ float rdc[64][2], rac[256][2];
  __shared__ float dc_cost[2][9];
  __shared__ int dc_ctx[2][9], dc_cand[2][9];
  __shared__ int s_lastdc;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // (uniform for the compiler too: the coding wave's branch is a scalar one)
  Q += (size_t)img * qstride;    // trellis_q_opt: every image has its own table set
  const MjhComp cc = C.c[0];
  const int qt = cc.qtbl;
  const int16_t *uq = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off;
  int16_t *q = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;
  const float *lam = lambda_in + (size_t)img * C.total_real_blocks + cc.blk_off;
  uint8_t *bk = back + ((size_t)img * C.total_real_blocks + cc.blk_off) * 16;
  const int q0 = Q->q[qt][0];
  int ncand = (2 + 60 / q0) | 1;
  if (ncand > 9) ncand = 9;
  // conditioning of component 0's tables (jget_arith_rates jcarith.c:949-951)
  const int dc_lo = (int)((1L << C.ari_L[cc.dctbl & 1]) >> 1), dc_hi = (int)((1L << C.ari_U[cc.dctbl & 1]) >> 1), ac_k = C.ari_K[cc.actbl & 1];
  AriCoder A;
  AriChain ch;
  AriModel M;
  AriScan sc;          // the single-component scan of the pass: whole blocks of component 0
  sc.ncomp = 1; sc.Ss = Ss; sc.Se = Se; sc.Ah = 0; sc.ri = restart_blocks; sc.frame_header = 0; sc.emit_dri = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { sc.comp[i] = 0; sc.td[i] = ARI_U(cc.dctbl); sc.ta[i] = ARI_U(cc.actbl); sc.comp_id[i] = 0; sc.h[i] = 1; sc.v[i] = 1; }
  ari_init_model(M, lane);
  for (int i = tid; i < 64 + 256; i += 256) s_state[i] = 0;   // (bins 245..255 do not exist: their rates are never read)
  A.lane0 = false; A.out = nullptr; A.pos = 0; A.cap = 0;     // nothing is written: only the statistics move
  A.reset();
  for (int i = 0; i < MJH_MAXC; i++) { ch.last_dc[i] = 0; ch.ctx[i] = 0; }
  ari_reset_stats(M, ch, sc, false, lane);
  ch.to_go = restart_blocks; ch.next_rst = 0;
  for (int br0 = 0; br0 < cc.hib; br0 += cc.v) {
    const int rows = br0 + cc.v <= cc.hib ? cc.v : cc.hib - br0;
    // ---- rates of this iMCU row (jget_arith_rates): the coding wave's bins -> LDS -> 320 table look-ups
    if (wave == 0) {      // bin 3 p + kind of jcarith.c's AC numbering = lane p of register `kind`, 189 + x = lane x of the tail register
      const int am = -(cc.actbl & 1), dm = -(cc.dctbl & 1);
      const int be = (M.ac[1][ARI_E] & am) | (M.ac[0][ARI_E] & ~am), bz = (M.ac[1][ARI_Z] & am) | (M.ac[0][ARI_Z] & ~am);
      const int bm = (M.ac[1][ARI_M] & am) | (M.ac[0][ARI_M] & ~am), bx = (M.ac[1][ARI_X] & am) | (M.ac[0][ARI_X] & ~am);
      if (lane < 63) {
        s_state[64 + 3 * lane] = (unsigned char)be; s_state[64 + 3 * lane + 1] = (unsigned char)bz; s_state[64 + 3 * lane + 2] = (unsigned char)bm;
      }
      if (lane < 56) s_state[64 + 189 + lane] = (unsigned char)bx;
      s_state[lane] = (unsigned char)((M.dc[1] & dm) | (M.dc[0] & ~dm));
    }
    __syncthreads();
    for (int i = tid; i < 64 + 256; i += 256) {
      const int state = s_state[i];
      float *o = i < 64 ? rdc[i] : rac[i - 64];
      o[0] = rate_tab->r[state][0];
      o[1] = rate_tab->r[state][1];
    }
    __syncthreads();
    // ---- AC: one thread per block of the row group (jcdctmgr.c:1512-1631).  The reference's DP with its float operations in
    // the reference's order, arranged around the LIVE entries (the start and the positions given a non-zero value so far --
    // the only predecessors its loop does not skip): the run's bits towards a predecessor j are a left-to-right float sum that
    // starts at j, so each entry carries its own partial sum and is brought up to the current position when it is next
    // consulted (the reference re-adds the whole run for every (j, i): cubic); a candidate's own bits do not depend on j.
    for (int idx = tid; idx < rows * cc.wib; idx += 256) {
      const int blk = br0 * cc.wib + idx;
      const float lambda = lam[blk];
      float lsum[64], lazd[64], lcost[64];     // entry e: run bits from its position up to `upto`, azd and cost at its position
      short lval[64], coef[64];
      unsigned char lpos[64], lback[64];
      int nl = 1, upto = Ss, fresh = 0;        // fresh: the newest entry was made at step `upto` (its run starts one position later)
      lsum[0] = rac[3 * (Ss - 1)][0]; lazd[0] = 0.0f; lcost[0] = 0.0f; lpos[0] = (unsigned char)(Ss - 1); lval[0] = 0; lback[0] = 0;
      float azd_run = 0.0f;
      for (int i = Ss; i <= Se; i++) {
        const int xs = uq[(size_t)i * cc.kstride + blk];
        const int sign = xs < 0 ? -1 : 0, x = xs < 0 ? -xs : xs;
        const int dq = Q->dq8[qt][i];
        const float lt = Q->lambda_tbl[qt][i];
        float t = (float)mul24(x, x) * lambda;
        t = t * lt;
        const float azd_im1 = azd_run;
        azd_run = t + azd_run;
        const int qval = (x + (dq >> 1)) / dq;
        if (qval == 0) continue;
        int cand[2];
        float cdist[2], cbits[2];
        int ncd = 1;
        cand[0] = qval;
        { const int delta = cand[0] * dq - x; float d = (float)(delta * delta) * lambda; cdist[0] = d * lt; }
        cand[1] = qval - 1; cdist[1] = 0.0f; cbits[1] = 0.0f;
        if (qval > 1) { const int delta = cand[1] * dq - x; float d = (float)(delta * delta) * lambda; cdist[1] = d * lt; ncd = 2; }
        for (int k = 0; k < ncd; k++) {
          float coef_bits = 1.0f;
          int vv = cand[k], m = 0, st = 3 * (i - 1) + 2;
          if (vv -= 1) {
            coef_bits += rac[st][1];
            m = 1;
            int v2 = vv;
            if (v2 >>= 1) {
              coef_bits += rac[st][1];
              m <<= 1;
              st = i <= ac_k ? 189 : 217;
              while (v2 >>= 1) { coef_bits += rac[st][1]; m <<= 1; st++; }
            }
          }
          coef_bits += rac[st][0];
          st += 14;
          while (m >>= 1) coef_bits += rac[st][(m & vv) ? 1 : 0];
          cbits[k] = coef_bits;
        }
        const float nz_bit = rac[3 * (i - 1) + 1][1];
        float best = 1e38f;
        int bj = 0, bv = 0;
        bool found = false;
        for (int e = 0; e < nl; e++) {
          float run_bits = lsum[e];
          for (int k = upto + ((fresh && e == nl - 1) ? 1 : 0); k < i; k++) run_bits += rac[3 * (k - 1) + 1][0];
          lsum[e] = run_bits;
          run_bits += nz_bit;
          float rhs = azd_im1 - lazd[e];
          rhs = rhs + lcost[e];
          for (int k = 0; k < ncd; k++) {
            const int rate = (int)(cbits[k] + run_bits);        // `int rate` (jcdctmgr.c:1349, :1583): the estimate is truncated
            float cost = (float)rate + cdist[k];
            cost = cost + rhs;
            if (cost < best) { best = cost; bv = (cand[k] ^ sign) - sign; bj = e; found = true; }
          }
        }
        upto = i; fresh = 0;
        if (found) {
          lsum[nl] = rac[3 * i][0]; lazd[nl] = azd_run; lcost[nl] = best;
          lpos[nl] = (unsigned char)i; lval[nl] = (short)bv; lback[nl] = (unsigned char)bj;
          nl++; fresh = 1;
        }
      }
      int last = 0;
      float best_cost = azd_run + rac[0][1];
      for (int e = 1; e < nl; e++) {
        const int i = lpos[e];
        float cost = lcost[e] + azd_run;
        cost = cost - lazd[e];
        if (i < Se) cost = cost + rac[3 * (i - 1)][1];
        if (cost < best_cost) { best_cost = cost; last = e; }
      }
      for (int k = 0; k < 64; k++) coef[k] = 0;
      for (int e = last; e > 0; e = lback[e]) coef[lpos[e]] = lval[e];
      for (int k = Ss; k <= Se; k++) q[(size_t)k * cc.kstride + blk] = coef[k];
    }
    // ---- DC along each block row of the group (jcdctmgr.c:1416-1509, :1643-1665): lane k of wave 0 = candidate k
    if (quant_dc) {
      if (tid == 0) s_lastdc = 0;
      __syncthreads();
      for (int rr = 0; rr < rows; rr++) {
        const int row0 = (br0 + rr) * cc.wib;
        if (wave == 0) {
          const int dq = 8 * q0;
          const float lt0 = Q->lambda_tbl[qt][0];
          const bool vert = rr > 0 && delta_dc_weight > 0.0f;
          // forward, 64 blocks at a time: the lanes fetch what the chunk's blocks need in one go, the steps of the chain
          // read it with v_readlane (a global load per step would put its latency into every link of the chain)
          for (int c0 = 0; c0 < cc.wib; c0 += 64) {
            const int nb = cc.wib - c0 < 64 ? cc.wib - c0 : 64;
            const int bl = c0 + (lane < nb ? lane : nb - 1);
            const int xs_l = uq[row0 + bl];
            const float lam_l = lam[row0 + bl] * lt0;
            int ao_l = 0, ar_l = 0;
            if (vert) { ao_l = uq[row0 - cc.wib + bl]; ar_l = (int)q[row0 - cc.wib + bl] * dq; }
            for (int b = 0; b < nb; b++) {
              const int bi = c0 + b;
              const int cur = bi & 1, prv = cur ^ 1;
              const int xs = rl(xs_l, b);
              const float lambda_dc = __int_as_float(rl(__float_as_int(lam_l), b));
              const int above_orig = rl(ao_l, b), above_recon = rl(ar_l, b);
              if (lane < ncand) {
                const int sign = xs < 0 ? -1 : 0, x = xs < 0 ? -xs : xs;
                const int qval = (x + (dq >> 1)) / dq;
                int cnd = qval - ncand / 2 + lane;
                int delta = cnd * dq - x;
                float dist = (float)(delta * delta) * lambda_dc;
                cnd *= 1 + 2 * sign;
                if (vert) {       // the block above inside the iMCU row (jcdctmgr.c:1440-1456)
                  delta = (above_orig - xs) - (above_recon - cnd * dq);
                  const float vertical = (float)(delta * delta) * lambda_dc;
                  float t = vertical - dist;
                  t = delta_dc_weight * t;
                  dist = dist + t;
                }
                float best = 0.0f;
                int bb = -1, bctx = 0;
                const int nl = bi == 0 ? 1 : ncand;
                for (int l = 0; l < nl; l++) {
                  const int pred = bi == 0 ? s_lastdc : dc_cand[prv][l];
                  int upd;
                  const float bits = ari_dc_bits(rdc, bi == 0 ? 0 : dc_ctx[prv][l], cnd - pred, upd, dc_lo, dc_hi);
                  float cost = bits + dist;
                  if (bi != 0) cost += dc_cost[prv][l];
                  if (l == 0 || cost < best) { best = cost; bb = bi == 0 ? -1 : l; bctx = upd; }
                }
                dc_cost[cur][lane] = best; dc_ctx[cur][lane] = bctx; dc_cand[cur][lane] = cnd;
                bk[(size_t)(row0 + bi) * 16 + lane] = (uint8_t)(bb < 0 ? 0 : bb);
              }
              __builtin_amdgcn_wave_barrier();
              __threadfence_block();
            }
          }
          // back-track, 64 blocks at a time from the end of the row: the chunk's back pointers (16 bytes per block) and
          // values are fetched by the lanes, the walk itself touches registers only
          int j = 0;
          {
            const int cur = (cc.wib - 1) & 1;
            for (int i = 1; i < ncand; i++) if (dc_cost[cur][i] < dc_cost[cur][j]) j = i;
            j = __builtin_amdgcn_readfirstlane(j);
          }
          int lastdc_new = 0;
          for (int c0 = ((cc.wib - 1) >> 6) << 6; c0 >= 0; c0 -= 64) {
            const int nb = cc.wib - c0 < 64 ? cc.wib - c0 : 64;
            const int bl = c0 + (lane < nb ? lane : nb - 1);
            const int xs_l = uq[row0 + bl];
            const int sign_l = xs_l < 0 ? -1 : 0, x_l = xs_l < 0 ? -xs_l : xs_l;
            const int base_l = (x_l + (dq >> 1)) / dq - ncand / 2;          // candidate 0 of the lane's block, before the sign
            const uint4 bk4 = *reinterpret_cast<const uint4 *>(bk + (size_t)(row0 + bl) * 16);
            int qout = 0;
            for (int b = nb - 1; b >= 0; b--) {
              const int cnd = (rl(base_l, b) + j) * (1 + 2 * rl(sign_l, b));
              if (lane == b) qout = cnd;
              if (c0 + b == cc.wib - 1) lastdc_new = cnd;
              const unsigned w = j < 4 ? bk4.x : (j < 8 ? bk4.y : bk4.z);       // (j <= 8: nine candidates at most)
              j = (int)(((unsigned)rl((int)w, b) >> (8 * (j & 3))) & 0xFFu);
            }
            if (lane < nb) q[row0 + c0 + lane] = (int16_t)qout;
          }
          if (lane == 0) s_lastdc = lastdc_new;
        }
        __threadfence_block();
        __syncthreads();
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- the row group goes through the coder: its statistics move on (compress_output -> encode_mcu, output discarded)
    // (wave 0 loads, thread 0 codes, every wave takes part in the barriers; emit_restart consults the FILE's mode, jcarith.c:328-341)
    ari_run(C, sc, 0, true, prog_file != 0, coef_q + (size_t)img * C.coefs_per_image, (long long)br0 * cc.wib, (long long)(br0 + rows) * cc.wib, 1, A, M, ch,
            s_blk, lane, wave == 0);
    __syncthreads();
  }
}

// =============================================================================================
// launch wrappers
// =============================================================================================
void mjh_launch_arith_scans(const MjhConst &C, const void *scans, const int *list, int nlist, void *ctl, const void *q,
                            const uint8_t *frame_hdr, int frame_hdr_len, const uint8_t *file_hdr, int file_hdr_len,
                            uint8_t *out, size_t out_stride, unsigned *sizes, int whole_blocks, int mode, int n, hipStream_t s)
{
  // mode 0: size the scans of the list; 1: write the scans of the final order (nlist = upper bound of its length); 2: one scan, one pass
  if (mode == 0)
    hipLaunchKernelGGL((k_arith_scan<false>), dim3(n, nlist), dim3(64), 0, s, C, (const MjhProgScan *)scans, list, (MjhProgCtl *)ctl, (const int16_t *)q,
                       frame_hdr, frame_hdr_len, file_hdr, file_hdr_len, out, out_stride, sizes, whole_blocks, 0);
  else
    hipLaunchKernelGGL((k_arith_scan<true>), dim3(n, nlist), dim3(64), 0, s, C, (const MjhProgScan *)scans, list, (MjhProgCtl *)ctl, (const int16_t *)q,
                       frame_hdr, frame_hdr_len, file_hdr, file_hdr_len, out, out_stride, sizes, whole_blocks, mode == 2 ? 1 : 0);
}

void mjh_launch_arith_layout(void *ctl, const uint8_t *file_hdr, int file_hdr_len, uint8_t *out, size_t out_stride, unsigned *sizes, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_arith_layout, dim3((n + 63) / 64), dim3(64), 0, s, (MjhProgCtl *)ctl, file_hdr, file_hdr_len, out, out_stride, sizes, n);
}

void mjh_launch_trellis_arith(const MjhConst &C, const MjhQuant *Q, int qstride, const void *uq, void *q, const float *lambda, const void *rate_tab, void *back,
                              int Ss, int Se, int quant_dc, float delta_dc_weight, int restart_blocks, int prog_file, int n, hipStream_t s)
{
  hipLaunchKernelGGL(k_trellis_arith, dim3(n), dim3(256), 0, s, C, Q, qstride, (const int16_t *)uq, (int16_t *)q, lambda, (const MjhArithRates *)rate_tab, (uint8_t *)back,
                     Ss, Se, quant_dc, delta_dc_weight, restart_blocks, prog_file);
}
