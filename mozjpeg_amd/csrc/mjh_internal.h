// mjh_internal.h -- structures shared by the host pipeline (mjh_encoder.cpp) and the gfx950
// kernels (mjh_kernels.hip).  Not part of the public ABI.
#ifndef MJH_INTERNAL_H
#define MJH_INTERNAL_H

#include <stdint.h>
#include <stddef.h>

#define MJH_MAXC 4

// Huffman table slot as it lives in HBM (one per image and per role).
// counts[] is the gather histogram (a10), bits/huffval the JHUFF_TBL content (a11),
// ehufsi/ehufco the derived code lengths / codes (jchuff.c:231-318).
struct MjhDhtPlan { int slots[8], ids[8]; };   // the tables one DHT position of a sequential file carries: table slot, Tc/Th byte

struct alignas(16) MjhHuffTable {
  uint32_t counts[260];   // 257 used
  uint8_t ehufsi[256];    // 16-byte aligned: the trellis reads whole 16-symbol rows (run r: symbols 16r..16r+15)
  uint16_t ehufco[256];
  uint8_t huffval[256];
  uint8_t bits[20];       // bits[1..16]
  uint32_t nsyms;         // sum(bits[1..16])
  uint32_t pad[2];
};

// geometry of one component (initial_setup jcmaster.c:237-259)
struct MjhComp {
  int h, v;               // sampling factors
  int hexp, vexp;         // max_h/h, max_v/v
  int wib, hib;           // real blocks across / down
  int wpad, hpad;         // rounded up to the sampling factors (dummy blocks, jccoefct.c:587-601)
  int pw, ph;             // sample plane = wib*8 x hib*8
  int nblk;               // wib*hib
  int kstride;            // elements between consecutive zig-zag planes (nblk rounded up to 64)
  int qtbl, dctbl, actbl; // table numbers; dctbl: number | class << 8 -- the progressive kernels keep two DC tables per scan, class = which of the
                          // image's (at most two) distinct DC table numbers this is, in order of first use
  int mcu_blk0;           // index of this component's first block inside an interleaved MCU
  long long plane_off;    // sample offset of this component's plane inside one image's plane set
  long long coef_off;     // element offset of this component's coefficient planes inside one image's set
  long long blk_off;      // offset of this component in per-block arrays (sum of nblk of previous comps)
};

struct MjhConst {
  int W, H;
  int in_comps;           // 3 or 1
  int px_size, off_r, off_g, off_b;   // input pixel layout (extended RGB formats), in SAMPLES
  int precision;          // 8 or 12 (12: samples and planes are uint16)
  int ncomp;
  int maxh, maxv;
  int mcus_per_row, mcu_rows;
  int groups_x, groups_y; // colour-conversion groups (maxh x maxv pixels each) across / down
  int real_groups_y;      // ceil(H / maxv): groups below replicate the last downsampled row
  int blocks_per_mcu;
  int total_mcu_blocks;   // mcus * blocks_per_mcu (dummy blocks included)
  int total_real_blocks;  // sum of nblk
  int no_ycc;             // MJH_COLOR_NONE: input samples become the components unconverted (null_convert jccolor.c:479)
  int smoothing;          // smoothing_factor (0 = off): k_color_smooth replaces k_color
  int deringing;
  int trellis;            // trellis_quant: the FDCT kernel also emits the per-block lambda
  int trellis_dc;
  float delta_dc_weight;  // trellis_delta_dc_weight (> 0: the DC trellis adds the vertical-gradient term, jcdctmgr.c:1069-1084)
  int restart_interval;   // of the final interleaved scan, in MCUs (0 = none)
  int ari_L[2], ari_U[2], ari_K[2];   // arithmetic coding: conditioning of tables 0 / 1 (cinfo->arith_dc_L / arith_dc_U / arith_ac_K)
  float lambda_log_scale1, lambda_log_scale2;
  double pow_scale1, pow_scale2;  // pow(2, s1) [or pow(2, s1-12) when s2 <= 0], pow(2, s2): host libm
  long long planes_per_image;     // samples
  long long coefs_per_image;      // int16 elements
  MjhComp c[MJH_MAXC];
};

// caller-supplied component planes (jpeg_write_raw_data path): one entry per component
struct MjhPlaneSrc {
  const void *base[MJH_MAXC];        // plane of image 0
  long long pitch[MJH_MAXC];         // bytes between rows
  long long stride[MJH_MAXC];        // bytes between images
  int w[MJH_MAXC], h[MJH_MAXC];      // valid samples; beyond them the last sample / row is replicated
};

// caller-supplied quantized coefficients (jpeg_write_coefficients path): one entry per component
struct MjhCoefSrc {
  const void *base[MJH_MAXC];        // first block of image 0: [height_in_blocks][blocks_per_row][64] int16, natural order
  long long blocks_per_row[MJH_MAXC];
  long long stride[MJH_MAXC];        // bytes between images
};

// per-table-slot constant data uploaded once per encoder
struct MjhQuant {
  uint16_t q[4][64];        // zig-zag order quantizer step
  int dq8[4][64];           // 8*q as a 32-bit word: wave-uniform reads become scalar loads (no 16-bit s_load)
  float rcp8q[4][64];       // 1.0f / (8*q) for the exact-division helper
  float lambda_tbl[4][64];  // (float)(1.0 / (q*q)), zig-zag order (jcdctmgr.c:1017-1021)
  // exact division by 8q through ONE multiply-high (tables with every q <= 255 only: fastdiv[t] != 0):
  // floor(n / 8q) == ((n << sdiv) * mdiv) >> 32 for 0 <= n < 2^16, with k = min(32, 22 + bitlen(8q)), mdiv = floor(2^k / 8q) + 1
  // (< 2^24), sdiv = 32 - k (n << sdiv < 2^24: the full-rate 24-bit multiply-high applies); checked exhaustively for every
  // q in 1..255 (tools/check_fastdiv.py)
  uint32_t mdiv[4][64];
  int sdiv[4][64];
  int fastdiv[4];
  // The divisor of the CONVENTIONAL quantizer (forward_DCT).  For 8-bit samples the reference hands `quantval << 3` to
  // compute_reciprocal(UINT16 divisor, ..) (jcdctmgr.c:182, :278-282): a step of 8192 or more wraps -- q = 8450 (quality 1) divides
  // by 67600 mod 65536 = 2064 -- and that is what its files contain; 12-bit samples keep the full value (:284).  The trellis reads
  // quantval itself (8 * q as an int, :1009-1015): dq8 / rcp8q above stay unwrapped.  (At the end: the offsets of the fields
  // above, and with them the machine code of the kernels that do not quantize conventionally, stay what they were.)
  int dqc8[4][64];
  float rcpc8q[4][64];
  // the smallest |x| the trellis' conventional quantization (jcdctmgr.c:1011-1015 applied to 8q) turns into a non-zero value:
  // x + (8q >> 1) >= 8q  <=>  x >= 8q - (8q >> 1), as a float (exact: 8q < 2^24) for a compare on the converted coefficient
  float thr8[4][64];
};

// per-image bookkeeping written by the encode kernels
struct MjhImageMeta {
  unsigned total_bits;     // entropy-coded bits of the scan
  unsigned hdr_len;        // bytes before the entropy-coded data
  unsigned stuffed_len;    // entropy-coded bytes after stuffing
  unsigned file_len;       // whole file
  unsigned bad_coef;       // coefficient input only: a value no Huffman symbol exists for (JERR_BAD_DCT_COEF, jchuff.c:489,596,624)
};


// ---- progressive mode (jcphuff.c, scan scripts jcparam.c:733-1004, scan search jcmaster.c:773-962) ----
#define MJH_MAX_PROG_SCANS 72   // 64 script scans + the 3 per-component trellis statistics passes

struct MjhProgScan {
  int ncomp;
  int comp[MJH_MAXC];
  int comp_id[MJH_MAXC];   // SOS component ids
  int td[MJH_MAXC], ta[MJH_MAXC];
  int Ss, Se, Ah, Al;
  int al_sel;              // 0: Al as given; 1: MjhProgCtl.best_Al_luma; 2: best_Al_chroma (jcmaster.c:487-497)
  int cond;                // scan search: 0 = always coded; k > 0 = only for images whose luma Al search is still improving after level k
                           // (MjhProgCtl.al_continue >= k) -- select_scans stops coding candidates at the first level that does not pay (jcmaster.c:799-818)
  int slot[2];             // DC scans: slot of DC table number 0 / 1; AC scans: slot[0] = AC table
  int seed;                // statistics of a trellis pass: every (run,size<12) count starts at 1 (jcphuff.c:257-264)
  int frame_header;        // 1: this scan's buffer starts with DQT + SOF (scan 0)
  int ndht;                // tables to emit in the scan's DHT, in order
  int dht_slot[2], dht_id[2];
  // restart intervals of THIS scan (per_scan_setup jcmaster.c:595-600, T10): ri in the scan's MCUs (= blocks for a
  // single-component scan), nrst markers, their byte positions at mpos_off in the image's marker-position list
  int ri, nrst, mpos_off;
  int emit_dri;            // write_scan_header jcmarker.c:778-781: DRI when the interval differs from the previous scan's
};

// parallel chain (scans without restart intervals, mjh_prog.hip): summary of one 2048-block chunk -- where its non-empty
// blocks begin and end, and what k_pp_resolve found on either side of it
#define MJH_PSTAT_BLOCKS 2048
struct MjhProgChunk {
  int first_ne, last_ne, e_last, nblk;
  int carry_p, carry_e;      // last non-empty block before the chunk (index in the scan, -1: none) and its "ends in zeros" flag
  int next_after;            // first non-empty block behind the chunk (-1: none)
  int pad;
};

// per (scan of the list, image) pair: the run / correction bits pending at the end of the scan and the total number of
// correction bits of a refinement scan (sizes its bit stream)
struct MjhProgPair { unsigned final_run, final_be; int fallback; unsigned corr_total; };
struct MjhProgPE {         // device buffers of that path, [scan of the list][image][...]
  uint16_t *len16, *run16; // bits of every block / unit (own symbols + the flush in front of it); the run it flushes
  uint16_t *tail16, *be16; // refinement scans: trailing correction bits of every block; correction bits in front of a non-empty block
  unsigned *off32, *sums, *totals;     // prefix sum of len16
  unsigned *T32, *tsums, *ttotals;     // prefix sum of tail16
  unsigned long long *ne_bits, *e_bits;   // [pair][chunk][32]: non-empty / ends-in-zeros bitmaps
  unsigned long long *ne2_bits;           // non-empty blocks + forced-flush marks = the flush points
  unsigned long long *rmask;              // refinement scans: [pair - first pair of the other kinds][3][nblk_pad] newly non-zero / already non-zero / sign-or-correction-bit masks of every block, kept by the statistics pass for the sizes and the bits
  unsigned *chist;                        // first-pass AC scans: [pair][chunk][256] symbol counts of the chunk (sizes its bits once the table exists)
  MjhProgPair *info;
  MjhProgChunk *chunks;
  int chunks_per_scan, nblk_pad;   // nblk_pad = chunks_per_scan * MJH_PSTAT_BLOCKS entries per pair
};

struct MjhProgCtl {        // per image, lives in HBM
  int best_Al_luma, best_Al_chroma, best_fs_luma, best_fs_chroma;
  int al_continue;            // luma successive-approximation search: levels that improved so far (gates the scans with cond > 0)
  unsigned pool_words_used;   // running allocation in the bit-stream pool
  unsigned pool_zero_from;    // first word handed out by the current phase: [pool_zero_from, pool_words_used) is zeroed before the bit writers run
  unsigned out_bytes_used;    // running allocation in the scan-buffer pool
  unsigned error;             // 1: pool overflow
  unsigned pad;
  unsigned scan_bits[MJH_MAX_PROG_SCANS];      // entropy-coded bits (incl. final pad)
  unsigned scan_words_off[MJH_MAX_PROG_SCANS]; // word offset of the scan's bit stream in the pool
  unsigned scan_out_off[MJH_MAX_PROG_SCANS];   // byte offset of the scan's buffer (headers + stuffed data)
  unsigned scan_hdr_len[MJH_MAX_PROG_SCANS];
  unsigned scan_size[MJH_MAX_PROG_SCANS];      // master->scan_size[]: bytes of the whole scan buffer
  int order[MJH_MAX_PROG_SCANS];               // final scan order
  int norder;
  int pad2[3];
  unsigned scan_us[2][MJH_MAX_PROG_SCANS];     // introspection: duration of the statistics [0] / encode [1] workgroup of each scan, microseconds
};

#endif
