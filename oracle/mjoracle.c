/*
 * mjoracle.c -- TEST INFRASTRUCTURE ONLY (see mjoracle.h for the parity pin statement).
 *
 * Plain-C restatement of the mozjpeg encode hot path as a whole-image planar encoder.
 * Every function names the reference file:line whose behaviour it follows
 * (paths relative to /root/reference).  Nothing here is used by the product.
 *
 * Float discipline: built with -ffp-contract=off; every float expression of
 * the reference is spelled out one IEEE operation at a time in the reference's
 * evaluation order, including the float/double promotion points.
 */
#include "mjoracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* zig-zag -> natural order, jutils.c:59 (the standard JPEG zig-zag sequence) */
static const int ZZ[64] = {
  0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

static int nbits_of(unsigned v) /* JPEG_NBITS, jpeg_nbits.h:34-38 */
{
  int n = 0;
  while (v) { n++; v >>= 1; }
  return n;
}

static long div_round_up(long a, long b) { return (a + b - 1) / b; } /* jutils.c:79 */

/* ------------------------------------------------------------------------------------------
 * Parameters: jpeg_set_defaults (jcparam.c:386-519), jpeg_set_quality (jcparam.c:360-380),
 * jpeg_add_quant_table (jcparam.c:30-68), jpeg_set_colorspace (jcparam.c:573-650).
 * Base tables: index 0 = Annex K (jcparam.c:76-99 luma, :180-190 chroma),
 *              index 3 = the max-compression default (jcparam.c:111-122 == :218-229).
 * ------------------------------------------------------------------------------------------ */
static const unsigned BASE_LUMA0[64] = {
  16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
  14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
  18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
  49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99
};
static const unsigned BASE_CHROMA0[64] = {
  17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
  24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99
};
static const unsigned BASE_3[64] = {
  16, 16, 16, 18, 25, 37, 56, 85, 16, 17, 20, 27, 34, 40, 53, 75,
  16, 20, 24, 31, 43, 62, 91, 135, 18, 27, 31, 40, 53, 74, 106, 156,
  25, 34, 43, 53, 69, 94, 131, 189, 37, 40, 62, 74, 94, 124, 169, 238,
  56, 53, 91, 106, 131, 169, 226, 311, 85, 75, 135, 156, 189, 238, 311, 418
};

static void add_quant_table(uint16_t *dst, const unsigned *base, int scale, int force_baseline)
{
  int i;
  for (i = 0; i < 64; i++) {
    long t = ((long)base[i] * scale + 50L) / 100L;
    if (t <= 0) t = 1;
    if (t > 32767) t = 32767;
    if (force_baseline && t > 255) t = 255;
    dst[i] = (uint16_t)t;
  }
}

void mjo_default_params(mjo_params *p, int width, int height, int input_components,
                        int gray_output, int quality, int force_baseline, int profile_fastest,
                        int hsamp, int vsamp, int base_quant_tbl_idx)
{
  float q = (float)quality;
  int scale, i;
  memset(p, 0, sizeof(*p));
  p->width = width;
  p->height = height;
  p->input_components = input_components;
  p->fastest_profile = profile_fastest;
  /* jpeg_float_quality_scaling, jcparam.c:340-357, truncated to int (rdswitch.c:545) */
  if (q <= 0.f) q = 1.f;
  if (q > 100.f) q = 100.f;
  if (q < 50.f) q = 5000.f / q; else q = 200.f - q * 2.f;
  scale = (int)q;
  if (base_quant_tbl_idx < 0) base_quant_tbl_idx = profile_fastest ? 0 : 3;
  if (base_quant_tbl_idx == 3) {
    add_quant_table(p->qtbl[0], BASE_3, scale, force_baseline);
    add_quant_table(p->qtbl[1], BASE_3, scale, force_baseline);
  } else {
    add_quant_table(p->qtbl[0], BASE_LUMA0, scale, force_baseline);
    add_quant_table(p->qtbl[1], BASE_CHROMA0, scale, force_baseline);
  }
  if (input_components == 1 || gray_output) {
    p->num_components = 1;
    p->component_id[0] = 1;
    p->h_samp[0] = p->v_samp[0] = 1;
  } else {
    p->num_components = 3;
    for (i = 0; i < 3; i++) {
      p->component_id[i] = i + 1;
      p->h_samp[i] = p->v_samp[i] = 1;
      p->quant_tbl_no[i] = p->dc_tbl_no[i] = p->ac_tbl_no[i] = (i > 0);
    }
    p->h_samp[0] = hsamp;
    p->v_samp[0] = vsamp;
  }
  p->write_jfif = 1;
  p->optimize_coding = !profile_fastest;
  p->trellis_quant = !profile_fastest;
  p->trellis_quant_dc = 1;
  { int t; for (t = 0; t < 4; t++) { p->arith_dc_L[t] = 0; p->arith_dc_U[t] = 1; p->arith_ac_K[t] = 5; } }   /* jcparam.c:417-419 */
  p->overshoot_deringing = !profile_fastest;
  p->lambda_log_scale1 = 14.75f;
  p->lambda_log_scale2 = 16.5f;
  p->num_scans = 0;
  p->optimize_scans = 0;
}

void mjo_set_rgb_output(mjo_params *p)
{ /* jpeg_set_colorspace jcparam.c:611-619 */
  int i;
  p->rgb_output = 1;
  p->write_jfif = 0;
  p->num_components = 3;
  for (i = 0; i < 3; i++) {
    p->component_id[i] = "RGB"[i];
    p->h_samp[i] = p->v_samp[i] = 1;
    p->quant_tbl_no[i] = p->dc_tbl_no[i] = p->ac_tbl_no[i] = 0;
  }
}

static mjo_scan *fill_a_scan(mjo_scan *s, int ci, int Ss, int Se, int Ah, int Al)
{
  s->comps_in_scan = 1; s->component_index[0] = ci;
  s->Ss = Ss; s->Se = Se; s->Ah = Ah; s->Al = Al;
  return s + 1;
}
static mjo_scan *fill_dc_scans(mjo_scan *s, int ncomps, int Ah, int Al)
{ /* jcparam.c:700-725 (ncomps <= MAX_COMPS_IN_SCAN: one interleaved DC scan) */
  int ci;
  s->comps_in_scan = ncomps;
  for (ci = 0; ci < ncomps; ci++) s->component_index[ci] = ci;
  s->Ss = s->Se = 0; s->Ah = Ah; s->Al = Al;
  return s + 1;
}

/* jpeg_simple_progression, jcparam.c:859-1004 (dc_scan_opt_mode :887-892, :934-947) */
void mjo_simple_progression(mjo_params *p)
{
  mjo_scan *s = p->scans;
  int ci, nc = p->num_components;
  p->optimize_scans = 0;
  if (nc == 3 && p->rgb_output) {   /* all-purpose script for other colour spaces, jcparam.c:985-1003 */
    const int mx = !p->fastest_profile;
    s = fill_dc_scans(s, nc, 0, mx ? 0 : 1);
    for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, mx ? 8 : 5, 0, 2);
    for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, mx ? 9 : 6, 63, 0, 2);
    for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 63, 2, 1);
    if (!mx) s = fill_dc_scans(s, nc, 1, 0);
    for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 63, 1, 0);
  } else if (nc == 3) {
    if (!p->fastest_profile) {
      if (p->dc_scan_opt_mode == 0) s = fill_dc_scans(s, nc, 0, 0);          /* one DC scan for all components */
      else if (p->dc_scan_opt_mode == 1) {                                   /* one per component */
        s = fill_a_scan(s, 0, 0, 0, 0, 0);
        s = fill_a_scan(s, 1, 0, 0, 0, 0);
        s = fill_a_scan(s, 2, 0, 0, 0, 0);
      } else {                                                               /* luma, then Cb+Cr interleaved (fill_a_scan_pair) */
        s = fill_dc_scans(s, 1, 0, 0);
        s->comps_in_scan = 2; s->component_index[0] = 1; s->component_index[1] = 2;
        s->Ss = s->Se = s->Ah = s->Al = 0; s++;
      }
      s = fill_a_scan(s, 0, 1, 8, 0, 2);
      s = fill_a_scan(s, 1, 1, 8, 0, 0);
      s = fill_a_scan(s, 2, 1, 8, 0, 0);
      s = fill_a_scan(s, 0, 9, 63, 0, 2);
      s = fill_a_scan(s, 0, 1, 63, 2, 1);
      s = fill_a_scan(s, 0, 1, 63, 1, 0);
      s = fill_a_scan(s, 1, 9, 63, 0, 0);
      s = fill_a_scan(s, 2, 9, 63, 0, 0);
    } else {
      s = fill_dc_scans(s, nc, 0, 1);
      s = fill_a_scan(s, 0, 1, 5, 0, 2);
      s = fill_a_scan(s, 2, 1, 63, 0, 1);
      s = fill_a_scan(s, 1, 1, 63, 0, 1);
      s = fill_a_scan(s, 0, 6, 63, 0, 2);
      s = fill_a_scan(s, 0, 1, 63, 2, 1);
      s = fill_dc_scans(s, nc, 1, 0);
      s = fill_a_scan(s, 2, 1, 63, 1, 0);
      s = fill_a_scan(s, 1, 1, 63, 1, 0);
      s = fill_a_scan(s, 0, 1, 63, 1, 0);
    }
  } else {
    if (!p->fastest_profile) {
      s = fill_dc_scans(s, nc, 0, 0);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 8, 0, 2);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 9, 63, 0, 2);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 63, 2, 1);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 63, 1, 0);
    } else {
      s = fill_dc_scans(s, nc, 0, 1);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 5, 0, 2);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 6, 63, 0, 2);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 63, 2, 1);
      s = fill_dc_scans(s, nc, 1, 0);
      for (ci = 0; ci < nc; ci++) s = fill_a_scan(s, ci, 1, 63, 1, 0);
    }
  }
  p->num_scans = (int)(s - p->scans);
}

/* jpeg_search_progression, jcparam.c:733-852 (dc_scan_opt_mode :791-794: scan 0 is the luma DC alone unless mode 0) */
void mjo_search_progression(mjo_params *p)
{
  static const int fs[5] = { 2, 8, 5, 12, 18 };
  mjo_scan *s = p->scans;
  int Al, i, nc = p->num_components;
  if (nc == 3 && p->rgb_output) { mjo_simple_progression(p); return; }   /* the search knows YCbCr and gray only (jcparam.c:749-757) */
  p->optimize_scans = 1;
  s = fill_dc_scans(s, p->dc_scan_opt_mode == 0 ? nc : 1, 0, 0);
  s = fill_a_scan(s, 0, 1, 8, 0, 0);
  s = fill_a_scan(s, 0, 9, 63, 0, 0);
  for (Al = 0; Al < 3; Al++) {
    s = fill_a_scan(s, 0, 1, 63, Al + 1, Al);
    s = fill_a_scan(s, 0, 1, 8, 0, Al + 1);
    s = fill_a_scan(s, 0, 9, 63, 0, Al + 1);
  }
  s = fill_a_scan(s, 0, 1, 63, 0, 0);
  for (i = 0; i < 5; i++) {
    s = fill_a_scan(s, 0, 1, fs[i], 0, 0);
    s = fill_a_scan(s, 0, fs[i] + 1, 63, 0, 0);
  }
  if (nc == 3) {
    s->comps_in_scan = 2; s->component_index[0] = 1; s->component_index[1] = 2;
    s->Ss = s->Se = s->Ah = s->Al = 0; s++;
    s = fill_a_scan(s, 1, 0, 0, 0, 0);
    s = fill_a_scan(s, 2, 0, 0, 0, 0);
    s = fill_a_scan(s, 1, 1, 8, 0, 0);
    s = fill_a_scan(s, 1, 9, 63, 0, 0);
    s = fill_a_scan(s, 2, 1, 8, 0, 0);
    s = fill_a_scan(s, 2, 9, 63, 0, 0);
    for (Al = 0; Al < 2; Al++) {
      s = fill_a_scan(s, 1, 1, 63, Al + 1, Al);
      s = fill_a_scan(s, 2, 1, 63, Al + 1, Al);
      s = fill_a_scan(s, 1, 1, 8, 0, Al + 1);
      s = fill_a_scan(s, 1, 9, 63, 0, Al + 1);
      s = fill_a_scan(s, 2, 1, 8, 0, Al + 1);
      s = fill_a_scan(s, 2, 9, 63, 0, Al + 1);
    }
    s = fill_a_scan(s, 1, 1, 63, 0, 0);
    s = fill_a_scan(s, 2, 1, 63, 0, 0);
    for (i = 0; i < 5; i++) {
      s = fill_a_scan(s, 1, 1, fs[i], 0, 0);
      s = fill_a_scan(s, 1, fs[i] + 1, 63, 0, 0);
      s = fill_a_scan(s, 2, 1, fs[i], 0, 0);
      s = fill_a_scan(s, 2, fs[i] + 1, 63, 0, 0);
    }
  }
  p->num_scans = (int)(s - p->scans);
}

void mjo_set_dc_scan_opt_mode(mjo_params *p, int mode)
{ /* jcext.c:192, then what cjpeg does next: jpeg_simple_progression (cjpeg.c:747-749), which builds the search script
   * when optimize_scans is on (jcparam.c:866-869) */
  p->dc_scan_opt_mode = mode;
  if (p->num_scans > 0) {
    if (p->optimize_scans) mjo_search_progression(p);
    else mjo_simple_progression(p);
  }
}

/* initial_setup, jcmaster.c:163-259; per_scan_setup :561-566; jccoefct.c:587-601 (padding) */
void mjo_geometry(const mjo_params *p, mjo_geom g[MJO_MAX_COMPS], int *mcus_per_row, int *mcu_rows)
{
  int ci, maxh = 1, maxv = 1;
  for (ci = 0; ci < p->num_components; ci++) {
    if (p->h_samp[ci] > maxh) maxh = p->h_samp[ci];
    if (p->v_samp[ci] > maxv) maxv = p->v_samp[ci];
  }
  if (p->num_components == 1) {   /* one component: never interleaved -- no dummy blocks, an MCU is a block (per_scan_setup jcmaster.c:548-575; encode_core) */
    g[0].wib = g[0].wpad = (int)div_round_up((long)p->width, 8);
    g[0].hib = g[0].hpad = (int)div_round_up((long)p->height, 8);
    g[0].pw = g[0].wib * 8; g[0].ph = g[0].hib * 8;
    if (mcus_per_row) *mcus_per_row = g[0].wib;
    if (mcu_rows) *mcu_rows = g[0].hib;
    return;
  }
  for (ci = 0; ci < p->num_components; ci++) {
    g[ci].wib = (int)div_round_up((long)p->width * p->h_samp[ci], (long)maxh * 8);
    g[ci].hib = (int)div_round_up((long)p->height * p->v_samp[ci], (long)maxv * 8);
    g[ci].wpad = (int)(div_round_up(g[ci].wib, p->h_samp[ci]) * p->h_samp[ci]);
    g[ci].hpad = (int)(div_round_up(g[ci].hib, p->v_samp[ci]) * p->v_samp[ci]);
    g[ci].pw = g[ci].wib * 8;
    g[ci].ph = g[ci].hib * 8;
  }
  if (mcus_per_row) *mcus_per_row = (int)div_round_up(p->width, (long)maxh * 8);
  if (mcu_rows) *mcu_rows = (int)div_round_up(p->height, (long)maxv * 8);
}

/* ------------------------------------------------------------------------------------------
 * a1-a3  Colour conversion + downsampling + edge replication.
 *   rgb_ycc_convert  jccolext.c:30-75, tables jccolor.c:213-246 (FIX(x) = (int)(x*65536+0.5))
 *   rgb_gray_convert jccolext.c:88-118 (same Y row of the table)
 *   downsampling     jcsample.c:151-295 (int_downsample covers every ratio; h2v1/h2v2 are the
 *                    special cases with the alternating bias :247,:286)
 *   edges            expand_right_edge jcsample.c:98; expand_bottom_edge jcprepct.c:113,
 *                    call sites :161-168 (input rows, BEFORE downsampling) and :180-190
 *                    (downsampled rows, to the iMCU height)
 * Whole-image formulation: every replicated sample is an index clamp.
 * ------------------------------------------------------------------------------------------ */
#define FIXC(x) ((int)((x) * 65536.0 + 0.5))

static int prec_of(const mjo_params *p) { return p->data_precision == 12 ? 12 : 8; }

static void convert_pixel(const mjo_params *p, const uint8_t *px, int out[3])
{
  const int P = prec_of(p);
  const int center = 1 << (P - 1);
  int v[3], i;
  /* 12-bit samples are uint16 and are masked to 12 bits (RANGE_LIMIT, jccolor.c:96-101) */
  for (i = 0; i < p->input_components; i++)
    v[i] = P == 12 ? (((const uint16_t *)px)[i] & 0xFFF) : px[i];
  if (p->input_components == 1) {
    out[0] = P == 12 ? ((const uint16_t *)px)[0] : v[0];   /* grayscale_convert / null path: no masking */
    return;
  }
  if (p->ycc_input && p->num_components == 1) {   /* YCbCr in, grayscale out: grayscale_convert takes the Y samples (jccolor.c:448-466) */
    out[0] = P == 12 ? ((const uint16_t *)px)[0] : px[0];
    return;
  }
  if (p->rgb_output || (p->ycc_input && p->num_components == 3)) {   /* null_convert jccolor.c:479: samples copied as they are */
    for (i = 0; i < 3; i++) out[i] = P == 12 ? ((const uint16_t *)px)[i] : px[i];
    return;
  }
  {
    long r = v[0], g = v[1], b = v[2];
    out[0] = (int)((FIXC(0.29900) * r + FIXC(0.58700) * g + FIXC(0.11400) * b + 32768) >> 16);
    if (p->num_components == 3) {
      out[1] = (int)((-FIXC(0.16874) * r - FIXC(0.33126) * g + FIXC(0.50000) * b + ((long)center << 16) + 32767) >> 16);
      out[2] = (int)((FIXC(0.50000) * r - FIXC(0.41869) * g - FIXC(0.08131) * b + ((long)center << 16) + 32767) >> 16);
    }
  }
}

static void color_downsample16(const mjo_params *p, const uint8_t *pixels, size_t row_stride, uint16_t *planes[MJO_MAX_COMPS])
{
  mjo_geom g[MJO_MAX_COMPS];
  int ci, maxh = 1, maxv = 1, W = p->width, H = p->height;
  int groups; /* number of input row groups, jcprepct.c:135-192 */
  uint16_t *full[3] = { 0, 0, 0 };
  int x, y;
  const int bps = prec_of(p) == 12 ? 2 : 1;
  mjo_geometry(p, g, NULL, NULL);
  for (ci = 0; ci < p->num_components; ci++) {
    if (p->h_samp[ci] > maxh) maxh = p->h_samp[ci];
    if (p->v_samp[ci] > maxv) maxv = p->v_samp[ci];
  }
  groups = (int)div_round_up(H, maxv);
  /* full-resolution converted planes (the colour buffer of jcprepct.c, whole image) */
  for (ci = 0; ci < p->num_components; ci++) full[ci] = (uint16_t *)malloc((size_t)W * H * 2);
  for (y = 0; y < H; y++) {
    const uint8_t *row = pixels + (size_t)y * row_stride;
    for (x = 0; x < W; x++) {
      int v[3];
      convert_pixel(p, row + (size_t)x * p->input_components * bps, v);
      for (ci = 0; ci < p->num_components; ci++) full[ci][(size_t)y * W + x] = (uint16_t)v[ci];
    }
  }
  for (ci = 0; ci < p->num_components; ci++) {
    int hexp = maxh / p->h_samp[ci], vexp = maxv / p->v_samp[ci];
    int v = p->v_samp[ci];
    int real_rows = groups * v; /* downsampled rows that exist before the iMCU padding */
    int numpix = hexp * vexp;
    int r, c;
    if (p->smoothing_factor) {
      /* Input smoothing (cjpeg -smooth N).  need_context_rows switches the WHOLE preprocessor to
       * pre_process_context (jcprepct.c:200-262): rows above the image are copies of row 0 (:224-232), every row
       * below it is a copy of the last input row (:243-249, the padding groups are downsampled like real ones --
       * there is no "replicate the last downsampled row" step in this mode), columns are replicated to the right by
       * expand_right_edge and column -1 counts as column 0: all of it is index clamping.  Only the full-size and the
       * 2x2 downsamplers have smoothing variants (jinit_downsampler jcsample.c:486-535). */
      const long sf = p->smoothing_factor;
      for (r = 0; r < g[ci].ph; r++)
        for (c = 0; c < g[ci].pw; c++) {
#define PX(yy, xx) ((long)full[ci][(size_t)((yy) < 0 ? 0 : (yy) > H - 1 ? H - 1 : (yy)) * W + ((xx) < 0 ? 0 : (xx) > W - 1 ? W - 1 : (xx))])
          long val;
          if (hexp == 1 && vexp == 1) {             /* fullsize_smooth_downsample jcsample.c:400-455 */
            long member = PX(r, c), neigh = 0;
            int dy, dx;
            for (dy = -1; dy <= 1; dy++)
              for (dx = -1; dx <= 1; dx++)
                if (dy || dx) neigh += PX(r + dy, c + dx);
            val = (member * (65536L - sf * 512L) + neigh * (sf * 64) + 32768) >> 16;
          } else if (hexp == 2 && vexp == 2) {      /* h2v2_smooth_downsample jcsample.c:304-391 */
            int y0 = 2 * r, x0 = 2 * c;
            long member = PX(y0, x0) + PX(y0, x0 + 1) + PX(y0 + 1, x0) + PX(y0 + 1, x0 + 1);
            long edge = PX(y0 - 1, x0) + PX(y0 - 1, x0 + 1) + PX(y0 + 2, x0) + PX(y0 + 2, x0 + 1) +
                        PX(y0, x0 - 1) + PX(y0, x0 + 2) + PX(y0 + 1, x0 - 1) + PX(y0 + 1, x0 + 2);
            long corner = PX(y0 - 1, x0 - 1) + PX(y0 - 1, x0 + 2) + PX(y0 + 2, x0 - 1) + PX(y0 + 2, x0 + 2);
            val = (member * (16384 - sf * 80) + (2 * edge + corner) * (sf * 16) + 32768) >> 16;
          } else {                                   /* no smoothing variant: the plain downsamplers, context-mode rows */
            long sum = 0;
            int hh, vv;
            for (vv = 0; vv < vexp; vv++)
              for (hh = 0; hh < hexp; hh++) sum += PX(r * vexp + vv, c * hexp + hh);
            if (hexp == 2 && vexp == 1) val = (sum + (c & 1)) >> 1;
            else val = (sum + numpix / 2) / numpix;
          }
#undef PX
          planes[ci][(size_t)r * g[ci].pw + c] = (uint16_t)val;
        }
      continue;
    }
    for (r = 0; r < g[ci].ph; r++) {
      int rr = r < real_rows ? r : real_rows - 1; /* jcprepct.c:180-190 */
      int grp = rr / v, sub = rr % v;
      int in_row0 = grp * maxv + sub * vexp;
      for (c = 0; c < g[ci].pw; c++) {
        int sum = 0, hh, vv, val;
        for (vv = 0; vv < vexp; vv++) {
          int iy = in_row0 + vv;
          if (iy > H - 1) iy = H - 1; /* jcprepct.c:161-168 */
          for (hh = 0; hh < hexp; hh++) {
            int ix = c * hexp + hh;
            if (ix > W - 1) ix = W - 1; /* jcsample.c:98-116 */
            sum += full[ci][(size_t)iy * W + ix];
          }
        }
        if (hexp == 1 && vexp == 1) val = sum;                       /* fullsize_downsample :199 */
        else if (hexp == 2 && vexp == 1) val = (sum + (c & 1)) >> 1; /* h2v1 :226, bias 0,1,0,1 */
        else if (hexp == 2 && vexp == 2) val = (sum + 1 + (c & 1)) >> 2; /* h2v2 :263, bias 1,2,1,2 */
        else val = (sum + numpix / 2) / numpix;                      /* int_downsample :151 */
        planes[ci][(size_t)r * g[ci].pw + c] = (uint16_t)val;
      }
    }
  }
  for (ci = 0; ci < p->num_components; ci++) free(full[ci]);
}

void mjo_color_downsample(const mjo_params *p, const uint8_t *pixels, size_t row_stride, uint8_t *planes[MJO_MAX_COMPS])
{ /* 8-bit public tap */
  mjo_geom g[MJO_MAX_COMPS];
  uint16_t *p16[MJO_MAX_COMPS] = { 0, 0, 0, 0 };
  int ci;
  size_t i;
  mjo_geometry(p, g, NULL, NULL);
  for (ci = 0; ci < p->num_components; ci++) p16[ci] = (uint16_t *)malloc((size_t)g[ci].pw * g[ci].ph * 2);
  color_downsample16(p, pixels, row_stride, p16);
  for (ci = 0; ci < p->num_components; ci++) {
    for (i = 0; i < (size_t)g[ci].pw * g[ci].ph; i++) planes[ci][i] = (uint8_t)p16[ci][i];
    free(p16[ci]);
  }
}

/* ------------------------------------------------------------------------------------------
 * a5  preprocess_deringing jcdctmgr.c:416-498, catmull_rom :387-403
 * ------------------------------------------------------------------------------------------ */
static float catmull_rom(int v1, int v2, int v3, int v4, float t, int size)
{
  const int tan1 = (v3 - v1) * size;
  const int tan2 = (v4 - v2) * size;
  const float t2 = t * t;
  const float t3 = t2 * t;
  const float f1 = ((2.f * t3) - (3.f * t2)) + 1.f;
  const float f2 = (-2.f * t3) + (3.f * t2);
  const float f3 = (t3 - (2.f * t2)) + t;
  const float f4 = t3 - t2;
  float r = (float)v2 * f1;
  r = r + (float)tan1 * f3;
  r = r + (float)v3 * f2;
  r = r + (float)tan2 * f4;
  return r;
}

void mjo_deringing(int data[64], int q0)
{
  const int maxsample = 255 - 128;
  const int size = 64;
  int sum = 0, cnt = 0, i, n, maxovershoot, a, b;
  for (i = 0; i < size; i++) {
    sum += data[i];
    if (data[i] >= maxsample) cnt++;
  }
  if (!cnt || cnt == size) return;
  a = 2 * q0 < 31 ? 2 * q0 : 31;
  b = (maxsample * size - sum) / cnt;
  maxovershoot = maxsample + (a < b ? a : b);
  n = 0;
  do {
    int start, end, length, f1, f2, l1, l2, fslope, lslope;
    float step, position;
    if (data[ZZ[n]] < maxsample) { n++; continue; }
    start = n;
    while (++n < size && data[ZZ[n]] >= maxsample) {}
    end = n;
    f1 = data[ZZ[start >= 1 ? start - 1 : 0]];
    f2 = data[ZZ[start >= 2 ? start - 2 : 0]];
    l1 = data[ZZ[end < size - 1 ? end : size - 1]];
    l2 = data[ZZ[end < size - 2 ? end + 1 : size - 1]];
    fslope = (f1 - f2) > (maxsample - f1) ? (f1 - f2) : (maxsample - f1);
    lslope = (l1 - l2) > (maxsample - l1) ? (l1 - l2) : (maxsample - l1);
    if (start == 0) fslope = lslope;
    if (end == size) lslope = fslope;
    length = end - start;
    step = 1.f / (float)(length + 1);
    position = step;
    for (i = start; i < end; i++, position += step) {
      int tmp = (int)ceilf(catmull_rom(maxsample - fslope, maxsample, maxsample, maxsample - lslope, position, length));
      data[ZZ[i]] = tmp < maxovershoot ? tmp : maxovershoot;
    }
    n++;
  } while (n < size);
}

/* ------------------------------------------------------------------------------------------
 * a6  jpeg_fdct_islow jfdctint.c:142-286 (8-bit: CONST_BITS 13, PASS1_BITS 2)
 * ------------------------------------------------------------------------------------------ */
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
static void fdct_1d(int *d, int stride, int pass, int p1)
{ /* p1 = PASS1_BITS: 2 for 8-bit, 1 for 12-bit samples (jfdctint.c:80-86) */
  int t0 = d[0] + d[7 * stride], t7 = d[0] - d[7 * stride];
  int t1 = d[stride] + d[6 * stride], t6 = d[stride] - d[6 * stride];
  int t2 = d[2 * stride] + d[5 * stride], t5 = d[2 * stride] - d[5 * stride];
  int t3 = d[3 * stride] + d[4 * stride], t4 = d[3 * stride] - d[4 * stride];
  int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  int z1, z2, z3, z4, z5;
  int sh = pass == 0 ? 13 - p1 : 13 + p1;
  if (pass == 0) {
    d[0] = (t10 + t11) * (1 << p1);
    d[4 * stride] = (t10 - t11) * (1 << p1);
  } else {
    d[0] = DESCALE(t10 + t11, p1);
    d[4 * stride] = DESCALE(t10 - t11, p1);
  }
  z1 = (t12 + t13) * 4433;
  d[2 * stride] = DESCALE(z1 + t13 * 6270, sh);
  d[6 * stride] = DESCALE(z1 + t12 * (-15137), sh);
  z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
  z5 = (z3 + z4) * 9633;
  t4 *= 2446; t5 *= 16819; t6 *= 25172; t7 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
  z3 += z5; z4 += z5;
  d[7 * stride] = DESCALE(t4 + z1 + z3, sh);
  d[5 * stride] = DESCALE(t5 + z2 + z4, sh);
  d[3 * stride] = DESCALE(t6 + z2 + z3, sh);
  d[stride] = DESCALE(t7 + z1 + z4, sh);
}

static void fdct_islow_p(int data[64], int p1)
{
  int i;
  for (i = 0; i < 8; i++) fdct_1d(data + 8 * i, 1, 0, p1);
  for (i = 0; i < 8; i++) fdct_1d(data + i, 8, 1, p1);
}

void mjo_fdct_islow(int data[64]) { fdct_islow_p(data, 2); }

/* ------------------------------------------------------------------------------------------
 * jpeg_fdct_ifast jfdctfst.c:117-227: Arai / Agui / Nakajima, five multiplies per 1-D pass by constants of 8 fractional
 * bits, the product shifted down WITHOUT rounding (DESCALE is a plain right shift there, :101-104); both passes alike, no
 * scaling between them.  DCTELEM is an int in the C build (jdct.h), nothing wraps.  The output carries the AA&N scale
 * factors, which the divisors absorb (mjo_ifast_divisor).
 * ------------------------------------------------------------------------------------------ */
static void fdct_ifast_1d(int *d, int stride)
{
  const int t0 = d[0] + d[7 * stride], t7 = d[0] - d[7 * stride], t1 = d[stride] + d[6 * stride], t6 = d[stride] - d[6 * stride];
  const int t2 = d[2 * stride] + d[5 * stride], t5 = d[2 * stride] - d[5 * stride], t3 = d[3 * stride] + d[4 * stride], t4 = d[3 * stride] - d[4 * stride];
  const int e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
  const int o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
  int z1, z2, z3, z4, z5, z11, z13;
  d[0] = e0 + e1;
  d[4 * stride] = e0 - e1;
  z1 = ((e2 + e3) * 181) >> 8;
  d[2 * stride] = e3 + z1;
  d[6 * stride] = e3 - z1;
  z5 = ((o0 - o2) * 98) >> 8;
  z2 = ((o0 * 139) >> 8) + z5;
  z4 = ((o2 * 334) >> 8) + z5;
  z3 = (o1 * 181) >> 8;
  z11 = t7 + z3; z13 = t7 - z3;
  d[5 * stride] = z13 + z2;
  d[3 * stride] = z13 - z2;
  d[stride] = z11 + z4;
  d[7 * stride] = z11 - z4;
}
static void fdct_ifast(int data[64])
{
  int i;
  for (i = 0; i < 8; i++) fdct_ifast_1d(data + 8 * i, 1);
  for (i = 0; i < 8; i++) fdct_ifast_1d(data + i, 8);
}
/* scalefactor[row] * scalefactor[col] * 2^14, natural order (jcdctmgr.c:302-312, the same table again at :733-743) */
static const short kAanScales[64] = {
  16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
  21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906, 19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
  16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 12873, 17855, 16819, 15137, 12873, 10114, 6967, 3552,
  8867, 12299, 11585, 10426, 8867, 6967, 4799, 2446, 4520, 6270, 5906, 5315, 4520, 3552, 2446, 1247 };
/* the divisor of natural-order position i: quantval * aanscale * 8 / 2^14, rounded (:327-335), as compute_reciprocal's UINT16 argument */
int mjo_ifast_divisor(int quantval, int i) { return (int)((((long)quantval * kAanScales[i] + 1024L) >> 11) & 0xFFFF); }
/* 12-bit samples: the same product kept as a DCTELEM (an int), divided by directly (jcdctmgr.c:337-341, quantize :655-684) */
static int ifast_divisor12(int quantval, int i) { return (int)(((long)quantval * kAanScales[i] + 1024L) >> 11); }

/* ------------------------------------------------------------------------------------------
 * a4,a7,a8  convsamp jcdctmgr.c:576, quantize :611 (the reciprocal form there is, for d = 8*q,
 * identical to sign(x)*((|x| + d/2) / d): SURVEY 8a row a7), forward_DCT :693-772,
 * compress_first_pass jccoefct.c:262-353 (dummy blocks :312-345).
 * ------------------------------------------------------------------------------------------ */
static void build_dummies(const mjo_params *p, const mjo_geom *g, int ci, int16_t *coef)
{
  int h = p->h_samp[ci];
  int r, c;
  for (r = 0; r < g->hib; r++) {
    for (c = g->wib; c < g->wpad; c++) {
      int16_t *b = coef + ((size_t)r * g->wpad + c) * 64;
      memset(b, 0, 128);
      b[0] = coef[((size_t)r * g->wpad + g->wib - 1) * 64];
    }
  }
  for (r = g->hib; r < g->hpad; r++) {
    for (c = 0; c < g->wpad; c++) {
      int16_t *b = coef + ((size_t)r * g->wpad + c) * 64;
      int src = (c / h) * h + h - 1;
      memset(b, 0, 128);
      b[0] = coef[((size_t)(r - 1) * g->wpad + src) * 64];
    }
  }
}

static void forward16(const mjo_params *p, uint16_t *const planes[MJO_MAX_COMPS],
                      int16_t *coef_uq[MJO_MAX_COMPS], int16_t *coef_q[MJO_MAX_COMPS])
{
  const int P = prec_of(p), center = 1 << (P - 1), maxval = (1 << (P + 2)) - 1;
  mjo_geom g[MJO_MAX_COMPS];
  int ci;
  mjo_geometry(p, g, NULL, NULL);
  for (ci = 0; ci < p->num_components; ci++) {
    const uint16_t *qt = p->qtbl[p->quant_tbl_no[ci]];
    int br, bc, i;
    memset(coef_uq[ci], 0, (size_t)g[ci].hpad * g[ci].wpad * 128);
    for (br = 0; br < g[ci].hib; br++) {
      for (bc = 0; bc < g[ci].wib; bc++) {
        int ws[64];
        int16_t *uq = coef_uq[ci] + ((size_t)br * g[ci].wpad + bc) * 64;
        int16_t *q = coef_q[ci] + ((size_t)br * g[ci].wpad + bc) * 64;
        for (i = 0; i < 64; i++)
          ws[i] = (int)planes[ci][(size_t)(br * 8 + i / 8) * g[ci].pw + bc * 8 + (i & 7)] - center;
        if (p->overshoot_deringing) mjo_deringing(ws, qt[0]);
        if (p->dct_method == 1) fdct_ifast(ws);
        else fdct_islow_p(ws, P == 12 ? 1 : 2);
        for (i = 0; i < 64; i++) {
          /* 8-bit build: the divisor reaches compute_reciprocal as a UINT16 (jcdctmgr.c:182, :278-282), so a step of 8192 or more
           * wraps -- q = 8450 (quality 1) divides by 67600 mod 65536 = 2064; the 12-bit build keeps the value (:284).  (A wrapped
           * divisor of 0 makes the reference divide by zero: mjo_encode refuses such tables.)  The reciprocal form equals this
           * rounding division for every divisor 8 .. 65528 and |x| <= 32767 (checked exhaustively). */
          int d = p->dct_method == 1 ? (P == 12 ? ifast_divisor12(qt[i], i) : mjo_ifast_divisor(qt[i], i)) : P == 12 ? 8 * qt[i] : (int)((8u * (unsigned)qt[i]) & 0xFFFFu), x = ws[i], v;
          uq[i] = (int16_t)x;   /* (12-bit: may wrap; only the 8-bit trellis reads it) */
          if (p->dct_method == 1) {   /* what the trellis gets to see: the AA&N factors taken out again, forward_DCT jcdctmgr.c:745-750 (C division: towards zero) */
            const int sc = kAanScales[i];
            uq[i] = (int16_t)(x >= 0 ? (x * 32768 + sc) / (2 * sc) : (x * 32768 - sc) / (2 * sc));
          }
          v = ((x < 0 ? -x : x) + d / 2) / d;
          if (x < 0) v = -v;
          if (p->overshoot_deringing) { /* jcdctmgr.c:761-770 */
            if (v < -maxval) v = -maxval;
            if (v > maxval) v = maxval;
          }
          q[i] = (int16_t)v;
        }
      }
    }
    build_dummies(p, &g[ci], ci, coef_q[ci]);
  }
}

void mjo_forward(const mjo_params *p, uint8_t *const planes[MJO_MAX_COMPS],
                 int16_t *coef_uq[MJO_MAX_COMPS], int16_t *coef_q[MJO_MAX_COMPS])
{ /* 8-bit public tap */
  mjo_geom g[MJO_MAX_COMPS];
  uint16_t *p16[MJO_MAX_COMPS] = { 0, 0, 0, 0 };
  int ci;
  size_t i;
  mjo_geometry(p, g, NULL, NULL);
  for (ci = 0; ci < p->num_components; ci++) {
    p16[ci] = (uint16_t *)malloc((size_t)g[ci].pw * g[ci].ph * 2);
    for (i = 0; i < (size_t)g[ci].pw * g[ci].ph; i++) p16[ci][i] = planes[ci][i];
  }
  forward16(p, p16, coef_uq, coef_q);
  for (ci = 0; ci < p->num_components; ci++) free(p16[ci]);
}

/* ------------------------------------------------------------------------------------------
 * a11  jpeg_gen_optimal_table jchuff.c:947-1106; jpeg_make_c_derived_tbl :231-318
 * ------------------------------------------------------------------------------------------ */
void mjo_gen_optimal_table(long freq_in[257], uint8_t bits_out[17], uint8_t huffval[256])
{
  uint8_t bits[33];
  int bit_pos[33];
  int codesize[257], nz_index[257], others[257];
  long freq[257];
  int c1, c2, p, i, j, nnz = 0;
  long v, v2;
  memset(bits, 0, sizeof(bits));
  memset(codesize, 0, sizeof(codesize));
  for (i = 0; i < 257; i++) others[i] = -1;
  freq_in[256] = 1;
  for (i = 0; i < 257; i++) {
    if (freq_in[i]) {
      nz_index[nnz] = i;
      freq[nnz] = freq_in[i];
      nnz++;
    }
  }
  for (;;) {
    c1 = -1; c2 = -1;
    v = 1000000000L; v2 = 1000000000L;
    for (i = 0; i < nnz; i++) {
      if (freq[i] <= v2) {
        if (freq[i] <= v) { c2 = c1; v2 = v; v = freq[i]; c1 = i; }
        else { v2 = freq[i]; c2 = i; }
      }
    }
    if (c2 < 0) break;
    freq[c1] += freq[c2];
    freq[c2] = 1000000001L;
    codesize[c1]++;
    while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
    others[c1] = c2;
    codesize[c2]++;
    while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
  }
  for (i = 0; i < nnz; i++) bits[codesize[i]]++;
  p = 0;
  for (i = 1; i <= 32; i++) { bit_pos[i] = p; p += bits[i]; }
  for (i = 32; i > 16; i--) {
    while (bits[i] > 0) {
      j = i - 2;
      while (bits[j] == 0) j--;
      bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
    }
  }
  while (bits[i] == 0) i--;
  bits[i]--;
  memcpy(bits_out, bits, 17);
  for (i = 0; i < nnz - 1; i++) {
    huffval[bit_pos[codesize[i]]] = (uint8_t)nz_index[i];
    bit_pos[codesize[i]]++;
  }
}

typedef struct { uint8_t bits[17]; uint8_t huffval[256]; int sent; } htbl;
typedef struct { unsigned code[256]; uint8_t size[256]; } dtbl;

static void make_derived(const htbl *h, dtbl *d)
{
  char huffsize[257];
  unsigned huffcode[257], code = 0;
  int p = 0, l, i, si, lastp;
  for (l = 1; l <= 16; l++) { i = h->bits[l]; while (i--) huffsize[p++] = (char)l; }
  huffsize[p] = 0; lastp = p;
  si = huffsize[0]; p = 0;
  while (huffsize[p]) {
    while ((int)huffsize[p] == si) { huffcode[p++] = code; code++; }
    code <<= 1; si++;
  }
  memset(d, 0, sizeof(*d));
  for (p = 0; p < lastp; p++) {
    d->code[h->huffval[p]] = huffcode[p];
    d->size[h->huffval[p]] = (uint8_t)huffsize[p];
  }
}

/* Annex K.3 tables, jstdhuff.c:54-131 */
static const uint8_t STD_DC_L_BITS[17] = { 0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t STD_DC_C_BITS[17] = { 0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t STD_DC_VAL[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t STD_AC_L_BITS[17] = { 0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t STD_AC_L_VAL[162] = {
  0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
  0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
  0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
  0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
  0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
  0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
  0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
  0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
  0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
  0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
  0xf9, 0xfa
};
static const uint8_t STD_AC_C_BITS[17] = { 0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t STD_AC_C_VAL[162] = {
  0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
  0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
  0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
  0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
  0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
  0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
  0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
  0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
  0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
  0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
  0xf9, 0xfa
};

/* ------------------------------------------------------------------------------------------
 * Encoder state
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  uint8_t *buf;
  size_t len, cap;
} bytebuf;

static void bb_put(bytebuf *b, int v)
{
  if (b->len == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 4096;
    b->buf = (uint8_t *)realloc(b->buf, b->cap);
  }
  b->buf[b->len++] = (uint8_t)v;
}
static void bb_put2(bytebuf *b, int v) { bb_put(b, (v >> 8) & 0xFF); bb_put(b, v & 0xFF); }

typedef struct {
  const mjo_params *p;
  mjo_geom g[MJO_MAX_COMPS];
  int mcus_per_row, mcu_rows;
  int16_t *uq[MJO_MAX_COMPS], *q[MJO_MAX_COMPS];
  htbl dc[4], ac[4];
  int qsent[4];
  int last_restart_interval; /* jcmarker.c:660 */
  int progressive;
  int sof_hv0;     /* one component sampled other than 1x1: the SOF's sampling byte (encode_core) */
  int chain_v;     /* ... and its V: the trellis passes walk iMCU rows of V block rows (0: the component's own v_samp) */
  /* trellis_q_opt: sums over the blocks of sum(raw * quantized) and sum(8 * quantized^2) per table and coefficient
   * (jcdctmgr.c:1299-1306).  Every term is an integer and the totals stay far below 2^53, so the double sums are
   * exact in ANY order -- a parallel reduction reproduces them bit for bit */
  double norm_src[4][64], norm_coef[4][64];
} enc_t;

/* an entropy sink: counts (gather) or bits (emit) */
typedef struct {
  int gather;
  long *dc_count[4], *ac_count[4]; /* gather */
  const dtbl *dcd[4], *acd[4];     /* emit */
  bytebuf *out;
  uint64_t acc; /* bit accumulator, msb first */
  int nacc;
} sink_t;

static void put_bits(sink_t *s, unsigned code, int size)
{
  if (s->gather || size == 0) return;
  s->acc = (s->acc << size) | (code & ((1u << size) - 1));
  s->nacc += size;
  while (s->nacc >= 8) {
    int c = (int)((s->acc >> (s->nacc - 8)) & 0xFF);
    bb_put(s->out, c);
    if (c == 0xFF) bb_put(s->out, 0); /* jchuff.c:354-358 / jcphuff.c:347-352 */
    s->nacc -= 8;
  }
}
static void flush_bits(sink_t *s)
{ /* jchuff.c:505-514 and jcphuff.c:362-367: pad the partial byte with one-bits */
  if (s->gather) return;
  put_bits(s, 0x7F, 7);
  s->acc = 0;
  s->nacc = 0;
}

/* The scan being coded: per_scan_setup jcmaster.c:518-601 */
typedef struct {
  int ncomp;
  int comp[MJO_MAX_COMPS];
  int Ss, Se, Ah, Al;
  int mcus_per_row, mcu_rows;
  int restart_interval;
} scan_t;

static void setup_scan(const enc_t *e, scan_t *sc, const mjo_scan *ms)
{
  const mjo_params *p = e->p;
  int i;
  sc->ncomp = ms->comps_in_scan;
  for (i = 0; i < sc->ncomp; i++) sc->comp[i] = ms->component_index[i];
  sc->Ss = ms->Ss; sc->Se = ms->Se; sc->Ah = ms->Ah; sc->Al = ms->Al;
  if (sc->ncomp == 1) {
    sc->mcus_per_row = e->g[sc->comp[0]].wib;
    sc->mcu_rows = e->g[sc->comp[0]].hib;
  } else {
    sc->mcus_per_row = e->mcus_per_row;
    sc->mcu_rows = e->mcu_rows;
  }
  sc->restart_interval = p->restart_interval;
  if (p->restart_in_rows > 0) { /* jcmaster.c:595-600 */
    long nominal = (long)p->restart_in_rows * sc->mcus_per_row;
    sc->restart_interval = (int)(nominal < 65535L ? nominal : 65535L);
  }
}

/* ---- sequential Huffman coder: jchuff.c encode_one_block :563, htest_one_block :812,
 *      encode_mcu_huff :693, encode_mcu_gather :886, emit_restart :668 ---------------------- */
static void seq_block(sink_t *s, const int16_t *blk, int last_dc, int dctbl, int actbl)
{
  int temp = blk[0] - last_dc, temp2 = temp, nb, k, r;
  if (temp < 0) { temp = -temp; temp2--; }
  nb = nbits_of((unsigned)temp);
  if (s->gather) s->dc_count[dctbl][nb]++;
  else {
    put_bits(s, s->dcd[dctbl]->code[nb], s->dcd[dctbl]->size[nb]);
    put_bits(s, (unsigned)temp2, nb);
  }
  r = 0;
  for (k = 1; k < 64; k++) {
    temp = blk[ZZ[k]];
    if (temp == 0) { r++; continue; }
    while (r > 15) {
      if (s->gather) s->ac_count[actbl][0xF0]++;
      else put_bits(s, s->acd[actbl]->code[0xF0], s->acd[actbl]->size[0xF0]);
      r -= 16;
    }
    temp2 = temp;
    if (temp < 0) { temp = -temp; temp2--; }
    nb = nbits_of((unsigned)temp);
    if (s->gather) s->ac_count[actbl][(r << 4) + nb]++;
    else {
      put_bits(s, s->acd[actbl]->code[(r << 4) + nb], s->acd[actbl]->size[(r << 4) + nb]);
      put_bits(s, (unsigned)temp2, nb);
    }
    r = 0;
  }
  if (r > 0) {
    if (s->gather) s->ac_count[actbl][0]++;
    else put_bits(s, s->acd[actbl]->code[0], s->acd[actbl]->size[0]);
  }
}

/* ---- progressive coder state, jcphuff.c:90-130 -------------------------------------------- */
typedef struct {
  unsigned EOBRUN;
  unsigned BE;
  char bit_buffer[1000]; /* MAX_CORR_BITS jcphuff.c:108 */
  int ac_tbl;
} prog_t;

static void prog_symbol(sink_t *s, int isdc, int tbl, int sym)
{
  if (s->gather) { if (isdc) s->dc_count[tbl][sym]++; else s->ac_count[tbl][sym]++; }
  else {
    const dtbl *d = isdc ? s->dcd[tbl] : s->acd[tbl];
    put_bits(s, d->code[sym], d->size[sym]);
  }
}
static void prog_emit_eobrun(sink_t *s, prog_t *pg)
{ /* jcphuff.c:409-431 */
  if (pg->EOBRUN > 0) {
    int nb = nbits_of(pg->EOBRUN) - 1;
    unsigned i;
    prog_symbol(s, 0, pg->ac_tbl, nb << 4);
    if (nb) put_bits(s, pg->EOBRUN, nb);
    pg->EOBRUN = 0;
    for (i = 0; i < pg->BE; i++) put_bits(s, (unsigned)pg->bit_buffer[i], 1);
    pg->BE = 0;
  }
}

/* iterate over the MCUs of a scan in coding order: compress_output jccoefct.c:498-553 */
static void code_scan(enc_t *e, const scan_t *sc, sink_t *s)
{
  const mjo_params *p = e->p;
  int last_dc[MJO_MAX_COMPS] = { 0, 0, 0, 0 };
  int restarts_to_go = sc->restart_interval, next_restart = 0;
  prog_t pg;
  int mr, mc, ci;
  memset(&pg, 0, sizeof(pg));
  if (sc->ncomp == 1) pg.ac_tbl = p->ac_tbl_no[sc->comp[0]];

  for (mr = 0; mr < sc->mcu_rows; mr++) {
    for (mc = 0; mc < sc->mcus_per_row; mc++) {
      /* restart handling: jchuff.c:710-714,:894-903 / jcphuff.c:485-488, emit_restart :438 */
      if (sc->restart_interval && restarts_to_go == 0) {
        if (e->progressive) prog_emit_eobrun(s, &pg);
        if (!s->gather) {
          flush_bits(s);
          bb_put(s->out, 0xFF);
          bb_put(s->out, 0xD0 + next_restart);
        }
        if (!e->progressive || sc->Ss == 0) for (ci = 0; ci < sc->ncomp; ci++) last_dc[ci] = 0;
        if (e->progressive && sc->Ss != 0) { pg.EOBRUN = 0; pg.BE = 0; }
        restarts_to_go = sc->restart_interval;
        next_restart = (next_restart + 1) & 7;
      }
      for (ci = 0; ci < sc->ncomp; ci++) {
        int c = sc->comp[ci];
        int mw = sc->ncomp == 1 ? 1 : p->h_samp[c];
        int mh = sc->ncomp == 1 ? 1 : p->v_samp[c];
        int yi, xi;
        for (yi = 0; yi < mh; yi++) {
          for (xi = 0; xi < mw; xi++) {
            const int16_t *blk = e->q[c] + ((size_t)(mr * mh + yi) * e->g[c].wpad + (mc * mw + xi)) * 64;
            if (!e->progressive) {
              seq_block(s, blk, last_dc[ci], p->dc_tbl_no[c], p->ac_tbl_no[c]);
              last_dc[ci] = blk[0];
            } else if (sc->Ss == 0 && sc->Ah == 0) {
              /* encode_mcu_DC_first jcphuff.c:468-552 */
              int t2 = blk[0] >> sc->Al, t = t2 - last_dc[ci], nb;
              last_dc[ci] = t2;
              t2 = t;
              if (t < 0) { t = -t; t2--; }
              nb = nbits_of((unsigned)t);
              prog_symbol(s, 1, p->dc_tbl_no[c], nb);
              if (nb) put_bits(s, (unsigned)t2, nb);
            } else if (sc->Ss == 0) {
              /* encode_mcu_DC_refine jcphuff.c:746-790 */
              put_bits(s, (unsigned)(blk[0] >> sc->Al), 1);
            } else if (sc->Ah == 0) {
              /* encode_mcu_AC_first jcphuff.c:648-737 (+prepare :580-625) */
              int k, r = 0, any = 0;
              for (k = sc->Ss; k <= sc->Se; k++) {
                int t = blk[ZZ[k]], t2, nb;
                if (t == 0) { r++; continue; }
                if (t < 0) { t = -t; t >>= sc->Al; t2 = ~t; } else { t >>= sc->Al; t2 = t; }
                if (t == 0) { r++; continue; }
                if (!any && pg.EOBRUN > 0) prog_emit_eobrun(s, &pg);
                any = 1;
                while (r > 15) { prog_symbol(s, 0, pg.ac_tbl, 0xF0); r -= 16; }
                nb = nbits_of((unsigned)t);
                prog_symbol(s, 0, pg.ac_tbl, (r << 4) + nb);
                put_bits(s, (unsigned)t2, nb);
                r = 0;
              }
              if (r > 0) {
                pg.EOBRUN++;
                if (pg.EOBRUN == 0x7FFF) prog_emit_eobrun(s, &pg);
              }
            } else {
              /* encode_mcu_AC_refine jcphuff.c:918-1025 (+prepare :817-870) */
              int absv[64], k, EOB = 0, r = 0;
              unsigned BR = 0;
              char *BR_buffer = pg.bit_buffer + pg.BE;
              for (k = sc->Ss; k <= sc->Se; k++) {
                int t = blk[ZZ[k]];
                if (t < 0) t = -t;
                t >>= sc->Al;
                absv[k] = t;
                if (t == 1) EOB = k;
              }
              for (k = sc->Ss; k <= sc->Se; k++) {
                int t = absv[k];
                if (t == 0) { r++; continue; }
                while (r > 15 && k <= EOB) {
                  unsigned i;
                  prog_emit_eobrun(s, &pg);
                  prog_symbol(s, 0, pg.ac_tbl, 0xF0);
                  r -= 16;
                  for (i = 0; i < BR; i++) put_bits(s, (unsigned)BR_buffer[i], 1);
                  BR_buffer = pg.bit_buffer;
                  BR = 0;
                }
                if (t > 1) { BR_buffer[BR++] = (char)(t & 1); continue; }
                prog_emit_eobrun(s, &pg);
                prog_symbol(s, 0, pg.ac_tbl, (r << 4) + 1);
                put_bits(s, blk[ZZ[k]] < 0 ? 0u : 1u, 1);
                {
                  unsigned i;
                  for (i = 0; i < BR; i++) put_bits(s, (unsigned)BR_buffer[i], 1);
                }
                BR_buffer = pg.bit_buffer;
                BR = 0;
                r = 0;
              }
              if (r > 0 || BR > 0) {
                pg.EOBRUN++;
                pg.BE += BR;
                if (pg.EOBRUN == 0x7FFF || pg.BE > (1000 - 64 + 1)) prog_emit_eobrun(s, &pg);
              }
            }
          }
        }
      }
      if (sc->restart_interval) restarts_to_go--;
    }
  }
  /* finish_pass: jchuff.c:772-800 / jcphuff.c:1033-1048,:1060 */
  if (e->progressive) prog_emit_eobrun(s, &pg);
  flush_bits(s);
}

/* ------------------------------------------------------------------------------------------
 * a9  quantize_trellis jcdctmgr.c:936-1329 driven by compress_trellis_pass jccoefct.c:356-486.
 * One component; dctbl/actbl are the code LENGTHS of the component's current tables (T7).
 * ------------------------------------------------------------------------------------------ */
static void q_opt_accumulate(enc_t *e, int ci)
{ /* jcdctmgr.c:1299-1306 (run per block row there; the order does not matter, see enc_t) */
  const mjo_params *p = e->p;
  const mjo_geom *g = &e->g[ci];
  const int t = p->quant_tbl_no[ci];
  int br, bi, i;
  for (br = 0; br < g->hib; br++)
    for (bi = 0; bi < g->wib; bi++) {
      const int16_t *src = e->uq[ci] + ((size_t)br * g->wpad + bi) * 64;
      const int16_t *coef = e->q[ci] + ((size_t)br * g->wpad + bi) * 64;
      for (i = 1; i < 64; i++) {
        e->norm_src[t][i] += (int)src[i] * (int)coef[i];
        e->norm_coef[t][i] += 8 * (int)coef[i] * (int)coef[i];
      }
    }
}

static void trellis_component(enc_t *e, int ci, const dtbl *dctbl, const dtbl *actbl, int Ss, int Se)
{ /* Ss..Se: 1..63, or one of the two bands of use_scans_in_trellis (select_scan_parameters jcmaster.c:451-467) */
  const mjo_params *p = e->p;
  const mjo_geom *g = &e->g[ci];
  const uint16_t *qt = p->qtbl[p->quant_tbl_no[ci]];
  const int v = e->chain_v ? e->chain_v : p->v_samp[ci];
  /* trellis_eob_opt state of one block row (jcdctmgr.c:977-993) */
  float *azbc = NULL, *abc = NULL;
  int *block_run_start = NULL, *requires_eob = NULL;
  int ncand = 2 + 60 / qt[0]; /* get_num_dc_trellis_candidates :930-933 */
  float lambda_tbl[64];
  float *acc_dc[9];
  int *back_dc[9];
  int16_t *cand_dc[9];
  int i, j, k, l, br, bi;
  int last_dc = 0;
  int run_start[64];
  ncand |= 1;
  if (ncand > 9) ncand = 9;
  for (i = 0; i < 64; i++) lambda_tbl[i] = (float)(1.0 / (double)((int)qt[i] * (int)qt[i])); /* :1017-1021 */
  for (i = 0; i < 9; i++) {
    acc_dc[i] = (float *)malloc(sizeof(float) * g->wib);
    back_dc[i] = (int *)malloc(sizeof(int) * g->wib);
    cand_dc[i] = (int16_t *)malloc(sizeof(int16_t) * g->wib);
  }
  memset(run_start, 0, sizeof(run_start));
  if (p->trellis_eob_opt) {
    azbc = (float *)malloc(sizeof(float) * (g->wib + 1));
    abc = (float *)malloc(sizeof(float) * (g->wib + 1));
    block_run_start = (int *)calloc(g->wib, sizeof(int));
    requires_eob = (int *)malloc(sizeof(int) * (g->wib + 1));
  }

  for (br = 0; br < g->hib; br++) {
    if (br % v == 0) last_dc = 0; /* jccoefct.c:418: per iMCU row */
    if (p->trellis_eob_opt) { azbc[0] = 0.0f; abc[0] = 0.0f; requires_eob[0] = 0; }
    for (bi = 0; bi < g->wib; bi++) {
      const int16_t *src = e->uq[ci] + ((size_t)br * g->wpad + bi) * 64;
      int16_t *coef = e->q[ci] + ((size_t)br * g->wpad + bi) * 64;
      float azd[64], acost[64];
      float norm = 0.0f, lambda, lambda_dc, best_cost, cost_all_zeros, best_cost_skip;
      int last_coeff_idx, has_eob;
      for (i = 1; i < 64; i++) norm = norm + (float)((int)src[i] * (int)src[i]); /* :1027-1031 */
      norm = (float)((double)norm / 63.0);
      if (p->lambda_log_scale2 > 0.0f)
        lambda = (float)(pow(2.0, (double)p->lambda_log_scale1) * (double)1.0f /
                         (pow(2.0, (double)p->lambda_log_scale2) + (double)norm));
      else
        lambda = (float)(pow(2.0, (double)p->lambda_log_scale1 - 12.0) * (double)1.0f);
      lambda_dc = lambda * lambda_tbl[0];
      azd[Ss - 1] = 0.0f;
      acost[Ss - 1] = 0.0f;

      if (p->trellis_quant_dc) { /* :1044-1118 */
        int sign = src[0] >> 15 ? -1 : 0; /* src[bi][0] >> 31 on the promoted int */
        int x = src[0] < 0 ? -src[0] : src[0];
        int q = 8 * qt[0];
        int qval = (x + q / 2) / q;
        for (k = 0; k < ncand; k++) {
          int delta, dc_delta, bits;
          float dist, cost;
          int cnd = qval - ncand / 2 + k;
          if (cnd >= 1024) cnd = 1023;
          if (cnd <= -1024) cnd = -1023;
          delta = cnd * q - x;
          dist = (float)(delta * delta) * lambda_dc;
          cnd *= 1 + 2 * sign;
          cand_dc[k][bi] = (int16_t)cnd;
          if (br % v != 0 && p->trellis_delta_dc_weight > 0.0f) {
            /* :1069-1084: the block above exists only inside an iMCU row (compress_trellis_pass jccoefct.c:426-427 hands
             * over buffer[block_row-1], NULL for the first row); its quantized DC is final (that row's DC trellis ran) */
            const int dc_above_orig = e->uq[ci][((size_t)(br - 1) * g->wpad + bi) * 64];
            const int dc_above_recon = e->q[ci][((size_t)(br - 1) * g->wpad + bi) * 64] * q;
            const int dc_orig = src[0];
            const int dc_recon = cnd * q;
            float vertical_dist, t;
            delta = (dc_above_orig - dc_orig) - (dc_above_recon - dc_recon);
            vertical_dist = (float)(delta * delta) * lambda_dc;
            t = vertical_dist - dist;
            t = p->trellis_delta_dc_weight * t;
            dist = dist + t;
          }
          if (bi == 0) {
            dc_delta = cnd - last_dc;
            bits = nbits_of((unsigned)(dc_delta < 0 ? -dc_delta : dc_delta));
            cost = (float)(bits + dctbl->size[bits]) + dist;
            acc_dc[k][0] = cost;
            back_dc[k][0] = -1;
          } else {
            for (l = 0; l < ncand; l++) {
              dc_delta = cnd - cand_dc[l][bi - 1];
              bits = nbits_of((unsigned)(dc_delta < 0 ? -dc_delta : dc_delta));
              cost = (float)(bits + dctbl->size[bits]) + dist;
              cost = cost + acc_dc[l][bi - 1];
              if (l == 0 || cost < acc_dc[k][bi]) {
                acc_dc[k][bi] = cost;
                back_dc[k][bi] = l;
              }
            }
          }
        }
      }

      /* AC, :1120-1185 */
      for (i = Ss; i <= Se; i++) {
        int z = ZZ[i];
        int sign = src[z] < 0 ? -1 : 0;
        int x = src[z] < 0 ? -src[z] : src[z];
        int q = 8 * qt[z];
        int cand[16], cbits[16], ncd, qval;
        float cdist[16], t;
        t = (float)(x * x) * lambda;
        t = t * lambda_tbl[z];
        azd[i] = t + azd[i - 1];
        qval = (x + q / 2) / q;
        if (qval == 0) {
          coef[z] = 0;
          acost[i] = 1e38f;
          continue;
        }
        if (qval >= 1024) qval = 1023;
        ncd = nbits_of((unsigned)qval);
        for (k = 0; k < ncd; k++) {
          int delta;
          cand[k] = (k < ncd - 1) ? (2 << k) - 1 : qval;
          delta = cand[k] * q - x;
          cbits[k] = k + 1;
          t = (float)(delta * delta) * lambda;
          cdist[k] = t * lambda_tbl[z];
        }
        acost[i] = 1e38f;
        for (j = Ss - 1; j < i; j++) {
          int zz = ZZ[j], zero_run, run_bits;
          if (j != Ss - 1 && coef[zz] == 0) continue;
          zero_run = i - 1 - j;
          if ((zero_run >> 4) && actbl->size[0xF0] == 0) continue;
          run_bits = (zero_run >> 4) * actbl->size[0xF0];
          zero_run &= 15;
          for (k = 0; k < ncd; k++) {
            int coef_bits = actbl->size[16 * zero_run + cbits[k]];
            int rate;
            float cost, rhs;
            if (coef_bits == 0) continue;
            rate = coef_bits + cbits[k] + run_bits;
            cost = (float)rate + cdist[k];
            rhs = azd[i - 1] - azd[j];
            rhs = rhs + acost[j];
            cost = cost + rhs;
            if (cost < acost[i]) {
              coef[z] = (int16_t)((cand[k] ^ sign) - sign);
              acost[i] = cost;
              run_start[i] = j;
            }
          }
        }
      }
      /* EOB choice :1187-1207 */
      last_coeff_idx = Ss - 1;
      best_cost = azd[Se] + (float)actbl->size[0];
      cost_all_zeros = azd[Se];
      best_cost_skip = cost_all_zeros;
      for (i = Ss; i <= Se; i++) {
        int z = ZZ[i];
        if (coef[z] != 0) {
          float cost = acost[i] + azd[Se], cost_wo_eob;
          cost = cost - azd[i];
          cost_wo_eob = cost;
          if (i < Se) cost = cost + (float)actbl->size[0];
          if (cost < best_cost) { best_cost = cost; last_coeff_idx = i; best_cost_skip = cost_wo_eob; }
        }
      }
      has_eob = (last_coeff_idx < Se) + (last_coeff_idx == Ss - 1);
      /* back-track :1211-1222 */
      i = Se;
      while (i >= Ss) {
        while (i > last_coeff_idx) { coef[ZZ[i]] = 0; i--; }
        last_coeff_idx = run_start[i];
        i--;
      }
      if (p->trellis_eob_opt) { /* :1224-1256: cheapest way to reach block bi through runs of all-zero blocks */
        azbc[bi + 1] = azbc[bi];
        azbc[bi + 1] += cost_all_zeros;
        requires_eob[bi + 1] = has_eob;
        best_cost = 1e38f;
        if (has_eob != 2) {
          for (i = 0; i <= bi; i++) {
            int zero_block_run, nb;
            float cost;
            if (requires_eob[i] == 2) continue;
            cost = best_cost_skip;
            cost += azbc[bi];
            cost -= azbc[i];
            cost += abc[i];
            zero_block_run = bi - i + requires_eob[i];
            nb = nbits_of((unsigned)zero_block_run);
            cost += (float)(actbl->size[16 * nb] + nb);
            if (cost < best_cost) { block_run_start[bi] = i; best_cost = cost; abc[bi + 1] = cost; }
          }
        }
      }
    }
    if (p->trellis_eob_opt) { /* :1259-1293: choose the end of the last run, then zero the blocks inside the chosen runs */
      const int num_blocks = g->wib;
      int last_block = num_blocks;
      float best = 1e38f;
      for (i = 0; i <= num_blocks; i++) {
        int zero_block_run, nb;
        float cost = 0.0f;
        if (requires_eob[i] == 2) continue;
        cost += azbc[num_blocks];
        cost -= azbc[i];
        zero_block_run = num_blocks - i + requires_eob[i];
        nb = nbits_of((unsigned)zero_block_run);
        cost += (float)(actbl->size[16 * nb] + nb);
        if (cost < best) { best = cost; last_block = i; }
      }
      last_block--;
      bi = num_blocks - 1;
      while (bi >= 0) {
        while (bi > last_block) {
          int16_t *coef = e->q[ci] + ((size_t)br * g->wpad + bi) * 64;
          for (j = Ss; j <= Se; j++) coef[ZZ[j]] = 0;
          bi--;
        }
        if (bi < 0) break;
        last_block = block_run_start[bi] - 1;
        bi--;
      }
    }
    if (p->trellis_quant_dc) { /* :1308-1327 */
      j = 0;
      for (i = 1; i < ncand; i++)
        if (acc_dc[i][g->wib - 1] < acc_dc[j][g->wib - 1]) j = i;
      for (bi = g->wib - 1; bi >= 0; bi--) {
        e->q[ci][((size_t)br * g->wpad + bi) * 64] = cand_dc[j][bi];
        j = back_dc[j][bi];
      }
      last_dc = e->q[ci][((size_t)br * g->wpad + g->wib - 1) * 64];
    }
  }
  build_dummies(p, g, ci, e->q[ci]); /* jccoefct.c:443-476 */
  for (i = 0; i < 9; i++) { free(acc_dc[i]); free(back_dc[i]); free(cand_dc[i]); }
  free(azbc); free(abc); free(block_run_start); free(requires_eob);
}


/* ------------------------------------------------------------------------------------------
 * f4  arithmetic entropy coding: jcarith.c.  The QM coder of ITU-T T.81 Annex D with the
 * statistics models of F.1.4 / G.1.3; one adaptive state per scan, so a scan is ONE sequential chain.
 * ------------------------------------------------------------------------------------------ */
#include "mjo_arith_table.h"

/* conditioning: mjo_params.arith_dc_L / arith_dc_U / arith_ac_K (defaults 0 / 1 / 5, jcparam.c:417-419; the DAC marker carries them) */

typedef struct {
  long c, a, sc, zc;     /* code register, interval, stacked 0xFF bytes, pending 0x00 bytes (jcarith.c:31-38) */
  int ct, buffer;
  bytebuf *out;          /* NULL: nothing is written, only the statistics adapt (trellis passes, jcarith.c:127-129) */
  unsigned char dc_stats[4][64], ac_stats[4][256];
  unsigned char fixed_bin[4];
  int last_dc_val[MJO_MAX_COMPS], dc_context[MJO_MAX_COMPS];
  const mjo_params *p;   /* the conditioning values */
} arith_t;

static void ari_byte(arith_t *A, int v) { if (A->out) bb_put(A->out, v); }

static void ari_zeros(arith_t *A) { while (A->zc) { ari_byte(A, 0x00); A->zc--; } }

/* the byte that left the code register (renormalisation D.1.6) or is flushed at the end (D.1.8): `over` = a carry
 * propagates into the bytes held back (jcarith.c:278-316 and :160-190 share this logic) */
static void ari_shift_out(arith_t *A, long temp, int final)
{
  if (final ? (A->c & 0xF8000000L) != 0 : temp > 0xFF) {
    if (A->buffer >= 0) {
      ari_zeros(A);
      ari_byte(A, A->buffer + 1);
      if (A->buffer + 1 == 0xFF) ari_byte(A, 0x00);
    }
    A->zc += A->sc;        /* the carry turns the stacked 0xFF bytes into 0x00 */
    A->sc = 0;
    if (!final) A->buffer = (int)(temp & 0xFF);
  } else if (!final && temp == 0xFF) {
    A->sc++;
  } else {
    if (A->buffer == 0) A->zc++;
    else if (A->buffer >= 0) { ari_zeros(A); ari_byte(A, A->buffer); }
    if (A->sc) {
      ari_zeros(A);
      do { ari_byte(A, 0xFF); ari_byte(A, 0x00); } while (--A->sc);
    }
    if (!final) A->buffer = (int)(temp & 0xFF);
  }
}

static void ari_finish(arith_t *A)
{ /* finish_pass jcarith.c:142-203: the value in the interval with the most trailing zero bits, pending bytes, then
   * the last two bytes unless they are zero ("Pacman" termination) */
  long temp = (A->a - 1 + A->c) & 0xFFFF0000L;
  A->c = temp < A->c ? temp + 0x8000L : temp;
  A->c <<= A->ct;
  ari_shift_out(A, 0, 1);
  if (A->c & 0x7FFF800L) {
    ari_zeros(A);
    ari_byte(A, (int)((A->c >> 19) & 0xFF));
    if (((A->c >> 19) & 0xFF) == 0xFF) ari_byte(A, 0x00);
    if (A->c & 0x7F800L) {
      ari_byte(A, (int)((A->c >> 11) & 0xFF));
      if (((A->c >> 11) & 0xFF) == 0xFF) ari_byte(A, 0x00);
    }
  }
}

static void ari_encode(arith_t *A, unsigned char *st, int val)
{ /* arith_encode jcarith.c:229-320 */
  const int sv = *st, idx = sv & 0x7F;
  const long qe = mjo_ari_qe[idx];
  A->a -= qe;
  if (val != (sv >> 7)) {                 /* less probable symbol */
    if (A->a >= qe) { A->c += A->a; A->a = qe; }
    *st = (unsigned char)((sv & 0x80) ^ mjo_ari_nlps[idx]);
  } else {
    if (A->a >= 0x8000L) return;          /* no renormalisation, no adaptation */
    if (A->a < qe) { A->c += A->a; A->a = qe; }
    *st = (unsigned char)((sv & 0x80) ^ mjo_ari_nmps[idx]);
  }
  do {
    A->a <<= 1;
    A->c <<= 1;
    if (--A->ct == 0) {
      ari_shift_out(A, A->c >> 19, 0);
      A->c &= 0x7FFFFL;
      A->ct += 8;
    }
  } while (A->a < 0x8000L);
}

static void ari_reset_coder(arith_t *A)
{ /* start_pass jcarith.c:878-884, emit_restart :345-351 */
  A->c = 0; A->a = 0x10000L; A->sc = 0; A->zc = 0; A->ct = 11; A->buffer = -1;
}

/* statistics areas of the scan's components: DC where the scan codes DC differences, AC where it has an AC band
 * (`progressive` as the CALLER of the reference sees it: start_pass switches it off during trellis passes, :826,
 * emit_restart does not, :328-341) */
static void ari_reset_stats(arith_t *A, const enc_t *e, const scan_t *sc, int progressive)
{
  int i;
  for (i = 0; i < sc->ncomp; i++) {
    const int c = sc->comp[i];
    if (!progressive || (sc->Ss == 0 && sc->Ah == 0)) {
      memset(A->dc_stats[e->p->dc_tbl_no[c]], 0, 64);
      A->last_dc_val[i] = 0;
      A->dc_context[i] = 0;
    }
    if (!progressive || sc->Se) memset(A->ac_stats[e->p->ac_tbl_no[c]], 0, 256);
  }
}

/* magnitude category + magnitude bits of v >= 1 (Figures F.8, F.9): st = the first magnitude bin (SP / SN / the bin behind
 * the sign), x1 = where the category bins continue (X1 = 20 for DC; for AC the bin itself once more, then 189 / 217);
 * returns the category mask m (the DC conditioning needs it) */
static int ari_magnitude(arith_t *A, unsigned char *stats, unsigned char *st, int v, int ac, int k, int kx)
{
  int m = 0, v2;
  if (v -= 1) {
    ari_encode(A, st, 1);
    m = 1;
    v2 = v;
    if (ac) {
      if (v2 >>= 1) {
        ari_encode(A, st, 1);
        m <<= 1;
        st = stats + (k <= kx ? 189 : 217);
        while (v2 >>= 1) { ari_encode(A, st, 1); m <<= 1; st++; }
      }
    } else {
      st = stats + 20;
      while (v2 >>= 1) { ari_encode(A, st, 1); m <<= 1; st++; }
    }
  }
  ari_encode(A, st, 0);
  st += 14;
  {
    int mm = m;
    while (mm >>= 1) ari_encode(A, st, (mm & v) ? 1 : 0);
  }
  return m;
}

static void ari_dc(arith_t *A, int tbl, int ci, int value)
{ /* Encode_DC_DIFF, jcarith.c:402-448 / :715-762 */
  unsigned char *stats = A->dc_stats[tbl], *st = stats + A->dc_context[ci];
  int v = value - A->last_dc_val[ci], m;
  if (v == 0) {
    ari_encode(A, st, 0);
    A->dc_context[ci] = 0;
    return;
  }
  A->last_dc_val[ci] = value;
  ari_encode(A, st, 1);
  if (v > 0) { ari_encode(A, st + 1, 0); st += 2; A->dc_context[ci] = 4; }
  else { v = -v; ari_encode(A, st + 1, 1); st += 3; A->dc_context[ci] = 8; }
  m = ari_magnitude(A, stats, st, v, 0, 0, 0);
  if (m < (int)((1L << A->p->arith_dc_L[tbl]) >> 1)) A->dc_context[ci] = 0;
  else if (m > (int)((1L << A->p->arith_dc_U[tbl]) >> 1)) A->dc_context[ci] += 8;
}

static void ari_ac_first(arith_t *A, int tbl, const int16_t *blk, int Ss, int Se, int Al)
{ /* Encode_AC_Coefficients: encode_mcu_AC_first jcarith.c:456-552, and with Ss = 1, Se = 63, Al = 0 the AC part of
   * the sequential encode_mcu :764-817 */
  unsigned char *stats = A->ac_stats[tbl], *st;
  int k, ke, v;
  for (ke = Se; ke > 0; ke--) {
    v = blk[ZZ[ke]];
    if (v < 0) v = -v;
    if (v >> Al) break;
  }
  for (k = Ss; k <= ke; k++) {
    int neg;
    st = stats + 3 * (k - 1);
    ari_encode(A, st, 0);                 /* not the end of the block */
    for (;;) {
      v = blk[ZZ[k]];
      neg = v < 0;
      if (neg) v = -v;
      v >>= Al;
      if (v) break;
      ari_encode(A, st + 1, 0);
      st += 3;
      k++;
    }
    ari_encode(A, st + 1, 1);
    ari_encode(A, A->fixed_bin, neg);
    ari_magnitude(A, stats, st + 2, v, 1, k, A->p->arith_ac_K[tbl]);
  }
  if (k <= Se) ari_encode(A, stats + 3 * (k - 1), 1);
}

static void ari_ac_refine(arith_t *A, int tbl, const int16_t *blk, int Ss, int Se, int Ah, int Al)
{ /* encode_mcu_AC_refine jcarith.c:596-687 */
  unsigned char *stats = A->ac_stats[tbl], *st;
  int k, ke, kex, v;
  for (ke = Se; ke > 0; ke--) {
    v = blk[ZZ[ke]];
    if (v < 0) v = -v;
    if (v >> Al) break;
  }
  for (kex = ke; kex > 0; kex--) {
    v = blk[ZZ[kex]];
    if (v < 0) v = -v;
    if (v >> Ah) break;
  }
  for (k = Ss; k <= ke; k++) {
    st = stats + 3 * (k - 1);
    if (k > kex) ari_encode(A, st, 0);
    for (;;) {
      int neg;
      v = blk[ZZ[k]];
      neg = v < 0;
      if (neg) v = -v;
      v >>= Al;
      if (v) {
        if (v >> 1) ari_encode(A, st + 2, v & 1);          /* was non-zero before: its next bit */
        else { ari_encode(A, st + 1, 1); ari_encode(A, A->fixed_bin, neg); }
        break;
      }
      ari_encode(A, st + 1, 0);
      st += 3;
      k++;
    }
  }
  if (k <= Se) ari_encode(A, stats + 3 * (k - 1), 1);
}

/* one scan (or, `sequential_blocks`: the whole blocks of a trellis pass, jcarith.c:824-826) over the MCU rows r0..r1-1 */
static void ari_code_rows(enc_t *e, const scan_t *sc, arith_t *A, int sequential_blocks, int r0, int r1, int *restarts_to_go, int *next_restart)
{
  const mjo_params *p = e->p;
  int mr, mc, ci;
  for (mr = r0; mr < r1; mr++)
    for (mc = 0; mc < sc->mcus_per_row; mc++) {
      if (sc->restart_interval) {
        if (*restarts_to_go == 0) { /* emit_restart :322-352 */
          ari_finish(A);
          ari_byte(A, 0xFF); ari_byte(A, 0xD0 + *next_restart);
          ari_reset_stats(A, e, sc, e->progressive);
          ari_reset_coder(A);
          *restarts_to_go = sc->restart_interval;
          *next_restart = (*next_restart + 1) & 7;
        }
        (*restarts_to_go)--;
      }
      for (ci = 0; ci < sc->ncomp; ci++) {
        const int c = sc->comp[ci];
        const int mw = sc->ncomp == 1 ? 1 : p->h_samp[c], mh = sc->ncomp == 1 ? 1 : p->v_samp[c];
        int yi, xi;
        for (yi = 0; yi < mh; yi++)
          for (xi = 0; xi < mw; xi++) {
            const int16_t *blk = e->q[c] + ((size_t)(mr * mh + yi) * e->g[c].wpad + (mc * mw + xi)) * 64;
            if (sequential_blocks || !e->progressive) {
              ari_dc(A, p->dc_tbl_no[c], ci, blk[0]);
              ari_ac_first(A, p->ac_tbl_no[c], blk, 1, 63, 0);
            } else if (sc->Ss == 0 && sc->Ah == 0) ari_dc(A, p->dc_tbl_no[c], ci, blk[0] >> sc->Al);   /* IRIGHT_SHIFT: arithmetic */
            else if (sc->Ss == 0) ari_encode(A, A->fixed_bin, (blk[0] >> sc->Al) & 1);                  /* encode_mcu_DC_refine :560-590 */
            else if (sc->Ah == 0) ari_ac_first(A, p->ac_tbl_no[c], blk, sc->Ss, sc->Se, sc->Al);
            else ari_ac_refine(A, p->ac_tbl_no[c], blk, sc->Ss, sc->Se, sc->Ah, sc->Al);
          }
      }
    }
}

static void ari_start(arith_t *A, const enc_t *e, const scan_t *sc, int progressive, bytebuf *out)
{
  memset(A, 0, sizeof(*A));
  A->p = e->p;
  A->out = out;
  A->fixed_bin[0] = 113;
  ari_reset_stats(A, e, sc, progressive);
  ari_reset_coder(A);
}

static void code_scan_arith(enc_t *e, const scan_t *sc, bytebuf *out)
{
  arith_t A;
  int rtg = sc->restart_interval, nr = 0;
  ari_start(&A, e, sc, e->progressive, out);
  ari_code_rows(e, sc, &A, 0, 0, sc->mcu_rows, &rtg, &nr);
  ari_finish(&A);
}

/* ---- quantize_trellis_arith jcdctmgr.c:1334-1667, driven by compress_trellis_pass jccoefct.c:356-486 ----------------
 * Rate estimates come from the CURRENT state of the adaptive coder (jget_arith_rates jcarith.c:944-976), read once per
 * iMCU row; after a row has been quantized it is run through the coder (its output discarded), which moves the state the
 * next row's estimates are read from.  The pass sequencing of jcmaster.c only ever selects component 0 for these passes
 * when arithmetic coding is on (prepare_for_pass's trellis_pass case does not re-select the scan, and no statistics pass
 * sits in between: jcmaster.c:686-702, :1001-1005), every one of them starts from a zeroed state and the unquantized
 * coefficients, so they all give the same result: ONE pass over component 0 is what the reference's passes amount to. */
typedef struct { float dc[64][2], ac[256][2]; } ari_rates;

static void ari_get_rates(const arith_t *A, int dctbl, int actbl, ari_rates *r)
{
  int i;
  for (i = 0; i < 64 + 256; i++) {
    const int state = i < 64 ? A->dc_stats[dctbl][i] : A->ac_stats[actbl][i - 64];
    const int mps = state >> 7;
    const float prob_lps = (float)((double)mjo_ari_qe[state & 0x7F] / 46340.95);
    const float prob_0 = mps ? prob_lps : (float)(1.0 - (double)prob_lps);
    const float prob_1 = (float)(1.0 - (double)prob_0);
    float *o = i < 64 ? r->dc[i] : r->ac[i - 64];
    o[0] = (float)(-log((double)prob_0) / log(2.0));
    o[1] = (float)(-log((double)prob_1) / log(2.0));
  }
}

static void trellis_row_arith(enc_t *e, int ci, const ari_rates *r, int br, int Ss, int Se, int *last_dc_io,
                              float *acc_dc[9], int *back_dc[9], int16_t *cand_dc[9], int *ctx_dc[9])
{
  const mjo_params *p = e->p;
  const mjo_geom *g = &e->g[ci];
  const uint16_t *qt = p->qtbl[p->quant_tbl_no[ci]];
  const int v = e->chain_v ? e->chain_v : p->v_samp[ci];
  int ncand = (2 + 60 / qt[0]) | 1;
  float lambda_tbl[64];
  int run_start[64];
  int i, j, k, l, bi;
  if (ncand > 9) ncand = 9;
  memset(run_start, 0, sizeof(run_start));
  for (i = 0; i < 64; i++) lambda_tbl[i] = (float)(1.0 / (double)((int)qt[i] * (int)qt[i]));
  for (bi = 0; bi < g->wib; bi++) {
    const int16_t *src = e->uq[ci] + ((size_t)br * g->wpad + bi) * 64;
    int16_t *coef = e->q[ci] + ((size_t)br * g->wpad + bi) * 64;
    float azd[64], acost[64];
    float norm = 0.0f, lambda, lambda_dc, best_cost;
    int last_coeff_idx;
    for (i = 1; i < 64; i++) norm = norm + (float)((int)src[i] * (int)src[i]);
    norm = (float)((double)norm / 63.0);
    if (p->lambda_log_scale2 > 0.0f)
      lambda = (float)(pow(2.0, (double)p->lambda_log_scale1) * (double)1.0f / (pow(2.0, (double)p->lambda_log_scale2) + (double)norm));
    else
      lambda = (float)(pow(2.0, (double)p->lambda_log_scale1 - 12.0) * (double)1.0f);
    lambda_dc = lambda * lambda_tbl[0];
    azd[Ss - 1] = 0.0f;
    acost[Ss - 1] = 0.0f;

    if (p->trellis_quant_dc) { /* :1416-1509 */
      const int sign = src[0] < 0 ? -1 : 0;
      const int x = src[0] < 0 ? -src[0] : src[0];
      const int q = 8 * qt[0];
      const int qval = (x + q / 2) / q;
      for (k = 0; k < ncand; k++) {
        int cnd = qval - ncand / 2 + k, delta;
        float dist;
        delta = cnd * q - x;
        dist = (float)(delta * delta) * lambda_dc;
        cnd *= 1 + 2 * sign;
        cand_dc[k][bi] = (int16_t)cnd;
        if (br % v != 0 && p->trellis_delta_dc_weight > 0.0f) {   /* :1440-1456 */
          const int dc_above_orig = e->uq[ci][((size_t)(br - 1) * g->wpad + bi) * 64];
          const int dc_above_recon = e->q[ci][((size_t)(br - 1) * g->wpad + bi) * 64] * q;
          const int dc_orig = src[0], dc_recon = cnd * q;
          float vertical_dist, t;
          delta = (dc_above_orig - dc_orig) - (dc_above_recon - dc_recon);
          vertical_dist = (float)(delta * delta) * lambda_dc;
          t = vertical_dist - dist;
          t = p->trellis_delta_dc_weight * t;
          dist = dist + t;
        }
        for (l = 0; l < (bi == 0 ? 1 : ncand); l++) {
          const int dc_pred = bi == 0 ? *last_dc_io : cand_dc[l][bi - 1];
          int st = bi == 0 ? 0 : ctx_dc[l][bi - 1], upd = 0, dc_delta = cnd - dc_pred, m, v2;
          float bits = r->dc[st][dc_delta != 0], cost;
          if (dc_delta != 0) {
            bits += r->dc[st + 1][dc_delta < 0];
            st += 2 + (dc_delta < 0);
            upd = dc_delta < 0 ? 8 : 4;
            if (dc_delta < 0) dc_delta = -dc_delta;
            m = 0;
            if (dc_delta -= 1) {
              bits += r->dc[st][1];
              st = 20;
              m = 1;
              v2 = dc_delta;
              while (v2 >>= 1) { bits += r->dc[st][1]; m <<= 1; st++; }
            }
            bits += r->dc[st][0];
            if (m < (int)((1L << p->arith_dc_L[p->dc_tbl_no[ci]]) >> 1)) upd = 0;
            else if (m > (int)((1L << p->arith_dc_U[p->dc_tbl_no[ci]]) >> 1)) upd += 8;
            st += 14;
            while (m >>= 1) bits += r->dc[st][(m & dc_delta) ? 1 : 0];
          }
          cost = bits + dist;
          if (bi != 0) cost += acc_dc[l][bi - 1];
          if (l == 0 || cost < acc_dc[k][bi]) {
            acc_dc[k][bi] = cost;
            back_dc[k][bi] = bi == 0 ? -1 : l;
            ctx_dc[k][bi] = upd;
          }
        }
      }
    }

    for (i = Ss; i <= Se; i++) { /* :1512-1601 */
      const int z = ZZ[i];
      const int sign = src[z] < 0 ? -1 : 0;
      const int x = src[z] < 0 ? -src[z] : src[z];
      const int q = 8 * qt[z];
      int cand[2], ncd, qval, delta;
      float cdist[2], t;
      t = (float)(x * x) * lambda;
      t = t * lambda_tbl[z];
      azd[i] = t + azd[i - 1];
      qval = (x + q / 2) / q;
      if (qval == 0) { coef[z] = 0; acost[i] = 1e38f; continue; }
      cand[0] = qval;
      delta = cand[0] * q - x;
      t = (float)(delta * delta) * lambda;
      cdist[0] = t * lambda_tbl[z];
      ncd = 1;
      if (qval > 1) {
        cand[1] = qval - 1;
        delta = cand[1] * q - x;
        t = (float)(delta * delta) * lambda;
        cdist[1] = t * lambda_tbl[z];
        ncd = 2;
      }
      acost[i] = 1e38f;
      for (j = Ss - 1; j < i; j++) {
        float run_bits;
        if (j != Ss - 1 && coef[ZZ[j]] == 0) continue;
        run_bits = r->ac[3 * j][0];
        for (k = j + 1; k < i; k++) run_bits += r->ac[3 * (k - 1) + 1][0];
        run_bits += r->ac[3 * (i - 1) + 1][1];
        for (k = 0; k < ncd; k++) {
          float coef_bits = 1.0f, cost, rhs;
          int vv = cand[k], v2, m = 0, st = 3 * (i - 1) + 2, rate;
          if (vv -= 1) {
            coef_bits += r->ac[st][1];
            m = 1;
            v2 = vv;
            if (v2 >>= 1) {
              coef_bits += r->ac[st][1];
              m <<= 1;
              st = i <= p->arith_ac_K[p->ac_tbl_no[ci]] ? 189 : 217;
              while (v2 >>= 1) { coef_bits += r->ac[st][1]; m <<= 1; st++; }
            }
          }
          coef_bits += r->ac[st][0];
          st += 14;
          while (m >>= 1) coef_bits += r->ac[st][(m & vv) ? 1 : 0];
          rate = (int)(coef_bits + run_bits);       /* `int rate` in the reference: the estimate is truncated (:1349, :1583) */
          cost = (float)rate + cdist[k];
          rhs = azd[i - 1] - azd[j];
          rhs = rhs + acost[j];
          cost = cost + rhs;
          if (cost < acost[i]) {
            coef[z] = (int16_t)((cand[k] ^ sign) - sign);
            acost[i] = cost;
            run_start[i] = j;
          }
        }
      }
    }
    last_coeff_idx = Ss - 1; /* :1603-1631 */
    best_cost = azd[Se] + r->ac[0][1];
    for (i = Ss; i <= Se; i++)
      if (coef[ZZ[i]] != 0) {
        float cost = acost[i] + azd[Se];
        cost = cost - azd[i];
        if (i < Se) cost = cost + r->ac[3 * (i - 1)][1];
        if (cost < best_cost) { best_cost = cost; last_coeff_idx = i; }
      }
    i = Se;
    while (i >= Ss) {
      while (i > last_coeff_idx) { coef[ZZ[i]] = 0; i--; }
      last_coeff_idx = run_start[i];
      i--;
    }
  }
  if (p->trellis_quant_dc) { /* :1643-1665 */
    j = 0;
    for (i = 1; i < ncand; i++)
      if (acc_dc[i][g->wib - 1] < acc_dc[j][g->wib - 1]) j = i;
    for (bi = g->wib - 1; bi >= 0; bi--) {
      e->q[ci][((size_t)br * g->wpad + bi) * 64] = cand_dc[j][bi];
      j = back_dc[j][bi];
    }
    *last_dc_io = e->q[ci][((size_t)br * g->wpad + g->wib - 1) * 64];
  }
}

static void trellis_component_arith(enc_t *e, int ci, int Ss, int Se)
{
  const mjo_params *p = e->p;
  const mjo_geom *g = &e->g[ci];
  const int v = e->chain_v ? e->chain_v : p->v_samp[ci];
  float *acc_dc[9];
  int *back_dc[9], *ctx_dc[9];
  int16_t *cand_dc[9];
  arith_t A;
  ari_rates rates;
  mjo_scan ms;
  scan_t sc;
  int i, br, rtg, nr = 0;
  if (Se < Ss) return;
  for (i = 0; i < 9; i++) {
    acc_dc[i] = (float *)malloc(sizeof(float) * g->wib);
    back_dc[i] = (int *)malloc(sizeof(int) * g->wib);
    ctx_dc[i] = (int *)malloc(sizeof(int) * g->wib);
    cand_dc[i] = (int16_t *)malloc(sizeof(int16_t) * g->wib);
  }
  memset(&ms, 0, sizeof(ms));
  ms.comps_in_scan = 1; ms.component_index[0] = ci; ms.Ss = Ss; ms.Se = Se;
  setup_scan(e, &sc, &ms);
  ari_start(&A, e, &sc, 0, NULL);           /* start_pass: "progressive mode off" during trellis passes (:824-826) */
  rtg = sc.restart_interval;
  for (br = 0; br < g->hib; br += v) {
    const int rows = br + v <= g->hib ? v : g->hib - br;
    int last_dc = 0, rr;
    ari_get_rates(&A, p->dc_tbl_no[ci], p->ac_tbl_no[ci], &rates);
    for (rr = 0; rr < rows; rr++) trellis_row_arith(e, ci, &rates, br + rr, Ss, Se, &last_dc, acc_dc, back_dc, cand_dc, ctx_dc);
    ari_code_rows(e, &sc, &A, 1, br, br + rows, &rtg, &nr);   /* compress_output of this iMCU row: the state moves on */
  }
  build_dummies(p, g, ci, e->q[ci]);
  for (i = 0; i < 9; i++) { free(acc_dc[i]); free(back_dc[i]); free(ctx_dc[i]); free(cand_dc[i]); }
}

/* ------------------------------------------------------------------------------------------
 * a15  markers: jcmarker.c
 * ------------------------------------------------------------------------------------------ */
static void emit_file_header(const enc_t *e, bytebuf *o)
{ /* write_file_header :649-666, emit_jfif_app0 :534-565 */
  bb_put(o, 0xFF); bb_put(o, 0xD8);
  if (e->p->write_jfif) {
    bb_put(o, 0xFF); bb_put(o, 0xE0);
    bb_put2(o, 16);
    bb_put(o, 'J'); bb_put(o, 'F'); bb_put(o, 'I'); bb_put(o, 'F'); bb_put(o, 0);
    bb_put(o, 1); bb_put(o, 1);
    bb_put(o, 0);
    bb_put2(o, 1); bb_put2(o, 1);
    bb_put(o, 0); bb_put(o, 0);
  }
  if (e->p->rgb_output) {   /* emit_adobe_app14 :452-486: "Adobe", version 100, flags0 0, flags1 0, transform 0 */
    static const unsigned char ad[16] = { 0xFF, 0xEE, 0, 14, 'A', 'd', 'o', 'b', 'e', 0, 100, 0, 0, 0, 0, 0 };
    int i;
    for (i = 0; i < 16; i++) bb_put(o, ad[i]);
  }
}

static void emit_frame_header(enc_t *e, bytebuf *o)
{ /* write_frame_header :674-734, emit_multi_dqt :189-254, emit_dqt :140-186, emit_sof :464-490 */
  const mjo_params *p = e->p;
  int ci, i, prec_any = 0, is_baseline;
  int prec[MJO_MAX_COMPS];
  int multi = !p->fastest_profile;
  for (ci = 0; ci < p->num_components; ci++) {
    const uint16_t *qt = p->qtbl[p->quant_tbl_no[ci]];
    prec[ci] = 0;
    for (i = 0; i < 64; i++) if (qt[i] > 255) prec[ci] = 1;
    if (e->qsent[p->quant_tbl_no[ci]]) multi = 0;
  }
  if (multi) {
    int seen[4] = { 0, 0, 0, 0 }, size = 2;
    bb_put(o, 0xFF); bb_put(o, 0xDB);
    for (ci = 0; ci < p->num_components; ci++) {
      int t = p->quant_tbl_no[ci];
      if (!seen[t]) { size += 64 * (prec[ci] + 1) + 1; seen[t] = 1; }
      prec_any += prec[ci];
    }
    bb_put2(o, size);
    for (ci = 0; ci < p->num_components; ci++) {
      int t = p->quant_tbl_no[ci];
      if (e->qsent[t]) continue;
      bb_put(o, t + (prec[ci] << 4));
      for (i = 0; i < 64; i++) {
        unsigned qv = p->qtbl[t][ZZ[i]];
        if (prec[ci]) bb_put(o, (int)(qv >> 8));
        bb_put(o, (int)(qv & 0xFF));
      }
      e->qsent[t] = 1;
    }
  } else {
    for (ci = 0; ci < p->num_components; ci++) {
      int t = p->quant_tbl_no[ci];
      prec_any += prec[ci];
      if (e->qsent[t]) continue;
      bb_put(o, 0xFF); bb_put(o, 0xDB);
      bb_put2(o, prec[ci] ? 64 * 2 + 1 + 2 : 64 + 1 + 2);
      bb_put(o, t + (prec[ci] << 4));
      for (i = 0; i < 64; i++) {
        unsigned qv = p->qtbl[t][ZZ[i]];
        if (prec[ci]) bb_put(o, (int)(qv >> 8));
        bb_put(o, (int)(qv & 0xFF));
      }
      e->qsent[t] = 1;
    }
  }
  is_baseline = !e->progressive && prec_of(p) == 8;
  if (is_baseline) {
    for (ci = 0; ci < p->num_components; ci++)
      if (p->dc_tbl_no[ci] > 1 || p->ac_tbl_no[ci] > 1) is_baseline = 0;
    if (prec_any) is_baseline = 0;
  }
  bb_put(o, 0xFF);
  if (p->arith_code) bb_put(o, e->progressive ? 0xCA : 0xC9);   /* SOF10 / SOF9, jcmarker.c:720-725 */
  else bb_put(o, e->progressive ? 0xC2 : (is_baseline ? 0xC0 : 0xC1));
  bb_put2(o, 3 * p->num_components + 2 + 5 + 1);
  bb_put(o, prec_of(p));
  bb_put2(o, p->height);
  bb_put2(o, p->width);
  bb_put(o, p->num_components);
  for (ci = 0; ci < p->num_components; ci++) {
    bb_put(o, p->component_id[ci]);
    bb_put(o, e->sof_hv0 ? e->sof_hv0 : (p->h_samp[ci] << 4) + p->v_samp[ci]);
    bb_put(o, p->quant_tbl_no[ci]);
  }
}

static void emit_one_dht(bytebuf *o, htbl *h, int index)
{ /* emit_dht :257-290 */
  int length = 0, i;
  if (h->sent) return;
  bb_put(o, 0xFF); bb_put(o, 0xC4);
  for (i = 1; i <= 16; i++) length += h->bits[i];
  bb_put2(o, length + 2 + 1 + 16);
  bb_put(o, index);
  for (i = 1; i <= 16; i++) bb_put(o, h->bits[i]);
  for (i = 0; i < length; i++) bb_put(o, h->huffval[i]);
  h->sent = 1;
}

static void emit_scan_header(enc_t *e, const scan_t *sc, bytebuf *o)
{ /* write_scan_header :744-784, emit_multi_dht :293-401, emit_dri :452, emit_sos :494-531 */
  const mjo_params *p = e->p;
  int i, j;
  if (p->arith_code) { /* emit_dac jcmarker.c:404-448: the conditioning parameters of the tables this scan uses, in one marker */
    int dc_in_use[4] = { 0, 0, 0, 0 }, ac_in_use[4] = { 0, 0, 0, 0 }, length = 0;
    for (i = 0; i < sc->ncomp; i++) {
      if (sc->Ss == 0 && sc->Ah == 0) dc_in_use[p->dc_tbl_no[sc->comp[i]]] = 1;
      if (sc->Se) ac_in_use[p->ac_tbl_no[sc->comp[i]]] = 1;
    }
    for (i = 0; i < 4; i++) length += dc_in_use[i] + ac_in_use[i];
    if (length) {
      bb_put(o, 0xFF); bb_put(o, 0xCC);
      bb_put2(o, length * 2 + 2);
      for (i = 0; i < 4; i++) {
        if (dc_in_use[i]) { bb_put(o, i); bb_put(o, p->arith_dc_L[i] + (p->arith_dc_U[i] << 4)); }
        if (ac_in_use[i]) { bb_put(o, i + 0x10); bb_put(o, p->arith_ac_K[i]); }
      }
    }
  } else if (!p->fastest_profile) {
    int length = 2, dclens[4] = { 0, 0, 0, 0 }, aclens[4] = { 0, 0, 0, 0 };
    htbl *dcseen[4] = { 0, 0, 0, 0 }, *acseen[4] = { 0, 0, 0, 0 };
    for (i = 0; i < sc->ncomp; i++) {
      htbl *d = &e->dc[p->dc_tbl_no[sc->comp[i]]], *a = &e->ac[p->ac_tbl_no[sc->comp[i]]];
      int seen = 0;
      if (sc->Ss == 0 && sc->Ah == 0) {
        if (d->sent) continue;
        for (j = 0; j < 4; j++) seen += (d == dcseen[j]);
        if (seen) continue;
        dcseen[i] = d;
        for (j = 1; j <= 16; j++) dclens[i] += d->bits[j];
        length += dclens[i] + 16 + 1;
      }
      if (sc->Se) {
        if (a->sent) continue;
        seen = 0;
        for (j = 0; j < 4; j++) seen += (a == acseen[j]);
        if (seen) continue;
        acseen[i] = a;
        for (j = 1; j <= 16; j++) aclens[i] += a->bits[j];
        length += aclens[i] + 16 + 1;
      }
    }
    bb_put(o, 0xFF); bb_put(o, 0xC4);
    bb_put2(o, length);
    for (i = 0; i < sc->ncomp; i++) {
      int dcidx = p->dc_tbl_no[sc->comp[i]], acidx = p->ac_tbl_no[sc->comp[i]];
      htbl *d = &e->dc[dcidx], *a = &e->ac[acidx];
      if (sc->Ss == 0 && sc->Ah == 0 && !d->sent) {
        bb_put(o, dcidx);
        for (j = 1; j <= 16; j++) bb_put(o, d->bits[j]);
        for (j = 0; j < dclens[i]; j++) bb_put(o, d->huffval[j]);
        d->sent = 1;
      }
      if (sc->Se && !a->sent) {
        bb_put(o, acidx + 0x10);
        for (j = 1; j <= 16; j++) bb_put(o, a->bits[j]);
        for (j = 0; j < aclens[i]; j++) bb_put(o, a->huffval[j]);
        a->sent = 1;
      }
    }
  } else {
    for (i = 0; i < sc->ncomp; i++) {
      int c = sc->comp[i];
      if (sc->Ss == 0 && sc->Ah == 0) emit_one_dht(o, &e->dc[p->dc_tbl_no[c]], p->dc_tbl_no[c]);
      if (sc->Se) emit_one_dht(o, &e->ac[p->ac_tbl_no[c]], p->ac_tbl_no[c] + 0x10);
    }
  }
  if (sc->restart_interval != e->last_restart_interval) {
    bb_put(o, 0xFF); bb_put(o, 0xDD);
    bb_put2(o, 4);
    bb_put2(o, sc->restart_interval);
    e->last_restart_interval = sc->restart_interval;
  }
  bb_put(o, 0xFF); bb_put(o, 0xDA);
  bb_put2(o, 2 * sc->ncomp + 2 + 1 + 3);
  bb_put(o, sc->ncomp);
  for (i = 0; i < sc->ncomp; i++) {
    int c = sc->comp[i];
    int td = (sc->Ss == 0 && sc->Ah == 0) ? p->dc_tbl_no[c] : 0;
    int ta = sc->Se ? p->ac_tbl_no[c] : 0;
    bb_put(o, p->component_id[c]);
    bb_put(o, (td << 4) + ta);
  }
  bb_put(o, sc->Ss);
  bb_put(o, sc->Se);
  bb_put(o, (sc->Ah << 4) + sc->Al);
}

/* ------------------------------------------------------------------------------------------
 * a16  pass control: jcmaster.c prepare_for_pass :612, finish_pass_master :968,
 *      select_scan_parameters :443, select_scans :773; gather -> table: finish_pass_gather
 *      jchuff.c:1113 / finish_pass_gather_phuff jcphuff.c:1055.
 * ------------------------------------------------------------------------------------------ */
static void gather_and_build(enc_t *e, const scan_t *sc, int trellis_pass)
{
  const mjo_params *p = e->p;
  if (p->arith_code) return;   /* the arithmetic coder adapts while it codes: no statistics pass (jcmaster.c:1088-1089) */
  long dcc[4][257], acc[4][257];
  sink_t s;
  int t, i, j, ci;
  int did_dc[4] = { 0, 0, 0, 0 }, did_ac[4] = { 0, 0, 0, 0 };
  memset(&s, 0, sizeof(s));
  memset(dcc, 0, sizeof(dcc));
  memset(acc, 0, sizeof(acc));
  s.gather = 1;
  for (t = 0; t < 4; t++) { s.dc_count[t] = dcc[t]; s.ac_count[t] = acc[t]; }
  if (e->progressive && trellis_pass) { /* jcphuff.c:257-264 */
    for (ci = 0; ci < sc->ncomp; ci++) {
      long *c = sc->Ss == 0 ? dcc[p->dc_tbl_no[sc->comp[ci]]] : acc[p->ac_tbl_no[sc->comp[ci]]];
      for (i = 0; i < 16; i++) for (j = 0; j < 12; j++) c[16 * i + j] = 1;
    }
  }
  code_scan(e, sc, &s);
  for (ci = 0; ci < sc->ncomp; ci++) {
    int c = sc->comp[ci];
    int d = p->dc_tbl_no[c], a = p->ac_tbl_no[c];
    int want_dc = !e->progressive || (sc->Ss == 0 && sc->Ah == 0);
    int want_ac = !e->progressive || sc->Ss != 0;
    if (want_dc && !did_dc[d]) {
      mjo_gen_optimal_table(dcc[d], e->dc[d].bits, e->dc[d].huffval);
      e->dc[d].sent = 0;
      did_dc[d] = 1;
    }
    if (want_ac && !did_ac[a]) {
      mjo_gen_optimal_table(acc[a], e->ac[a].bits, e->ac[a].huffval);
      e->ac[a].sent = 0;
      did_ac[a] = 1;
    }
  }
}

static void output_scan(enc_t *e, const scan_t *sc, int first_scan, bytebuf *o)
{
  const mjo_params *p = e->p;
  dtbl dcd[4], acd[4];
  sink_t s;
  int ci;
  memset(&s, 0, sizeof(s));
  for (ci = 0; ci < sc->ncomp; ci++) {
    int c = sc->comp[ci];
    make_derived(&e->dc[p->dc_tbl_no[c]], &dcd[p->dc_tbl_no[c]]);
    make_derived(&e->ac[p->ac_tbl_no[c]], &acd[p->ac_tbl_no[c]]);
    s.dcd[p->dc_tbl_no[c]] = &dcd[p->dc_tbl_no[c]];
    s.acd[p->ac_tbl_no[c]] = &acd[p->ac_tbl_no[c]];
  }
  s.out = o;
  if (first_scan) emit_frame_header(e, o);
  emit_scan_header(e, sc, o);
  if (p->arith_code) code_scan_arith(e, sc, o);
  else code_scan(e, sc, &s);
}

static void bb_append(bytebuf *dst, const bytebuf *src)
{
  size_t i;
  for (i = 0; i < src->len; i++) bb_put(dst, src->buf[i]);
}

/* The whole encode from the point where the sample planes exist.  `fill` produces them: colour
 * conversion + downsampling of pixels (jpeg_write_scanlines) or a copy of caller-supplied component
 * planes (jpeg_write_raw_data). */
typedef struct {
  const uint8_t *pixels; size_t row_stride;                 /* pixel input */
  const uint8_t *const *src; const size_t *src_stride;      /* plane input (NULL = pixels) */
  const int *src_w, *src_h;
  const int16_t *const *coef; const size_t *coef_pitch;     /* quantized coefficient input (jpeg_write_coefficients) */
} plane_source;

static void import_planes16(const mjo_params *p, const plane_source *ps, uint16_t *planes[MJO_MAX_COMPS])
{
  /* jpeg_write_raw_data jcapistd.c:145-199 hands the rows straight to the coefficient controller, which
   * reads width_in_blocks*8 samples of height_in_blocks*8 rows per component (compress_first_pass
   * jccoefct.c:262-353).  A caller whose planes are smaller replicates the last sample / row up to that
   * size first, as tj3CompressFromYUVPlanes8 does (turbojpeg.c:1295-1316): modelled by the clamps. */
  mjo_geom g[MJO_MAX_COMPS];
  int ci, r, c;
  const int bps = prec_of(p) == 12 ? 2 : 1;
  mjo_geometry(p, g, NULL, NULL);
  for (ci = 0; ci < p->num_components; ci++)
    for (r = 0; r < g[ci].ph; r++) {
      const int sr = r < ps->src_h[ci] ? r : ps->src_h[ci] - 1;
      const uint8_t *row = ps->src[ci] + (size_t)sr * ps->src_stride[ci];
      for (c = 0; c < g[ci].pw; c++) {
        const int sc = c < ps->src_w[ci] ? c : ps->src_w[ci] - 1;
        planes[ci][(size_t)r * g[ci].pw + c] = bps == 2 ? ((const uint16_t *)row)[sc] : row[sc];
      }
    }
}

static size_t encode_core(const mjo_params *p_in, const plane_source *ps, uint8_t *out, size_t cap, mjo_taps *taps)
{
  enc_t e;
  bytebuf o = { 0, 0, 0 };
  uint16_t *planes[MJO_MAX_COMPS] = { 0, 0, 0, 0 };
  int ci, t;
  size_t n;
  mjo_params pp = *p_in;
  const mjo_params *p = &pp;
  if (prec_of(p) == 12) { pp.optimize_coding = 1; if (pp.trellis_quant) return 0; }   /* jcparam.c:452, SURVEY F1 */
  if (p->arith_code) pp.optimize_coding = 0;                                           /* jcmaster.c:1088-1089 */

  memset(&e, 0, sizeof(e));
  e.p = p;
  if (pp.num_components == 1 && (pp.h_samp[0] != 1 || pp.v_samp[0] != 1)) {
    /* ONE component sampled HxV (cjpeg gives a gray image 2x1 for qualities 80..89, set_quality_ratings rdswitch.c:566-570): its
     * scans are non-interleaved -- an MCU is one block, MCUs_per_row = width_in_blocks, no dummy blocks (per_scan_setup
     * jcmaster.c:548-575) -- and max_samp = its own factors (initial_setup :210-259), so width / height_in_blocks, the
     * downsampler (h_expand = v_expand = 1: fullsize_downsample / fullsize_smooth_downsample, jcsample.c:507-518) and the restart
     * rows are those of 1x1: only the SOF byte differs.  The exception is the trellis: compress_trellis_pass walks iMCU rows
     * of V block rows (lastDC and the row above chain over them, jccoefct.c:418-441): chain_v. */
    if (pp.h_samp[0] < 1 || pp.h_samp[0] > 4 || pp.v_samp[0] < 1 || pp.v_samp[0] > 4) return 0;
    e.chain_v = pp.v_samp[0];
    e.sof_hv0 = (pp.h_samp[0] << 4) + pp.v_samp[0];
    pp.h_samp[0] = pp.v_samp[0] = 1;
  }
  /* validate_script jcmaster.c:309-330: a script whose scans are all Ss = 0, Se = 63 is a SEQUENTIAL multi-scan file (SOF0 / SOF1,
   * whole blocks per scan); any other script is progressive */
  e.progressive = p->num_scans > 0 && !(p->scans[0].Ss == 0 && p->scans[0].Se == 63 && !p->optimize_scans);
  mjo_geometry(p, e.g, &e.mcus_per_row, &e.mcu_rows);
  for (ci = 0; ci < p->num_components; ci++) {
    planes[ci] = (uint16_t *)malloc((size_t)e.g[ci].pw * e.g[ci].ph * 2);
    e.uq[ci] = (int16_t *)calloc((size_t)e.g[ci].hpad * e.g[ci].wpad * 64, 2);
    e.q[ci] = (int16_t *)calloc((size_t)e.g[ci].hpad * e.g[ci].wpad * 64, 2);
  }
  /* std_huff_tables, jstdhuff.c:54 */
  memcpy(e.dc[0].bits, STD_DC_L_BITS, 17); memcpy(e.dc[0].huffval, STD_DC_VAL, 12);
  memcpy(e.dc[1].bits, STD_DC_C_BITS, 17); memcpy(e.dc[1].huffval, STD_DC_VAL, 12);
  memcpy(e.ac[0].bits, STD_AC_L_BITS, 17); memcpy(e.ac[0].huffval, STD_AC_L_VAL, 162);
  memcpy(e.ac[1].bits, STD_AC_C_BITS, 17); memcpy(e.ac[1].huffval, STD_AC_C_VAL, 162);
  /* slots 2 / 3: empty in the library until the application defines them, and a component that names an empty slot aborts
   * (JERR_NO_HUFF_TABLE); this restatement has no notion of application tables, so it assumes what oracle/refenc.c does for
   * -dctbl / -actbl and what the test harness hands the GPU library: copies of the standard tables of the same parity */
  e.dc[2] = e.dc[0]; e.dc[3] = e.dc[1]; e.ac[2] = e.ac[0]; e.ac[3] = e.ac[1];

  if (ps->coef) {
    /* jpeg_write_coefficients jctrans.c:44: the caller's quantized blocks (width_in_blocks x height_in_blocks per
     * component, natural order) are entropy-coded as they are; dummy blocks are made on the fly with the same
     * rule as the pixel path (compress_output jctrans.c:322-373).  No unquantized data exists, so no trellis
     * (jpeg_copy_critical_parameters switches it off, jctrans.c:102). */
    int r;
    if (p->trellis_quant) return 0;
    for (ci = 0; ci < p->num_components; ci++) {
      for (r = 0; r < e.g[ci].hib; r++)
        memcpy(e.q[ci] + (size_t)r * e.g[ci].wpad * 64, ps->coef[ci] + (size_t)r * ps->coef_pitch[ci] * 64, (size_t)e.g[ci].wib * 128);
      build_dummies(p, &e.g[ci], ci, e.q[ci]);
    }
  } else {
    if (prec_of(p) != 12)     /* compute_reciprocal(0): the reference divides by zero (see forward16) */
      for (ci = 0; ci < p->num_components; ci++) {
        int k;
        for (k = 0; k < 64; k++) if ((p->dct_method == 1 ? (unsigned)mjo_ifast_divisor(p->qtbl[p->quant_tbl_no[ci]][k], k) : (8u * (unsigned)p->qtbl[p->quant_tbl_no[ci]][k]) & 0xFFFFu) == 0u) return 0;
      }
    if (ps->src) import_planes16(p, ps, planes);
    else color_downsample16(p, ps->pixels, ps->row_stride, planes);
    forward16(p, planes, e.uq, e.q);
  }
  if (taps) {
    for (ci = 0; ci < p->num_components; ci++) {
      size_t nb = (size_t)e.g[ci].hpad * e.g[ci].wpad * 128;
      if (taps->planes[ci]) { size_t i; for (i = 0; i < (size_t)e.g[ci].pw * e.g[ci].ph; i++) taps->planes[ci][i] = (uint8_t)planes[ci][i]; }
      if (taps->coef_uq[ci]) memcpy(taps->coef_uq[ci], e.uq[ci], nb);
      if (taps->coef_q0[ci]) memcpy(taps->coef_q0[ci], e.q[ci], nb);
    }
  }

  emit_file_header(&e, &o);

  /* trellis passes (pass numbers < pass_number_scan_opt_base): SURVEY 3.3 table */
  if (p->trellis_quant && p->arith_code) {
    /* component 0 only, band of pass 0's scan selection (see trellis_component_arith): passes 0 .. T-1 with
     * T = pass_number_scan_opt_base = (1 or 2) * num_components * trellis_num_loops + 1 (jcmaster.c:1135-1138, :1010), all of them
     * the same pass as long as the tables stay what they are.  trellis_q_opt: the sums are zeroed in front of every pass with
     * pass_number % M == 1 and the tables re-estimated behind every pass with (pass_number + 1) % M == 0, M = (2 or 4) *
     * num_components (prepare_for_pass jcmaster.c:687-698, finish_pass_master :1016-1030) -- whether such a pass exists depends
     * on T: none for three components and one loop, one for a gray image and one loop (the estimate then only reaches the
     * DQT marker: no trellis pass follows it), one or more with further loops; component 0's table is the only one with
     * non-zero sums.  Restated pass by pass, as the reference runs them. */
    const int split = p->trellis_freq_split > 0 ? p->trellis_freq_split : 8;
    const int nb = p->use_scans_in_trellis ? 2 : 1;
    const int nloops = p->trellis_num_loops > 1 ? p->trellis_num_loops : 1;
    const int T = nb * p->num_components * nloops + 1, M = 2 * nb * p->num_components;
    const int Se = p->use_scans_in_trellis ? split : 63;
    if (!p->trellis_q_opt) trellis_component_arith(&e, 0, 1, Se);
    else {
      int pass, stale = 1;
      for (pass = 0; pass < T; pass++) {
        if (pass % M == 1) { memset(e.norm_src, 0, sizeof(e.norm_src)); memset(e.norm_coef, 0, sizeof(e.norm_coef)); }
        if (stale) { trellis_component_arith(&e, 0, 1, Se); stale = 0; }   /* (a pass with unchanged tables repeats the previous one) */
        if (Se >= 1) q_opt_accumulate(&e, 0);
        if ((pass + 1) % M == 0) {
          int ti, j;
          for (ti = 0; ti < 4; ti++)
            for (j = 1; j < 64; j++)
              if (e.norm_coef[ti][j] != 0.0) {
                int q = (int)(e.norm_src[ti][j] / e.norm_coef[ti][j] + 0.5);
                if (q > 254) q = 254;
                if (q < 1) q = 1;
                pp.qtbl[ti][j] = (uint16_t)q;
              }
          stale = 1;
        }
      }
    }
  } else if (p->trellis_quant) {
    for (ci = 0; ci < p->num_components; ci++) {
      mjo_scan ms;
      scan_t sc;
      dtbl dcd, acd;
      memset(&ms, 0, sizeof(ms));
      ms.comps_in_scan = 1; ms.component_index[0] = ci;
      ms.Ss = 1; ms.Se = 63; ms.Ah = 0; ms.Al = 0;
      int loop;
      setup_scan(&e, &sc, &ms);
      /* trellis_num_loops (gather, trellis) pass pairs per component: pass_number / (2 * trellis_num_loops) selects
       * the component (jcmaster.c:462-466); every trellis pass restarts from the unquantized coefficients with the
       * tables gathered from the previous loop's result */
      const int nloops = p->trellis_num_loops > 1 ? p->trellis_num_loops : 1;
      const int nbands = p->use_scans_in_trellis ? 2 : 1;          /* jcmaster.c:451-467 */
      const int split = p->trellis_freq_split > 0 ? p->trellis_freq_split : 8;
      const int ppc = 2 * nbands;                                   /* passes per component and loop */
      for (loop = 0; loop < nloops; loop++) {
        int band;
        for (band = 0; band < nbands; band++) {
          const int pass_number = (ci * nloops + loop) * ppc + 2 * band + 1;   /* of this trellis pass (its gather pass is one before) */
          const int bSs = nbands == 1 ? 1 : (band == 0 ? 1 : split + 1);
          const int bSe = nbands == 1 ? 63 : (band == 0 ? split : 63);
          ms.Ss = bSs; ms.Se = bSe;
          setup_scan(&e, &sc, &ms);
          gather_and_build(&e, &sc, 1);
          make_derived(&e.dc[p->dc_tbl_no[ci]], &dcd);
          make_derived(&e.ac[p->ac_tbl_no[ci]], &acd);
          if (p->trellis_q_opt && pass_number % (p->num_components * ppc) == 1) { /* prepare_for_pass jcmaster.c:687-698 */
            memset(e.norm_src, 0, sizeof(e.norm_src));
            memset(e.norm_coef, 0, sizeof(e.norm_coef));
          }
          if (bSe >= bSs) trellis_component(&e, ci, &dcd, &acd, bSs, bSe);   /* quantize_trellis returns at once for an empty band (:979-980) ... */
          if (p->trellis_q_opt) {
            if (bSe >= bSs) q_opt_accumulate(&e, ci);                        /* ... before it reaches the sums (:1299-1306) */
            if ((pass_number + 1) % (p->num_components * ppc) == 0) { /* finish_pass_master jcmaster.c:1014-1030 */
              int ti, j;
              for (ti = 0; ti < 4; ti++)
                for (j = 1; j < 64; j++)
                  if (e.norm_coef[ti][j] != 0.0) {
                    int q = (int)(e.norm_src[ti][j] / e.norm_coef[ti][j] + 0.5);
                    if (q > 254) q = 254;
                    if (q < 1) q = 1;
                    pp.qtbl[ti][j] = (uint16_t)q;
                  }
            }
          }
          gather_and_build(&e, &sc, 1);
        }
      }
    }
  }
  if (taps)
    for (ci = 0; ci < p->num_components; ci++)
      if (taps->coef_q[ci]) memcpy(taps->coef_q[ci], e.q[ci], (size_t)e.g[ci].hpad * e.g[ci].wpad * 128);

  if (!e.progressive && p->num_scans > 0) {
    /* sequential, several scans: every scan has its statistics pass and its own tables (jcmaster.c:1090-1101: two passes per scan
     * with optimize_coding), whole blocks of its components in its own MCU order */
    int si;
    for (si = 0; si < p->num_scans; si++) {
      scan_t sc;
      setup_scan(&e, &sc, &p->scans[si]);
      if (p->optimize_coding) gather_and_build(&e, &sc, 0);
      output_scan(&e, &sc, si == 0, &o);
    }
  } else if (!e.progressive) {
    mjo_scan ms;
    scan_t sc;
    memset(&ms, 0, sizeof(ms));
    ms.comps_in_scan = p->num_components;
    for (ci = 0; ci < p->num_components; ci++) ms.component_index[ci] = ci;
    ms.Ss = 0; ms.Se = 63;
    setup_scan(&e, &sc, &ms);
    if (p->optimize_coding) gather_and_build(&e, &sc, 0);
    output_scan(&e, &sc, 1, &o);
  } else if (!p->optimize_scans) {
    int si;
    for (si = 0; si < p->num_scans; si++) {
      scan_t sc;
      setup_scan(&e, &sc, &p->scans[si]);
      if (sc.Ss != 0 || sc.Ah == 0) gather_and_build(&e, &sc, 0);
      output_scan(&e, &sc, si == 0, &o);
    }
  } else {
    /* scan search: select_scans jcmaster.c:773-962; constants from jpeg_search_progression */
    bytebuf sb[MJO_MAX_SCANS];
    unsigned long size[MJO_MAX_SCANS];
    const int nsl_dc = 1, Al_max_luma = 3, nfs = 5;
    const int nsl = nsl_dc + (3 * Al_max_luma + 2) + (2 * nfs + 1); /* 23 */
    const int Al_max_chroma = p->num_components == 3 ? 2 : 0;
    const int nsc_dc = p->num_components == 3 ? 3 : 0;
    const int luma_fs_start = nsl_dc + 3 * Al_max_luma + 2;            /* 12 */
    const int chroma_fs_start = nsl + nsc_dc + (6 * Al_max_chroma + 4); /* 42 */
    unsigned long best_cost = 0;
    int best_Al_luma = 0, best_Al_chroma = 0, best_fs_luma = 0, best_fs_chroma = 0;
    int sn = 0, i, Al, min_Al, base, interleave_chroma_dc = 0;
    memset(sb, 0, sizeof(sb));
    memset(size, 0, sizeof(size));
    while (sn < p->num_scans) {
      mjo_scan ms = p->scans[sn];
      scan_t sc;
      int next;
      if (sn >= luma_fs_start && sn < nsl) ms.Al = best_Al_luma;             /* jcmaster.c:487-491 */
      if (sn >= chroma_fs_start && sn < p->num_scans) ms.Al = best_Al_chroma; /* :492-497 */
      setup_scan(&e, &sc, &ms);
      if (sc.Ss != 0 || sc.Ah == 0) gather_and_build(&e, &sc, 0);
      output_scan(&e, &sc, sn == 0, &sb[sn]);
      size[sn] = (unsigned long)sb[sn].len;
      next = sn + 1;
      if (next > 1 && next <= luma_fs_start) {
        if ((next - 1) % 3 == 2) {
          unsigned long cost = size[next - 2] + size[next - 1];
          Al = (next - 1) / 3;
          for (i = 0; i < Al; i++) cost += size[3 + 3 * i];
          if (Al == 0 || cost < best_cost) { best_cost = cost; best_Al_luma = Al; }
          else sn = luma_fs_start - 1;
        }
      } else if (next > luma_fs_start && next <= nsl) {
        if (next == luma_fs_start + 1) { best_fs_luma = 0; best_cost = size[next - 1]; }
        else if ((next - luma_fs_start) % 2 == 1) {
          int idx = (next - luma_fs_start) >> 1;
          unsigned long cost = size[next - 2] + size[next - 1];
          if (cost < best_cost) { best_cost = cost; best_fs_luma = idx; }
          if ((idx == 2 && best_fs_luma == 0) || (idx == 3 && best_fs_luma != 2) ||
              (idx == 4 && best_fs_luma != 4))
            sn = nsl - 1;
        }
      } else if (p->num_scans > nsl) {
        if (next == nsl + nsc_dc) {
          interleave_chroma_dc = size[nsl] <= size[nsl + 1] + size[nsl + 2];   /* jcmaster.c:836-838 */
        } else if (next > nsl + nsc_dc && next <= chroma_fs_start) {
          base = nsl + nsc_dc;
          if ((next - base) % 6 == 4) {
            unsigned long cost = size[next - 4] + size[next - 3] + size[next - 2] + size[next - 1];
            Al = (next - base) / 6;
            for (i = 0; i < Al; i++) cost += size[base + 4 + 6 * i] + size[base + 5 + 6 * i];
            if (Al == 0 || cost < best_cost) { best_cost = cost; best_Al_chroma = Al; }
            else sn = chroma_fs_start - 1;
          }
        } else if (next > chroma_fs_start && next <= p->num_scans) {
          if (next == chroma_fs_start + 2) {
            best_fs_chroma = 0;
            best_cost = size[next - 2] + size[next - 1];
          } else if ((next - chroma_fs_start) % 4 == 2) {
            int idx = (next - chroma_fs_start) >> 2;
            unsigned long cost = size[next - 4] + size[next - 3] + size[next - 2] + size[next - 1];
            if (cost < best_cost) { best_cost = cost; best_fs_chroma = idx; }
            if ((idx == 2 && best_fs_chroma == 0) || (idx == 3 && best_fs_chroma != 2) ||
                (idx == 4 && best_fs_chroma != 4))
              sn = p->num_scans - 1;
          }
        }
      }
      sn++;
    }
    /* final assembly, jcmaster.c:898-956 */
    min_Al = best_Al_luma < best_Al_chroma ? best_Al_luma : best_Al_chroma;
    bb_append(&o, &sb[0]);
    if (p->num_scans > nsl && p->dc_scan_opt_mode != 0) {   /* :904-913 */
      if (interleave_chroma_dc && p->dc_scan_opt_mode != 1) bb_append(&o, &sb[nsl]);
      else { bb_append(&o, &sb[nsl + 1]); bb_append(&o, &sb[nsl + 2]); }
    }
    if (best_fs_luma == 0) bb_append(&o, &sb[luma_fs_start]);
    else {
      bb_append(&o, &sb[luma_fs_start + 2 * (best_fs_luma - 1) + 1]);
      bb_append(&o, &sb[luma_fs_start + 2 * (best_fs_luma - 1) + 2]);
    }
    for (Al = best_Al_luma - 1; Al >= min_Al; Al--) bb_append(&o, &sb[3 + 3 * Al]);
    base = nsl + nsc_dc;
    if (p->num_scans > nsl) {
      if (best_fs_chroma == 0) {
        bb_append(&o, &sb[chroma_fs_start]);
        bb_append(&o, &sb[chroma_fs_start + 1]);
      } else {
        for (i = 2; i <= 5; i++) bb_append(&o, &sb[chroma_fs_start + 4 * (best_fs_chroma - 1) + i]);
      }
      for (Al = best_Al_chroma - 1; Al >= min_Al; Al--) {
        bb_append(&o, &sb[base + 6 * Al + 4]);
        bb_append(&o, &sb[base + 6 * Al + 5]);
      }
    }
    for (Al = min_Al - 1; Al >= 0; Al--) {
      bb_append(&o, &sb[3 + 3 * Al]);
      if (p->num_scans > nsl) {
        bb_append(&o, &sb[base + 6 * Al + 4]);
        bb_append(&o, &sb[base + 6 * Al + 5]);
      }
    }
    for (i = 0; i < p->num_scans; i++) free(sb[i].buf);
  }
  bb_put(&o, 0xFF); bb_put(&o, 0xD9); /* write_file_trailer jcmarker.c:791 */

  if (taps) {
    for (t = 0; t < 4; t++) {
      memcpy(taps->dc_bits[t], e.dc[t].bits, 17); memcpy(taps->dc_vals[t], e.dc[t].huffval, 256);
      memcpy(taps->ac_bits[t], e.ac[t].bits, 17); memcpy(taps->ac_vals[t], e.ac[t].huffval, 256);
    }
  }
  n = o.len;
  if (n <= cap) memcpy(out, o.buf, n); else n = 0;
  free(o.buf);
  for (ci = 0; ci < p->num_components; ci++) { free(planes[ci]); free(e.uq[ci]); free(e.q[ci]); }
  return n;
}


size_t mjo_encode(const mjo_params *p, const uint8_t *pixels, size_t row_stride,
                  uint8_t *out, size_t cap, mjo_taps *taps)
{
  plane_source ps;
  memset(&ps, 0, sizeof(ps));
  ps.pixels = pixels; ps.row_stride = row_stride;
  return encode_core(p, &ps, out, cap, taps);
}

size_t mjo_encode_planes(const mjo_params *p, const uint8_t *const src[MJO_MAX_COMPS],
                         const size_t src_stride[MJO_MAX_COMPS], const int src_w[MJO_MAX_COMPS],
                         const int src_h[MJO_MAX_COMPS], uint8_t *out, size_t cap, mjo_taps *taps)
{
  plane_source ps;
  memset(&ps, 0, sizeof(ps));
  ps.src = src; ps.src_stride = src_stride; ps.src_w = src_w; ps.src_h = src_h;
  return encode_core(p, &ps, out, cap, taps);
}

size_t mjo_encode_coefficients(const mjo_params *p, const int16_t *const coef[MJO_MAX_COMPS],
                               const size_t blocks_per_row[MJO_MAX_COMPS], uint8_t *out, size_t cap)
{
  plane_source ps;
  memset(&ps, 0, sizeof(ps));
  ps.coef = coef; ps.coef_pitch = blocks_per_row;
  return encode_core(p, &ps, out, cap, NULL);
}
