/* api_fuzz.c -- TEST INFRASTRUCTURE: one libjpeg client whose calls are drawn from a seed (public API only), so that the same
 * binary can be run against the reference's libjpeg.so.62 (expected output), with the interposing library in front of it, and
 * against the stand-alone library; tools/simt/fuzz_api.py compares what the three runs print.  What it reaches that cjpeg's
 * switches do not: every in_color_space / jpeg_set_colorspace pair of the path, quantization tables and table slots of the
 * application's own, jpeg_set_linear_quality, dc / ac table numbers, the extension parameters in any combination, JFIF fields,
 * COM / APPn markers, rows handed over in uneven chunks, jpeg_write_raw_data, a small destination buffer, a second image from the
 * same object; abbreviated datastreams (jpeg_write_tables, jpeg_suppress_tables and flags of single tables, jpeg_start_compress
 * with write_all_tables FALSE) and further images from the object WITHOUT setting the parameters again (libjpeg.txt "Abbreviated
 * datastreams and multiple images") -- those decisions come from a second generator, so that the cases of older seeds keep
 * their parameters.
 *   usage: api_fuzz SEED INDEX     -> one line per image: "<index>.<k> <bytes> <fnv64>" ("<index>.<k>t ..." for a tables-only stream) */
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jpeglib.h"
#include "jerror.h"

static unsigned long long rs;
static unsigned rnd(void) { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rs >> 33); }
static int ri(int lo, int hi) { return lo + (int)(rnd() % (unsigned)(hi - lo + 1)); }      /* inclusive */
static int chance(int pct) { return (int)(rnd() % 100u) < pct; }
static unsigned long long rs2;       /* the table / re-use decisions */
static unsigned rnd2(void) { rs2 = rs2 * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rs2 >> 33); }
static int chance2(int pct) { return (int)(rnd2() % 100u) < pct; }

static unsigned long long fnv(const unsigned char *b, size_t n) { unsigned long long h = 1469598103934665603ull; size_t i; for (i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull; return h; }

/* a destination manager with a small buffer that appends to a growing array (empty_output_buffer is called often) */
typedef struct { struct jpeg_destination_mgr pub; unsigned char *chunk; size_t chunk_size; unsigned char *all; size_t n, cap; } small_dest;
static void sd_append(small_dest *d, size_t k)
{
  if (d->n + k > d->cap) { d->cap = (d->n + k) * 2 + 1024; d->all = (unsigned char *)realloc(d->all, d->cap); }
  memcpy(d->all + d->n, d->chunk, k); d->n += k;
}
static void sd_init(j_compress_ptr c) { small_dest *d = (small_dest *)c->dest; d->pub.next_output_byte = d->chunk; d->pub.free_in_buffer = d->chunk_size; }
static boolean sd_empty(j_compress_ptr c) { small_dest *d = (small_dest *)c->dest; sd_append(d, d->chunk_size); sd_init(c); return TRUE; }
static void sd_term(j_compress_ptr c) { small_dest *d = (small_dest *)c->dest; sd_append(d, d->chunk_size - d->pub.free_in_buffer); }

static const J_COLOR_SPACE kRgbFamily[] = { JCS_RGB, JCS_EXT_RGB, JCS_EXT_RGBX, JCS_EXT_BGR, JCS_EXT_BGRX, JCS_EXT_XBGR, JCS_EXT_XRGB, JCS_EXT_RGBA, JCS_EXT_BGRA, JCS_EXT_ABGR, JCS_EXT_ARGB };
static const int kRgbSize[] = { 3, 3, 4, 3, 4, 4, 4, 4, 4, 4, 4 };
static const int kFactors[][6] = { { 1, 1, 1, 1, 1, 1 }, { 2, 1, 1, 1, 1, 1 }, { 2, 2, 1, 1, 1, 1 }, { 1, 2, 1, 1, 1, 1 }, { 4, 1, 1, 1, 1, 1 }, { 2, 2, 2, 1, 1, 1 },
                                   { 2, 1, 1, 1, 1, 2 }, { 1, 1, 2, 2, 2, 2 }, { 4, 2, 1, 1, 1, 1 }, { 2, 2, 1, 2, 2, 1 }, { 3, 1, 1, 1, 1, 1 }, { 2, 4, 1, 1, 1, 1 } };

/* a Huffman table of the application's own in slot t: the Annex K table of that kind with its symbols dealt anew over the code words
 * (any assignment of symbols to a legal set of code lengths is a legal table), now and then with its last symbols missing */
static void own_huff_table(struct jpeg_compress_struct *c, int t, int is_ac)
{
  JHUFF_TBL **slot = is_ac ? &c->ac_huff_tbl_ptrs[t] : &c->dc_huff_tbl_ptrs[t];
  const JHUFF_TBL *std = is_ac ? c->ac_huff_tbl_ptrs[t & 1] : c->dc_huff_tbl_ptrs[t & 1];
  JHUFF_TBL tmp;
  int nv = 0, l, i;
  if (std == NULL) return;
  tmp = *std;
  for (l = 1; l <= 16; l++) nv += tmp.bits[l];
  for (i = nv - 1; i > 0; i--) { const int j = (int)(rnd2() % (unsigned)(i + 1)); const UINT8 v = tmp.huffval[i]; tmp.huffval[i] = tmp.huffval[j]; tmp.huffval[j] = v; }
  if (chance2(20)) { int drop = 1 + (int)(rnd2() % 3u); for (l = 16; l >= 1 && drop > 0; l--) while (tmp.bits[l] > 0 && drop > 0 && nv > 2) { tmp.bits[l]--; nv--; drop--; } }
  if (*slot == NULL) *slot = jpeg_alloc_huff_table((j_common_ptr)c);
  memcpy((*slot)->bits, tmp.bits, sizeof(tmp.bits));
  memcpy((*slot)->huffval, tmp.huffval, sizeof(tmp.huffval));
  (*slot)->sent_table = FALSE;
}

/* keep: the object's parameters stay as the previous image left them (same size and pixel format, new pixels) */
static void one_image(struct jpeg_compress_struct *c, int tag, int k, int keep)
{
  static int w, h, ps, raw;
  static J_COLOR_SPACE in_cs;
  int x, y, i, ci, small, noise, all_tables = TRUE;
  unsigned char *img, *out = NULL;
  unsigned long outn = 0;
  small_dest sd;
  if (!keep) {
    int cs_kind;
    w = chance(10) ? ri(1, 900) : ri(1, 150); h = chance(10) ? ri(1, 12) : ri(1, 120);
    cs_kind = ri(0, 9); raw = 0; small = chance(30); noise = chance(35);
    if (cs_kind <= 5) { i = ri(0, 10); in_cs = kRgbFamily[i]; ps = kRgbSize[i]; }
    else if (cs_kind <= 7) { in_cs = JCS_GRAYSCALE; ps = 1; }
    else { in_cs = JCS_YCbCr; ps = 3; }
  } else { small = chance2(30); noise = chance2(35); }
  img = (unsigned char *)malloc((size_t)w * h * ps + 16);
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++)
      for (i = 0; i < ps; i++) {
        const unsigned r = keep ? rnd2() : rnd();
        img[((size_t)y * w + x) * ps + i] = noise ? (unsigned char)r : (unsigned char)(((x * (3 + i) + y * (5 - i)) & 0xFF) / 2 + (r & 0x1F) + (((x / 16 + y / 16) % 5) == 0 ? 64 : 0));
      }
  memset(&sd, 0, sizeof(sd));
  if (small) {
    sd.chunk_size = keep ? (size_t)(1 + rnd2() % 700u) : (size_t)ri(1, 700); sd.chunk = (unsigned char *)malloc(sd.chunk_size);
    sd.pub.init_destination = sd_init; sd.pub.empty_output_buffer = sd_empty; sd.pub.term_destination = sd_term;
  }
  if (keep) goto start;
  if (small) c->dest = &sd.pub;
  else { c->dest = NULL; jpeg_mem_dest(c, &out, &outn); }
  if (chance(35)) jpeg_c_set_int_param(c, JINT_COMPRESS_PROFILE, JCP_FASTEST);
  else jpeg_c_set_int_param(c, JINT_COMPRESS_PROFILE, JCP_MAX_COMPRESSION);
  c->image_width = w; c->image_height = h; c->input_components = ps; c->in_color_space = in_cs;
  jpeg_set_defaults(c);
  jpeg_c_set_bool_param(c, JBOOLEAN_TRELLIS_EOB_OPT, FALSE);    /* (the one extension parameter jpeg_set_defaults leaves as the previous image set it, jcparam.c:495-518) */
  c->dct_method = JDCT_ISLOW;
  if (chance(20) && ps != 1) {      /* (the script is rebuilt for the new components, as cjpeg does after its colour-space switches) */
    jpeg_set_colorspace(c, in_cs == JCS_YCbCr || chance(50) ? JCS_GRAYSCALE : JCS_RGB);
    if (c->num_scans > 0) jpeg_simple_progression(c);
  }
  if (chance(8)) c->write_Adobe_marker = !c->write_Adobe_marker;
  if (chance(15)) jpeg_c_set_int_param(c, JINT_BASE_QUANT_TBL_IDX, ri(0, 8));
  i = ri(0, 9);
  if (i <= 5) jpeg_set_quality(c, ri(1, 100), chance(60));
  else if (i <= 7) jpeg_set_linear_quality(c, ri(1, 600), chance(60));
  else {       /* tables of the application's own, in any slots, assigned per component */
    unsigned int tbl[64];
    int t, nt = ri(1, 4), hi = chance(70) ? 255 : 1200;
    for (t = 0; t < nt; t++) { for (i = 0; i < 64; i++) tbl[i] = (unsigned)ri(1, hi); jpeg_add_quant_table(c, t, tbl, ri(20, 300), chance(50)); }
    for (ci = 0; ci < c->num_components; ci++) c->comp_info[ci].quant_tbl_no = ri(0, nt - 1);
  }
  if (c->num_components == 3 && chance(70)) {
    const int *f = kFactors[ri(0, 11)];
    for (ci = 0; ci < 3; ci++) { c->comp_info[ci].h_samp_factor = f[2 * ci]; c->comp_info[ci].v_samp_factor = f[2 * ci + 1]; }
  } else if (c->num_components == 1 && chance(30)) { c->comp_info[0].h_samp_factor = ri(1, 4); c->comp_info[0].v_samp_factor = chance(70) ? 1 : ri(1, 4); }
  if (chance(15)) for (ci = 0; ci < c->num_components; ci++) { c->comp_info[ci].dc_tbl_no = ri(0, 1); c->comp_info[ci].ac_tbl_no = ri(0, 1); }
  if (chance(c->optimize_coding ? 6 : 25)) c->optimize_coding = !c->optimize_coding;   /* (switching it OFF under the max-compression profile leaves the trellis without optimal tables: refused, INTEGRATION.md 1a') */
  if (chance(12)) c->arith_code = TRUE;
  i = ri(0, 9);
  if (i <= 2) { c->num_scans = 0; c->scan_info = NULL; }
  else if (i <= 4) jpeg_simple_progression(c);
  if (chance(25)) { if (chance(50)) c->restart_interval = (unsigned)ri(1, 40); else c->restart_in_rows = ri(1, 4); }
  if (chance(12)) c->smoothing_factor = ri(1, 100);
  if (chance(20)) jpeg_c_set_bool_param(c, JBOOLEAN_TRELLIS_QUANT, chance(50));
  if (chance(20)) jpeg_c_set_bool_param(c, JBOOLEAN_TRELLIS_QUANT_DC, chance(50));
  if (chance(15)) jpeg_c_set_bool_param(c, JBOOLEAN_TRELLIS_EOB_OPT, TRUE);
  if (chance(15)) { jpeg_c_set_bool_param(c, JBOOLEAN_USE_SCANS_IN_TRELLIS, TRUE); if (chance(60)) jpeg_c_set_int_param(c, JINT_TRELLIS_FREQ_SPLIT, ri(1, 62)); }
  if (chance(12)) jpeg_c_set_bool_param(c, JBOOLEAN_TRELLIS_Q_OPT, TRUE);
  if (chance(15)) jpeg_c_set_int_param(c, JINT_TRELLIS_NUM_LOOPS, ri(1, 3));
  if (chance(15)) jpeg_c_set_bool_param(c, JBOOLEAN_OVERSHOOT_DERINGING, chance(50));
  if (chance(15)) jpeg_c_set_bool_param(c, JBOOLEAN_OPTIMIZE_SCANS, chance(50));
  if (chance(12)) jpeg_c_set_bool_param(c, JBOOLEAN_USE_LAMBDA_WEIGHT_TBL, chance(50));
  if (chance(12)) { jpeg_c_set_float_param(c, JFLOAT_LAMBDA_LOG_SCALE1, (float)ri(-2, 20) + 0.25f * (float)ri(0, 3)); jpeg_c_set_float_param(c, JFLOAT_LAMBDA_LOG_SCALE2, chance(30) ? 0.0f : (float)ri(6, 22) + 0.5f); }
  if (chance(12)) jpeg_c_set_float_param(c, JFLOAT_TRELLIS_DELTA_DC_WEIGHT, 0.25f * (float)ri(0, 12));
  if (chance(15)) { jpeg_c_set_int_param(c, JINT_DC_SCAN_OPT_MODE, ri(0, 2)); if (chance(60) && c->num_scans > 0) jpeg_simple_progression(c); }
  if (chance(15)) { c->write_JFIF_header = chance(50); c->JFIF_minor_version = (UINT8)ri(1, 2); c->density_unit = (UINT8)ri(0, 2); c->X_density = (UINT16)ri(1, 600); c->Y_density = (UINT16)ri(1, 600); }
  if (c->data_precision == 8 && c->num_components == 3 && !c->raw_data_in && in_cs != JCS_YCbCr) (void)chance(0);   /* (a draw that decides nothing: kept, under its old condition, so that the seeds recorded in profiles/r05z_dropin_fuzz.md give the same cases) */
  if (in_cs == JCS_YCbCr && c->jpeg_color_space == JCS_YCbCr && chance(40)) raw = 1;      /* planes through jpeg_write_raw_data (the samples are components already) */
  if (raw) c->raw_data_in = TRUE;
  if (getenv("API_FUZZ_VERBOSE")) {
    fprintf(stderr, "image %d.%d: %dx%d in_cs %d ps %d -> jpeg_cs %d ncomp %d raw %d small %d profile %s optimize %d arith %d scans %d optscans %d restart %u/%d smooth %d\n", tag, k, w, h, (int)in_cs, ps,
            (int)c->jpeg_color_space, c->num_components, raw, small, jpeg_c_get_int_param(c, JINT_COMPRESS_PROFILE) == JCP_FASTEST ? "fastest" : "max", c->optimize_coding, c->arith_code, c->num_scans,
            jpeg_c_get_bool_param(c, JBOOLEAN_OPTIMIZE_SCANS), c->restart_interval, c->restart_in_rows, c->smoothing_factor);
    fprintf(stderr, "   trellis %d dc %d eob %d scans_in %d split %d qopt %d loops %d dering %d l1 %g l2 %g dcw %g dcscan %d jfif %d\n", jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_QUANT), jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_QUANT_DC),
            jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_EOB_OPT), jpeg_c_get_bool_param(c, JBOOLEAN_USE_SCANS_IN_TRELLIS), jpeg_c_get_int_param(c, JINT_TRELLIS_FREQ_SPLIT), jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_Q_OPT),
            jpeg_c_get_int_param(c, JINT_TRELLIS_NUM_LOOPS), jpeg_c_get_bool_param(c, JBOOLEAN_OVERSHOOT_DERINGING), jpeg_c_get_float_param(c, JFLOAT_LAMBDA_LOG_SCALE1), jpeg_c_get_float_param(c, JFLOAT_LAMBDA_LOG_SCALE2),
            jpeg_c_get_float_param(c, JFLOAT_TRELLIS_DELTA_DC_WEIGHT), jpeg_c_get_int_param(c, JINT_DC_SCAN_OPT_MODE), c->write_JFIF_header);
    for (ci = 0; ci < c->num_components; ci++) fprintf(stderr, "   comp %d: %dx%d q %d dc %d ac %d\n", ci, c->comp_info[ci].h_samp_factor, c->comp_info[ci].v_samp_factor, c->comp_info[ci].quant_tbl_no, c->comp_info[ci].dc_tbl_no, c->comp_info[ci].ac_tbl_no);
  }
  if (chance2(14)) {      /* Huffman tables of its own, table numbers up to 3 */
    const int hi = chance2(50) ? 1 : 3;
    int t;
    for (t = 0; t <= hi; t++) { if (t > 1 || chance2(60)) own_huff_table(c, t, 0); if (t > 1 || chance2(60)) own_huff_table(c, t, 1); }
    if (chance2(60)) for (ci = 0; ci < c->num_components; ci++) { c->comp_info[ci].dc_tbl_no = (int)(rnd2() % (unsigned)(hi + 1)); c->comp_info[ci].ac_tbl_no = (int)(rnd2() % (unsigned)(hi + 1)); }
    if (chance2(50)) c->optimize_coding = FALSE;
  }
  if (chance2(25)) c->dct_method = JDCT_IFAST;
start:
  {
    /* tables: mostly the whole file; otherwise a tables-only stream first, flags of single tables, or write_all_tables FALSE */
    const int mode = (int)(rnd2() % 100u);
    int t;
    if (mode >= 60 && mode < 72) {
      unsigned char *tb = NULL;
      unsigned long tn = 0;
      c->dest = NULL; jpeg_mem_dest(c, &tb, &tn);
      jpeg_write_tables(c);
      printf("%d.%dt %lu %016llx\n", tag, k, tn, fnv(tb, tn));
      free(tb);
      all_tables = chance2(15);
    } else if (mode >= 72 && mode < 84) {
      jpeg_suppress_tables(c, chance2(70));
      for (t = 0; t < NUM_QUANT_TBLS; t++) {
        if (c->quant_tbl_ptrs[t] && chance2(25)) c->quant_tbl_ptrs[t]->sent_table = chance2(50);
        if (c->dc_huff_tbl_ptrs[t] && chance2(25)) c->dc_huff_tbl_ptrs[t]->sent_table = chance2(50);
        if (c->ac_huff_tbl_ptrs[t] && chance2(25)) c->ac_huff_tbl_ptrs[t]->sent_table = chance2(50);
      }
      all_tables = FALSE;
    } else if (mode >= 84) all_tables = FALSE;
    if (keep || mode >= 60) {       /* (the destination again: a tables-only stream used it, or this is a further image) */
      if (small) c->dest = &sd.pub;
      else { c->dest = NULL; jpeg_mem_dest(c, &out, &outn); }
    }
    if (getenv("API_FUZZ_VERBOSE")) fprintf(stderr, "image %d.%d: keep %d tables mode %d write_all_tables %d\n", tag, k, keep, mode, all_tables);
  }
  if (getenv("API_FUZZ_DUMP") && k == 1) {     /* (debugging aid: the object's state in front of the second image) */
    FILE *f = fopen(getenv("API_FUZZ_DUMP"), "w");
    const unsigned char *b = (const unsigned char *)c;
    size_t j;
    int t;
    for (j = 0; j < sizeof(*c); j += 8) { unsigned long long v; memcpy(&v, b + j, 8); fprintf(f, "cinfo+%zu: %016llx\n", j, v); }
    for (t = 0; t < 4; t++) {
      if (c->quant_tbl_ptrs[t]) { fprintf(f, "qtbl %d sent %d:", t, c->quant_tbl_ptrs[t]->sent_table); for (j = 0; j < 64; j++) fprintf(f, " %u", c->quant_tbl_ptrs[t]->quantval[j]); fprintf(f, "\n"); }
      if (c->dc_huff_tbl_ptrs[t]) { fprintf(f, "dc %d:", t); for (j = 0; j < 17; j++) fprintf(f, " %u", c->dc_huff_tbl_ptrs[t]->bits[j]); for (j = 0; j < 16; j++) fprintf(f, " %u", c->dc_huff_tbl_ptrs[t]->huffval[j]); fprintf(f, "\n"); }
      if (c->ac_huff_tbl_ptrs[t]) { fprintf(f, "ac %d:", t); for (j = 0; j < 17; j++) fprintf(f, " %u", c->ac_huff_tbl_ptrs[t]->bits[j]); for (j = 0; j < 32; j++) fprintf(f, " %u", c->ac_huff_tbl_ptrs[t]->huffval[j]); fprintf(f, "\n"); }
    }
    { const unsigned char *m = (const unsigned char *)c->master; for (j = 0; j < 4400; j += 8) { unsigned long long v; memcpy(&v, m + j, 8); fprintf(f, "master+%zu: %016llx\n", j, v); } }
    fclose(f);
  }
  jpeg_start_compress(c, all_tables);
  if (chance(25)) { unsigned char com[300]; const int n = ri(0, 300); for (i = 0; i < n; i++) com[i] = (unsigned char)rnd(); jpeg_write_marker(c, chance(50) ? JPEG_COM : JPEG_APP0 + ri(1, 15), com, (unsigned)n); }
  if (chance(10)) { const int n = ri(1, 40); jpeg_write_m_header(c, JPEG_APP0 + 5, (unsigned)n); for (i = 0; i < n; i++) jpeg_write_m_byte(c, ri(0, 255)); }
  if (!raw) {
    while (c->next_scanline < c->image_height) {
      JSAMPROW r[24];
      int n = ri(1, 24);
      for (i = 0; i < n; i++) { const JDIMENSION row = c->next_scanline + (JDIMENSION)i; r[i] = img + (size_t)(row < c->image_height ? row : c->image_height - 1) * w * ps; }
      if (c->next_scanline + (JDIMENSION)n > c->image_height) n = (int)(c->image_height - c->next_scanline);
      jpeg_write_scanlines(c, r, (JDIMENSION)n);
    }
  } else {
    /* one iMCU row per call: component ci hands over v_samp*8 rows of width_in_blocks*8 samples, here its share of the pixels
     * taken at the component's own resolution (nearest sample), the edges replicated */
    JSAMPARRAY planes[3];
    JSAMPROW rowp[3][32];
    unsigned char *buf[3];
    for (ci = 0; ci < 3; ci++) {
      const jpeg_component_info *ce = &c->comp_info[ci];
      buf[ci] = (unsigned char *)malloc((size_t)ce->width_in_blocks * 8 * ce->v_samp_factor * 8);
      for (i = 0; i < ce->v_samp_factor * 8; i++) rowp[ci][i] = buf[ci] + (size_t)i * ce->width_in_blocks * 8;
      planes[ci] = rowp[ci];
    }
    while (c->next_scanline < c->image_height) {
      for (ci = 0; ci < 3; ci++) {
        const jpeg_component_info *ce = &c->comp_info[ci];
        const int hs = c->max_h_samp_factor / ce->h_samp_factor, vs = c->max_v_samp_factor / ce->v_samp_factor;
        const int row0 = (int)(c->next_scanline / (unsigned)c->max_v_samp_factor) * ce->v_samp_factor / 1;
        for (y = 0; y < ce->v_samp_factor * 8; y++)
          for (x = 0; x < (int)ce->width_in_blocks * 8; x++) {
            int sx = x * hs, sy = ((int)c->next_scanline / (c->max_v_samp_factor) * ce->v_samp_factor + y) * vs;
            (void)row0;
            if (sx >= w) sx = w - 1;
            if (sy >= h) sy = h - 1;
            rowp[ci][y][x] = img[((size_t)sy * w + sx) * 3 + ci];
          }
      }
      jpeg_write_raw_data(c, planes, (JDIMENSION)c->max_v_samp_factor * 8);
    }
    for (ci = 0; ci < 3; ci++) free(buf[ci]);
  }
  jpeg_finish_compress(c);
  if (getenv("API_FUZZ_SAVE")) {      /* (debugging aid: the files themselves) */
    char name[512];
    FILE *f;
    snprintf(name, sizeof(name), "%s.%d.jpg", getenv("API_FUZZ_SAVE"), k);
    if ((f = fopen(name, "wb")) != NULL) { fwrite(small ? sd.all : out, 1, small ? sd.n : (size_t)outn, f); fclose(f); }
  }
  if (small) { printf("%d.%d %lu %016llx\n", tag, k, (unsigned long)sd.n, fnv(sd.all, sd.n)); free(sd.all); free(sd.chunk); }
  else { printf("%d.%d %lu %016llx\n", tag, k, outn, fnv(out, outn)); free(out); }
  fflush(stdout);
  free(img);
}

int main(int argc, char **argv)
{
  struct jpeg_compress_struct c;
  struct jpeg_error_mgr err;
  int index, k, nimg, keep_from = 1000;
  if (argc != 3) { fprintf(stderr, "usage: api_fuzz SEED INDEX\n"); return 2; }
  index = atoi(argv[2]);
  rs = (unsigned long long)atoll(argv[1]) * 1000003ull + (unsigned long long)index * 7919ull + 12345ull;
  rs2 = rs ^ 0x9E3779B97F4A7C15ull;
  for (k = 0; k < 4; k++) { rnd(); rnd2(); }
  c.err = jpeg_std_error(&err);
  jpeg_create_compress(&c);
  nimg = chance(30) ? 2 : 1;          /* a second image from the same object (parameters set anew, as libjpeg.txt asks) */
  if (chance2(35)) { nimg += 1 + (int)(rnd2() % 2u); keep_from = nimg - (chance2(50) ? 1 : 2); if (keep_from < 1) keep_from = 1; }   /* further images with the parameters left alone */
  for (k = 0; k < nimg; k++) {
    /* API_FUZZ_FRESH=1: a new object for the second image.  The REFERENCE's bytes for a second image from the same object depend on
     * the first one: select_scan_parameters sets Ss / Se for the trellis passes but not Ah / Al (jcmaster.c:451-466), so the trellis'
     * statistics passes of a progressive image run with the Ah / Al of the previous image's last coded scan (refinement statistics
     * after a simple script, AC-first statistics at the previous image's best chroma Al after a scan search); a fresh object has
     * 0 / 0.  The device path codes every image as a fresh object would (known divergences): tools/simt/fuzz_api.py
     * runs the reference with API_FUZZ_FRESH=1 and the libraries under test without it. */
    if (k > 0 && getenv("API_FUZZ_FRESH")) { jpeg_destroy_compress(&c); jpeg_create_compress(&c); }
    one_image(&c, index, k, k >= keep_from);
  }
  jpeg_destroy_compress(&c);
  return 0;
}
