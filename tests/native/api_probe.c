/* api_probe.c -- prints everything the host-side libjpeg COMPRESS API leaves in a compress object (parameters, tables,
 * scan scripts, message texts, memory-manager behaviour, tables-only datastream).  The same binary is run once against
 * the reference's libjpeg.so.62 and once against mozjpeg_amd/standalone/libjpeg.so.62: the outputs must be identical
 * (tests/test_standalone_api.py).  No pixel is compressed here, so it runs without a GPU. */
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jpeglib.h"
#include "jerror.h"

static jmp_buf env;
static void my_exit(j_common_ptr cinfo)
{
  char buf[JMSG_LENGTH_MAX];
  (*cinfo->err->format_message) (cinfo, buf);
  printf("ERROR %d: %s\n", cinfo->err->msg_code, buf);
  longjmp(env, 1);
}
static void my_output(j_common_ptr cinfo)
{
  char buf[JMSG_LENGTH_MAX];
  (*cinfo->err->format_message) (cinfo, buf);
  printf("MESSAGE: %s\n", buf);
}

static void dump(j_compress_ptr c, const char *title)
{
  int i, k;
  printf("== %s\n", title);
  printf("size %ux%u in_comps %d in_cs %d jpeg_cs %d ncomp %d prec %d\n", c->image_width, c->image_height, c->input_components,
         c->in_color_space, c->jpeg_color_space, c->num_components, c->data_precision);
  printf("opt %d arith %d raw %d ccir %d smooth %d dct %d restart %u/%d jfif %d %d.%d unit %d dens %dx%d adobe %d\n", c->optimize_coding,
         c->arith_code, c->raw_data_in, c->CCIR601_sampling, c->smoothing_factor, c->dct_method, c->restart_interval, c->restart_in_rows,
         c->write_JFIF_header, c->JFIF_major_version, c->JFIF_minor_version, c->density_unit, c->X_density, c->Y_density, c->write_Adobe_marker);
  for (i = 0; i < c->num_components; i++) {
    jpeg_component_info *p = &c->comp_info[i];
    printf(" comp %d id %d samp %dx%d q %d dc %d ac %d\n", i, p->component_id, p->h_samp_factor, p->v_samp_factor, p->quant_tbl_no, p->dc_tbl_no, p->ac_tbl_no);
  }
  for (i = 0; i < NUM_QUANT_TBLS; i++)
    if (c->quant_tbl_ptrs[i]) { printf(" qtbl %d sent %d:", i, c->quant_tbl_ptrs[i]->sent_table); for (k = 0; k < 64; k++) printf(" %u", c->quant_tbl_ptrs[i]->quantval[k]); printf("\n"); }
  for (i = 0; i < NUM_HUFF_TBLS; i++) {
    JHUFF_TBL *t[2] = { c->dc_huff_tbl_ptrs[i], c->ac_huff_tbl_ptrs[i] };
    int j;
    for (j = 0; j < 2; j++)
      if (t[j]) { int n = 0; printf(" huff %s%d sent %d bits", j ? "ac" : "dc", i, t[j]->sent_table); for (k = 1; k <= 16; k++) { printf(" %d", t[j]->bits[k]); n += t[j]->bits[k]; }
                  printf(" vals"); for (k = 0; k < n; k++) printf(" %d", t[j]->huffval[k]); printf("\n"); }
  }
  printf(" scans %d:", c->num_scans);
  for (i = 0; i < c->num_scans; i++) {
    const jpeg_scan_info *s = &c->scan_info[i];
    printf(" [");
    for (k = 0; k < s->comps_in_scan; k++) printf("%d%s", s->component_index[k], k + 1 < s->comps_in_scan ? "," : "");
    printf(":%d-%d:%d,%d]", s->Ss, s->Se, s->Ah, s->Al);
  }
  printf("\n");
  printf(" ext: optscans %d trellis %d dc %d eob %d lwt %d sit %d qopt %d dering %d | profile %x dcmode %d base %d split %d loops %d | l1 %g l2 %g ddw %g\n",
         jpeg_c_get_bool_param(c, JBOOLEAN_OPTIMIZE_SCANS), jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_QUANT), jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_QUANT_DC),
         jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_EOB_OPT), jpeg_c_get_bool_param(c, JBOOLEAN_USE_LAMBDA_WEIGHT_TBL), jpeg_c_get_bool_param(c, JBOOLEAN_USE_SCANS_IN_TRELLIS),
         jpeg_c_get_bool_param(c, JBOOLEAN_TRELLIS_Q_OPT), jpeg_c_get_bool_param(c, JBOOLEAN_OVERSHOOT_DERINGING),
         (unsigned)jpeg_c_get_int_param(c, JINT_COMPRESS_PROFILE), jpeg_c_get_int_param(c, JINT_DC_SCAN_OPT_MODE), jpeg_c_get_int_param(c, JINT_BASE_QUANT_TBL_IDX),
         jpeg_c_get_int_param(c, JINT_TRELLIS_FREQ_SPLIT), jpeg_c_get_int_param(c, JINT_TRELLIS_NUM_LOOPS),
         jpeg_c_get_float_param(c, JFLOAT_LAMBDA_LOG_SCALE1), jpeg_c_get_float_param(c, JFLOAT_LAMBDA_LOG_SCALE2), jpeg_c_get_float_param(c, JFLOAT_TRELLIS_DELTA_DC_WEIGHT));
}

int main(void)
{
  struct jpeg_compress_struct c;
  struct jpeg_error_mgr err;
  int q, idx, mode;

  c.err = jpeg_std_error(&err);
  err.error_exit = my_exit;
  err.output_message = my_output;
  if (setjmp(env)) { printf("unexpected exit\n"); return 1; }
  jpeg_create_compress(&c);
  c.image_width = 227; c.image_height = 149; c.input_components = 3; c.in_color_space = JCS_RGB;
  jpeg_set_defaults(&c);
  dump(&c, "defaults (max compression), RGB in");
  jpeg_set_quality(&c, 75, TRUE);
  dump(&c, "+ set_quality 75 baseline (now with the profile's base table)");
  for (q = 1; q <= 100; q += 33) { jpeg_set_quality(&c, q, FALSE); printf("q%d:", q); for (idx = 0; idx < 64; idx += 9) printf(" %u/%u", c.quant_tbl_ptrs[0]->quantval[idx], c.quant_tbl_ptrs[1]->quantval[idx]); printf("\n"); }
  for (idx = 0; idx <= 9; idx++) {
    jpeg_c_set_int_param(&c, JINT_BASE_QUANT_TBL_IDX, idx);
    jpeg_set_quality(&c, 60, TRUE);
    printf("base %d -> %d:", idx, jpeg_c_get_int_param(&c, JINT_BASE_QUANT_TBL_IDX));
    for (q = 0; q < 64; q++) printf(" %u", c.quant_tbl_ptrs[0]->quantval[q]);
    for (q = 0; q < 64; q++) printf(" %u", c.quant_tbl_ptrs[1]->quantval[q]);
    printf("\n");
  }
  printf("scaling: %d %d %d %d %g %g\n", jpeg_quality_scaling(0), jpeg_quality_scaling(25), jpeg_quality_scaling(50), jpeg_quality_scaling(100),
         jpeg_float_quality_scaling(33.3f), jpeg_float_quality_scaling(87.5f));
  jpeg_set_linear_quality(&c, 37, FALSE);
  dump(&c, "linear quality 37");
  for (mode = 0; mode <= 2; mode++) {
    jpeg_c_set_int_param(&c, JINT_DC_SCAN_OPT_MODE, mode);
    jpeg_c_set_bool_param(&c, JBOOLEAN_OPTIMIZE_SCANS, FALSE);
    jpeg_simple_progression(&c);
    dump(&c, "simple progression, dc_scan_opt_mode");
    jpeg_c_set_bool_param(&c, JBOOLEAN_OPTIMIZE_SCANS, TRUE);
    jpeg_simple_progression(&c);
    dump(&c, "search progression, dc_scan_opt_mode");
  }
  jpeg_set_colorspace(&c, JCS_GRAYSCALE);
  jpeg_simple_progression(&c);
  dump(&c, "grayscale + search progression");
  jpeg_set_colorspace(&c, JCS_RGB);
  jpeg_simple_progression(&c);
  dump(&c, "RGB output + progression (all-purpose script)");
  c.input_components = 4; c.in_color_space = JCS_CMYK;
  jpeg_default_colorspace(&c);
  jpeg_simple_progression(&c);
  dump(&c, "CMYK + progression");
  jpeg_set_colorspace(&c, JCS_YCCK);
  dump(&c, "YCCK");
  c.input_components = 2; jpeg_set_colorspace(&c, JCS_UNKNOWN);
  dump(&c, "unknown, 2 components");
  /* fastest profile on a fresh object */
  jpeg_destroy_compress(&c);
  jpeg_create_compress(&c);
  jpeg_c_set_int_param(&c, JINT_COMPRESS_PROFILE, JCP_FASTEST);
  c.input_components = 1; c.in_color_space = JCS_GRAYSCALE; c.image_width = 9; c.image_height = 9;
  jpeg_set_defaults(&c);
  dump(&c, "defaults (fastest), gray in");
  c.input_components = 3; c.in_color_space = JCS_EXT_BGRX; c.data_precision = 12;
  jpeg_set_defaults(&c);
  jpeg_simple_progression(&c);
  dump(&c, "fastest, BGRX, 12-bit, progression");
  printf("supported: %d %d %d %d %d %d\n", jpeg_c_bool_param_supported(&c, JBOOLEAN_TRELLIS_Q_OPT), jpeg_c_bool_param_supported(&c, (J_BOOLEAN_PARAM)1),
         jpeg_c_float_param_supported(&c, JFLOAT_LAMBDA_LOG_SCALE1), jpeg_c_float_param_supported(&c, (J_FLOAT_PARAM)2),
         jpeg_c_int_param_supported(&c, JINT_DC_SCAN_OPT_MODE), jpeg_c_int_param_supported(&c, (J_INT_PARAM)3));
  /* errors: message texts and parameters */
  if (!setjmp(env)) jpeg_c_set_int_param(&c, JINT_COMPRESS_PROFILE, 12345);
  if (!setjmp(env)) jpeg_c_set_bool_param(&c, (J_BOOLEAN_PARAM)7, TRUE);
  if (!setjmp(env)) { unsigned int t[64] = { 0 }; jpeg_add_quant_table(&c, 9, t, 100, TRUE); }
  if (!setjmp(env)) jpeg_write_marker(&c, JPEG_COM, (const JOCTET *)"x", 1);
  if (!setjmp(env)) { c.in_color_space = (J_COLOR_SPACE)99; jpeg_default_colorspace(&c); }
  if (!setjmp(env)) jpeg_set_colorspace(&c, (J_COLOR_SPACE)77);
  if (!setjmp(env)) { c.global_state = 101; jpeg_set_defaults(&c); }
  c.global_state = 100;
  if (!setjmp(env)) { struct jpeg_compress_struct d; d.err = c.err; jpeg_CreateCompress(&d, 61, sizeof(d)); }
  if (!setjmp(env)) { struct jpeg_compress_struct d; d.err = c.err; jpeg_CreateCompress(&d, JPEG_LIB_VERSION, sizeof(d) - 8); }
  err.trace_level = 0; WARNMS(&c, JWRN_TOO_MUCH_DATA); WARNMS(&c, JWRN_TOO_MUCH_DATA); printf("warnings %ld\n", err.num_warnings);
  err.trace_level = 3; WARNMS(&c, JWRN_TOO_MUCH_DATA); TRACEMS2(&c, 2, JTRC_DRI, 7, 8); TRACEMS(&c, 4, JTRC_EOI); printf("warnings %ld\n", err.num_warnings);
  (*err.reset_error_mgr) ((j_common_ptr)&c); printf("after reset %ld %d\n", err.num_warnings, err.msg_code);
  err.trace_level = 0;
  {
    int codes[] = { JERR_BAD_PRECISION, JERR_BAD_STATE, JERR_TOO_LITTLE_DATA, JERR_NOT_COMPILED, JERR_OUT_OF_MEMORY, JERR_BAD_VIRTUAL_ACCESS, JMSG_COPYRIGHT, JMSG_VERSION, 9999 };
    unsigned i;
    for (i = 0; i < sizeof(codes) / sizeof(codes[0]); i++) { char buf[JMSG_LENGTH_MAX]; err.msg_code = codes[i]; err.msg_parm.i[0] = 11; err.msg_parm.i[1] = 22; (*err.format_message) ((j_common_ptr)&c, buf);
      if (codes[i] != JMSG_VERSION && codes[i] != JMSG_COPYRIGHT) printf("msg %d: %s\n", codes[i], buf); else printf("msg %d: (%s)\n", codes[i], buf[0] ? "non-empty" : "empty"); }
  }
  /* memory manager */
  {
    JSAMPARRAY sa = (*c.mem->alloc_sarray) ((j_common_ptr)&c, JPOOL_IMAGE, 100, 5);
    JBLOCKARRAY ba = (*c.mem->alloc_barray) ((j_common_ptr)&c, JPOOL_IMAGE, 7, 3);
    jvirt_barray_ptr vb = (*c.mem->request_virt_barray) ((j_common_ptr)&c, JPOOL_IMAGE, TRUE, 11, 20, 4);
    jvirt_sarray_ptr vs = (*c.mem->request_virt_sarray) ((j_common_ptr)&c, JPOOL_IMAGE, FALSE, 33, 10, 2);
    JBLOCKARRAY r;
    JSAMPARRAY s;
    int i, nz = 0;
    memset(sa[4], 1, 100 * 2); ba[2][6][63] = 5;      /* 12-bit object: rows hold 2-byte samples */
    printf("rows distinct %d %d\n", sa[0] != sa[1], ba[0] != ba[1]);
    (*c.mem->realize_virt_arrays) ((j_common_ptr)&c);
    r = (*c.mem->access_virt_barray) ((j_common_ptr)&c, vb, 8, 4, FALSE);
    for (i = 0; i < 11 * 64; i++) nz += r[3][0][i] != 0;
    printf("pre-zeroed read: %d nonzero\n", nz);
    r = (*c.mem->access_virt_barray) ((j_common_ptr)&c, vb, 0, 2, TRUE); r[1][10][63] = 42;
    r = (*c.mem->access_virt_barray) ((j_common_ptr)&c, vb, 1, 1, FALSE); printf("read back %d\n", r[0][10][63]);
    if (!setjmp(env)) (*c.mem->access_virt_barray) ((j_common_ptr)&c, vb, 18, 4, FALSE);
    if (!setjmp(env)) (*c.mem->access_virt_barray) ((j_common_ptr)&c, vb, 0, 5, FALSE);
    if (!setjmp(env)) (*c.mem->access_virt_sarray) ((j_common_ptr)&c, vs, 0, 2, FALSE);   /* undefined rows, not pre-zeroed */
    s = (*c.mem->access_virt_sarray) ((j_common_ptr)&c, vs, 0, 2, TRUE); s[1][32] = 9;
    s = (*c.mem->access_virt_sarray) ((j_common_ptr)&c, vs, 1, 1, FALSE); printf("sample back %d\n", s[0][32]);
    if (!setjmp(env)) (*c.mem->access_virt_sarray) ((j_common_ptr)&c, vs, 4, 2, TRUE);    /* writing beyond the defined part leaves a hole */
    if (!setjmp(env)) (*c.mem->request_virt_barray) ((j_common_ptr)&c, JPOOL_PERMANENT, TRUE, 1, 1, 1);
    if (!setjmp(env)) (*c.mem->alloc_small) ((j_common_ptr)&c, 5, 10);
    printf("limits %ld\n", c.mem->max_alloc_chunk);
  }
  /* tables-only datastream into a memory destination that has to grow */
  c.data_precision = 8;
  {
    unsigned char *buf = NULL;
    unsigned long n = 0, i, h = 0;
    jpeg_abort_compress(&c);
    jpeg_mem_dest(&c, &buf, &n);
    jpeg_set_quality(&c, 3, FALSE);        /* 16-bit table entries */
    jpeg_suppress_tables(&c, FALSE);
    jpeg_write_tables(&c);
    for (i = 0; i < n; i++) h = h * 131 + buf[i];
    printf("tables-only %lu bytes, hash %lu, head %02x%02x%02x%02x tail %02x%02x\n", n, h, buf[0], buf[1], buf[2], buf[3], buf[n - 2], buf[n - 1]);
    dump(&c, "after write_tables (sent flags)");
    free(buf);
  }
  jpeg_destroy_compress(&c);
  printf("destroyed: mem %s state %d\n", c.mem ? "set" : "null", c.global_state);
  printf("utils %ld %ld %ld\n", jdiv_round_up(17, 8), jround_up(17, 8), jround_up(16, 8));
  return 0;
}
