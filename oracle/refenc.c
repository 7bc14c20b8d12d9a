/*
 * refenc.c -- TEST INFRASTRUCTURE: in-memory driver around the REAL reference
 * library (mozjpeg compiled from /root/reference into oracle/_ref/libjpeg.so.62).
 *
 * Why it exists: (1) tjbench / TurboJPEG force JCP_FASTEST (turbojpeg.c:336) and so can
 * never reach the trellis path; cjpeg reaches it but only file->file.  This driver makes
 * the same libjpeg API calls cjpeg makes (cjpeg.c:813-1023) with jpeg_mem_dest, so the
 * reference CPU path can be timed in memory ("cpu_baseline.kind = reference") and its
 * bytes used as goldens.  (2) -dumpcoef re-reads the produced file with
 * jpeg_read_coefficients (jpeglib.h:1177) to give the post-trellis quantized
 * coefficients as a stage-level oracle.
 *
 * usage: refenc [switches] in.(ppm|rgb) out.jpg
 *   -raw W H        input is headerless interleaved RGB (or gray with -grayin), 8-bit
 *   -grayin         raw input has 1 component
 *   -quality N  -baseline  -revert  -optimize  -progressive  -fastcrush
 *   -notrellis  -notrellis-dc  -noovershoot  -sample HxV  -restart N[B]  -gray
 *   -yccin          the input samples are Y, Cb, Cr already (in_color_space = JCS_YCbCr)
 *   -graysample HxV sampling factors of a gray image's one component (-sample is applied to three-component images only)
 *   -quant-table N  -lambda1 F -lambda2 F
 *   -reps N         encode N times, report best and mean wall time
 *   -dumpcoef FILE  dump quantized coefficients of the output
 * Not part of the product; nothing under mozjpeg_amd/ links to or executes it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "jpeglib.h"

static double now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static unsigned char *read_ppm(const char *fn, int *w, int *h, int *nc)
{
  FILE *f = fopen(fn, "rb");
  char magic[3] = { 0 };
  int maxv, c;
  unsigned char *buf;
  if (!f) { perror(fn); exit(2); }
  if (fscanf(f, "%2s", magic) != 1) exit(2);
  *nc = (magic[1] == '6') ? 3 : 1;
  /* skip comments */
  for (;;) {
    c = fgetc(f);
    if (c == '#') { while ((c = fgetc(f)) != '\n' && c != EOF) {} }
    else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
    else { ungetc(c, f); break; }
  }
  if (fscanf(f, "%d %d %d", w, h, &maxv) != 3 || maxv != 255) { fprintf(stderr, "bad ppm\n"); exit(2); }
  fgetc(f);
  buf = malloc((size_t)(*w) * (*h) * (*nc));
  if (fread(buf, 1, (size_t)(*w) * (*h) * (*nc), f) != (size_t)(*w) * (*h) * (*nc)) { fprintf(stderr, "short ppm\n"); exit(2); }
  fclose(f);
  return buf;
}

int main(int argc, char **argv)
{
  int quality = 75, baseline = 0, revert = 0, optimize = 0, progressive = 0, fastcrush = 0;
  int notrellis = 0, notrellis_dc = 0, noovershoot = 0, gray = 0, rgbout = 0, grayin = 0, qtbl = -1;
  const char *dctbl = NULL, *actbl = NULL;
  int no_optimize = 0;
  int ghs = 0, gvs = 0, yccin = 0;
  int hs = 2, vs = 2, hs1 = 1, vs1 = 1, hs2 = 1, vs2 = 1, nsamp = 2, restart = 0, restart_blocks = 0, reps = 1, rawW = 0, rawH = 0;
  double l1 = -1e9, l2 = -1e9;
  int dc_scan_opt = -1;
  double dc_ver_weight = -1e9;
  int precision = 8, yuvin = 0, dct_fast = 0, arithmetic = 0, trellis_loops = 0, smooth = 0, trellis_q_opt = 0, eob_opt = 0, scans_in_trellis = 0, freq_split = 0;
  const char *arith_cond = NULL, *scanspec = NULL;
  static jpeg_scan_info user_scans[64];
  const char *dump = NULL, *in = NULL, *out = NULL;
  int i, w, h, nc;
  unsigned char *img;
  unsigned char *jbuf = NULL;
  unsigned long jsize = 0;
  double best = 1e30, total = 0;

  for (i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (!strcmp(a, "-quality")) quality = atoi(argv[++i]);
    else if (!strcmp(a, "-baseline")) baseline = 1;
    else if (!strcmp(a, "-revert")) revert = 1;
    else if (!strcmp(a, "-optimize")) optimize = 1;
    else if (!strcmp(a, "-progressive")) progressive = 1;
    else if (!strcmp(a, "-fastcrush")) fastcrush = 1;
    else if (!strcmp(a, "-notrellis")) notrellis = 1;
    else if (!strcmp(a, "-notrellis-dc")) notrellis_dc = 1;
    else if (!strcmp(a, "-noovershoot")) noovershoot = 1;
    else if (!strcmp(a, "-gray")) gray = 1;
    else if (!strcmp(a, "-rgb")) rgbout = 1;   /* cjpeg -rgb: jpeg_set_colorspace(JCS_RGB), samples unconverted */
    else if (!strcmp(a, "-grayin")) grayin = 1;
    else if (!strcmp(a, "-yccin")) yccin = 1;   /* the three input samples are Y, Cb, Cr: in_color_space = JCS_YCbCr (set before jpeg_set_defaults) */
    else if (!strcmp(a, "-quant-table")) qtbl = atoi(argv[++i]);
    else if (!strcmp(a, "-lambda1")) l1 = atof(argv[++i]);
    else if (!strcmp(a, "-lambda2")) l2 = atof(argv[++i]);
    else if (!strcmp(a, "-sample")) {   /* cjpeg -sample HxV[,HxV,HxV] (set_sample_factors rdswitch.c): the luma factors, or all three components' */
      nsamp = sscanf(argv[++i], "%dx%d,%dx%d,%dx%d", &hs, &vs, &hs1, &vs1, &hs2, &vs2);
    }
    else if (!strcmp(a, "-restart")) {
      char ch = 'x'; long v = 0;
      sscanf(argv[++i], "%ld%c", &v, &ch);
      restart = (int)v; restart_blocks = (ch == 'b' || ch == 'B');
    }
    else if (!strcmp(a, "-reps")) reps = atoi(argv[++i]);
    else if (!strcmp(a, "-graysample")) sscanf(argv[++i], "%dx%d", &ghs, &gvs);   /* factors of a gray image's component (-sample is only applied to three) */
    else if (!strcmp(a, "-precision")) precision = atoi(argv[++i]);   /* 12: raw input is uint16 samples */
    else if (!strcmp(a, "-raw")) { rawW = atoi(argv[++i]); rawH = atoi(argv[++i]); }
    else if (!strcmp(a, "-dumpcoef")) dump = argv[++i];
    else if (!strcmp(a, "-trellis-loops")) trellis_loops = atoi(argv[++i]);   /* JINT_TRELLIS_NUM_LOOPS: API-only parameter */
    else if (!strcmp(a, "-trellis-eob-opt")) eob_opt = 1;                 /* JBOOLEAN_TRELLIS_EOB_OPT */
    else if (!strcmp(a, "-use-scans-in-trellis")) scans_in_trellis = 1;   /* JBOOLEAN_USE_SCANS_IN_TRELLIS */
    else if (!strcmp(a, "-trellis-freq-split")) freq_split = atoi(argv[++i]);   /* JINT_TRELLIS_FREQ_SPLIT */
    else if (!strcmp(a, "-trellis-q-opt")) trellis_q_opt = 1;   /* JBOOLEAN_TRELLIS_Q_OPT: API-only parameter */
    else if (!strcmp(a, "-dc-scan-opt")) dc_scan_opt = atoi(argv[++i]);   /* cjpeg -dc-scan-opt N (cjpeg.c:494-499) */
    else if (!strcmp(a, "-trellis-dc-ver-weight")) dc_ver_weight = atof(argv[++i]);   /* cjpeg.c:667-672 */
    else if (!strcmp(a, "-smooth")) smooth = atoi(argv[++i]);   /* cjpeg -smooth N (cjpeg.c: cinfo->smoothing_factor) */
    else if (!strcmp(a, "-arithmetic")) arithmetic = 1;   /* cjpeg -arithmetic (cjpeg.c:371-376): cinfo->arith_code */
    else if (!strcmp(a, "-scanspec")) scanspec = argv[++i];   /* a scan script (cjpeg -scans file, read_scan_script rdswitch.c) on the command line: "c[,c..]:Ss-Se:Ah:Al;..." */
    else if (!strcmp(a, "-dct")) dct_fast = !strcmp(argv[++i], "fast");
    else if (!strcmp(a, "-arith-cond")) arith_cond = argv[++i];   /* L0,U0,K0,L1,U1,K1: cinfo->arith_dc_L / arith_dc_U / arith_ac_K of tables 0 and 1 (API-only fields, jpeglib.h:447-449) */
    else if (!strcmp(a, "-no-optimize")) no_optimize = 1;   /* cinfo->optimize_coding = FALSE by hand, whatever the profile set (API-only) */
    else if (!strcmp(a, "-dctbl")) dctbl = argv[++i];   /* a,b,c: cinfo->comp_info[i].dc_tbl_no (API-only; jpeg_set_colorspace assigns 0,1,1) */
    else if (!strcmp(a, "-actbl")) actbl = argv[++i];
    else if (!strcmp(a, "-yuvin")) yuvin = 1;   /* -raw W H input holds component planes: jpeg_write_raw_data */
    else if (!in) in = a;
    else out = a;
  }
  if (!in || !out) { fprintf(stderr, "usage: refenc [switches] in out.jpg\n"); return 2; }

  if (rawW) {
    FILE *f = fopen(in, "rb");
    size_t n;
    if (!f) { perror(in); return 2; }
    w = rawW; h = rawH; nc = grayin ? 1 : 3;
    n = (size_t)w * h * nc * (precision == 12 ? 2 : 1);
    if (yuvin) {   /* planes are at most (w+3)*(h+3) samples each */
      fseek(f, 0, SEEK_END); n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
      nc = 3;
    }
    img = malloc(n);
    if (fread(img, 1, n, f) != n) { fprintf(stderr, "short raw\n"); return 2; }
    fclose(f);
  } else
    img = read_ppm(in, &w, &h, &nc);

  for (i = 0; i < reps; i++) {
    struct jpeg_compress_struct cinfo;
    struct jpeg_error_mgr jerr;
    JSAMPROW *rows;
    int y;
    double t0, t1;

    if (jbuf) { free(jbuf); jbuf = NULL; jsize = 0; }
    t0 = now();
    cinfo.err = jpeg_std_error(&jerr);
    jpeg_create_compress(&cinfo);
    cinfo.in_color_space = (nc == 3) ? (yccin ? JCS_YCbCr : JCS_RGB) : JCS_GRAYSCALE;
    cinfo.input_components = nc;
    if (revert)
      jpeg_c_set_int_param(&cinfo, JINT_COMPRESS_PROFILE, JCP_FASTEST);
    jpeg_set_defaults(&cinfo);
    cinfo.image_width = w;
    cinfo.image_height = h;
    cinfo.dct_method = dct_fast ? JDCT_IFAST : JDCT_ISLOW;    /* cjpeg -dct fast / -dct int (cjpeg.c:389-404) */
    cinfo.data_precision = precision;   /* cjpeg.c:533 sets it after jpeg_set_defaults as well */
    if (qtbl >= 0) jpeg_c_set_int_param(&cinfo, JINT_BASE_QUANT_TBL_IDX, qtbl);
    if (l1 > -1e8) jpeg_c_set_float_param(&cinfo, JFLOAT_LAMBDA_LOG_SCALE1, (float)l1);
    if (l2 > -1e8) jpeg_c_set_float_param(&cinfo, JFLOAT_LAMBDA_LOG_SCALE2, (float)l2);
    if (gray) jpeg_set_colorspace(&cinfo, JCS_GRAYSCALE);
    if (rgbout) jpeg_set_colorspace(&cinfo, JCS_RGB);
    jpeg_set_quality(&cinfo, quality, baseline ? TRUE : FALSE);
    if (baseline) { cinfo.num_scans = 0; cinfo.scan_info = NULL; }
    if (optimize) cinfo.optimize_coding = TRUE;
    if (arithmetic) cinfo.arith_code = TRUE;
    if (arith_cond) {
      int v[6] = { 0, 1, 5, 0, 1, 5 }, t;
      sscanf(arith_cond, "%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]);
      for (t = 0; t < 2; t++) { cinfo.arith_dc_L[t] = (UINT8)v[3 * t]; cinfo.arith_dc_U[t] = (UINT8)v[3 * t + 1]; cinfo.arith_ac_K[t] = (UINT8)v[3 * t + 2]; }
    }
    if (fastcrush) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_OPTIMIZE_SCANS, FALSE);
    if (dc_scan_opt >= 0) jpeg_c_set_int_param(&cinfo, JINT_DC_SCAN_OPT_MODE, dc_scan_opt);   /* a switch: before the script is rebuilt */
    if (dc_ver_weight > -1e8) jpeg_c_set_float_param(&cinfo, JFLOAT_TRELLIS_DELTA_DC_WEIGHT, (float)dc_ver_weight);
    /* cjpeg.c re-runs jpeg_simple_progression after all colourspace switches (simple_progressive
     * is TRUE by default in the max-compression profile, cjpeg.c:345-347,:767-768) */
    if (progressive || fastcrush || (!revert && !baseline)) jpeg_simple_progression(&cinfo);
    cinfo.smoothing_factor = smooth;
    if (trellis_q_opt) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_TRELLIS_Q_OPT, TRUE);
    if (eob_opt) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_TRELLIS_EOB_OPT, TRUE);
    if (scans_in_trellis) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_USE_SCANS_IN_TRELLIS, TRUE);
    if (freq_split) jpeg_c_set_int_param(&cinfo, JINT_TRELLIS_FREQ_SPLIT, freq_split);
    if (trellis_loops) jpeg_c_set_int_param(&cinfo, JINT_TRELLIS_NUM_LOOPS, trellis_loops);
    if (notrellis) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_TRELLIS_QUANT, FALSE);
    if (notrellis_dc) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_TRELLIS_QUANT_DC, FALSE);
    if (noovershoot) jpeg_c_set_bool_param(&cinfo, JBOOLEAN_OVERSHOOT_DERINGING, FALSE);
    if (cinfo.num_components == 1 && ghs > 0) {   /* -graysample HxV: the one component's factors (cjpeg: -sample on a gray image, or quality 80..89) */
      cinfo.comp_info[0].h_samp_factor = ghs;
      cinfo.comp_info[0].v_samp_factor = gvs;
    }
    if (cinfo.num_components == 3 && !rgbout) {
      cinfo.comp_info[0].h_samp_factor = hs;
      cinfo.comp_info[0].v_samp_factor = vs;
      cinfo.comp_info[1].h_samp_factor = cinfo.comp_info[1].v_samp_factor = 1;
      cinfo.comp_info[2].h_samp_factor = cinfo.comp_info[2].v_samp_factor = 1;
      if (nsamp == 6) {
        cinfo.comp_info[1].h_samp_factor = hs1; cinfo.comp_info[1].v_samp_factor = vs1;
        cinfo.comp_info[2].h_samp_factor = hs2; cinfo.comp_info[2].v_samp_factor = vs2;
      }
    }
    if (no_optimize) cinfo.optimize_coding = FALSE;
    if (dctbl || actbl) {   /* table numbers of the application's own */
      int t[4] = { 0, 0, 0, 0 }, ci;
      if (dctbl) { sscanf(dctbl, "%d,%d,%d,%d", &t[0], &t[1], &t[2], &t[3]); for (ci = 0; ci < cinfo.num_components; ci++) cinfo.comp_info[ci].dc_tbl_no = t[ci]; }
      if (actbl) { sscanf(actbl, "%d,%d,%d,%d", &t[0], &t[1], &t[2], &t[3]); for (ci = 0; ci < cinfo.num_components; ci++) cinfo.comp_info[ci].ac_tbl_no = t[ci]; }
      /* slots 2 / 3 are empty after jpeg_set_defaults and the library insists on a table in every slot a component names
       * (JERR_NO_HUFF_TABLE, also where it only needs rates): what an application would do -- a copy of the standard table of the
       * same parity */
      for (ci = 0; ci < cinfo.num_components; ci++) {
        const int d = cinfo.comp_info[ci].dc_tbl_no, a = cinfo.comp_info[ci].ac_tbl_no;
        if (d > 1 && cinfo.dc_huff_tbl_ptrs[d] == NULL) {
          cinfo.dc_huff_tbl_ptrs[d] = jpeg_alloc_huff_table((j_common_ptr)&cinfo);
          *cinfo.dc_huff_tbl_ptrs[d] = *cinfo.dc_huff_tbl_ptrs[d & 1];
          cinfo.dc_huff_tbl_ptrs[d]->sent_table = FALSE;
        }
        if (a > 1 && cinfo.ac_huff_tbl_ptrs[a] == NULL) {
          cinfo.ac_huff_tbl_ptrs[a] = jpeg_alloc_huff_table((j_common_ptr)&cinfo);
          *cinfo.ac_huff_tbl_ptrs[a] = *cinfo.ac_huff_tbl_ptrs[a & 1];
          cinfo.ac_huff_tbl_ptrs[a]->sent_table = FALSE;
        }
      }
    }
    if (restart) {
      if (restart_blocks) { cinfo.restart_interval = restart; cinfo.restart_in_rows = 0; }
      else cinfo.restart_in_rows = restart;
    }
    if (scanspec) {      /* cjpeg -scans: the script replaces whatever the profile chose, the scan search is off (cjpeg.c:738-743) */
      const char *s = scanspec;
      int ns = 0;
      while (*s && ns < 64) {
        jpeg_scan_info *sc = &user_scans[ns];
        int nc = 0, Ss = 0, Se = 63, Ah = 0, Al = 0, used = 0;
        while (*s >= '0' && *s <= '9') { sc->component_index[nc++] = (int)strtol(s, (char **)&s, 10); if (*s == ',') s++; }
        if (sscanf(s, ":%d-%d:%d:%d%n", &Ss, &Se, &Ah, &Al, &used) == 4) s += used;
        sc->comps_in_scan = nc; sc->Ss = Ss; sc->Se = Se; sc->Ah = Ah; sc->Al = Al;
        ns++;
        if (*s == ';') s++;
      }
      jpeg_c_set_bool_param(&cinfo, JBOOLEAN_OPTIMIZE_SCANS, FALSE);
      cinfo.scan_info = user_scans;
      cinfo.num_scans = ns;
    }
    jpeg_mem_dest(&cinfo, &jbuf, &jsize);
    if (yuvin) {
      /* Planar input the way TurboJPEG's YUV entry points feed it (turbojpeg.c:1222-1335): plane ci is
       * PAD(w,maxh)*h_i/maxh x PAD(h,maxv)*v_i/maxv samples, tightly packed, planes back to back; the
       * last sample / row is replicated out to whole iMCU rows; one jpeg_write_raw_data call per iMCU row. */
      int ci, maxh = 1, maxv = 1, r, c, imcu;
      unsigned char *pl[4], *tmp[4];
      JSAMPROW *rowp[4];
      JSAMPARRAY data[4];
      int pw[4], ph[4], iw[4], th[4];
      size_t off = 0;
      cinfo.raw_data_in = TRUE;
      jpeg_start_compress(&cinfo, TRUE);
      for (ci = 0; ci < cinfo.num_components; ci++) {
        if (cinfo.comp_info[ci].h_samp_factor > maxh) maxh = cinfo.comp_info[ci].h_samp_factor;
        if (cinfo.comp_info[ci].v_samp_factor > maxv) maxv = cinfo.comp_info[ci].v_samp_factor;
      }
      for (ci = 0; ci < cinfo.num_components; ci++) {
        jpeg_component_info *cp = &cinfo.comp_info[ci];
        pw[ci] = (w + maxh - 1) / maxh * maxh * cp->h_samp_factor / maxh;
        ph[ci] = (h + maxv - 1) / maxv * maxv * cp->v_samp_factor / maxv;
        iw[ci] = cp->width_in_blocks * 8;
        th[ci] = cp->v_samp_factor * 8;
        pl[ci] = img + off;
        off += (size_t)pw[ci] * ph[ci];
        tmp[ci] = malloc((size_t)iw[ci] * th[ci]);
        rowp[ci] = malloc(sizeof(JSAMPROW) * th[ci]);
        for (r = 0; r < th[ci]; r++) rowp[ci][r] = tmp[ci] + (size_t)r * iw[ci];
        data[ci] = rowp[ci];
      }
      for (imcu = 0; imcu * maxv * 8 < h; imcu++) {
        for (ci = 0; ci < cinfo.num_components; ci++)
          for (r = 0; r < th[ci]; r++) {
            int sr = imcu * th[ci] + r;
            if (sr > ph[ci] - 1) sr = ph[ci] - 1;
            for (c = 0; c < iw[ci]; c++) tmp[ci][(size_t)r * iw[ci] + c] = pl[ci][(size_t)sr * pw[ci] + (c < pw[ci] ? c : pw[ci] - 1)];
          }
        jpeg_write_raw_data(&cinfo, data, maxv * 8);
      }
      jpeg_finish_compress(&cinfo);
      jpeg_destroy_compress(&cinfo);
      for (ci = 0; ci < 4 && ci < 3; ci++) if (ci < (gray ? 1 : 3)) { free(tmp[ci]); free(rowp[ci]); }
      t1 = now();
      if (t1 - t0 < best) best = t1 - t0;
      total += t1 - t0;
      continue;
    }
    jpeg_start_compress(&cinfo, TRUE);
    rows = malloc(sizeof(JSAMPROW) * h);
    for (y = 0; y < h; y++) rows[y] = img + (size_t)y * w * nc * (precision == 12 ? 2 : 1);
    while (cinfo.next_scanline < cinfo.image_height) {
      if (precision == 12)
        jpeg12_write_scanlines(&cinfo, (J12SAMPARRAY)(rows + cinfo.next_scanline), cinfo.image_height - cinfo.next_scanline);
      else
        jpeg_write_scanlines(&cinfo, rows + cinfo.next_scanline, cinfo.image_height - cinfo.next_scanline);
    }
    jpeg_finish_compress(&cinfo);
    jpeg_destroy_compress(&cinfo);
    free(rows);
    t1 = now();
    if (t1 - t0 < best) best = t1 - t0;
    total += t1 - t0;
  }
  {
    FILE *f = fopen(out, "wb");
    if (!f) { perror(out); return 2; }
    fwrite(jbuf, 1, jsize, f);
    fclose(f);
  }
  printf("{\"width\": %d, \"height\": %d, \"bytes\": %lu, \"reps\": %d, \"best_s\": %.6f, \"mean_s\": %.6f, \"mpix_per_s_best\": %.3f}\n",
         w, h, jsize, reps, best, total / reps, (double)w * h / best / 1e6);

  if (dump) {
    struct jpeg_decompress_struct d;
    struct jpeg_error_mgr jerr;
    jvirt_barray_ptr *arrs;
    FILE *f = fopen(dump, "wb");
    int ci;
    d.err = jpeg_std_error(&jerr);
    jpeg_create_decompress(&d);
    jpeg_mem_src(&d, jbuf, jsize);
    jpeg_read_header(&d, TRUE);
    arrs = jpeg_read_coefficients(&d);
    for (ci = 0; ci < d.num_components; ci++) {
      jpeg_component_info *c = &d.comp_info[ci];
      int hdr[2];
      JDIMENSION r;
      hdr[0] = c->height_in_blocks; hdr[1] = c->width_in_blocks;
      fwrite(hdr, sizeof(int), 2, f);
      for (r = 0; r < c->height_in_blocks; r++) {
        JBLOCKARRAY ba = (*d.mem->access_virt_barray)((j_common_ptr)&d, arrs[ci], r, 1, FALSE);
        fwrite(ba[0], sizeof(JBLOCK), c->width_in_blocks, f);
      }
    }
    fclose(f);
    jpeg_finish_decompress(&d);
    jpeg_destroy_decompress(&d);
  }
  free(jbuf);
  free(img);
  return 0;
}
