/* shim_harness.c -- libjpeg client scenarios the drop-in has to survive, written against the public API
 * only.  The same binary runs against the reference's libjpeg.so.62 (expected output), with the preload shim in front of
 * it, and against the stand-alone library; tests/test_gpu_dropin.py compares the printed hashes.
 *   scenario names on the command line; image = deterministic synthetic RGB.                                        */
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jpeglib.h"
#include "jerror.h"

static jmp_buf env;
static void my_exit(j_common_ptr cinfo) { (void)cinfo; longjmp(env, 1); }

static unsigned char *make_image(int w, int h, int seed)
{
  unsigned char *p = (unsigned char *)malloc((size_t)w * h * 3);
  unsigned s = 12345u + (unsigned)seed * 7919u;
  int x, y, c;
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++)
      for (c = 0; c < 3; c++) {
        s = s * 1664525u + 1013904223u;
        p[((size_t)y * w + x) * 3 + c] = (unsigned char)(((x * (3 + c) + y * (5 - c)) & 0xFF) / 2 + ((s >> 24) & 0x3F) + (((x / 16 + y / 16) % 5) == 0 ? 64 : 0));
      }
  return p;
}

static unsigned long hash(const unsigned char *b, unsigned long n) { unsigned long h = 1469598103934665603ul, i; for (i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ul; return h; }

static void setup(struct jpeg_compress_struct *c, int w, int h, int quality, int baseline)
{
  c->image_width = w; c->image_height = h; c->input_components = 3; c->in_color_space = JCS_RGB;
  jpeg_set_defaults(c);
  c->dct_method = JDCT_ISLOW;
  jpeg_set_quality(c, quality, TRUE);
  if (baseline) { c->num_scans = 0; c->scan_info = NULL; }
}

static void rows(struct jpeg_compress_struct *c, unsigned char *img, int upto)
{
  while ((int)c->next_scanline < upto) {
    JSAMPROW r = img + (size_t)c->next_scanline * c->image_width * 3;
    jpeg_write_scanlines(c, &r, 1);
  }
}

static long rss_kb(void)
{
  long pages = 0, dummy = 0;
  FILE *f = fopen("/proc/self/statm", "r");
  if (f) { if (fscanf(f, "%ld %ld", &dummy, &pages) != 2) pages = 0; fclose(f); }
  return pages * 4;
}

int main(int argc, char **argv)
{
  struct jpeg_error_mgr err;
  int a;
  for (a = 1; a < argc; a++) {
    const char *sc = argv[a];
    if (!strcmp(sc, "interleaved")) {
      /* two compress objects of different geometry, started and fed alternately on one thread (thumbnail + main image) */
      struct jpeg_compress_struct c1, c2;
      unsigned char *o1 = NULL, *o2 = NULL, *i1 = make_image(160, 120, 1), *i2 = make_image(333, 211, 2);
      unsigned long n1 = 0, n2 = 0;
      c1.err = c2.err = jpeg_std_error(&err);
      jpeg_create_compress(&c1); jpeg_create_compress(&c2);
      jpeg_mem_dest(&c1, &o1, &n1); jpeg_mem_dest(&c2, &o2, &n2);
      setup(&c1, 160, 120, 80, 1); setup(&c2, 333, 211, 60, 1);
      jpeg_start_compress(&c1, TRUE); jpeg_start_compress(&c2, TRUE);
      rows(&c1, i1, 60); rows(&c2, i2, 100); rows(&c1, i1, 120); rows(&c2, i2, 211);
      jpeg_finish_compress(&c1); jpeg_finish_compress(&c2);
      printf("interleaved %lu %016lx %lu %016lx\n", n1, hash(o1, n1), n2, hash(o2, n2));
      jpeg_destroy_compress(&c1); jpeg_destroy_compress(&c2);
      free(o1); free(o2); free(i1); free(i2);
    } else if (!strcmp(sc, "abort_reuse")) {
      /* the application's error_exit longjmps out of jpeg_finish_compress (too little data), the object is aborted and
       * then used again; afterwards many start/abort and create/destroy cycles must not accumulate anything */
      struct jpeg_compress_struct c;
      unsigned char *o = NULL, *img = make_image(200, 150, 3);
      unsigned long n = 0;
      long r0, r1;
      int k;
      c.err = jpeg_std_error(&err);
      err.error_exit = my_exit;
      jpeg_create_compress(&c);
      jpeg_mem_dest(&c, &o, &n);
      setup(&c, 200, 150, 75, 1);
      if (!setjmp(env)) { jpeg_start_compress(&c, TRUE); rows(&c, img, 70); jpeg_finish_compress(&c); printf("abort_reuse: finish did not fail\n"); }
      else { printf("abort_reuse: error %d caught\n", err.msg_code == JERR_TOO_LITTLE_DATA); jpeg_abort_compress(&c); }
      jpeg_mem_dest(&c, &o, &n);
      jpeg_start_compress(&c, TRUE); rows(&c, img, 150); jpeg_finish_compress(&c);
      printf("abort_reuse %lu %016lx\n", n, hash(o, n));
      for (k = 0; k < 30; k++) { jpeg_start_compress(&c, TRUE); rows(&c, img, 10); jpeg_abort_compress(&c); }
      r0 = rss_kb();
      for (k = 0; k < 300; k++) { jpeg_start_compress(&c, TRUE); rows(&c, img, 10); jpeg_abort_compress(&c); }
      for (k = 0; k < 100; k++) {
        struct jpeg_compress_struct d;
        unsigned char *od = NULL; unsigned long nd = 0;
        d.err = c.err; jpeg_create_compress(&d); jpeg_mem_dest(&d, &od, &nd); setup(&d, 200, 150, 75, 1);
        jpeg_start_compress(&d, TRUE); rows(&d, img, 5); jpeg_destroy_compress(&d); free(od);
      }
      r1 = rss_kb();
      printf("abort_reuse growth_ok %d\n", r1 - r0 < 20000);
      jpeg_start_compress(&c, TRUE); rows(&c, img, 150); jpeg_finish_compress(&c);
      printf("abort_reuse again %lu %016lx\n", n, hash(o, n));
      jpeg_destroy_compress(&c); free(o); free(img);
    } else if (!strcmp(sc, "abort_midway")) {
      /* an image is abandoned after more than 512 of its scanlines were written (the drop-in has sent them to the device
       * by then), then the same object compresses a DIFFERENT image: nothing of the abandoned one may show; the same again
       * with the error_exit longjmp out of jpeg_finish_compress, and twice in a row (both staging buffers of the object) */
      struct jpeg_compress_struct c;
      unsigned char *o = NULL, *a = make_image(640, 700, 21), *b = make_image(640, 700, 22), *d = make_image(640, 700, 23);
      unsigned long n = 0;
      int k;
      c.err = jpeg_std_error(&err);
      err.error_exit = my_exit;
      jpeg_create_compress(&c);
      jpeg_mem_dest(&c, &o, &n);
      setup(&c, 640, 700, 75, 1);
      jpeg_start_compress(&c, TRUE); rows(&c, a, 600); jpeg_abort_compress(&c);
      jpeg_mem_dest(&c, &o, &n);
      jpeg_start_compress(&c, TRUE); rows(&c, b, 700); jpeg_finish_compress(&c);
      printf("abort_midway first %lu %016lx\n", n, hash(o, n));
      for (k = 0; k < 2; k++) {
        if (!setjmp(env)) { jpeg_start_compress(&c, TRUE); rows(&c, k ? b : d, 520 + k * 100); jpeg_finish_compress(&c); printf("abort_midway: finish did not fail\n"); }
        else jpeg_abort_compress(&c);
      }
      jpeg_mem_dest(&c, &o, &n);
      jpeg_start_compress(&c, TRUE); rows(&c, a, 700); jpeg_finish_compress(&c);
      printf("abort_midway second %lu %016lx\n", n, hash(o, n));
      jpeg_mem_dest(&c, &o, &n);
      jpeg_start_compress(&c, TRUE); rows(&c, d, 700); jpeg_finish_compress(&c);
      printf("abort_midway third %lu %016lx\n", n, hash(o, n));
      jpeg_destroy_compress(&c); free(o); free(a); free(b); free(d);
    } else if (!strcmp(sc, "markers")) {
      /* COM + APPn + ICC markers written by the application between start and the first scanline */
      struct jpeg_compress_struct c;
      unsigned char *o = NULL, *img = make_image(97, 61, 4), icc[70000];
      unsigned long n = 0;
      unsigned k;
      for (k = 0; k < sizeof(icc); k++) icc[k] = (unsigned char)(k * 31 + (k >> 8));
      c.err = jpeg_std_error(&err);
      jpeg_create_compress(&c);
      jpeg_mem_dest(&c, &o, &n);
      setup(&c, 97, 61, 85, 0);           /* default mode: progressive with scan search */
      c.density_unit = 1; c.X_density = 300; c.Y_density = 150; c.JFIF_minor_version = 2;
      jpeg_start_compress(&c, TRUE);
      jpeg_write_marker(&c, JPEG_COM, (const JOCTET *)"made by the harness", 19);
      jpeg_write_m_header(&c, JPEG_APP0 + 5, 3); jpeg_write_m_byte(&c, 1); jpeg_write_m_byte(&c, 2); jpeg_write_m_byte(&c, 3);
      jpeg_write_icc_profile(&c, icc, sizeof(icc));
      rows(&c, img, 61);
      jpeg_finish_compress(&c);
      printf("markers %lu %016lx\n", n, hash(o, n));
      jpeg_destroy_compress(&c); free(o); free(img);
    } else if (!strcmp(sc, "stdio")) {
      struct jpeg_compress_struct c;
      unsigned char *img = make_image(640, 400, 5), *buf;
      FILE *f = tmpfile();
      long n;
      c.err = jpeg_std_error(&err);
      jpeg_create_compress(&c);
      jpeg_stdio_dest(&c, f);
      setup(&c, 640, 400, 90, 1);
      jpeg_start_compress(&c, TRUE); rows(&c, img, 400); jpeg_finish_compress(&c);
      jpeg_destroy_compress(&c);
      n = ftell(f); rewind(f);
      buf = (unsigned char *)malloc((size_t)n);
      if (fread(buf, 1, (size_t)n, f) != (size_t)n) n = 0;
      printf("stdio %ld %016lx\n", n, hash(buf, (unsigned long)n));
      free(buf); fclose(f); free(img);
    } else if (!strcmp(sc, "ext_params")) {
      /* the extension parameters no cjpeg switch reaches (jpeglib.h:321-356, jcext.c): trellis_eob_opt, use_scans_in_trellis
       * + trellis_freq_split, trellis_q_opt, trellis_delta_dc_weight, dc_scan_opt_mode changed AFTER the script was built
       * (scan 0 keeps all components and the search appends the separate chroma DC scans, jcmaster.c:904-913) */
      int v;
      for (v = 0; v < 4; v++) {
        struct jpeg_compress_struct c;
        unsigned char *o = NULL, *img = make_image(208, 136, 6 + v);
        unsigned long n = 0;
        c.err = jpeg_std_error(&err);
        jpeg_create_compress(&c);
        jpeg_mem_dest(&c, &o, &n);
        setup(&c, 208, 136, 70 + 5 * v, v == 1);
        if (v == 0) { jpeg_c_set_bool_param(&c, JBOOLEAN_TRELLIS_EOB_OPT, TRUE); jpeg_c_set_bool_param(&c, JBOOLEAN_TRELLIS_Q_OPT, TRUE); }
        if (v == 1) { jpeg_c_set_bool_param(&c, JBOOLEAN_USE_SCANS_IN_TRELLIS, TRUE); jpeg_c_set_int_param(&c, JINT_TRELLIS_FREQ_SPLIT, 12);
                      jpeg_c_set_float_param(&c, JFLOAT_TRELLIS_DELTA_DC_WEIGHT, 0.75f); }
        if (v == 2) { jpeg_c_set_int_param(&c, JINT_DC_SCAN_OPT_MODE, 1); }
        if (v == 3) { jpeg_c_set_int_param(&c, JINT_DC_SCAN_OPT_MODE, 2); jpeg_simple_progression(&c); jpeg_c_set_bool_param(&c, JBOOLEAN_TRELLIS_EOB_OPT, TRUE);
                      jpeg_c_set_bool_param(&c, JBOOLEAN_USE_SCANS_IN_TRELLIS, TRUE); }
        jpeg_start_compress(&c, TRUE); rows(&c, img, 136); jpeg_finish_compress(&c);
        printf("ext_params %d %lu %016lx\n", v, n, hash(o, n));
        jpeg_destroy_compress(&c);
        free(o); free(img);
      }
    } else if (!strcmp(sc, "custom_huffman")) {
      /* Huffman tables of the application's own with optimize_coding off (two symbols of equal code length swapped in AC table 0):
       * the reference codes with them (start_pass_huff, jchuff.c:190-196), and so does the device path (mjh_params.huff_tables_given) */
      struct jpeg_compress_struct c;
      unsigned char *o = NULL, *img = make_image(96, 64, 31), t;
      unsigned long n = 0;
      c.err = jpeg_std_error(&err);
      jpeg_create_compress(&c);
      jpeg_mem_dest(&c, &o, &n);
      jpeg_c_set_int_param(&c, JINT_COMPRESS_PROFILE, JCP_FASTEST);
      setup(&c, 96, 64, 75, 1);
      c.optimize_coding = FALSE;
      t = c.ac_huff_tbl_ptrs[0]->huffval[0]; c.ac_huff_tbl_ptrs[0]->huffval[0] = c.ac_huff_tbl_ptrs[0]->huffval[1]; c.ac_huff_tbl_ptrs[0]->huffval[1] = t;
      jpeg_start_compress(&c, TRUE); rows(&c, img, 64); jpeg_finish_compress(&c);
      printf("custom_huffman %lu %016lx\n", n, hash(o, n));
      jpeg_destroy_compress(&c);
      free(o); free(img);
    } else if (!strcmp(sc, "abbreviated")) {
      /* libjpeg.txt "Abbreviated datastreams and multiple images": the tables once, then frames without them -- how an MJPEG writer
       * or libtiff's JPEG codec drives the library.  (a) jpeg_write_tables + three frames with write_all_tables FALSE, parameters set
       * once (fastest profile: Annex K Huffman tables); (b) the same with optimal Huffman tables: each frame carries its own DHT, no
       * DQT; (c) mozjpeg's defaults (progressive, scan search, trellis): the first frame whole, then two with FALSE -- what the object
       * keeps from frame to frame (table flags, optimal DC tables, Ah / Al) decides the bytes; (d) jpeg_suppress_tables(TRUE) with
       * the luminance quantization table alone marked unsent again */
      int v, k;
      for (v = 0; v < 4; v++) {
        struct jpeg_compress_struct c;
        unsigned char *o = NULL;
        unsigned long n = 0;
        c.err = jpeg_std_error(&err);
        jpeg_create_compress(&c);
        jpeg_mem_dest(&c, &o, &n);
        if (v != 2) jpeg_c_set_int_param(&c, JINT_COMPRESS_PROFILE, JCP_FASTEST);
        setup(&c, 176, 112, v == 2 ? 85 : 70, v != 2);
        if (v == 1) c.optimize_coding = TRUE;
        if (v == 0 || v == 1) {
          jpeg_write_tables(&c);
          printf("abbreviated %d tables %lu %016lx\n", v, n, hash(o, n));
        }
        if (v == 3) { jpeg_suppress_tables(&c, TRUE); c.quant_tbl_ptrs[0]->sent_table = FALSE; }
        for (k = 0; k < 3; k++) {
          unsigned char *img = make_image(176, 112, 40 + 4 * v + k);
          free(o); o = NULL; n = 0;
          c.dest = NULL;
          jpeg_mem_dest(&c, &o, &n);
          jpeg_start_compress(&c, v == 2 && k == 0 ? TRUE : FALSE); rows(&c, img, 112); jpeg_finish_compress(&c);
          printf("abbreviated %d frame %d %lu %016lx\n", v, k, n, hash(o, n));
          free(img);
        }
        jpeg_destroy_compress(&c);
        free(o);
      }
    } else if (!strcmp(sc, "color_spaces")) {
      /* what an application sets in in_color_space / per component besides plain RGB: samples that are YCbCr already (null_convert,
       * jccolor.c:687-692 -- e.g. Pillow's "YCbCr" mode), an extended pixel order with a pad byte, and ONE component whose sampling
       * factors the application chose (cjpeg itself: 2x1 at qualities 80..89), YCbCr samples into a grayscale file, an Adobe marker on a YCbCr file */
      int v;
      for (v = 0; v < 7; v++) {
        struct jpeg_compress_struct c;
        const int w = 173, h = 121, ps = v == 2 ? 4 : (v == 3 || v == 4 ? 1 : 3);
        unsigned char *o = NULL, *rgb = make_image(w, h, 20 + v), *img = (unsigned char *)malloc((size_t)w * h * 4);
        unsigned long n = 0;
        int x, y;
        for (y = 0; y < h; y++)
          for (x = 0; x < w; x++) {
            const unsigned char *p = rgb + ((size_t)y * w + x) * 3;
            unsigned char *q = img + ((size_t)y * w + x) * ps;
            if (ps == 4) { q[0] = 0xA5; q[1] = p[2]; q[2] = p[1]; q[3] = p[0]; }   /* JCS_EXT_XBGR */
            else if (ps == 1) q[0] = p[1];
            else { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; }
          }
        c.err = jpeg_std_error(&err);
        jpeg_create_compress(&c);
        jpeg_mem_dest(&c, &o, &n);
        c.image_width = w; c.image_height = h; c.input_components = ps;
        c.in_color_space = v < 2 || v == 5 ? JCS_YCbCr : v == 2 ? JCS_EXT_XBGR : v == 6 ? JCS_RGB : JCS_GRAYSCALE;
        jpeg_set_defaults(&c);
        c.dct_method = JDCT_ISLOW;
        jpeg_set_quality(&c, 70 + 4 * v, TRUE);
        if (v == 0 || v == 3) { c.num_scans = 0; c.scan_info = NULL; }
        if (v == 1) { c.comp_info[0].h_samp_factor = 2; c.comp_info[0].v_samp_factor = 1; c.restart_in_rows = 2; }
        if (v == 3) { c.comp_info[0].h_samp_factor = 2; c.comp_info[0].v_samp_factor = 1; }
        if (v == 5) { jpeg_set_colorspace(&c, JCS_GRAYSCALE); jpeg_simple_progression(&c); }      /* (the script is rebuilt for the new component count, as cjpeg does) YCbCr samples into a grayscale file: the Y samples (grayscale_convert jccolor.c:448-466) */
        if (v == 6) c.write_Adobe_marker = TRUE;                 /* an Adobe APP14 marker next to the JFIF one on a YCbCr file (transform 1, jcmarker.c:597-598) */
        if (v == 4) { c.comp_info[0].h_samp_factor = 1; c.comp_info[0].v_samp_factor = 2; jpeg_c_set_bool_param(&c, JBOOLEAN_TRELLIS_QUANT, FALSE); }
        jpeg_start_compress(&c, TRUE);
        while (c.next_scanline < c.image_height) { JSAMPROW r = img + (size_t)c.next_scanline * w * ps; jpeg_write_scanlines(&c, &r, 1); }
        jpeg_finish_compress(&c);
        printf("color_spaces %d %lu %016lx\n", v, n, hash(o, n));
        jpeg_destroy_compress(&c);
        free(o); free(img); free(rgb);
      }
    }
  }
  return 0;
}
