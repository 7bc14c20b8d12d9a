/* mt_bench.c -- drop-in throughput of a libjpeg client (MEASUREMENT TOOL, public API only): T threads, each
 * compressing N images of WxH RGB through jpeg_CreateCompress / jpeg_set_defaults / jpeg_set_quality /
 * jpeg_start_compress / jpeg_write_scanlines / jpeg_finish_compress into a jpeg_mem_dest buffer -- exactly what an
 * application does.  The same binary is timed against the reference's libjpeg.so.62 (CPU), with the preload shim in
 * front of it, and against the stand-alone library (tools/bench_dropin.py picks the library through
 * LD_LIBRARY_PATH / LD_PRELOAD).      usage: mt_bench THREADS IMAGES_PER_THREAD WIDTH HEIGHT QUALITY [baseline]       */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "jpeglib.h"

static int W, H, Q, NIMG, BASE;
static unsigned char *image;
static unsigned long long total_bytes[256];
static unsigned long long hashes[256];

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void *worker(void *arg)
{
  int id = (int)(long)arg, i, y;
  struct jpeg_compress_struct c;
  struct jpeg_error_mgr err;
  JSAMPROW *rowp = (JSAMPROW *)malloc(sizeof(JSAMPROW) * H);
  for (y = 0; y < H; y++) rowp[y] = image + (size_t)y * W * 3;
  c.err = jpeg_std_error(&err);
  jpeg_create_compress(&c);
  for (i = 0; i < NIMG; i++) {
    unsigned char *out = NULL; unsigned long n = 0, k; unsigned long long h = 1469598103934665603ull;
    jpeg_mem_dest(&c, &out, &n);
    c.image_width = W; c.image_height = H; c.input_components = 3; c.in_color_space = JCS_RGB;
    jpeg_set_defaults(&c);
    jpeg_set_quality(&c, Q, TRUE);
    if (BASE) { c.num_scans = 0; c.scan_info = NULL; }
    jpeg_start_compress(&c, TRUE);
    while (c.next_scanline < c.image_height)
      jpeg_write_scanlines(&c, rowp + c.next_scanline, c.image_height - c.next_scanline);
    jpeg_finish_compress(&c);
    total_bytes[id] += n;
    if (i == 0) { for (k = 0; k < n; k++) h = (h ^ out[k]) * 1099511628211ull; hashes[id] = h; }
    free(out);
  }
  jpeg_destroy_compress(&c);
  free(rowp);
  return NULL;
}

int main(int argc, char **argv)
{
  int T, t, x, y, ch; unsigned s = 99991u; pthread_t th[256]; double t0, t1, tw; unsigned long long bytes = 0;
  if (argc < 6) { fprintf(stderr, "usage: mt_bench THREADS IMAGES_PER_THREAD WIDTH HEIGHT QUALITY [baseline]\n"); return 2; }
  T = atoi(argv[1]); NIMG = atoi(argv[2]); W = atoi(argv[3]); H = atoi(argv[4]); Q = atoi(argv[5]); BASE = argc > 6;
  if (T < 1 || T > 256) return 2;
  image = (unsigned char *)malloc((size_t)W * H * 3);
  for (y = 0; y < H; y++)
    for (x = 0; x < W; x++)
      for (ch = 0; ch < 3; ch++) {
        s = s * 1664525u + 1013904223u;
        image[((size_t)y * W + x) * 3 + ch] = (unsigned char)(((x * (2 + ch) + y * (4 - ch)) & 0xFF) / 2 + ((s >> 25) & 0x3F) + (((x / 32 + y / 32) % 3) == 0 ? 48 : 0));
      }
  /* one untimed image per thread first: library load, device context, encoder construction */
  {
    int keep = NIMG;
    NIMG = 1; tw = now();
    for (t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, (void *)(long)t);
    for (t = 0; t < T; t++) pthread_join(th[t], NULL);
    tw = now() - tw; NIMG = keep;
    memset(total_bytes, 0, sizeof total_bytes);
  }
  t0 = now();
  for (t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, (void *)(long)t);
  for (t = 0; t < T; t++) pthread_join(th[t], NULL);
  t1 = now();
  for (t = 0; t < T; t++) bytes += total_bytes[t];
  for (t = 1; t < T; t++) if (hashes[t] != hashes[0]) { fprintf(stderr, "threads disagree on the file\n"); return 1; }
  printf("{\"threads\": %d, \"images\": %d, \"size\": \"%dx%d\", \"quality\": %d, \"baseline\": %s, \"first_image_s\": %.3f, "
         "\"seconds\": %.4f, \"images_per_s\": %.2f, \"mpix_per_s\": %.1f, \"jpeg_bytes_per_image\": %llu, \"fnv1a_first\": \"%016llx\"}\n",
         T, T * NIMG, W, H, Q, BASE ? "true" : "false", tw, t1 - t0, T * NIMG / (t1 - t0), (double)T * NIMG * W * H / (t1 - t0) / 1e6,
         bytes / ((unsigned long long)T * NIMG), hashes[0]);
  return 0;
}
