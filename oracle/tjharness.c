/*
 * tjharness.c -- TEST INFRASTRUCTURE: drives the REAL reference TurboJPEG library
 * (oracle/_ref/libturbojpeg.so.0, built from /root/reference/turbojpeg.c unchanged) through
 * tjCompress2 (turbojpeg.c:1169).  Run plainly it gives the reference bytes; run with
 * libmozjpeg_hip_jpeg62.so in front (LD_PRELOAD) the SAME unchanged tjCompress2 ->
 * tj3Compress8 -> jpeg_start_compress / jpeg_write_scanlines / jpeg_finish_compress call chain
 * (turbojpeg-mp.c:115-125) lands in the GPU path.
 * usage: tjharness W H PIXELFORMAT SUBSAMP QUALITY FLAGS in.raw out.jpg
 *        (PIXELFORMAT/SUBSAMP/FLAGS are the TJPF_ / TJSAMP_ / TJFLAG_ integers of turbojpeg.h)
 *        PIXELFORMAT = -1: in.raw is a planar YUV image (tjBufSizeYUV2 layout, align 1) and the call is
 *        tjCompressFromYUV (turbojpeg.c:1437 -> tj3CompressFromYUVPlanes8 :1222 -> jpeg_write_raw_data).
 */
#include <stdio.h>
#include <stdlib.h>
#include "turbojpeg.h"

int main(int argc, char **argv)
{
  int w, h, pf, ss, q, flags;
  size_t n;
  unsigned char *src, *jpeg = NULL;
  unsigned long size = 0;
  tjhandle tj;
  FILE *f;
  if (argc != 9) { fprintf(stderr, "usage: tjharness W H PF SUBSAMP Q FLAGS in.raw out.jpg\n"); return 2; }
  w = atoi(argv[1]); h = atoi(argv[2]); pf = atoi(argv[3]); ss = atoi(argv[4]); q = atoi(argv[5]); flags = atoi(argv[6]);
  n = pf < 0 ? tjBufSizeYUV2(w, 1, h, ss) : (size_t)w * h * tjPixelSize[pf];
  src = malloc(n);
  f = fopen(argv[7], "rb");
  if (!f || fread(src, 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", argv[7]); return 2; }
  fclose(f);
  tj = tjInitCompress();
  if (!tj) { fprintf(stderr, "tjInitCompress: %s\n", tjGetErrorStr()); return 1; }
  if (pf < 0) {
    if (tjCompressFromYUV(tj, src, w, 1, h, ss, &jpeg, &size, q, flags) != 0) {
      fprintf(stderr, "tjCompressFromYUV: %s\n", tjGetErrorStr2(tj));
      return 1;
    }
  } else if (tjCompress2(tj, src, w, 0, h, pf, &jpeg, &size, ss, q, flags) != 0) {
    fprintf(stderr, "tjCompress2: %s\n", tjGetErrorStr2(tj));
    return 1;
  }
  f = fopen(argv[8], "wb");
  fwrite(jpeg, 1, size, f);
  fclose(f);
  tjFree(jpeg);
  tjDestroy(tj);
  free(src);
  return 0;
}
