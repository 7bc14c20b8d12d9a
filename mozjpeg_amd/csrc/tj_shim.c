/*
 * tj_shim.c -- TurboJPEG-signature entry points on top of the MI355X batch encoder (libmozjpeg_hip_turbojpeg.so).
 *
 * A stock libturbojpeg carries a PRIVATE libjpeg (CMakeLists.txt:684-707), so the libjpeg drop-in cannot get in front of
 * it; this library interposes the TurboJPEG compress entry points themselves (turbojpeg.c:1169 tjCompress2,
 * turbojpeg-mp.c:69 tj3Compress8, turbojpeg.c:1222 tj3CompressFromYUVPlanes8 and their legacy wrappers).  Placed in front
 * of a libturbojpeg (LD_PRELOAD / link order) it serves compress handles on the GPU and forwards every handle it did not
 * create (decompress / transform instances) to the library behind it; used alone it is a compress-only TurboJPEG.
 *
 * What TurboJPEG asks of the codec (setCompDefaults turbojpeg.c:316-390): the JCP_FASTEST profile (turbojpeg.c:336) --
 * Annex K tables scaled by jpeg_set_quality(q, TRUE), standard Huffman tables unless TJPARAM_OPTIMIZE / progressive /
 * 12-bit, no trellis -- with the sampling factors of TJSAMP_*, the byte order of TJPF_*, restart intervals, JFIF
 * density.  Exactly these are mapped onto mjh_params; the output is the byte stream the reference TurboJPEG produces.
 * The LEGACY tjCompress2 selects the fast DCT (JDCT_IFAST, jfdctfst.c) unless quality >= 96 or TJFLAG_ACCURATEDCT is given
 * (processFlags turbojpeg.c:522-527), the 3.x API through TJPARAM_FASTDCT: mjh_params.dct_method, coded bit-exactly like the
 * accurate one, 8- and 12-bit samples.  Outside the GPU path (an ERROR, never a CPU fallback): CMYK / YCCK, lossless.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "turbojpeg.h"
#include "mozjpeg_hip.h"

#define TJS_MAGIC 0x4D4A4854u
#define PAD(v, p) (((v) + (p) - 1) & (~((p) - 1)))

typedef struct tjs {
  unsigned magic;
  struct tjs *self;
  /* TJPARAM_* state (tj3Set / legacy arguments) */
  int quality, subsamp, bottom_up, no_realloc, fast_dct, optimize, progressive, arithmetic, lossless, colorspace;
  int restart_blocks, restart_rows, xdensity, ydensity, density_units, stop_on_warning, precision;
  int jpeg_width, jpeg_height;
  /* one cached encoder (a handle is used by one thread at a time, like a tjinstance) */
  mjh_encoder *enc;
  mjh_params enc_params;
  int device;
  char err[200];
  int err_code;
} tjs;

static __thread char g_err[200] = "No error";

static const int kMcuW[TJ_NUMSAMP] = { 8, 16, 16, 8, 8, 32, 8 };
static const int kMcuH[TJ_NUMSAMP] = { 8, 8, 16, 8, 16, 8, 32 };
static const int kPixelSize[TJ_NUMPF] = { 3, 3, 4, 4, 4, 4, 1, 4, 4, 4, 4, 4 };
static const int kRed[TJ_NUMPF] = { 0, 2, 0, 2, 3, 1, -1, 0, 2, 3, 1, -1 };
static const int kGreen[TJ_NUMPF] = { 1, 1, 1, 1, 2, 2, -1, 1, 1, 2, 2, -1 };
static const int kBlue[TJ_NUMPF] = { 2, 0, 2, 0, 1, 3, -1, 2, 0, 1, 3, -1 };

static tjs *ours(tjhandle h) { tjs *t = (tjs *)h; return t && t->magic == TJS_MAGIC && t->self == t ? t : NULL; }
static void *next_sym(const char *name) { return dlsym(RTLD_NEXT, name); }

static int fail(tjs *t, const char *fn, const char *msg)
{
  snprintf(g_err, sizeof(g_err), "%s(): %s", fn, msg);
  if (t) { snprintf(t->err, sizeof(t->err), "%s(): %s", fn, msg); t->err_code = TJERR_FATAL; }
  return -1;
}

static void defaults(tjs *t)
{ /* tj3Init turbojpeg.c:540-600 */
  t->quality = -1; t->subsamp = TJSAMP_UNKNOWN; t->colorspace = -1; t->precision = 8;
  t->xdensity = 1; t->ydensity = 1; t->density_units = 0;
  t->jpeg_width = t->jpeg_height = -1;
}

static int pick_device(void)
{
  static int next = 0;
  const char *v = getenv("MOZJPEG_HIP_DEVICE");
  int n = mjh_device_count();
  if (n < 1) n = 1;
  if (v && *v >= '0' && *v <= '9') return atoi(v) % n;
  return __sync_fetch_and_add(&next, 1) % n;   /* handles are dealt round-robin over the GPUs */
}

DLLEXPORT tjhandle tj3Init(int initType)
{
  tjs *t;
  if (initType != TJINIT_COMPRESS) {
    tjhandle (*f)(int) = (tjhandle (*)(int))next_sym("tj3Init");
    if (f) return f(initType);
    snprintf(g_err, sizeof(g_err), "tj3Init(): this library serves compress instances only");
    return NULL;
  }
  t = (tjs *)calloc(1, sizeof(*t));
  if (!t) { snprintf(g_err, sizeof(g_err), "tj3Init(): Memory allocation failure"); return NULL; }
  t->magic = TJS_MAGIC; t->self = t;
  defaults(t);
  t->device = pick_device();
  snprintf(t->err, sizeof(t->err), "No error");
  return (tjhandle)t;
}
DLLEXPORT tjhandle tjInitCompress(void) { return tj3Init(TJINIT_COMPRESS); }

DLLEXPORT void tj3Destroy(tjhandle handle)
{
  tjs *t = ours(handle);
  if (!t) { void (*f)(tjhandle) = (void (*)(tjhandle))next_sym("tj3Destroy"); if (f && handle) f(handle); return; }
  if (t->enc) mjh_encoder_destroy(t->enc);
  t->magic = 0;
  free(t);
}
DLLEXPORT int tjDestroy(tjhandle handle)
{
  if (!ours(handle)) { int (*f)(tjhandle) = (int (*)(tjhandle))next_sym("tjDestroy"); if (f && handle) return f(handle); snprintf(g_err, sizeof(g_err), "tjDestroy(): Invalid handle"); return -1; }
  tj3Destroy(handle);
  return 0;
}

DLLEXPORT char *tj3GetErrorStr(tjhandle handle)
{
  tjs *t = ours(handle);
  if (t) return t->err;
  if (handle) { char *(*f)(tjhandle) = (char *(*)(tjhandle))next_sym("tj3GetErrorStr"); if (f) return f(handle); }
  return g_err;
}
DLLEXPORT char *tjGetErrorStr2(tjhandle handle) { return tj3GetErrorStr(handle); }
DLLEXPORT char *tjGetErrorStr(void) { return g_err; }
DLLEXPORT int tj3GetErrorCode(tjhandle handle)
{
  tjs *t = ours(handle);
  if (t) return t->err_code;
  if (handle) { int (*f)(tjhandle) = (int (*)(tjhandle))next_sym("tj3GetErrorCode"); if (f) return f(handle); }
  return TJERR_FATAL;
}
DLLEXPORT int tjGetErrorCode(tjhandle handle) { return tj3GetErrorCode(handle); }

DLLEXPORT void *tj3Alloc(size_t bytes) { return malloc(bytes); }
DLLEXPORT unsigned char *tjAlloc(int bytes) { return (unsigned char *)malloc((size_t)bytes); }
DLLEXPORT void tj3Free(void *buffer) { free(buffer); }
DLLEXPORT void tjFree(unsigned char *buffer) { free(buffer); }

/* ---- sizes (pure functions; turbojpeg.c:897-1147) ---- */
DLLEXPORT size_t tj3JPEGBufSize(int width, int height, int jpegSubsamp)
{
  unsigned long long r;
  int mw, mh, csf;
  if (width < 1 || height < 1 || jpegSubsamp < TJSAMP_UNKNOWN || jpegSubsamp >= TJ_NUMSAMP) { snprintf(g_err, sizeof(g_err), "tj3JPEGBufSize(): Invalid argument"); return 0; }
  if (jpegSubsamp == TJSAMP_UNKNOWN) jpegSubsamp = TJSAMP_444;
  mw = kMcuW[jpegSubsamp]; mh = kMcuH[jpegSubsamp];
  csf = jpegSubsamp == TJSAMP_GRAY ? 0 : 4 * 64 / (mw * mh);
  r = (unsigned long long)PAD(width, mw) * PAD(height, mh) * (2ULL + csf) + 2048ULL;
  return (size_t)r;
}
DLLEXPORT unsigned long tjBufSize(int width, int height, int jpegSubsamp)
{
  size_t r;
  if (jpegSubsamp < 0) { snprintf(g_err, sizeof(g_err), "tjBufSize(): Invalid argument"); return (unsigned long)-1; }
  r = tj3JPEGBufSize(width, height, jpegSubsamp);
  return r == 0 ? (unsigned long)-1 : (unsigned long)r;
}
DLLEXPORT int tj3YUVPlaneWidth(int componentID, int width, int subsamp)
{
  unsigned long long pw;
  if (width < 1 || subsamp < 0 || subsamp >= TJ_NUMSAMP || componentID < 0 || componentID >= (subsamp == TJSAMP_GRAY ? 1 : 3)) { snprintf(g_err, sizeof(g_err), "tj3YUVPlaneWidth(): Invalid argument"); return 0; }
  pw = PAD((unsigned long long)width, (unsigned long long)(kMcuW[subsamp] / 8));
  return (int)(componentID == 0 ? pw : pw * 8 / kMcuW[subsamp]);
}
DLLEXPORT int tj3YUVPlaneHeight(int componentID, int height, int subsamp)
{
  unsigned long long ph;
  if (height < 1 || subsamp < 0 || subsamp >= TJ_NUMSAMP || componentID < 0 || componentID >= (subsamp == TJSAMP_GRAY ? 1 : 3)) { snprintf(g_err, sizeof(g_err), "tj3YUVPlaneHeight(): Invalid argument"); return 0; }
  ph = PAD((unsigned long long)height, (unsigned long long)(kMcuH[subsamp] / 8));
  return (int)(componentID == 0 ? ph : ph * 8 / kMcuH[subsamp]);
}
DLLEXPORT int tjPlaneWidth(int componentID, int width, int subsamp) { int r = tj3YUVPlaneWidth(componentID, width, subsamp); return r == 0 ? -1 : r; }
DLLEXPORT int tjPlaneHeight(int componentID, int height, int subsamp) { int r = tj3YUVPlaneHeight(componentID, height, subsamp); return r == 0 ? -1 : r; }
DLLEXPORT size_t tj3YUVPlaneSize(int componentID, int width, int stride, int height, int subsamp)
{
  int pw, ph;
  if (width < 1 || height < 1 || subsamp < 0 || subsamp >= TJ_NUMSAMP) { snprintf(g_err, sizeof(g_err), "tj3YUVPlaneSize(): Invalid argument"); return 0; }
  pw = tj3YUVPlaneWidth(componentID, width, subsamp); ph = tj3YUVPlaneHeight(componentID, height, subsamp);
  if (pw == 0 || ph == 0) return 0;
  if (stride == 0) stride = pw; else stride = abs(stride);
  return (size_t)stride * (ph - 1) + pw;
}
DLLEXPORT unsigned long tjPlaneSizeYUV(int componentID, int width, int stride, int height, int subsamp) { size_t r = tj3YUVPlaneSize(componentID, width, stride, height, subsamp); return r == 0 ? (unsigned long)-1 : (unsigned long)r; }
DLLEXPORT size_t tj3YUVBufSize(int width, int align, int height, int subsamp)
{
  unsigned long long r = 0;
  int nc, i;
  if (align < 1 || (align & (align - 1)) != 0 || subsamp < 0 || subsamp >= TJ_NUMSAMP) { snprintf(g_err, sizeof(g_err), "tj3YUVBufSize(): Invalid argument"); return 0; }
  nc = subsamp == TJSAMP_GRAY ? 1 : 3;
  for (i = 0; i < nc; i++) {
    const int pw = tj3YUVPlaneWidth(i, width, subsamp), ph = tj3YUVPlaneHeight(i, height, subsamp);
    if (pw == 0 || ph == 0) return 0;
    r += (unsigned long long)PAD(pw, align) * ph;
  }
  return (size_t)r;
}
DLLEXPORT unsigned long tjBufSizeYUV2(int width, int align, int height, int subsamp) { size_t r = tj3YUVBufSize(width, align, height, subsamp); return r == 0 ? (unsigned long)-1 : (unsigned long)r; }

/* ---- parameters (tj3Set / tj3Get turbojpeg.c:660-860, the compress-side ones) ---- */
DLLEXPORT int tj3Set(tjhandle handle, int param, int value)
{
  tjs *t = ours(handle);
  if (!t) { int (*f)(tjhandle, int, int) = (int (*)(tjhandle, int, int))next_sym("tj3Set"); return f && handle ? f(handle, param, value) : fail(NULL, "tj3Set", "Invalid handle"); }
#define RANGE(lo, hi) do { if (value < (lo) || value > (hi)) return fail(t, "tj3Set", "Parameter value out of range"); } while (0)
  switch (param) {
  case TJPARAM_STOPONWARNING: RANGE(0, 1); t->stop_on_warning = value; break;
  case TJPARAM_BOTTOMUP: RANGE(0, 1); t->bottom_up = value; break;
  case TJPARAM_NOREALLOC: RANGE(0, 1); t->no_realloc = value; break;
  case TJPARAM_QUALITY: RANGE(1, 100); t->quality = value; break;
  case TJPARAM_SUBSAMP: RANGE(0, TJ_NUMSAMP - 1); t->subsamp = value; break;
  case TJPARAM_FASTUPSAMPLE: RANGE(0, 1); break;
  case TJPARAM_FASTDCT: RANGE(0, 1); t->fast_dct = value; break;
  case TJPARAM_OPTIMIZE: RANGE(0, 1); t->optimize = value; break;
  case TJPARAM_PROGRESSIVE: RANGE(0, 1); t->progressive = value; break;
  case TJPARAM_SCANLIMIT: break;
  case TJPARAM_ARITHMETIC: RANGE(0, 1); t->arithmetic = value; break;
  case TJPARAM_LOSSLESS: RANGE(0, 1); t->lossless = value; break;
  case TJPARAM_LOSSLESSPSV: case TJPARAM_LOSSLESSPT: break;
  case TJPARAM_RESTARTBLOCKS: RANGE(0, 65535); t->restart_blocks = value; if (value) t->restart_rows = 0; break;
  case TJPARAM_RESTARTROWS: RANGE(0, 65535); t->restart_rows = value; if (value) t->restart_blocks = 0; break;
  case TJPARAM_XDENSITY: RANGE(1, 65535); t->xdensity = value; break;
  case TJPARAM_YDENSITY: RANGE(1, 65535); t->ydensity = value; break;
  case TJPARAM_DENSITYUNITS: RANGE(0, 2); t->density_units = value; break;
  case TJPARAM_COLORSPACE: RANGE(0, TJ_NUMCS - 1); t->colorspace = value; break;
  case TJPARAM_MAXMEMORY: case TJPARAM_MAXPIXELS: break;
  case TJPARAM_JPEGWIDTH: case TJPARAM_JPEGHEIGHT: case TJPARAM_PRECISION:
    return fail(t, "tj3Set", "Parameter is read-only in compression instances");
  default: return fail(t, "tj3Set", "Invalid parameter");
  }
#undef RANGE
  return 0;
}

DLLEXPORT int tj3Get(tjhandle handle, int param)
{
  tjs *t = ours(handle);
  if (!t) { int (*f)(tjhandle, int) = (int (*)(tjhandle, int))next_sym("tj3Get"); return f && handle ? f(handle, param) : -1; }
  switch (param) {
  case TJPARAM_STOPONWARNING: return t->stop_on_warning;
  case TJPARAM_BOTTOMUP: return t->bottom_up;
  case TJPARAM_NOREALLOC: return t->no_realloc;
  case TJPARAM_QUALITY: return t->quality;
  case TJPARAM_SUBSAMP: return t->subsamp;
  case TJPARAM_JPEGWIDTH: return t->jpeg_width;
  case TJPARAM_JPEGHEIGHT: return t->jpeg_height;
  case TJPARAM_PRECISION: return t->precision;
  case TJPARAM_COLORSPACE: return t->colorspace;
  case TJPARAM_FASTDCT: return t->fast_dct;
  case TJPARAM_OPTIMIZE: return t->optimize;
  case TJPARAM_PROGRESSIVE: return t->progressive;
  case TJPARAM_ARITHMETIC: return t->arithmetic;
  case TJPARAM_LOSSLESS: return t->lossless;
  case TJPARAM_RESTARTBLOCKS: return t->restart_blocks;
  case TJPARAM_RESTARTROWS: return t->restart_rows;
  case TJPARAM_XDENSITY: return t->xdensity;
  case TJPARAM_YDENSITY: return t->ydensity;
  case TJPARAM_DENSITYUNITS: return t->density_units;
  }
  return -1;
}

/* ---- the codec parameters TurboJPEG would leave in its cinfo (setCompDefaults turbojpeg.c:316-390) ---- */
static int build_params(tjs *t, const char *fn, int width, int height, int pixelFormat, int precision, mjh_params *p)
{
  int subsamp = t->subsamp, gray_out, in_comps = 3;
  if (t->lossless) return fail(t, fn, "lossless mode is outside the GPU path (no CPU fallback)");
  if (pixelFormat == TJPF_CMYK || t->colorspace == TJCS_CMYK || t->colorspace == TJCS_YCCK) return fail(t, fn, "CMYK / YCCK are outside the GPU path (no CPU fallback)");
  if (pixelFormat == TJPF_GRAY) in_comps = 1;
  gray_out = t->colorspace == TJCS_GRAY || (t->colorspace < 0 && subsamp == TJSAMP_GRAY) || in_comps == 1;
  if (in_comps == 1 && !(t->colorspace == TJCS_GRAY || (t->colorspace < 0 && subsamp == TJSAMP_GRAY)))
    return fail(t, fn, "grayscale pixels need TJSAMP_GRAY / TJCS_GRAY");
  if (mjh_params_defaults(p, width, height, in_comps, gray_out, MJH_PROFILE_FASTEST, gray_out ? 1 : kMcuW[subsamp] / 8, gray_out ? 1 : kMcuH[subsamp] / 8) != MJH_OK ||
      mjh_params_set_quality(p, t->quality, 1, 0) != MJH_OK)
    return fail(t, fn, mjh_last_error());
  if (in_comps == 3) {
    p->input_pixel_size = kPixelSize[pixelFormat];
    p->rgb_offset[0] = kRed[pixelFormat]; p->rgb_offset[1] = kGreen[pixelFormat]; p->rgb_offset[2] = kBlue[pixelFormat];
  }
  if (t->colorspace == TJCS_RGB && !gray_out) {   /* jpeg_set_colorspace(JCS_RGB): unconverted samples, ids 'R' 'G' 'B', tables 0, no JFIF */
    int i;
    p->color_transform = MJH_COLOR_NONE;
    p->write_JFIF_header = 0;
    for (i = 0; i < 3; i++) { p->component_id[i] = "RGB"[i]; p->quant_tbl_no[i] = p->dc_tbl_no[i] = p->ac_tbl_no[i] = 0; }
    p->h_samp_factor[0] = kMcuW[subsamp] / 8; p->v_samp_factor[0] = kMcuH[subsamp] / 8;   /* set after the colour space, like setCompDefaults */
  }
  p->data_precision = precision;
  p->dct_method = t->fast_dct ? 1 : 0;                        /* cinfo->dct_method = JDCT_FASTEST / JDCT_ISLOW, turbojpeg.c:358 */
  p->optimize_coding = precision == 12 ? 1 : t->optimize;     /* turbojpeg.c:375-376; 12-bit: jcparam.c:452-453 */
  p->restart_interval = (unsigned)t->restart_blocks;
  p->restart_in_rows = t->restart_rows;
  if (t->progressive && mjh_params_simple_progression(p) != MJH_OK) return fail(t, fn, mjh_last_error());
  if (t->arithmetic) { p->arith_code = 1; p->optimize_coding = 0; }   /* TJPARAM_ARITHMETIC -> cinfo->arith_code (turbojpeg.c setCompDefaults) */
  return 0;
}

static int get_encoder(tjs *t, const char *fn, const mjh_params *p)
{
  if (t->enc && memcmp(&t->enc_params, p, sizeof(*p)) == 0) return 0;
  if (t->enc) { mjh_encoder_destroy(t->enc); t->enc = NULL; }
  if (mjh_params_size() != sizeof(mjh_params)) return fail(t, fn, "libmozjpeg_hip.so was built with another mjh_params layout than this shim (rebuild both)");
  if (mjh_encoder_create(p, 1, t->device, &t->enc) != MJH_OK) return fail(t, fn, mjh_last_error());
  t->enc_params = *p;
  return 0;
}

/* hand the finished file to the caller the way jpeg_mem_dest_tj does (jdatadst-tj.c): in place if it fits / must,
 * else a buffer from tjAlloc; JFIF density patched in (the device writes the default 1:1) */
static int deliver(tjs *t, const char *fn, unsigned char **jpegBuf, size_t *jpegSize, size_t capacity_if_fixed)
{
  const void *base = NULL;
  const mjh_result *res = NULL;
  unsigned char *copy = NULL;
  const unsigned char *file;
  size_t n = 0;
  int cnt = 0;
  if (mjh_collect(t->enc, 0, &base, &res, &cnt) == MJH_OK && cnt == 1) { file = (const unsigned char *)base + res[0].offset; n = (size_t)res[0].size; }
  else if (mjh_get_jpeg_size(t->enc, 0, &n) == MJH_OK && (copy = (unsigned char *)malloc(n)) != NULL && mjh_get_jpeg(t->enc, 0, copy, n, &n) == MJH_OK) file = copy;
  else { free(copy); return fail(t, fn, mjh_last_error()); }
  if (t->no_realloc) {
    if (*jpegBuf == NULL || n > capacity_if_fixed) { free(copy); return fail(t, fn, "Buffer passed to JPEG library was too small"); }
  } else if (*jpegBuf == NULL || *jpegSize < n) {
    unsigned char *nb = (unsigned char *)malloc(n);
    if (!nb) { free(copy); return fail(t, fn, "Memory allocation failure"); }
    free(*jpegBuf);
    *jpegBuf = nb;
  }
  memcpy(*jpegBuf, file, n);
  if (n > 20 && (*jpegBuf)[2] == 0xFF && (*jpegBuf)[3] == 0xE0) {   /* emit_jfif_app0 jcmarker.c:422-449 */
    (*jpegBuf)[13] = (unsigned char)t->density_units;
    (*jpegBuf)[14] = (unsigned char)(t->xdensity >> 8); (*jpegBuf)[15] = (unsigned char)t->xdensity;
    (*jpegBuf)[16] = (unsigned char)(t->ydensity >> 8); (*jpegBuf)[17] = (unsigned char)t->ydensity;
  }
  *jpegSize = n;
  free(copy);
  return 0;
}

static int compress_pixels(tjs *t, const char *fn, const void *srcBuf, int width, int pitch, int height, int pixelFormat, int precision,
                           unsigned char **jpegBuf, size_t *jpegSize)
{ /* tj3Compress8 / 12 turbojpeg-mp.c:69-136 */
  mjh_params p;
  void *stage = NULL;
  size_t cap = 0, row_bytes;
  int y;
  if (srcBuf == NULL || width <= 0 || pitch < 0 || height <= 0 || pixelFormat < 0 || pixelFormat >= TJ_NUMPF || jpegBuf == NULL || jpegSize == NULL)
    return fail(t, fn, "Invalid argument");
  if (t->quality == -1) return fail(t, fn, "TJPARAM_QUALITY must be specified");
  if (t->subsamp == TJSAMP_UNKNOWN) return fail(t, fn, "TJPARAM_SUBSAMP must be specified");
  if (build_params(t, fn, width, height, pixelFormat, precision, &p) || get_encoder(t, fn, &p)) return -1;
  row_bytes = (size_t)width * kPixelSize[pixelFormat] * (precision == 12 ? 2 : 1);
  if (pitch == 0) pitch = width * kPixelSize[pixelFormat];
  if (mjh_host_staging(t->enc, &stage, &cap) != MJH_OK || cap < row_bytes * (size_t)height) return fail(t, fn, mjh_last_error());
  for (y = 0; y < height; y++) {   /* pitch counts SAMPLES (turbojpeg.h): bytes for 8-bit, 2-byte units for 12-bit */
    const size_t src_row = (size_t)(t->bottom_up ? height - 1 - y : y) * (size_t)pitch * (precision == 12 ? 2 : 1);
    memcpy((unsigned char *)stage + (size_t)y * row_bytes, (const unsigned char *)srcBuf + src_row, row_bytes);
  }
  if (mjh_encode_host(t->enc, stage, row_bytes, row_bytes * (size_t)height, 1) != MJH_OK) return fail(t, fn, mjh_last_error());
  t->jpeg_width = width; t->jpeg_height = height; t->precision = precision;
  return deliver(t, fn, jpegBuf, jpegSize, tj3JPEGBufSize(width, height, t->subsamp));
}

DLLEXPORT int tj3Compress8(tjhandle handle, const unsigned char *srcBuf, int width, int pitch, int height, int pixelFormat,
                           unsigned char **jpegBuf, size_t *jpegSize)
{
  tjs *t = ours(handle);
  if (!t) {
    int (*f)(tjhandle, const unsigned char *, int, int, int, int, unsigned char **, size_t *) =
      (int (*)(tjhandle, const unsigned char *, int, int, int, int, unsigned char **, size_t *))next_sym("tj3Compress8");
    return f && handle ? f(handle, srcBuf, width, pitch, height, pixelFormat, jpegBuf, jpegSize) : fail(NULL, "tj3Compress8", "Invalid handle");
  }
  return compress_pixels(t, "tj3Compress8", srcBuf, width, pitch, height, pixelFormat, 8, jpegBuf, jpegSize);
}

DLLEXPORT int tj3Compress12(tjhandle handle, const short *srcBuf, int width, int pitch, int height, int pixelFormat,
                            unsigned char **jpegBuf, size_t *jpegSize)
{
  tjs *t = ours(handle);
  if (!t) {
    int (*f)(tjhandle, const short *, int, int, int, int, unsigned char **, size_t *) =
      (int (*)(tjhandle, const short *, int, int, int, int, unsigned char **, size_t *))next_sym("tj3Compress12");
    return f && handle ? f(handle, srcBuf, width, pitch, height, pixelFormat, jpegBuf, jpegSize) : fail(NULL, "tj3Compress12", "Invalid handle");
  }
  return compress_pixels(t, "tj3Compress12", srcBuf, width, pitch, height, pixelFormat, 12, jpegBuf, jpegSize);
}

static void legacy_flags(tjs *t, int flags)
{ /* processFlags turbojpeg.c:507-534, COMPRESS */
  t->bottom_up = !!(flags & TJFLAG_BOTTOMUP);
  t->no_realloc = !!(flags & TJFLAG_NOREALLOC);
  t->fast_dct = !(t->quality >= 96 || (flags & TJFLAG_ACCURATEDCT));
  t->stop_on_warning = !!(flags & TJFLAG_STOPONWARNING);
  t->progressive = !!(flags & TJFLAG_PROGRESSIVE);
}

DLLEXPORT int tjCompress2(tjhandle handle, const unsigned char *srcBuf, int width, int pitch, int height, int pixelFormat,
                          unsigned char **jpegBuf, unsigned long *jpegSize, int jpegSubsamp, int jpegQual, int flags)
{ /* turbojpeg.c:1169-1195 */
  tjs *t = ours(handle);
  size_t size;
  int rc;
  if (!t) {
    int (*f)(tjhandle, const unsigned char *, int, int, int, int, unsigned char **, unsigned long *, int, int, int) =
      (int (*)(tjhandle, const unsigned char *, int, int, int, int, unsigned char **, unsigned long *, int, int, int))next_sym("tjCompress2");
    return f && handle ? f(handle, srcBuf, width, pitch, height, pixelFormat, jpegBuf, jpegSize, jpegSubsamp, jpegQual, flags) : fail(NULL, "tjCompress2", "Invalid handle");
  }
  if (jpegSize == NULL || jpegSubsamp < 0 || jpegSubsamp >= TJ_NUMSAMP || jpegQual < 0 || jpegQual > 100) return fail(t, "tjCompress2", "Invalid argument");
  t->quality = jpegQual; t->subsamp = jpegSubsamp;
  legacy_flags(t, flags);
  size = (size_t)*jpegSize;
  rc = compress_pixels(t, "tjCompress2", srcBuf, width, pitch, height, pixelFormat, 8, jpegBuf, &size);
  *jpegSize = (unsigned long)size;
  return rc;
}

DLLEXPORT int tjCompress(tjhandle handle, unsigned char *srcBuf, int width, int pitch, int height, int pixelSize,
                         unsigned char *jpegBuf, unsigned long *jpegSize, int jpegSubsamp, int jpegQual, int flags)
{ /* turbojpeg.c:1198-1222 (the TJ_YUV branch of the 1.0 API is not served) */
  int pf;
  if (flags & TJ_YUV) return fail(ours(handle), "tjCompress", "TJ_YUV is outside the GPU path (no CPU fallback)");
  if (pixelSize == 1) pf = TJPF_GRAY;
  else if (pixelSize == 3) pf = (flags & TJ_BGR) ? TJPF_BGR : TJPF_RGB;
  else if (pixelSize == 4) pf = (flags & TJ_ALPHAFIRST) ? ((flags & TJ_BGR) ? TJPF_XBGR : TJPF_XRGB) : ((flags & TJ_BGR) ? TJPF_BGRX : TJPF_RGBX);
  else return fail(ours(handle), "tjCompress", "Invalid argument");
  if (jpegSize == NULL) return fail(ours(handle), "tjCompress", "Invalid argument");
  return tjCompress2(handle, srcBuf, width, pitch, height, pf, &jpegBuf, jpegSize, jpegSubsamp, jpegQual, flags | TJFLAG_NOREALLOC);
}

/* ---- planar YUV input (tj3CompressFromYUVPlanes8 turbojpeg.c:1222-1340): jpeg_write_raw_data for whole images ---- */
static int compress_planes(tjs *t, const char *fn, const unsigned char *const *srcPlanes, int width, const int *strides, int height,
                           unsigned char **jpegBuf, size_t *jpegSize)
{
  mjh_params p;
  const void *pl[MJH_MAX_COMPS] = { 0, 0, 0, 0 };
  size_t pitch[MJH_MAX_COMPS] = { 0, 0, 0, 0 };
  int pw[MJH_MAX_COMPS] = { 0, 0, 0, 0 }, ph[MJH_MAX_COMPS] = { 0, 0, 0, 0 }, nc, i;
  if (!srcPlanes || !srcPlanes[0] || width <= 0 || height <= 0 || jpegBuf == NULL || jpegSize == NULL) return fail(t, fn, "Invalid argument");
  if (t->quality == -1) return fail(t, fn, "TJPARAM_QUALITY must be specified");
  if (t->subsamp == TJSAMP_UNKNOWN) return fail(t, fn, "TJPARAM_SUBSAMP must be specified");
  if (t->subsamp != TJSAMP_GRAY && (!srcPlanes[1] || !srcPlanes[2])) return fail(t, fn, "Invalid argument");
  if (build_params(t, fn, width, height, t->subsamp == TJSAMP_GRAY ? TJPF_GRAY : TJPF_RGB, 8, &p) || get_encoder(t, fn, &p)) return -1;
  nc = t->subsamp == TJSAMP_GRAY ? 1 : 3;
  for (i = 0; i < nc; i++) {
    pw[i] = tj3YUVPlaneWidth(i, width, t->subsamp); ph[i] = tj3YUVPlaneHeight(i, height, t->subsamp);
    if (strides && strides[i] < 0) return fail(t, fn, "negative plane strides are outside the GPU path");
    pitch[i] = (size_t)(strides && strides[i] != 0 ? strides[i] : pw[i]);
    pl[i] = srcPlanes[i];
  }
  if (mjh_encode_planes_host(t->enc, pl, pitch, NULL, pw, ph, 1) != MJH_OK) return fail(t, fn, mjh_last_error());
  t->jpeg_width = width; t->jpeg_height = height; t->precision = 8;
  return deliver(t, fn, jpegBuf, jpegSize, tj3JPEGBufSize(width, height, t->subsamp));
}

DLLEXPORT int tj3CompressFromYUVPlanes8(tjhandle handle, const unsigned char * const *srcPlanes, int width, const int *strides, int height,
                                        unsigned char **jpegBuf, size_t *jpegSize)
{
  tjs *t = ours(handle);
  if (!t) {
    int (*f)(tjhandle, const unsigned char * const *, int, const int *, int, unsigned char **, size_t *) =
      (int (*)(tjhandle, const unsigned char * const *, int, const int *, int, unsigned char **, size_t *))next_sym("tj3CompressFromYUVPlanes8");
    return f && handle ? f(handle, srcPlanes, width, strides, height, jpegBuf, jpegSize) : fail(NULL, "tj3CompressFromYUVPlanes8", "Invalid handle");
  }
  return compress_planes(t, "tj3CompressFromYUVPlanes8", srcPlanes, width, strides, height, jpegBuf, jpegSize);
}

static int split_yuv(tjs *t, const char *fn, const unsigned char *srcBuf, int width, int align, int height, const unsigned char *planes[3], int strides[3])
{ /* tj3CompressFromYUV8 turbojpeg.c:1350-1400: one buffer, planes back to back, rows padded to `align` */
  int pw0, ph0;
  if (srcBuf == NULL || width <= 0 || align < 1 || (align & (align - 1)) != 0 || height <= 0) return fail(t, fn, "Invalid argument");
  if (t->subsamp == TJSAMP_UNKNOWN) return fail(t, fn, "TJPARAM_SUBSAMP must be specified");
  pw0 = tj3YUVPlaneWidth(0, width, t->subsamp); ph0 = tj3YUVPlaneHeight(0, height, t->subsamp);
  planes[0] = srcBuf; strides[0] = PAD(pw0, align);
  if (t->subsamp == TJSAMP_GRAY) { strides[1] = strides[2] = 0; planes[1] = planes[2] = NULL; }
  else {
    const int pw1 = tj3YUVPlaneWidth(1, width, t->subsamp), ph1 = tj3YUVPlaneHeight(1, height, t->subsamp);
    strides[1] = strides[2] = PAD(pw1, align);
    planes[1] = planes[0] + (size_t)strides[0] * ph0;
    planes[2] = planes[1] + (size_t)strides[1] * ph1;
  }
  return 0;
}

DLLEXPORT int tj3CompressFromYUV8(tjhandle handle, const unsigned char *srcBuf, int width, int align, int height, unsigned char **jpegBuf, size_t *jpegSize)
{
  tjs *t = ours(handle);
  const unsigned char *planes[3];
  int strides[3];
  if (!t) {
    int (*f)(tjhandle, const unsigned char *, int, int, int, unsigned char **, size_t *) =
      (int (*)(tjhandle, const unsigned char *, int, int, int, unsigned char **, size_t *))next_sym("tj3CompressFromYUV8");
    return f && handle ? f(handle, srcBuf, width, align, height, jpegBuf, jpegSize) : fail(NULL, "tj3CompressFromYUV8", "Invalid handle");
  }
  if (split_yuv(t, "tj3CompressFromYUV8", srcBuf, width, align, height, planes, strides)) return -1;
  return compress_planes(t, "tj3CompressFromYUV8", planes, width, strides, height, jpegBuf, jpegSize);
}

DLLEXPORT int tjCompressFromYUVPlanes(tjhandle handle, const unsigned char **srcPlanes, int width, const int *strides, int height, int subsamp,
                                      unsigned char **jpegBuf, unsigned long *jpegSize, int jpegQual, int flags)
{ /* turbojpeg.c:1405-1430 */
  tjs *t = ours(handle);
  size_t size;
  int rc;
  if (!t) {
    int (*f)(tjhandle, const unsigned char **, int, const int *, int, int, unsigned char **, unsigned long *, int, int) =
      (int (*)(tjhandle, const unsigned char **, int, const int *, int, int, unsigned char **, unsigned long *, int, int))next_sym("tjCompressFromYUVPlanes");
    return f && handle ? f(handle, srcPlanes, width, strides, height, subsamp, jpegBuf, jpegSize, jpegQual, flags) : fail(NULL, "tjCompressFromYUVPlanes", "Invalid handle");
  }
  if (subsamp < 0 || subsamp >= TJ_NUMSAMP || jpegSize == NULL || jpegQual < 0 || jpegQual > 100) return fail(t, "tjCompressFromYUVPlanes", "Invalid argument");
  t->quality = jpegQual; t->subsamp = subsamp;
  legacy_flags(t, flags);
  size = (size_t)*jpegSize;
  rc = compress_planes(t, "tjCompressFromYUVPlanes", (const unsigned char *const *)srcPlanes, width, strides, height, jpegBuf, &size);
  *jpegSize = (unsigned long)size;
  return rc;
}

DLLEXPORT int tjCompressFromYUV(tjhandle handle, const unsigned char *srcBuf, int width, int align, int height, int subsamp,
                                unsigned char **jpegBuf, unsigned long *jpegSize, int jpegQual, int flags)
{ /* turbojpeg.c:1433-1458 */
  tjs *t = ours(handle);
  const unsigned char *planes[3];
  int strides[3], rc;
  size_t size;
  if (!t) {
    int (*f)(tjhandle, const unsigned char *, int, int, int, int, unsigned char **, unsigned long *, int, int) =
      (int (*)(tjhandle, const unsigned char *, int, int, int, int, unsigned char **, unsigned long *, int, int))next_sym("tjCompressFromYUV");
    return f && handle ? f(handle, srcBuf, width, align, height, subsamp, jpegBuf, jpegSize, jpegQual, flags) : fail(NULL, "tjCompressFromYUV", "Invalid handle");
  }
  if (subsamp < 0 || subsamp >= TJ_NUMSAMP || jpegSize == NULL || jpegQual < 0 || jpegQual > 100) return fail(t, "tjCompressFromYUV", "Invalid argument");
  t->quality = jpegQual; t->subsamp = subsamp;
  legacy_flags(t, flags);
  if (split_yuv(t, "tjCompressFromYUV", srcBuf, width, align, height, planes, strides)) return -1;
  size = (size_t)*jpegSize;
  rc = compress_planes(t, "tjCompressFromYUV", planes, width, strides, height, jpegBuf, &size);
  *jpegSize = (unsigned long)size;
  return rc;
}
