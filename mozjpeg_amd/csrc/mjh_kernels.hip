// mjh_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the JPEG encode hot path.
//
// Data layout in HBM (one "image set" per image of the batch):
//   planes : uint8  per component [ph][pw]                     (downsampled samples)
//   coef   : int16  per component [64 zig-zag k][kstride]       (coefficient-major "SoA":
//            lane = block, so every per-block kernel reads/writes 128 contiguous bytes per
//            wave and per coefficient index -- fully coalesced, no LDS transpose needed)
//   tables : MjhHuffTable per image and role
// Every kernel is "one 8x8 block per lane" unless stated otherwise; no MFMA (nothing here is a
// dense contraction).  Float arithmetic of the trellis / deringing reproduces the reference's
// IEEE single/double operation order; this file MUST be built with -ffp-contract=off.
//
// Reference behaviour each kernel reproduces is cited as file:line under /root/reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include "mjh_internal.h"
#include "mjh_device.h"

#define WAVE 64

// zig-zag -> natural (jutils.c:59)
struct ZZTab { int v[64]; };
static constexpr ZZTab make_zz() {
  ZZTab t{};
  const int z[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
  for (int i = 0; i < 64; i++) t.v[i] = z[i];
  return t;
}
static constexpr ZZTab kZZ = make_zz();    // kZZ.v[k]  = natural index of zig-zag position k

// =============================================================================================
// K1  colour conversion + chroma downsampling + edge replication  
//   rgb_ycc_convert jccolext.c:30-75 (tables jccolor.c:213-246), h2v2/h2v1/int_downsample
//   jcsample.c:151-295, expand_right_edge jcsample.c:98, expand_bottom_edge jcprepct.c:113
//   with its two call sites :161-168 / :180-190.  Every replicated sample is an index clamp.
// One lane = one group of H0 x V0 pixels (one chroma sample).  Component 0 has the maximum
// sampling factors H0 x V0, components 1,2 are 1x1.
// =============================================================================================
#define FIXC(x) ((int)((x) * 65536.0 + 0.5))

template <int H0, int V0, class T>   // T = uint8_t (8-bit) or uint16_t (12-bit samples, jccolor.c:96-101)
__global__ void __launch_bounds__(256)
k_color(MjhConst C, const uint8_t *__restrict__ pix, size_t row_pitch, size_t img_stride,
        T *__restrict__ planes)
{
  const int gx = blockIdx.x * 256 + threadIdx.x;
  const int gy = blockIdx.y;
  const int img = blockIdx.z;
  if (gx >= C.groups_x) return;
  const uint8_t *p = pix + (size_t)img * img_stride;
  T *pl = planes + (size_t)img * C.planes_per_image;
  const int ic = C.in_comps;
  const int center = sizeof(T) == 2 ? 2048 : 128;
  const bool below = gy >= C.real_groups_y;          // replicate the last DOWNSAMPLED row
  const int gys = below ? C.real_groups_y - 1 : gy;
  int yv[V0][H0];
  int cbs = 0, crs = 0;
#pragma unroll
  for (int vy = 0; vy < V0; vy++) {
    int iy = gys * V0 + vy;
    if (iy > C.H - 1) iy = C.H - 1;                   // last INPUT row replicated (jcprepct.c:161)
    const T *row = reinterpret_cast<const T *>(p + (size_t)iy * row_pitch);
#pragma unroll
    for (int vx = 0; vx < H0; vx++) {
      int ix = gx * H0 + vx;
      if (ix > C.W - 1) ix = C.W - 1;                 // jcsample.c:98
      if (ic == 3) {
        const T *px = row + (size_t)ix * C.px_size;
        int r = px[C.off_r], g = px[C.off_g], b = px[C.off_b];
        if (C.no_ycc) { yv[vy][vx] = r; cbs += g; crs += b; }        // null_convert jccolor.c:479 (cjpeg -rgb)
        else {
          if (sizeof(T) == 2) { r &= 0xFFF; g &= 0xFFF; b &= 0xFFF; }   // RANGE_LIMIT of the 12-bit build
          yv[vy][vx] = (FIXC(0.29900) * r + FIXC(0.58700) * g + FIXC(0.11400) * b + 32768) >> 16;
          cbs += (-FIXC(0.16874) * r - FIXC(0.33126) * g + FIXC(0.50000) * b + (center << 16) + 32767) >> 16;
          crs += (FIXC(0.50000) * r - FIXC(0.41869) * g - FIXC(0.08131) * b + (center << 16) + 32767) >> 16;
        }
      } else {
        yv[vy][vx] = row[ix];
      }
    }
  }
  {
    const MjhComp &c0 = C.c[0];
    T *y = pl + c0.plane_off;
#pragma unroll
    for (int vy = 0; vy < V0; vy++) {
      const int r = gy * V0 + vy;
      const int svy = below ? V0 - 1 : vy;
#pragma unroll
      for (int vx = 0; vx < H0; vx++) {
        const int c = gx * H0 + vx;
        if (r < c0.ph && c < c0.pw) y[(size_t)r * c0.pw + c] = (T)yv[svy][vx];
      }
    }
  }
  if (C.ncomp == 3) {
    int cb, cr;
    if (H0 == 2 && V0 == 2) { const int bias = 1 + (gx & 1); cb = (cbs + bias) >> 2; cr = (crs + bias) >> 2; }       // h2v2 :263
    else if (H0 == 2 && V0 == 1) { const int bias = gx & 1; cb = (cbs + bias) >> 1; cr = (crs + bias) >> 1; }       // h2v1 :226
    else if (H0 == 1 && V0 == 1) { cb = cbs; cr = crs; }                                                             // fullsize :199
    else { const int n = H0 * V0; cb = (cbs + n / 2) / n; cr = (crs + n / 2) / n; }                                  // int_downsample :151
    const MjhComp &c1 = C.c[1];
    const MjhComp &c2 = C.c[2];
    if (gy < c1.ph && gx < c1.pw) {
      pl[c1.plane_off + (size_t)gy * c1.pw + gx] = (T)cb;
      pl[c2.plane_off + (size_t)gy * c2.pw + gx] = (T)cr;
    }
  }
}

// Input smoothing (cjpeg -smooth N): h2v2_smooth_downsample jcsample.c:304-391, fullsize_smooth_downsample :400-455.
// A smoothing downsampler asks for context rows, which switches the whole preprocessor to pre_process_context
// (jcprepct.c:200-262): rows above the image are copies of row 0, EVERY row below it is a copy of the last input
// row (the padding row groups are downsampled like real ones; this mode has no "replicate the last downsampled
// row" step), columns beyond the image repeat the last one and column -1 counts as column 0 -- all of it is index
// clamping here.  Ratios without a smoothing variant (2x1, 1x2 ...) use the plain filters on the same clamped rows.
// One lane = one output sample of one component; the colour conversion of its 3x3 / 4x4 neighbourhood is redone
// per lane (non-default mode: clarity over speed).
template <class T>
__device__ __forceinline__ int smooth_px(const MjhConst &C, const uint8_t *p, size_t row_pitch, int comp, int iy, int ix)
{
  iy = iy < 0 ? 0 : (iy > C.H - 1 ? C.H - 1 : iy);
  ix = ix < 0 ? 0 : (ix > C.W - 1 ? C.W - 1 : ix);
  const T *row = reinterpret_cast<const T *>(p + (size_t)iy * row_pitch);
  if (C.in_comps != 3) return row[ix];
  const int center = sizeof(T) == 2 ? 2048 : 128;
  const T *px = row + (size_t)ix * C.px_size;
  int r = px[C.off_r], g = px[C.off_g], b = px[C.off_b];
  if (C.no_ycc) return comp == 0 ? r : comp == 1 ? g : b;
  if (sizeof(T) == 2) { r &= 0xFFF; g &= 0xFFF; b &= 0xFFF; }
  if (comp == 0) return (FIXC(0.29900) * r + FIXC(0.58700) * g + FIXC(0.11400) * b + 32768) >> 16;
  if (comp == 1) return (-FIXC(0.16874) * r - FIXC(0.33126) * g + FIXC(0.50000) * b + (center << 16) + 32767) >> 16;
  return (FIXC(0.50000) * r - FIXC(0.41869) * g - FIXC(0.08131) * b + (center << 16) + 32767) >> 16;
}

template <class T>
__global__ void __launch_bounds__(256)
k_color_smooth(MjhConst C, const uint8_t *__restrict__ pix, size_t row_pitch, size_t img_stride, T *__restrict__ planes)
{
  const int comp = blockIdx.z % C.ncomp, img = blockIdx.z / C.ncomp;
  const MjhComp cc = C.c[comp];
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (r >= cc.ph || c >= cc.pw) return;
  const uint8_t *p = pix + (size_t)img * img_stride;
  const long long sf = C.smoothing;
  long long val;
#define SPX(yy, xx) ((long long)smooth_px<T>(C, p, row_pitch, comp, (yy), (xx)))
  if (cc.hexp == 1 && cc.vexp == 1) {
    const long long member = SPX(r, c);
    long long neigh = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++)
        if (dy || dx) neigh += SPX(r + dy, c + dx);
    val = (member * (65536ll - sf * 512ll) + neigh * (sf * 64ll) + 32768ll) >> 16;
  } else if (cc.hexp == 2 && cc.vexp == 2) {
    const int y0 = 2 * r, x0 = 2 * c;
    const long long member = SPX(y0, x0) + SPX(y0, x0 + 1) + SPX(y0 + 1, x0) + SPX(y0 + 1, x0 + 1);
    const long long edge = SPX(y0 - 1, x0) + SPX(y0 - 1, x0 + 1) + SPX(y0 + 2, x0) + SPX(y0 + 2, x0 + 1) +
                           SPX(y0, x0 - 1) + SPX(y0, x0 + 2) + SPX(y0 + 1, x0 - 1) + SPX(y0 + 1, x0 + 2);
    const long long corner = SPX(y0 - 1, x0 - 1) + SPX(y0 - 1, x0 + 2) + SPX(y0 + 2, x0 - 1) + SPX(y0 + 2, x0 + 2);
    val = (member * (16384ll - sf * 80ll) + (2 * edge + corner) * (sf * 16ll) + 32768ll) >> 16;
  } else {
    long long sum = 0;
    for (int vv = 0; vv < cc.vexp; vv++)
      for (int hh = 0; hh < cc.hexp; hh++) sum += SPX(r * cc.vexp + vv, c * cc.hexp + hh);
    const int n = cc.hexp * cc.vexp;
    if (cc.hexp == 2 && cc.vexp == 1) val = (sum + (c & 1)) >> 1;
    else val = (sum + n / 2) / n;
  }
#undef SPX
  planes[(size_t)img * C.planes_per_image + cc.plane_off + (size_t)r * cc.pw + c] = (T)val;
}

// Any legal set of sampling factors (cjpeg -sample HxV,HxV,HxV with chroma other than 1x1, or luma that is not the largest
// component): one lane = one output sample of one component, its hexp x vexp box converted pixel by pixel.  Row r of a
// component comes from input rows (r / v) * maxv + (r % v) * vexp ..., the rows behind the last real row group replicate the
// last DOWNSAMPLED row (jcprepct.c:180-190), input rows / columns beyond the image repeat the last one (jcprepct.c:161-168,
// jcsample.c:98-116); fullsize / h2v1 / h2v2 / int_downsample roundings (jcsample.c:199,226,263,151).  A rare configuration:
// clarity over speed (the common ones keep k_color / k_color_vec).
template <class T>
__global__ void __launch_bounds__(256)
k_color_generic(MjhConst C, const uint8_t *__restrict__ pix, size_t row_pitch, size_t img_stride, T *__restrict__ planes)
{
  const int comp = blockIdx.z % C.ncomp, img = blockIdx.z / C.ncomp;
  const MjhComp cc = C.c[comp];
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (r >= cc.ph || c >= cc.pw) return;
  const uint8_t *p = pix + (size_t)img * img_stride;
  const int real_rows = C.real_groups_y * cc.v;
  const int rr = r < real_rows ? r : real_rows - 1;
  const int in_row0 = (rr / cc.v) * C.maxv + (rr % cc.v) * cc.vexp;
  int sum = 0;
  for (int vv = 0; vv < cc.vexp; vv++)
    for (int hh = 0; hh < cc.hexp; hh++) sum += smooth_px<T>(C, p, row_pitch, comp, in_row0 + vv, c * cc.hexp + hh);   // (clamps to the image)
  int val;
  if (cc.hexp == 1 && cc.vexp == 1) val = sum;
  else if (cc.hexp == 2 && cc.vexp == 1) val = (sum + (c & 1)) >> 1;
  else if (cc.hexp == 2 && cc.vexp == 2) val = (sum + 1 + (c & 1)) >> 2;
  else { const int n = cc.hexp * cc.vexp; val = (sum + n / 2) / n; }
  planes[(size_t)img * C.planes_per_image + cc.plane_off + (size_t)r * cc.pw + c] = (T)val;
}

// 8 pixels of 3 bytes (six dwords of one row) -> 8 luma bytes, and the 4 x 2 chroma values of the row added to the running sums.
// Round 5: the bytes are taken out of the dwords as they are used (a byte array in between made the compiler pack and unpack
// 16-bit halves: 33 lane-instructions per pixel), RGB / BGR is a choice of COEFFICIENTS (uniform scalars) instead of two selects per
// pixel, and the luma bytes -- byte 2 of the 24-bit sums -- are gathered by v_perm_b32.
struct MjhYccCoef { int y0, y1, y2, cb0, cb1, cb2, cr0, cr1, cr2; };    // per byte position of the pixel
__device__ __forceinline__ MjhYccCoef ycc_coef(bool bgr)
{
  MjhYccCoef k;
  k.y0 = bgr ? FIXC(0.11400) : FIXC(0.29900);   k.y1 = FIXC(0.58700);   k.y2 = bgr ? FIXC(0.29900) : FIXC(0.11400);
  k.cb0 = bgr ? FIXC(0.50000) : -FIXC(0.16874); k.cb1 = -FIXC(0.33126); k.cb2 = bgr ? -FIXC(0.16874) : FIXC(0.50000);
  k.cr0 = bgr ? -FIXC(0.08131) : FIXC(0.50000); k.cr1 = -FIXC(0.41869); k.cr2 = bgr ? FIXC(0.50000) : -FIXC(0.08131);
  return k;
}
__device__ __forceinline__ uint2 ycc_row8(const unsigned (&w)[6], const MjhYccCoef &k, int (&cbs)[4], int (&crs)[4])
{
  unsigned ys[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int a = (int)((w[(3 * j) >> 2] >> (8 * ((3 * j) & 3))) & 0xFFu);
    const int g = (int)((w[(3 * j + 1) >> 2] >> (8 * ((3 * j + 1) & 3))) & 0xFFu);
    const int c = (int)((w[(3 * j + 2) >> 2] >> (8 * ((3 * j + 2) & 3))) & 0xFFu);
    ys[j] = (unsigned)(mul24(a, k.y0) + mul24(g, k.y1) + mul24(c, k.y2) + 32768);                      // < 2^24: Y is byte 2
    cbs[j >> 1] += (mul24(a, k.cb0) + mul24(g, k.cb1) + mul24(c, k.cb2) + (128 << 16) + 32767) >> 16;
    crs[j >> 1] += (mul24(a, k.cr0) + mul24(g, k.cr1) + mul24(c, k.cr2) + (128 << 16) + 32767) >> 16;
  }
  uint2 o;
  o.x = __builtin_amdgcn_perm(__builtin_amdgcn_perm(ys[3], ys[2], 0x0c0c0602u), __builtin_amdgcn_perm(ys[1], ys[0], 0x0c0c0602u), 0x05040100u);
  o.y = __builtin_amdgcn_perm(__builtin_amdgcn_perm(ys[7], ys[6], 0x0c0c0602u), __builtin_amdgcn_perm(ys[5], ys[4], 0x0c0c0602u), 0x05040100u);
  return o;
}
// the six dwords of 8 pixels: three 8-byte loads inside the image, clamped byte loads for a run that touches the right edge
__device__ __forceinline__ void load_px8(const uint8_t *row, int x0, int W, bool interior, unsigned (&w)[6])
{
  if (interior) {
    const uint2 *rv = reinterpret_cast<const uint2 *>(row + (size_t)x0 * 3);
#pragma unroll
    for (int i = 0; i < 3; i++) { const uint2 v = rv[i]; w[2 * i] = v.x; w[2 * i + 1] = v.y; }
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int ix = x0 + j;
      if (ix > W - 1) ix = W - 1;
#pragma unroll
      for (int c = 0; c < 3; c++) w[(3 * j + c) >> 2] |= (unsigned)row[ix * 3 + c] << (8 * ((3 * j + c) & 3));
    }
  }
}
// h2v2 / h2v1 roundings of four consecutive chroma samples (the first one at an even column): bias 1,2,1,2 / 0,1,0,1
template <int V0>
__device__ __forceinline__ unsigned chroma4(const int (&sum)[4])
{
  unsigned o = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) o |= (unsigned)(V0 == 2 ? (sum[j] + 1 + (j & 1)) >> 2 : (sum[j] + (j & 1)) >> 1) << (8 * j);
  return o;
}

// Vectorised variant for the common case (8-bit, 3 bytes per pixel, 2:1 horizontal chroma
// subsampling, 8-byte aligned rows): one lane converts 8 pixels x V0 rows = 4 chroma samples.  The
// 24 bytes per row arrive as three 8-byte loads, Y leaves as one 8-byte store per row, Cb/Cr as one
// 4-byte store each; lanes whose 8 pixels touch the right image edge take the clamped byte path.
template <int V0>
__global__ void __launch_bounds__(256)
k_color_vec(MjhConst C, const uint8_t *__restrict__ pix, size_t row_pitch, size_t img_stride,
            uint8_t *__restrict__ planes)
{
  const int g4 = blockIdx.x * 256 + threadIdx.x;   // index of a run of 4 groups (8 pixels)
  const int gy = blockIdx.y;
  const int img = blockIdx.z;
  if (g4 * 4 >= C.groups_x) return;
  const uint8_t *p = pix + (size_t)img * img_stride;
  uint8_t *pl = planes + (size_t)img * C.planes_per_image;
  const bool below = gy >= C.real_groups_y;
  const int gys = below ? C.real_groups_y - 1 : gy;
  const int x0 = g4 * 8;
  const bool interior = x0 + 7 <= C.W - 1;
  const MjhYccCoef k = ycc_coef(C.off_r == 2);
  uint2 yv[V0];
  int cbs[4] = { 0, 0, 0, 0 }, crs[4] = { 0, 0, 0, 0 };
  unsigned w[V0][6];
  const uint8_t *rowp[V0];
#pragma unroll
  for (int vy = 0; vy < V0; vy++) {
    int iy = gys * V0 + vy;
    if (iy > C.H - 1) iy = C.H - 1;
    rowp[vy] = p + (size_t)iy * row_pitch;
  }
  if (interior) {          // every load of the lane in flight before the first use (the kernel runs at the HBM rate)
#pragma unroll
    for (int vy = 0; vy < V0; vy++) load_px8(rowp[vy], x0, C.W, true, w[vy]);
  } else {
#pragma unroll
    for (int vy = 0; vy < V0; vy++) load_px8(rowp[vy], x0, C.W, false, w[vy]);
  }
#pragma unroll
  for (int vy = 0; vy < V0; vy++) yv[vy] = ycc_row8(w[vy], k, cbs, crs);
  const MjhComp &c0 = C.c[0];
  if (x0 < c0.pw) {
#pragma unroll
    for (int vy = 0; vy < V0; vy++) {
      const int r = gy * V0 + vy;
      if (r < c0.ph) *reinterpret_cast<uint2 *>(pl + c0.plane_off + (size_t)r * c0.pw + x0) = below ? yv[V0 - 1] : yv[vy];
    }
  }
  const MjhComp &c1 = C.c[1];
  const MjhComp &c2 = C.c[2];
  if (gy < c1.ph && g4 * 4 < c1.pw) {
    *reinterpret_cast<unsigned *>(pl + c1.plane_off + (size_t)gy * c1.pw + g4 * 4) = chroma4<V0>(cbs);
    *reinterpret_cast<unsigned *>(pl + c2.plane_off + (size_t)gy * c2.pw + g4 * 4) = chroma4<V0>(crs);
  }
}

// =============================================================================================
// K1b  plane import: jpeg_write_raw_data jcapistd.c:145-199 hands caller-made component
// planes straight to the coefficient controller, which reads width_in_blocks*8 samples of
// height_in_blocks*8 rows (compress_first_pass jccoefct.c:262-353).  Callers with smaller planes
// replicate the last sample / row first (tj3CompressFromYUVPlanes8 turbojpeg.c:1295-1316): the two
// clamps.  One lane = 4 consecutive samples of one plane row.
// =============================================================================================
template <typename T>
__global__ void __launch_bounds__(256)
k_import_planes(MjhConst C, MjhPlaneSrc S, T *__restrict__ planes)
{
  const int comp = blockIdx.z % C.ncomp, img = blockIdx.z / C.ncomp;
  const MjhComp cc = C.c[comp];
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4, r = blockIdx.y;
  if (r >= cc.ph || c0 >= cc.pw) return;        // pw is a multiple of 8, so the 4 samples are all inside
  const int sr = min(r, S.h[comp] - 1);
  const T *src = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(S.base[comp]) + (size_t)img * S.stride[comp] +
                                            (size_t)sr * S.pitch[comp]);
  T v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) v[j] = src[min(c0 + j, S.w[comp] - 1)];
  T *dst = planes + (size_t)img * C.planes_per_image + cc.plane_off + (size_t)r * cc.pw + c0;
#pragma unroll
  for (int j = 0; j < 4; j++) dst[j] = v[j];
}
// =============================================================================================
// K2  convsamp + overshoot deringing + islow FDCT + quantize   (rows a4-a8)
//   convsamp jcdctmgr.c:576, preprocess_deringing :416-498 (catmull_rom :387), jpeg_fdct_islow
//   jfdctint.c:142-286, quantize jcdctmgr.c:611 (== sign(x)*((|x|+d/2)/d), d = 8q), post-clamp
//   :761-770, unquantized copy :729-756.
// One lane = one block, whole 8x8 in registers.  Only blocks that touch the maximum sample
// value take the (rare, divergent) deringing path, which walks the block in zig-zag order
// through a per-lane LDS column ([64][64] ints, conflict-free: bank = lane).
// =============================================================================================
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

template <int PASS, int P1>   // P1 = PASS1_BITS: 2 for 8-bit, 1 for 12-bit samples (jfdctint.c:80-86)
__device__ __forceinline__ void fdct8(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5, int &d6, int &d7)
{
  const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
  const int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  constexpr int SH = PASS == 0 ? 13 - P1 : 13 + P1;
  if (PASS == 0) { d0 = (t10 + t11) * (1 << P1); d4 = (t10 - t11) * (1 << P1); }
  else { d0 = DESCALE(t10 + t11, P1); d4 = DESCALE(t10 - t11, P1); }
  // (every operand stays below 2^19 even for 12-bit samples in the second pass: 24-bit multiplies are exact and full rate)
  int z1 = mul24(t12 + t13, 4433);
  d2 = DESCALE(z1 + mul24(t13, 6270), SH);
  d6 = DESCALE(z1 + mul24(t12, -15137), SH);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = mul24(z3 + z4, 9633);
  const int a4 = mul24(t4, 2446), a5 = mul24(t5, 16819), a6 = mul24(t6, 25172), a7 = mul24(t7, 12299);
  z1 = mul24(z1, -7373); z2 = mul24(z2, -20995); z3 = mul24(z3, -16069); z4 = mul24(z4, -3196);
  z3 += z5; z4 += z5;
  d7 = DESCALE(a4 + z1 + z3, SH);
  d5 = DESCALE(a5 + z2 + z4, SH);
  d3 = DESCALE(a6 + z2 + z3, SH);
  d1 = DESCALE(a7 + z1 + z4, SH);
}

// jpeg_fdct_ifast jfdctfst.c:117-227: the AA&N butterflies, both passes alike; five multiplies by constants of 8 fractional bits,
// the products shifted down without rounding (DESCALE is a plain arithmetic shift there, :101-104).  DCTELEM is an int in the C
// build: nothing is narrowed between the steps.  |operand| < 2^15 and constants < 2^9: 24-bit multiplies are exact.
__device__ __forceinline__ void fdct8_ifast(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5, int &d6, int &d7)
{
  const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
  const int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  const int e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
  d0 = e0 + e1;
  d4 = e0 - e1;
  const int z1 = mul24(e2 + e3, 181) >> 8;
  d2 = e3 + z1;
  d6 = e3 - z1;
  const int o0 = t4 + t5, o1 = t5 + t6, o2 = t6 + t7;
  const int z5 = mul24(o0 - o2, 98) >> 8;
  const int z2 = (mul24(o0, 139) >> 8) + z5;
  const int z4 = (mul24(o2, 334) >> 8) + z5;
  const int z3 = mul24(o1, 181) >> 8;
  const int z11 = t7 + z3, z13 = t7 - z3;
  d5 = z13 + z2;
  d3 = z13 - z2;
  d1 = z11 + z4;
  d7 = z11 - z4;
}
// scalefactor[row] * scalefactor[col] * 2^14 in natural order (jcdctmgr.c:302-312 and again :733-743)
struct AanTab { int v[64]; };
static constexpr AanTab kAan = { {
  16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
  21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906, 19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
  16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 12873, 17855, 16819, 15137, 12873, 10114, 6967, 3552,
  8867, 12299, 11585, 10426, 8867, 6967, 4799, 2446, 4520, 6270, 5906, 5315, 4520, 3552, 2446, 1247 } };
// what the trellis gets to see of an AA&N coefficient: the scale factor taken out again (forward_DCT jcdctmgr.c:745-750; C division,
// towards zero; stored as a JCOEF).  N: natural index, a compile-time constant -- the division becomes a multiply-high.
template <int N>
__device__ __forceinline__ int aan_unscale(int x)
{
  constexpr int sc = kAan.v[N];
  return (int)(short)(x >= 0 ? (x * 32768 + sc) / (2 * sc) : (x * 32768 - sc) / (2 * sc));
}

template <int N = 0>
__device__ __forceinline__ void aan_unscale_all(const int (&d)[64], int (&du)[64])
{
  du[N] = aan_unscale<N>(d[N]);
  if constexpr (N < 63) aan_unscale_all<N + 1>(d, du);
}
__device__ __forceinline__ void aan_unscale_all(const int (&)[64], int (&)[1]) { }

__device__ __forceinline__ float catmull_rom(int v1, int v2, int v3, int v4, float t, int size)
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

// Fused extras (sequential mode with trellis quantization, the metric's configuration):
//  * STATS: the AC symbol statistics of the conventionally quantized block (passes 0,2,4 of SURVEY 3.3,
//    htest_one_block jchuff.c:812-915) are gathered here, while the values are in registers in zig-zag order, instead
//    of by a k_stats_ac pass that re-reads all 63 planes;
//    and, because the AC trellis recomputes every AC coefficient from coef_uq, the 63 quantized AC planes this kernel
//    would write are then never read: only plane 0 (DC statistics / DC trellis) is stored.
// FD: every quantizer step of the tables in use fits 8 bits -- the division by 8q is one shift + one 24-bit multiply-high
// (MjhQuant.mdiv / sdiv) instead of the float-reciprocal division with its integer fix-up; with STATS the AC coefficients are
// quantized for the statistics only, which need the magnitude category and nothing else (no sign, no signed clamp).
#define DCTQ_NB 4      // sets of 64 blocks per wave of the FDCT kernel with fused statistics (the others: one set -- a loop only cost them: 12-bit C5 477 -> 532 us)
// IFAST: dct_method JDCT_IFAST -- the AA&N transform; the host put its divisors (quantval x scale factors, jcdctmgr.c:291-345) where
// the conventional quantizer reads them (MjhQuant.dqc8 / rcpc8q), and the trellis' copy of the coefficients is unscaled again.
template <class T, bool STATS, bool FD, bool IFAST = false>   // uint8_t: 8-bit samples; uint16_t: 12-bit samples (no trellis: coef_uq / lambda are not produced)
__device__ __forceinline__ void dct_quant_body(const MjhConst &C, const MjhQuant *__restrict__ Q, const T *__restrict__ planes,
                                               int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out,
                                               MjhHuffTable *__restrict__ stat_tabs, int slots_per_image, int4 stat_slot_of_comp, uint8_t *__restrict__ nq8_out)
{
  constexpr bool W12 = sizeof(T) == 2;
  // A wave takes DCTQ_NB consecutive sets of 64 blocks (round 5; one set per wave before): the statistics histogram is zeroed and
  // flushed once per wave instead of once per set, and a quarter of the workgroups are launched (1.12 -> 1.07 ms per 64 frames;
  // 8 sets: the same; prefetching the next set's pixel rows into registers: 1.06 at 211 VGPRs = two waves per SIMD, 1.12 when held
  // to three waves with spills -- not kept; gpurun_out/r5h, profiles/r05h_fdct_variants.md).
  // LDS per wave: 8 KB of deringing columns (the ORIGINAL level-shifted samples of the lane's block in zig-zag order, 16 bits
  // each for 8- and 12-bit data) + 4 KB = 4 interleaved copies of the 256-bin statistics histogram (12 KB: 13 waves per CU).
  constexpr int LW = 32;
  __shared__ int lds_raw[64][LW];
  constexpr int NCOPY = 4;
  __shared__ unsigned hist_raw[STATS ? NCOPY * 256 : 1];
  typedef short dcol_t;
  typedef dcol_t __attribute__((may_alias)) dcol_alias;
  const int lane = (int)threadIdx.x;      // one wave per workgroup
  dcol_alias (*lds)[64] = reinterpret_cast<dcol_alias (*)[64]>(&lds_raw[0][0]);
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  constexpr int NB = STATS ? DCTQ_NB : 1;
  const int set0 = blockIdx.x * NB;
  if (set0 * 64 >= cc.nblk) return;       // whole workgroup outside (grid is sized for the largest component)
  constexpr bool stats = STATS;
  unsigned *hist = hist_raw;
  if (stats) {
#pragma unroll
    for (int j = 0; j < NCOPY * 4; j++) hist[j * 64 + lane] = 0u;
    __syncthreads();
  }
  // copy c of bin b lives at word b * NCOPY + c: the lanes that count the same symbol in different copies hit different banks
  unsigned *hh = hist + (lane & (NCOPY - 1));
  typedef typename std::conditional<W12, uint4, uint2>::type row_t;
  auto load_rows = [&](int set, row_t (&rows)[8]) {
    const int braw = set * 64 + lane;
    const int b = braw < cc.nblk ? braw : cc.nblk - 1;
    const int r0 = b / cc.wib, c0 = b - r0 * cc.wib;
    const T *src = planes + (size_t)img * C.planes_per_image + cc.plane_off + (size_t)(r0 * 8) * cc.pw + c0 * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) rows[r] = *reinterpret_cast<const row_t *>(src + (size_t)r * cc.pw);
  };
  row_t rows_cur[8];
#pragma unroll 1
  for (int it = 0; it < NB; it++) {
  const int set = set0 + it;
  if (set * 64 >= cc.nblk) break;                        // uniform
  load_rows(set, rows_cur);
  const int blk_raw = set * 64 + lane;
  const bool valid = blk_raw < cc.nblk;                  // tail lanes redo the last block (identical stores), they only stay out of the statistics
  const int blk = valid ? blk_raw : cc.nblk - 1;
  int d[64];
  unsigned ff = 0;     // 8-bit samples: some byte of the block is 255 (the only value that reaches maxsample after the level shift)
#pragma unroll
  for (int r = 0; r < 8; r++) {
    if (!W12) {
      const uint2 v = *reinterpret_cast<const uint2 *>(&rows_cur[r]);
      ff |= ((~v.x - 0x01010101u) & v.x) | ((~v.y - 0x01010101u) & v.y);     // bit 7 of a byte set: that byte (or one above a 255) is 255
#pragma unroll
      for (int i = 0; i < 4; i++) {
        d[r * 8 + i] = (int)((v.x >> (8 * i)) & 0xFF) - 128;
        d[r * 8 + 4 + i] = (int)((v.y >> (8 * i)) & 0xFF) - 128;
      }
    } else {
      const uint4 v = *reinterpret_cast<const uint4 *>(&rows_cur[r]);
      const unsigned w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
      for (int i = 0; i < 4; i++) {
        d[r * 8 + 2 * i] = (int)(w4[i] & 0xFFFF) - 2048;
        d[r * 8 + 2 * i + 1] = (int)(w4[i] >> 16) - 2048;
      }
    }
  }
  const uint16_t *qz = Q->q[cc.qtbl];
  if (C.deringing) {
    const int maxsample = 127;
    int sum = 0, cnt = 0;
    if (W12 || (ff & 0x80808080u) != 0u) {   // (most 8-bit blocks have no saturated sample: the word test above spares them 64 compares)
#pragma unroll
      for (int i = 0; i < 64; i++) { sum += d[i]; cnt += (d[i] >= maxsample); }
    }
    if (cnt != 0 && cnt != 64) {
      // preprocess_deringing jcdctmgr.c:416-498, restated for lanes in lock step: ONE pass over the 64 zig-zag positions with
      // the block in registers (the zig-zag order is a compile-time permutation of them).  A run of saturated samples needs
      //  * the two samples in front of it as they are AFTER earlier runs were rewritten (f1, f2): the last two final values,
      //    kept in registers as the pass goes;
      //  * its length and the two ORIGINAL samples behind it (l1, l2): the length is a count of consecutive ones in the lane's
      //    64-bit saturation mask, the samples are the only data-dependent reads -- they come from the lane's LDS column of
      //    originals, once per run;
      //  * the spline position, accumulated step by step in float exactly as the reference's loop does.
      // (12-bit data, where maxsample stays 127 and nearly every block takes this path, walked its runs with LDS reads and
      // lane-private loops before: 2.0 ms per 8192x8192 frame, 7 times the 8-bit kernel's time per block.)
      unsigned mlo = 0u, mhi = 0u;
#pragma unroll
      for (int i = 0; i < 64; i++) {
        lds[i][lane] = (dcol_t)d[kZZ.v[i]];
        if (i < 32) mlo |= (unsigned)(d[kZZ.v[i]] >= maxsample) << i;
        else mhi |= (unsigned)(d[kZZ.v[i]] >= maxsample) << (i - 32);
      }
      const unsigned long long sat = ((unsigned long long)mhi << 32) | mlo;
      const int q0 = qz[0];
      const int a = min(31, 2 * q0);
      const int b = (maxsample * 64 - sum) / cnt;   // C division truncates toward zero: negative for 12-bit data (T9)
      const int maxovershoot = maxsample + min(a, b);
      int prev1 = 0, prev2 = 0, fslope = 0, lslope = 0, length = 0;
      float step = 0.f, position = 0.f;
#pragma unroll
      for (int i = 0; i < 64; i++) {
        int v = d[kZZ.v[i]];
        if ((sat >> i) & 1ull) {
          if (i == 0 || !((sat >> (i - 1)) & 1ull)) {          // first sample of a run
            length = __builtin_ctzll(~(sat >> i));                // (zeros shifted in at the top end the count at position 64)
            const int end = i + length;
            const int f1 = i >= 1 ? prev1 : v;
            const int f2 = i >= 2 ? prev2 : i == 1 ? prev1 : v;
            const int l1 = lds[end < 63 ? end : 63][lane];
            const int l2 = lds[end < 62 ? end + 1 : 63][lane];
            fslope = max(f1 - f2, maxsample - f1);
            lslope = max(l1 - l2, maxsample - l1);
            if (i == 0) fslope = lslope;
            if (end == 64) lslope = fslope;
            step = 1.f / (float)(length + 1);
            position = step;
          }
          const int tmp = (int)ceilf(catmull_rom(maxsample - fslope, maxsample, maxsample, maxsample - lslope, position, length));
          v = min(tmp, maxovershoot);
          position += step;
          d[kZZ.v[i]] = v;
        }
        prev2 = prev1; prev1 = v;
      }
    }
  }
  if (IFAST) {
#pragma unroll
    for (int r = 0; r < 8; r++) fdct8_ifast(d[r * 8], d[r * 8 + 1], d[r * 8 + 2], d[r * 8 + 3], d[r * 8 + 4], d[r * 8 + 5], d[r * 8 + 6], d[r * 8 + 7]);
#pragma unroll
    for (int c = 0; c < 8; c++) fdct8_ifast(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
  } else {
#pragma unroll
  for (int r = 0; r < 8; r++)
    fdct8<0, W12 ? 1 : 2>(d[r * 8], d[r * 8 + 1], d[r * 8 + 2], d[r * 8 + 3], d[r * 8 + 4], d[r * 8 + 5], d[r * 8 + 6], d[r * 8 + 7]);
#pragma unroll
  for (int c = 0; c < 8; c++)
    fdct8<1, W12 ? 1 : 2>(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
  }
  // IFAST with the trellis: the raw coefficients as the trellis reads them (natural order; the quantizer below keeps the scaled ones)
  int du[IFAST ? 64 : 1];
  if (IFAST && !W12) aan_unscale_all(d, du);      // (12-bit samples have no trellis: nothing reads the copy)

  float lambda_blk = 0.0f;
  if (!W12 && C.trellis) {
    // per-block trellis lambda (jcdctmgr.c:1027-1037): norm of the 63 AC coefficients summed in
    // NATURAL index order in float, /63 in double, then lambda in double -> float.  pow(2, .) comes
    // from the host libm.  Both trellis kernels consume it.
    float norm = 0.0f;
#pragma unroll
    for (int n = 1; n < 64; n++) { const int rc = IFAST ? du[IFAST ? n : 0] : d[n]; norm = norm + squaref(rc); }   // |raw coefficient| <= 2^15: (float)rc * (float)rc == (float)(rc * rc)
    norm = (float)((double)norm / 63.0);
    float lambda;
    if (C.lambda_log_scale2 > 0.0f) lambda = (float)(C.pow_scale1 * 1.0 / (C.pow_scale2 + (double)norm));
    else lambda = (float)(C.pow_scale1 * 1.0);
    lambda_out[(size_t)img * C.total_real_blocks + cc.blk_off + blk] = lambda;
    lambda_blk = lambda;
  }
  int16_t *uq = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
  int16_t *qo = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
  const bool clampq = C.deringing != 0;
  int run = 0, nzc = 0;   // nzc: non-zero quantized AC coefficients = the AC trellis' queue length (its tile-sort key)
  // The quantizer's constants are wave-uniform (scalar loads).  Fetched where they are used -- inside the per-position branches --
  // every position paid two scalar-memory round trips (s_waitcnt lgkmcnt(0) on 8q, then on the divider pair) that two waves per
  // SIMD cannot hide; they are fetched for QCH positions at a time, unconditionally: one round trip per QCH positions.
  constexpr int QCH = 8;
  int dq_c[QCH], sdiv_c[QCH];
  unsigned mdiv_c[QCH];
  float rcp_c[QCH], thr_c[QCH];
#pragma unroll
  for (int k = 0; k < 64; k++) {
    if ((k % QCH) == 0) {
#pragma unroll
      for (int j = 0; j < QCH; j++) {
        // (FD: every step <= 255, nothing wraps; otherwise the conventional quantizer's own divisor -- MjhQuant.dqc8)
        if (FD) { dq_c[j] = Q->dq8[cc.qtbl][k + j]; sdiv_c[j] = Q->sdiv[cc.qtbl][k + j]; mdiv_c[j] = Q->mdiv[cc.qtbl][k + j]; if (STATS) thr_c[j] = Q->thr8[cc.qtbl][k + j]; }
        else { dq_c[j] = Q->dqc8[cc.qtbl][k + j]; rcp_c[j] = Q->rcpc8q[cc.qtbl][k + j]; }
      }
    }
    const int x = d[kZZ.v[k]];
    const int dq = dq_c[k % QCH];
    if (FD && STATS && k > 0) {
      // statistics of the conventionally quantized block only (the trellis recomputes the values): magnitude category of
      // min(floor((|x| + 4q) / 8q), 1023) -- the signed clamp to +-1023 (jcdctmgr.c:761-770) leaves the category of 1023.
      // "Quantizes to non-zero" (|x| + 4q >= 8q) is one conversion + one compare of |(float)x| with MjhQuant.thr8 (exact: both
      // are integers below 2^24); |x| as an integer only where the division happens
      uq[(size_t)k * cc.kstride] = (int16_t)x;
      if (valid) {
        const float xf = (float)x;
        if (__builtin_fabsf(xf) >= thr_c[k % QCH]) {
          const int ax = (int)__builtin_fabsf(xf);
          int qa = udiv_mh(ax + (dq >> 1), sdiv_c[k % QCH], mdiv_c[k % QCH]);
          if (clampq) qa = min(qa, 1023);
          nzc++;
          if (run > 15) { atomicAdd(&hh[0xF0 * NCOPY], (unsigned)(run >> 4)); run &= 15; }
          atomicAdd(&hh[((run << 4) + bitlen((unsigned)qa)) * NCOPY], 1u);
          run = 0;
        } else run++;
      }
      continue;
    }
    const int ax = x < 0 ? -x : x;
    int v = FD ? udiv_mh(ax + (dq >> 1), sdiv_c[k % QCH], mdiv_c[k % QCH]) : udiv_exact(ax + (dq >> 1), dq, rcp_c[k % QCH]);
    if (x < 0) v = -v;
    if (clampq) v = W12 ? max(-16383, min(16383, v)) : max(-1023, min(1023, v));
    if (!W12) uq[(size_t)k * cc.kstride] = (int16_t)(IFAST ? du[IFAST ? kZZ.v[k] : 0] : x);   // raw x8 coefficients only feed the (8-bit only) trellis
    if (k == 0 || !STATS) qo[(size_t)k * cc.kstride] = (int16_t)v;   // STATS: the AC planes would never be read
    if (!stats && !W12 && k > 0) nzc += (v != 0);
    if (stats && k > 0 && valid) {
      if (v == 0) run++;
      else {
        nzc++;
        if (run > 15) { atomicAdd(&hh[0xF0 * NCOPY], (unsigned)(run >> 4)); run &= 15; }
        const int nb = bitlen((unsigned)(v < 0 ? -v : v));
        atomicAdd(&hh[((run << 4) + nb) * NCOPY], 1u);
        run = 0;
      }
    }
  }
  if (!W12 && nq8_out && valid) nq8_out[(size_t)img * C.total_real_blocks + cc.blk_off + blk] = (uint8_t)nzc;
  if (stats && valid && run > 0) atomicAdd(&hh[0], 1u);
  }   // (sets of this wave)
  if (stats) {
    __syncthreads();
    const int slot = comp == 0 ? stat_slot_of_comp.x : comp == 1 ? stat_slot_of_comp.y : comp == 2 ? stat_slot_of_comp.z : stat_slot_of_comp.w;
    MjhHuffTable *T2 = stat_tabs + (size_t)img * slots_per_image + slot;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int bin = lane + 64 * j;
      unsigned sum = 0;
#pragma unroll
      for (int c2 = 0; c2 < NCOPY; c2++) sum += hist[bin * NCOPY + c2];
      if (sum) atomicAdd(&T2->counts[bin], sum);
    }
  }
}

template <class T, bool STATS, bool FD = false>
__global__ void __launch_bounds__(64)
k_dct_quant(MjhConst C, const MjhQuant *__restrict__ Q, const T *__restrict__ planes,
            int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out,
            MjhHuffTable *__restrict__ stat_tabs, int slots_per_image, int4 stat_slot_of_comp, uint8_t *__restrict__ nq8_out)
{
  dct_quant_body<T, STATS, FD>(C, Q, planes, coef_uq, coef_q, lambda_out, stat_tabs, slots_per_image, stat_slot_of_comp, nq8_out);
}

// JDCT_IFAST (legacy TurboJPEG calls below quality 96, `cjpeg -dct fast`): its own kernel name, the body above
__global__ void __launch_bounds__(64)
k_dct_quant_ifast(MjhConst C, const MjhQuant *__restrict__ Q, const uint8_t *__restrict__ planes,
                  int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out, uint8_t *__restrict__ nq8_out)
{
  dct_quant_body<uint8_t, false, false, true>(C, Q, planes, coef_uq, coef_q, lambda_out, nullptr, 0, make_int4(0, 0, 0, 0), nq8_out);
}

__global__ void __launch_bounds__(64)
k_dct_quant_ifast12(MjhConst C, const MjhQuant *__restrict__ Q, const uint16_t *__restrict__ planes,
                    int16_t *__restrict__ coef_uq, int16_t *__restrict__ coef_q, float *__restrict__ lambda_out, uint8_t *__restrict__ nq8_out)
{
  dct_quant_body<uint16_t, false, false, true>(C, Q, planes, coef_uq, coef_q, lambda_out, nullptr, 0, make_int4(0, 0, 0, 0), nq8_out);
}

// =============================================================================================
// K2b  coefficient import: jpeg_write_coefficients jctrans.c:44 entropy-codes blocks the
// caller already has (jpegtran, "jpegrescan").  The caller's arrays are block-major, natural order
// (JBLOCKARRAY); the pipeline's layout is coefficient-major in zig-zag order.  One lane = one block: 32
// dword loads of its 128 contiguous bytes, 64 coalesced plane stores.  Dummy blocks stay virtual
// (compress_output jctrans.c:322-373 builds them with the same rule as the pixel path).
// =============================================================================================
__global__ void __launch_bounds__(64)
k_import_coefs(MjhConst C, MjhCoefSrc S, int16_t *__restrict__ coef_q, MjhImageMeta *__restrict__ meta)
{
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int blk = blockIdx.x * 64 + threadIdx.x;
  if (blk >= cc.nblk) return;
  const int br = blk / cc.wib, bc = blk - br * cc.wib;
  const unsigned *src = reinterpret_cast<const unsigned *>(reinterpret_cast<const uint8_t *>(S.base[comp]) + (size_t)img * S.stride[comp] +
                                                          ((size_t)br * S.blocks_per_row[comp] + bc) * 128);
  int v[64];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const unsigned w = src[i];
    v[2 * i] = (int)(short)(w & 0xFFFFu);
    v[2 * i + 1] = (int)(short)(w >> 16);
  }
  int16_t *qo = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + blk;
  // untrusted input (jpegtran on arbitrary files): a coefficient beyond MAX_COEF_BITS has no Huffman symbol; the reference
  // raises JERR_BAD_DCT_COEF when it meets one (jchuff.c:596,624) -- flagged here, the DC difference in k_enc_len
  const int lim = C.precision == 12 ? 16383 : 1023;
  int worst = 0;
#pragma unroll
  for (int k = 1; k < 64; k++) { const int a = v[k] < 0 ? -v[k] : v[k]; worst = a > worst ? a : worst; }
  if (worst > lim) meta[img].bad_coef = 1u;
#pragma unroll
  for (int k = 0; k < 64; k++) qo[(size_t)k * cc.kstride] = (int16_t)v[kZZ.v[k]];
}
// =============================================================================================
// Dummy-block resolution (compress_first_pass jccoefct.c:312-345, compress_trellis_pass
// :443-476): dummy blocks are never stored.  A padded position (r,c) of a component maps to
// the real block whose DC it copies; its AC coefficients are zero.
// =============================================================================================
// =============================================================================================
// K3  symbol statistics (row a10): htest_one_block / encode_mcu_gather jchuff.c:812-915.
// AC symbols of a block do not depend on scan order, DC symbols do, so they are gathered by
// two kernels:
//   k_stats_ac : one lane per real block, LDS histogram (4 interleaved copies), one global
//                atomic per used symbol and workgroup.
//   k_stats_dc : one lane per block in scan order (component raster order for the
//                per-component passes, interleaved MCU order incl. dummy blocks for the final
//                scan); per-wave ballot counting, no LDS atomics.
// =============================================================================================
// Compact coefficient records (written by the COMPACT trellis kernels): `mask` = the block's non-zero positions, plane i+1 =
// its i-th non-zero value in position order.  f(position, value) is called for every non-zero coefficient in
// position order; values arrive in bursts of 8 plane loads, and a burst is skipped once no block of the wave has a
// value left for it.
template <class F>
__device__ __forceinline__ void for_each_nonzero(const int16_t *__restrict__ qb, size_t kstride, unsigned long long mask, bool active, F &&f)
{
  // (measured: this rolled form beats issuing every load of the wave up front -- the consumers are bound by their
  // per-coefficient work (LDS histogram atomics, table lookups, bit writer), not by the load latency)
  const int n = active ? __popcll(mask) : 0;
#pragma unroll 1
  for (int base = 0; base < 63; base += 8) {
    if (__builtin_amdgcn_ballot_w64(base < n) == 0ull) break;
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (base + j < n) ? (int)qb[(size_t)(base + j + 1) * kstride] : 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (base + j < n) {
        const int pos = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        f(pos, v[j]);
      }
  }
}

// k_stats_ac on compact records (the final statistics of the sequential scan)
#define STATS_AC_ITER 4
__global__ void __launch_bounds__(256)
k_stats_ac_compact(MjhConst C, const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, MjhHuffTable *__restrict__ tabs,
                   int slots_per_image, int4 slot_of_comp, int count_dummies)
{
  __shared__ unsigned h[256][16];   // 16 copies of every bin side by side (copy c of bin b in bank (16 b + c) mod 32): the common symbols would otherwise serialise the LDS atomics, and copy-major storage would put all copies of a bin in one bank
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int tid = threadIdx.x;
  // STATS_AC_ITER rounds of 256 blocks per workgroup: zeroing and reducing the 16 KB of histograms costs as much as
  // counting one round
  for (int i = tid; i < 4096; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < STATS_AC_ITER; it++) {
    const int blk = (blockIdx.x * STATS_AC_ITER + it) * 256 + tid;
    if ((blockIdx.x * STATS_AC_ITER + it) * 256 >= cc.nblk) break;   // uniform
    const bool in = blk < cc.nblk;
    const int b = blk < cc.nblk ? blk : cc.nblk - 1;
    const int16_t *q = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + b;
    const unsigned long long m = in ? nzmask[(size_t)img * C.total_real_blocks + cc.blk_off + b] : 0ull;
    unsigned *hh = &h[0][tid & 15];
    int prev = 0;
    for_each_nonzero(q, (size_t)cc.kstride, m, in, [&](int pos, int v) {
      int r = pos - prev - 1;
      prev = pos;
      if (r > 15) { atomicAdd(&hh[0xF0 * 16], (unsigned)(r >> 4)); r &= 15; }
      atomicAdd(&hh[((r << 4) + bitlen((unsigned)(v < 0 ? -v : v))) * 16], 1u);
    });
    if (in && prev < 63) atomicAdd(&hh[0], 1u);
  }
  __syncthreads();
  const int slot = comp == 0 ? slot_of_comp.x : comp == 1 ? slot_of_comp.y : comp == 2 ? slot_of_comp.z : slot_of_comp.w;
  MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  unsigned s = 0;
#pragma unroll
  for (int c2 = 0; c2 < 16; c2++) s += h[tid][c2];
  if (count_dummies && tid == 0 && blockIdx.x == 0)
    s += (unsigned)(cc.wpad * cc.hpad - cc.nblk);  // every dummy block codes one EOB (all-zero AC)
  if (s) atomicAdd(&T->counts[tid], s);
}

__global__ void __launch_bounds__(256)
k_stats_ac(MjhConst C, const int16_t *__restrict__ coef_q, MjhHuffTable *__restrict__ tabs,
           int slots_per_image, int4 slot_of_comp, int count_dummies)
{
  __shared__ unsigned h[4][256];
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const int blk = blockIdx.x * 256 + tid;
  {
    // all 63 plane loads in one burst (unconditional, index clamped for the tail lanes): a load
    // inside the data-dependent run-length loop would cost one memory latency per coefficient
    const int16_t *q = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off + (blk < cc.nblk ? blk : cc.nblk - 1);
    int x[64];
#pragma unroll
    for (int k = 1; k < 64; k++) x[k] = q[(size_t)k * cc.kstride];
    if (blk < cc.nblk) {
      unsigned *hh = h[tid & 3];
      int r = 0;
#pragma unroll
      for (int k = 1; k < 64; k++) {
        const int v = x[k];
        if (v == 0) r++;
        else {
          if (r > 15) { atomicAdd(&hh[0xF0], (unsigned)(r >> 4)); r &= 15; }
          const int nb = bitlen((unsigned)(v < 0 ? -v : v));
          atomicAdd(&hh[(r << 4) + nb], 1u);
          r = 0;
        }
      }
      if (r > 0) atomicAdd(&hh[0], 1u);
    }
  }
  __syncthreads();
  const int slot = comp == 0 ? slot_of_comp.x : comp == 1 ? slot_of_comp.y : comp == 2 ? slot_of_comp.z : slot_of_comp.w;
  MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  unsigned s = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
  if (count_dummies && tid == 0 && blockIdx.x == 0)
    s += (unsigned)(cc.wpad * cc.hpad - cc.nblk);  // every dummy block codes one EOB (all-zero AC)
  if (s) atomicAdd(&T->counts[tid], s);
}

#define STATS_DC_ITER 16
// component raster order (the per-component passes): one lane per block
__global__ void __launch_bounds__(256)
k_stats_dc(MjhConst C, const int16_t *__restrict__ coef_q, MjhHuffTable *__restrict__ tabs,
           int slots_per_image, int4 slot_of_comp, int4 comp_restart)
{
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int16_t *q0 = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;  // plane k = 0
  const int lane = threadIdx.x & 63;
  const int nitems = cc.nblk;
  const int ri = comp == 0 ? comp_restart.x : comp == 1 ? comp_restart.y : comp == 2 ? comp_restart.z : comp_restart.w;
  unsigned cnt = 0;   // lane s (< 16) counts symbol s for this wave
  for (int it = 0; it < STATS_DC_ITER; it++) {
    const int t = (blockIdx.x * STATS_DC_ITER + it) * 256 + threadIdx.x;
    if ((int)(blockIdx.x * STATS_DC_ITER + it) * 256 >= nitems) break;   // uniform
    int nb = -1;
    if (t < nitems) {
      const int dc = q0[t];
      const int pred = (t == 0 || (ri && (t % ri) == 0)) ? 0 : q0[t - 1];
      const int df = dc - pred;
      nb = bitlen((unsigned)(df < 0 ? -df : df));
    }
#pragma unroll
    for (int s = 0; s < 16; s++) {   // DC categories 0..11 for 8-bit, up to 15 for 12-bit samples
      const unsigned long long m = __ballot(nb == s);
      if (lane == s) cnt += (unsigned)__popcll(m);
    }
  }
  const int slot = comp == 0 ? slot_of_comp.x : comp == 1 ? slot_of_comp.y : comp == 2 ? slot_of_comp.z : slot_of_comp.w;
  MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  if (lane < 16 && cnt) atomicAdd(&T->counts[lane], cnt);
}

// interleaved MCU order incl. dummy blocks (the final scan; encode_mcu_gather jchuff.c:866-915 walks the MCUs, the blocks of
// a component inside an MCU row by row): ONE LANE PER MCU of one component.  A workgroup takes `rows_per_wg` MCU rows, the threads
// stride the MCUs of a row, so no thread divides anything; the lane loads the last block of the MCU in front (the prediction of
// its first block) and its own h x v blocks -- every load is issued before the first use, the other predictions are the
// lane's previous value.  Counts per symbol are wave-uniform (ballot + s_bcnt1); the four waves meet in LDS, so a (component,
// image) table sees at most ~128 atomics per symbol (one 8192 x 8192 frame with a workgroup per MCU row and an atomic per wave:
// 12 288 waves on 48 addresses = 119 us of serialised L2 atomics, profiles/r05m_c5_kernel_stats.csv).
__global__ void __launch_bounds__(256)
k_stats_dc_mcu(MjhConst C, const int16_t *__restrict__ coef_q, MjhHuffTable *__restrict__ tabs,
               int slots_per_image, int4 slot_of_comp, int rows_per_wg)
{
  __shared__ unsigned s_cnt[16];
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int16_t *q0 = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;  // plane k = 0
  const int lane = threadIdx.x & 63;
  const int h = cc.h, v = cc.v, wib = cc.wib, hib = cc.hib, mpr = C.mcus_per_row, ri = C.restart_interval;
  unsigned cnt[16];
#pragma unroll
  for (int s = 0; s < 16; s++) cnt[s] = 0;
  if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  for (int my = blockIdx.x * rows_per_wg; my < (int)(blockIdx.x + 1) * rows_per_wg && my < C.mcu_rows; my++)      // uniform
  for (int mx0 = 0; mx0 < mpr; mx0 += 256) {   // uniform
    const int mx = mx0 + (int)threadIdx.x;
    const bool in = mx < mpr;
    const int mxc = in ? mx : mpr - 1;
    // prediction of the MCU's first block: the last block (row v-1, column h-1) of the MCU in front
    const int m = my * mpr + mxc;
    bool has = m != 0;
    if (ri) has = has && (m % ri) != 0;
    int pmy = my, pmx = mxc - 1;
    if (pmx < 0) { pmx = mpr - 1; pmy = my - 1; }
    if (pmy < 0) { pmy = 0; pmx = 0; }
    int pr = pmy * v + v - 1, pc = pmx * h + h - 1;     // dc_source_block of it: a last column stays one under the row clamp
    if (pr >= hib) pr = hib - 1;
    if (pc > wib - 1) pc = wib - 1;
    int pred = q0[pr * wib + pc];
    if (!has) pred = 0;
#pragma unroll 1
    for (int yi = 0; yi < v; yi++) {
      int r = my * v + yi;
      const bool below = r >= hib;                      // dummy rows copy the DC of the MCU's last real column in the last row
      if (below) r = hib - 1;
#pragma unroll 1
      for (int xi0 = 0; xi0 < h; xi0 += 4) {
        int dcv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          int c = mxc * h + (below ? h - 1 : (xi0 + j < h ? xi0 + j : h - 1));
          if (c > wib - 1) c = wib - 1;
          dcv[j] = q0[r * wib + c];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (xi0 + j < h) {                            // uniform
            const int df = dcv[j] - pred;
            pred = dcv[j];
            const int nb = in ? bitlen((unsigned)(df < 0 ? -df : df)) : -1;
#pragma unroll
            for (int s = 0; s < 16; s++) cnt[s] += (unsigned)__popcll(__ballot(nb == s));   // DC categories 0..11 (8-bit), ..15 (12-bit)
          }
        }
      }
    }
  }
  const int slot = comp == 0 ? slot_of_comp.x : comp == 1 ? slot_of_comp.y : comp == 2 ? slot_of_comp.z : slot_of_comp.w;
  MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  unsigned mine = 0;
#pragma unroll
  for (int s = 0; s < 16; s++) mine = lane == s ? cnt[s] : mine;
  if (lane < 16 && mine) atomicAdd(&s_cnt[lane], mine);
  __syncthreads();
  if (threadIdx.x < 16 && s_cnt[threadIdx.x]) atomicAdd(&T->counts[threadIdx.x], s_cnt[threadIdx.x]);
}

// =============================================================================================
// K4  optimal Huffman table construction (row a11): jpeg_gen_optimal_table jchuff.c:947-1106,
// jpeg_make_c_derived_tbl :231-318.  ONE WAVE per table: the 257 symbols live 5 per lane.
//  * "two smallest, ties -> larger symbol index" (jchuff.c:990-1012) is a wave arg-min on the
//    key (freq, 511-symbol); the reference's chain walk over others[] is replaced by a group id
//    per symbol (every member of the two merged trees gets codesize+1), identical result.
//  * code-length limiting (K.2, :1073-1084) and pseudo-symbol removal are serial on lane 0.
//  * huffval order = (codesize, symbol) via per-length ballots (bit_pos, :1060-1064,:1099-1102).
// =============================================================================================
// minimum of a 64-bit key over the wave, the same value in every lane.  DPP row shifts (a running minimum over 1, 2, 4, 8
// lanes leaves each 16-lane row's minimum in its last lane) + four v_readlane: no LDS round trips -- the table
// construction below is one long dependent chain of these reductions (two per merge step), and with six ds_bpermute
// exchanges per reduction the two k_gen_tables launches cost 0.21 ms per batch although they do almost no work.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#define MJH_MIN_STEP(CTRL)                                                                                           \
  {                                                                                                                   \
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)v, CTRL, 0xF, 0xF, false);            \
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);     \
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;                                                  \
    v = w < v ? w : v;                                                                                                \
  }
  MJH_MIN_STEP(0x111) MJH_MIN_STEP(0x112) MJH_MIN_STEP(0x114) MJH_MIN_STEP(0x118)   // row_shr:1, 2, 4, 8 (lanes without a source read the identity)
#undef MJH_MIN_STEP
  unsigned long long m = ~0ull;
#pragma unroll
  for (int r = 15; r < 64; r += 16) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, r);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), r);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    m = w < m ? w : m;
  }
  return m;
}

__device__ __forceinline__ void gen_table_body(MjhHuffTable *__restrict__ T, int lane)
{
  __shared__ int s_bits[33];
  __shared__ int s_bitpos[33];
  __shared__ int s_cum[18];
  __shared__ int s_first[18];
  __shared__ unsigned char s_val[256];
  const unsigned INF = 1000000001u;
  unsigned f[5];
  int cs[5], g[5];
  bool nz[5];
#pragma unroll
  for (int t = 0; t < 5; t++) {
    const int s = lane + 64 * t;
    const unsigned c = s < 256 ? T->counts[s] : (s == 256 ? 1u : 0u);
    nz[t] = c != 0;
    f[t] = c ? c : INF;
    cs[t] = 0;
    g[t] = s;
  }
  for (;;) {
    unsigned long long k1 = ~0ull;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const unsigned long long key = ((unsigned long long)f[t] << 9) | (unsigned)(511 - (lane + 64 * t));
      k1 = key < k1 ? key : k1;
    }
    k1 = wave_min_u64(k1);
    const int c1 = 511 - (int)(k1 & 511);
    const unsigned f1 = (unsigned)(k1 >> 9);
    unsigned long long k2 = ~0ull;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int s = lane + 64 * t;
      const unsigned long long key = ((unsigned long long)f[t] << 9) | (unsigned)(511 - s);
      if (s != c1) k2 = key < k2 ? key : k2;
    }
    k2 = wave_min_u64(k2);
    const int c2 = 511 - (int)(k2 & 511);
    const unsigned f2 = (unsigned)(k2 >> 9);
    if (f1 > 1000000000u || f2 > 1000000000u) break;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int s = lane + 64 * t;
      if (s == c1) f[t] = f1 + f2;
      if (s == c2) f[t] = INF;
      if (g[t] == c1 || g[t] == c2) { cs[t]++; g[t] = c1; }
    }
  }
  // bits[L] = number of symbols (incl. pseudo-symbol 256) with code length L
  for (int L = lane; L < 33; L += 64) s_bits[L] = 0;
  __syncthreads();
  for (int L = 1; L <= 32; L++) {
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < 5; t++) cnt += __popcll(__ballot(nz[t] && cs[t] == L));
    if (lane == 0) s_bits[L] = cnt;
  }
  __syncthreads();
  if (lane == 0) {
    int p = 0;
    for (int L = 1; L <= 32; L++) { s_bitpos[L] = p; p += s_bits[L]; }
  }
  __syncthreads();
  // symbols sorted by (code length, symbol value); the pseudo-symbol is never listed
  for (int L = 1; L <= 32; L++) {
    int p = s_bitpos[L];
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int s = lane + 64 * t;
      const bool mine = nz[t] && cs[t] == L && s != 256;
      const unsigned long long m = __ballot(mine);
      if (mine) {
        const int pos = p + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < 256) s_val[pos] = (unsigned char)s;
      }
      p += __popcll(m);
    }
  }
  __syncthreads();
  if (lane == 0) {
    int i, j;
    for (i = 32; i > 16; i--) {
      while (s_bits[i] > 0) {
        j = i - 2;
        while (s_bits[j] == 0) j--;
        s_bits[i] -= 2; s_bits[i - 1]++; s_bits[j + 1] += 2; s_bits[j]--;
      }
    }
    i = 16;
    while (i > 0 && s_bits[i] == 0) i--;
    if (i > 0) s_bits[i]--;
    int code = 0, cum = 0;
    s_cum[0] = 0;
    for (int L = 1; L <= 16; L++) {
      s_first[L] = code;
      code = (code + s_bits[L]) << 1;
      cum += s_bits[L];
      s_cum[L] = cum;
      T->bits[L] = (uint8_t)s_bits[L];
    }
    T->bits[0] = 0;
    T->nsyms = (uint32_t)cum;
  }
  __syncthreads();
  const int nsyms = s_cum[16];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int p = lane + 64 * t;
    T->ehufsi[p] = 0;
    T->ehufco[p] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int p = lane + 64 * t;
    if (p < nsyms) {
      int L = 1;
      while (s_cum[L] <= p) L++;
      const int sym = s_val[p];
      T->huffval[p] = (uint8_t)sym;
      T->ehufsi[sym] = (uint8_t)L;
      T->ehufco[sym] = (uint16_t)(s_first[L] + (p - s_cum[L - 1]));
    }
  }
}


__global__ void __launch_bounds__(64)
k_gen_tables(MjhHuffTable *__restrict__ tabs, int slots_per_image, int4 slots_a, int4 slots_b)
{
  const int img = blockIdx.y;
  const int li = blockIdx.x;
  const int slot = li == 0 ? slots_a.x : li == 1 ? slots_a.y : li == 2 ? slots_a.z : li == 3 ? slots_a.w
                 : li == 4 ? slots_b.x : li == 5 ? slots_b.y : li == 6 ? slots_b.z : slots_b.w;
  gen_table_body(tabs + (size_t)img * slots_per_image + slot, threadIdx.x);
}

// same, slots from a device array (progressive mode: one table set per candidate scan)
__global__ void __launch_bounds__(64)
k_gen_tables_list(MjhHuffTable *__restrict__ tabs, int slots_per_image, const int *__restrict__ slot_list)
{
  gen_table_body(tabs + (size_t)blockIdx.y * slots_per_image + slot_list[blockIdx.x], threadIdx.x);
}

// =============================================================================================
// K6  DC trellis (row a9, DC part): quantize_trellis jcdctmgr.c:1044-1118 + :1308-1327, driven
// per iMCU row by compress_trellis_pass jccoefct.c:418-441 (lastDC = 0 at the start of each
// iMCU row, chained over its v_samp_factor block rows).
// The Viterbi recursion along a block row is strictly sequential in float (no associativity),
// so the parallelism is: one 16-lane group per (image, component, iMCU row) chain, lane = the
// candidate k (<= 9 live), the 9 predecessor costs come over cross-lane shuffles.  Back
// pointers go to a byte scratch array (16 B per block, one coalesced store per group) and the
// back-track walks 16 blocks per step with shuffles instead of dependent global loads.
// =============================================================================================
__device__ __forceinline__ int grp_shfl(int v, int srclane_in_group, int lane)
{
  return __shfl(v, (lane & ~15) | srclane_in_group, 64);
}
__device__ __forceinline__ float grp_shfl_f(float v, int srclane_in_group, int lane)
{
  return __shfl(v, (lane & ~15) | srclane_in_group, 64);
}

// ---------------------------------------------------------------------------------------------
// K6 v2: the recursion with the cross-lane traffic on DPP.  Measured on the first version (ds_bpermute shuffles, removed in round 5; profiles/r03a): a luma chain
// of 960 sequential steps ran at ~2250 cycles per step, because the compiler serialised its 18 ds_bpermute round trips (LDS
// latency each) inside the dependent chain; 1.69 ms per 64 4K frames even with the GPU to itself -- longer than the AC kernel
// it is supposed to hide under.  Here a 16-lane group is one DPP row: the nine predecessor (value, cost) pairs arrive through
// `row_newbcast` (a VALU move, no LDS round trip), the per-step inputs are fetched one step ahead (their ds_bpermute latency
// hides behind the previous step), the next 16 blocks' loads are issued before the current 16 are walked, and the arg-min
// is a min3 tree on the critical path with the index (first minimum, as the reference's strict '<' scan) recovered beside it.
// Same operations on the same values in the same order per candidate: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <int L> __device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + L, 0xF, 0xF, true); }   // row_newbcast:L
template <int L> __device__ __forceinline__ float row_bcast_f(float v) { return __int_as_float(row_bcast<L>(__float_as_int(v))); }

__global__ void __launch_bounds__(64)
k_trellis_dc2(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
              int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
              int4 dc_slot_of_comp, const float *__restrict__ lambda_in, uint8_t *__restrict__ back, int chain0, int chain1)
{
  const int img = blockIdx.y;
  const int lane = threadIdx.x;
  MJH_WAVE_GROUPS(16);
  const int k = lane & 15;
  const int chain = chain0 + blockIdx.x * 4 + (lane >> 4);     // chains [chain0, chain1) of the image: component-major, one per iMCU row
  if (chain >= chain1) return;   // whole 16-lane groups (= DPP rows) leave together
  const int comp = chain / C.mcu_rows, imcu = chain - comp * C.mcu_rows;
  const MjhComp cc = C.c[comp];
  const int slot = comp == 0 ? dc_slot_of_comp.x : comp == 1 ? dc_slot_of_comp.y : comp == 2 ? dc_slot_of_comp.z : dc_slot_of_comp.w;
  const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  unsigned long long rsi = 0;   // 12 x 5 bits: category + its code length = the rate of a DC difference of that category
  for (int s = 0; s < 12; s++) rsi |= (unsigned long long)((T->ehufsi[s] + s) & 31) << (5 * s);
  const int q0 = Q->q[cc.qtbl][0];
  const int dq = 8 * q0;
  const float rcp = Q->rcp8q[cc.qtbl][0];
  const float lt0 = Q->lambda_tbl[cc.qtbl][0];
  int ncand = (2 + 60 / q0) | 1;                 // get_num_dc_trellis_candidates :930-933
  if (ncand > 9) ncand = 9;
  const int16_t *uq0 = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off;  // plane k = 0
  int16_t *qo0 = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;
  const float *lam = lambda_in + (size_t)img * C.total_real_blocks + cc.blk_off;
  uint8_t *bk = back + ((size_t)img * C.total_real_blocks + cc.blk_off) * 16;
  int last_dc = 0;
  for (int sub = 0; sub < cc.v; sub++) {
    const int br = imcu * cc.v + sub;
    if (br >= cc.hib) break;
    const int row0 = br * cc.wib;
    int prev_c = 0;
    float prev_cost = 0.0f;
    const bool vert = sub > 0 && C.delta_dc_weight > 0.0f;
    auto fetch = [&](int b, int &xs_o, float &lam_o, int &ab_o) {
      xs_o = b < cc.wib ? (int)uq0[row0 + b] : 0;
      lam_o = b < cc.wib ? lam[row0 + b] : 0.0f;
      ab_o = 0;
      if (vert && b < cc.wib)   // raw DC above (low half) and the final quantized DC above (high half): written by this group's own back-track
        ab_o = ((int)uq0[row0 - cc.wib + b] & 0xFFFF) |
               ((int)__hip_atomic_load(reinterpret_cast<const unsigned short *>(qo0 + row0 - cc.wib + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 16);
    };
    int xs_l, ab_l;
    float lam_l;
    fetch(k, xs_l, lam_l, ab_l);
    for (int g = 0; g < cc.wib; g += 16) {
      int xs_n, ab_n;
      float lam_n;
      fetch(g + 16 + k, xs_n, lam_n, ab_n);          // the next 16 blocks: in flight while these 16 are walked
      const int steps = min(16, cc.wib - g);
      int xs_s = grp_shfl(xs_l, 0, lane), ab_s = vert ? grp_shfl(ab_l, 0, lane) : 0;
      float lam_s = grp_shfl_f(lam_l, 0, lane);
      for (int s = 0; s < steps; s++) {
        const int bi = g + s;
        const int xs = xs_s, ab = ab_s;
        const float lambda_dc = lam_s * lt0;
        // the next step's inputs (their LDS round trip hides behind this step)
        xs_s = grp_shfl(xs_l, (s + 1) & 15, lane);
        lam_s = grp_shfl_f(lam_l, (s + 1) & 15, lane);
        if (vert) ab_s = grp_shfl(ab_l, (s + 1) & 15, lane);
        const int x = xs < 0 ? -xs : xs;
        const int qval = udiv_exact(x + (dq >> 1), dq, rcp);
        int cnd = qval - ncand / 2 + k;
        cnd = min(1023, max(-1023, cnd));
        const int delta = mul24(cnd, dq) - x;
        float dist = (float)mul24(delta, delta) * lambda_dc;
        if (xs < 0) cnd = -cnd;
        if (vert) {   // jcdctmgr.c:1069-1084
          const int dc_above_orig = (int)(short)(ab & 0xFFFF), dc_above_recon = (ab >> 16) * dq;
          const int d2 = (dc_above_orig - xs) - (dc_above_recon - cnd * dq);
          const float vertical_dist = (float)(d2 * d2) * lambda_dc;
          float t = vertical_dist - dist;
          t = C.delta_dc_weight * t;
          dist = dist + t;
        }
        float best;
        int bb = 0;
        if (bi == 0) {
          const int df = cnd - last_dc;
          const int bits = bitlen((unsigned)(df < 0 ? -df : df));
          best = (float)(int)((rsi >> (5 * bits)) & 31) + dist;
        } else {
          float costs[9];
#define DC2_EVAL(L)                                                                       \
          {                                                                               \
            const int df = cnd - row_bcast<L>(prev_c);                                    \
            const int bits = bitlen((unsigned)(df < 0 ? -df : df));                       \
            const float cost = (float)(int)((rsi >> (5 * bits)) & 31) + dist;             \
            costs[L] = L < ncand ? cost + row_bcast_f<L>(prev_cost) : 3e38f;              \
          }
          DC2_EVAL(0) DC2_EVAL(1) DC2_EVAL(2) DC2_EVAL(3) DC2_EVAL(4) DC2_EVAL(5) DC2_EVAL(6) DC2_EVAL(7) DC2_EVAL(8)
#undef DC2_EVAL
          const float m = fminf(fminf(fminf(costs[0], costs[1]), fminf(costs[2], costs[3])), fminf(fminf(fminf(costs[4], costs[5]), fminf(costs[6], costs[7])), costs[8]));
          // the first l that attains the minimum = what the reference's strict '<' scan in increasing l keeps (:1100-1106)
          bb = 8;
#pragma unroll
          for (int l = 7; l >= 0; l--) bb = costs[l] == m ? l : bb;
          best = m;
        }
        prev_c = cnd;
        prev_cost = best;
        bk[(size_t)(row0 + bi) * 16 + k] = (uint8_t)bb;
      }
      xs_l = xs_n; lam_l = lam_n; ab_l = ab_n;
    }
    // first minimum over the live candidates (:1309-1313)
    int j = 0;
    {
      float bc = row_bcast_f<0>(prev_cost);
#define DC2_LAST(L) { const float c = row_bcast_f<L>(prev_cost); if (L < ncand && c < bc) { bc = c; j = L; } }
      DC2_LAST(1) DC2_LAST(2) DC2_LAST(3) DC2_LAST(4) DC2_LAST(5) DC2_LAST(6) DC2_LAST(7) DC2_LAST(8)
#undef DC2_LAST
    }
    __threadfence_block();
    // back-track, 16 blocks per step: lane k owns block top-k; the next 16 blocks' loads are issued before these are walked
    auto fetch_back = [&](int top, int &xs_o, uint4 &w_o) {
      const int b = top - k;
      xs_o = 0; w_o = make_uint4(0, 0, 0, 0);
      if (top >= 0 && b >= 0) {
        xs_o = uq0[row0 + b];
        w_o = *reinterpret_cast<const uint4 *>(bk + (size_t)(row0 + b) * 16);
      }
    };
    int bx;
    uint4 w;
    fetch_back(cc.wib - 1, bx, w);
    for (int top = cc.wib - 1; top >= 0; top -= 16) {
      int bx_n;
      uint4 w_n;
      fetch_back(top - 16, bx_n, w_n);
      const int b = top - k;
      const int x = bx < 0 ? -bx : bx;
      const int qv = udiv_exact(x + (dq >> 1), dq, rcp);
      int myj = 0;
      const int steps = min(16, top + 1);
#define DC2_BACK(S)                                                                        \
      if (S < steps) {                                                                     \
        if (k == S) myj = j;                                                               \
        const unsigned word = j < 4 ? w.x : (j < 8 ? w.y : w.z);                           \
        const int nj = (int)((word >> (8 * (j & 3))) & 0xFF);                              \
        j = row_bcast<S>(nj);                                                              \
      }
      DC2_BACK(0) DC2_BACK(1) DC2_BACK(2) DC2_BACK(3) DC2_BACK(4) DC2_BACK(5) DC2_BACK(6) DC2_BACK(7)
      DC2_BACK(8) DC2_BACK(9) DC2_BACK(10) DC2_BACK(11) DC2_BACK(12) DC2_BACK(13) DC2_BACK(14) DC2_BACK(15)
#undef DC2_BACK
      if (b >= 0) {
        int cnd = qv - ncand / 2 + myj;
        cnd = min(1023, max(-1023, cnd));
        if (bx < 0) cnd = -cnd;
        qo0[row0 + b] = (int16_t)cnd;
        if (b == cc.wib - 1) last_dc = cnd;
      }
      bx = bx_n; w = w_n;
    }
    last_dc = row_bcast<0>(last_dc);        // owner of block wib-1 is lane 0 of the first step
    if (C.delta_dc_weight > 0.0f) __threadfence();   // the next sub-row reads this row's final DC values back (other lanes of the group)
    else __threadfence_block();
  }
}

// ---------------------------------------------------------------------------------------------
// K6 v3: the transition costs through a sliding window.  Every kernel of this pipeline turned out to be bound by VALU
// issue (~4 cycles per wave instruction), and the recursion above spends most of its instructions on 9 x 9 = 81
// (difference -> category -> code length) evaluations per block, one per (candidate, predecessor) pair.  The candidates
// of a block are CONSECUTIVE integers (qv - h + k, jcdctmgr.c:1054-1062; the clamp to +-1023 cannot bind when 8q >= 40),
// so with the lanes of a group ordered by candidate VALUE (lane = k for a non-negative raw DC, ncand-1-k for a negative
// one, whose candidates are negated) lane L holds the value c0 + L and the difference to the predecessor in lane M is
// (c0 - c0_prev) + (L - M): 17 distinct differences per block instead of 81.  Lane j evaluates the rate of difference
// number j+1 (number 0 is needed by one pair only and is evaluated by every lane), and the nine rates a lane needs are
// its neighbours' values, fetched with DPP row shifts.  Costs are the same sums of the same floats, so the minimum is the
// same; "first minimum in candidate order" (the reference's strict '<' scan over l) is the lowest or the highest lane of
// the equality mask, depending on the orientation of the predecessor block.
// Used when every DC quantizer step 8q is >= 40 and the vertical-gradient term is off; k_trellis_dc2 otherwise.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dc_rate(unsigned long long rsi, int n)
{
  const int a = n < 0 ? -n : n;
  return (float)(int)((rsi >> (5 * bitlen((unsigned)a))) & 31);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float old, float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// the same with lanes that have no source reading 0 (bound_ctrl): a DPP operand the compiler may fold into the consuming
// instruction instead of a v_mov_b32_dpp of its own (4 issue cycles each, profiles/r04a_valu_rate_summary.md)
template <int CTRL> __device__ __forceinline__ float dpp_f0(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// dpp(v) + d in ONE instruction (v_add_f32 with a DPP source; this compiler does not fold a v_mov_b32_dpp into the add by
// itself, and each costs a 4-cycle issue slot): SHL = row_shl:n (lane i reads lane i + n of its row) with lanes that have no source reading 0.0, BC = row_newbcast:n.
// Same IEEE add as the two-instruction form.  `first`: the source may have been written by the instruction before (DPP read
// hazard: 2 wait states), which the hazard recognizer cannot see across inline asm.
#ifdef MJH_SIMT_HOST   // (tools/simt: the same three operations spelled with builtins)
template <int SHL> __device__ __forceinline__ float add_shl(float v, float d, bool first = false)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + SHL, 0xF, 0xF, true)) + d;
}
template <int BC> __device__ __forceinline__ float add_bcast(float v, float d, bool first = false)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + BC, 0xF, 0xF, false)) + d;
}
__device__ __forceinline__ float min3_f(float a, float b, float c) { return fminf(fminf(a, b), c); }
#else
template <int SHL> __device__ __forceinline__ float add_shl(float v, float d, bool first = false)
{
  float r;
  if (first) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(d), "n"(SHL));
  else asm volatile("v_add_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(d), "n"(SHL));
  return r;
}
template <int BC> __device__ __forceinline__ float add_bcast(float v, float d, bool first = false)
{
  float r;
  if (first) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(d), "n"(BC));
  else asm volatile("v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(d), "n"(BC));
  return r;
}
// The minimum of three costs in ONE instruction.  fminf() on values that come out of inline assembly costs a v_max_f32 x, x each
// (the compiler quiets a NaN it cannot rule out) in front of every v_min_f32: the nine-way minimum of a DC step was 8 + 6 + 2
// instructions; costs are never NaN (3e38 stands for "no candidate"), the minimum of the same nine floats is the same float.
__device__ __forceinline__ float min3_f(float a, float b, float c)
{
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
#endif
__device__ __forceinline__ float min9_f(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8)
{
  return min3_f(min3_f(a0, a1, a2), min3_f(a3, a4, a5), min3_f(a6, a7, a8));
}

__global__ void __launch_bounds__(64)
k_trellis_dc3(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
              int16_t *__restrict__ coef_q, const MjhHuffTable *__restrict__ tabs, int slots_per_image,
              int4 dc_slot_of_comp, const float *__restrict__ lambda_in, uint8_t *__restrict__ back, int chain0, int chain1)
{
  const int img = blockIdx.y;
  const int lane = threadIdx.x;
  MJH_WAVE_GROUPS(16);
  const int k = lane & 15;                      // lane in the group = rank of its candidate VALUE
  const int chain = chain0 + blockIdx.x * 4 + (lane >> 4);     // chains [chain0, chain1) of the image: component-major, one per iMCU row
  if (chain >= chain1) return;   // whole 16-lane groups (= DPP rows) leave together
  const int comp = chain / C.mcu_rows, imcu = chain - comp * C.mcu_rows;
  const MjhComp cc = C.c[comp];
  const int slot = comp == 0 ? dc_slot_of_comp.x : comp == 1 ? dc_slot_of_comp.y : comp == 2 ? dc_slot_of_comp.z : dc_slot_of_comp.w;
  const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  unsigned long long rsi = 0;   // 12 x 5 bits: category + its code length
  for (int s = 0; s < 12; s++) rsi |= (unsigned long long)((T->ehufsi[s] + s) & 31) << (5 * s);
  const int q0 = Q->q[cc.qtbl][0];
  const int dq = 8 * q0;
  const float rcp = Q->rcp8q[cc.qtbl][0];
  const float lt0 = Q->lambda_tbl[cc.qtbl][0];
  int ncand = (2 + 60 / q0) | 1;                 // get_num_dc_trellis_candidates :930-933
  if (ncand > 9) ncand = 9;
  const int h = ncand / 2;
  const bool vlane = k < ncand;
  const int16_t *uq0 = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off;  // plane k = 0
  int16_t *qo0 = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;
  const float *lam = lambda_in + (size_t)img * C.total_real_blocks + cc.blk_off;
  uint8_t *bk = back + ((size_t)img * C.total_real_blocks + cc.blk_off) * 16;
  int last_dc = 0;
  for (int sub = 0; sub < cc.v; sub++) {
    const int br = imcu * cc.v + sub;
    if (br >= cc.hib) break;
    const int row0 = br * cc.wib;
    int prev_c0 = 0, prev_neg = 0;
    float prev_cost = 0.0f;
    // per block, computed 16 blocks at a time (lane j: block g+j): |raw DC| | conventional quantized value << 16 | negative << 26,
    // and lambda * (1 / q0^2)
    auto fetch = [&](int b, unsigned &pk_o, float &lam_o) {
      const int xs = b < cc.wib ? (int)uq0[row0 + b] : 0;
      const float l = b < cc.wib ? lam[row0 + b] : 0.0f;
      const int x = xs < 0 ? -xs : xs;
      const int qv = udiv_exact(x + (dq >> 1), dq, rcp);
      pk_o = (unsigned)x | ((unsigned)qv << 16) | (xs < 0 ? 1u << 26 : 0u);
      lam_o = l * lt0;
    };
    unsigned pk_l;
    float lam_l;
    fetch(k, pk_l, lam_l);
    for (int g = 0; g < cc.wib; g += 16) {
      unsigned pk_n;
      float lam_n;
      fetch(g + 16 + k, pk_n, lam_n);          // the next 16 blocks: in flight while these 16 are walked
      const int steps = min(16, cc.wib - g);
      unsigned pk_s = (unsigned)grp_shfl((int)pk_l, 0, lane);
      float lam_s = grp_shfl_f(lam_l, 0, lane);
      for (int s = 0; s < steps; s++) {
        const int bi = g + s;
        const unsigned pk = pk_s;
        const float lambda_dc = lam_s;
        pk_s = (unsigned)grp_shfl((int)pk_l, (s + 1) & 15, lane);     // the next step's inputs (their LDS round trip hides behind this step)
        lam_s = grp_shfl_f(lam_l, (s + 1) & 15, lane);
        const int x = (int)(pk & 0xFFFFu), qv = (int)((pk >> 16) & 1023u), neg = (int)(pk >> 26);
        const int kk = neg ? ncand - 1 - k : k;            // candidate index held by this lane
        const int c0 = neg ? -(qv + h) : qv - h;           // value of lane 0's candidate; this lane's is c0 + k
        const int delta = mul24(qv - h + kk, dq) - x;
        const float dist = (float)mul24(delta, delta) * lambda_dc;
        float best;
        int bb = 0;
        if (bi == 0) {
          best = dc_rate(rsi, c0 + k - last_dc) + dist;
        } else {
          const int D = c0 - prev_c0;
          const float Rj = dc_rate(rsi, D + k - 7), R0 = dc_rate(rsi, D - 8);
          // predecessor in lane M: rate of difference number k - M + 8 = the value of lane k + 7 - M (number 0: R0)
          const float c_0 = add_bcast<0>(prev_cost, add_shl<7>(Rj, dist, true), true);
          const float c_1 = add_bcast<1>(prev_cost, add_shl<6>(Rj, dist));
          const float c_2 = add_bcast<2>(prev_cost, add_shl<5>(Rj, dist));
          const float c_3 = add_bcast<3>(prev_cost, add_shl<4>(Rj, dist));
          const float c_4 = add_bcast<4>(prev_cost, add_shl<3>(Rj, dist));
          const float c_5 = add_bcast<5>(prev_cost, add_shl<2>(Rj, dist));
          const float c_6 = add_bcast<6>(prev_cost, add_shl<1>(Rj, dist));
          const float c_7 = add_bcast<7>(prev_cost, Rj + dist);
          const float c_8 = add_bcast<8>(prev_cost, dpp_f<0x111>(R0, Rj) + dist);     // row_shr:1; lane 0 keeps R0
          const float m = min9_f(c_0, c_1, c_2, c_3, c_4, c_5, c_6, c_7, c_8);
          // first minimum in CANDIDATE order of the predecessor (:1100-1106): lanes of invalid candidates hold 3e38
          // (tried in round 6: the mask shifted in by an add-with-carry behind each compare, 18 instead of ~26 instructions -- a
          // chain of nine dependent adds through vcc, 0.3 % slower than the select / or tree)
          unsigned e = (c_0 == m ? 1u : 0u) | (c_1 == m ? 2u : 0u) | (c_2 == m ? 4u : 0u) | (c_3 == m ? 8u : 0u) | (c_4 == m ? 16u : 0u) |
                       (c_5 == m ? 32u : 0u) | (c_6 == m ? 64u : 0u) | (c_7 == m ? 128u : 0u) | (c_8 == m ? 256u : 0u);
          bb = prev_neg ? ncand - 1 - (31 - __clz((int)e)) : __ffs((int)e) - 1;
          best = m;
        }
        prev_cost = vlane ? best : 3e38f;
        prev_c0 = c0;
        prev_neg = neg;
        if (vlane) bk[(size_t)(row0 + bi) * 16 + kk] = (uint8_t)bb;
      }
      pk_l = pk_n; lam_l = lam_n;
    }
    // first minimum over the candidates of the last block, in candidate order (:1309-1313)
    int j;
    {
      const float p0 = row_bcast_f<0>(prev_cost), p1 = row_bcast_f<1>(prev_cost), p2 = row_bcast_f<2>(prev_cost), p3 = row_bcast_f<3>(prev_cost),
                  p4 = row_bcast_f<4>(prev_cost), p5 = row_bcast_f<5>(prev_cost), p6 = row_bcast_f<6>(prev_cost), p7 = row_bcast_f<7>(prev_cost),
                  p8 = row_bcast_f<8>(prev_cost);
      const float m = min9_f(p0, p1, p2, p3, p4, p5, p6, p7, p8);
      const unsigned e = (p0 == m ? 1u : 0u) | (p1 == m ? 2u : 0u) | (p2 == m ? 4u : 0u) | (p3 == m ? 8u : 0u) | (p4 == m ? 16u : 0u) |
                         (p5 == m ? 32u : 0u) | (p6 == m ? 64u : 0u) | (p7 == m ? 128u : 0u) | (p8 == m ? 256u : 0u);
      j = prev_neg ? ncand - 1 - (31 - __clz((int)e)) : __ffs((int)e) - 1;
    }
    __threadfence_block();
    // back-track, 16 blocks per step: lane k owns block top-k; the next 16 blocks' loads are issued before these are walked
    auto fetch_back = [&](int top, int &xs_o, uint4 &w_o) {
      const int b = top - k;
      xs_o = 0; w_o = make_uint4(0, 0, 0, 0);
      if (top >= 0 && b >= 0) {
        xs_o = uq0[row0 + b];
        w_o = *reinterpret_cast<const uint4 *>(bk + (size_t)(row0 + b) * 16);
      }
    };
    int bx;
    uint4 w;
    fetch_back(cc.wib - 1, bx, w);
    for (int top = cc.wib - 1; top >= 0; top -= 16) {
      int bx_n;
      uint4 w_n;
      fetch_back(top - 16, bx_n, w_n);
      const int b = top - k;
      const int x = bx < 0 ? -bx : bx;
      const int qv = udiv_exact(x + (dq >> 1), dq, rcp);
      int myj = 0;
      const int steps = min(16, top + 1);
#define DC3_BACK(S)                                                                        \
      if (S < steps) {                                                                     \
        if (k == S) myj = j;                                                               \
        const unsigned word = j < 4 ? w.x : (j < 8 ? w.y : w.z);                           \
        const int nj = (int)((word >> (8 * (j & 3))) & 0xFF);                              \
        j = row_bcast<S>(nj);                                                              \
      }
      DC3_BACK(0) DC3_BACK(1) DC3_BACK(2) DC3_BACK(3) DC3_BACK(4) DC3_BACK(5) DC3_BACK(6) DC3_BACK(7)
      DC3_BACK(8) DC3_BACK(9) DC3_BACK(10) DC3_BACK(11) DC3_BACK(12) DC3_BACK(13) DC3_BACK(14) DC3_BACK(15)
#undef DC3_BACK
      if (b >= 0) {
        int cnd = qv - h + myj;
        cnd = min(1023, max(-1023, cnd));
        if (bx < 0) cnd = -cnd;
        qo0[row0 + b] = (int16_t)cnd;
        if (b == cc.wib - 1) last_dc = cnd;
      }
      bx = bx_n; w = w_n;
    }
    last_dc = row_bcast<0>(last_dc);        // owner of block wib-1 is lane 0 of the first step
    __threadfence_block();
  }
}

// ---------------------------------------------------------------------------------------------
// K6 for one or two frames: block rows walked SPECULATIVELY (round 4).  With a single 4K frame on the chip the DC trellis is
// the critical path of the whole encode (402 of 690 us): a luma chain is the 2 block rows of an iMCU row walked one after the
// other, because the first block of the second row is rated against the FINAL last value of the first row (lastDC,
// jccoefct.c:418-441).  That value is one of the <= 9 candidates of the first row's last block, which depend on its raw DC
// only -- so the second (third, fourth) row is walked once per candidate, in parallel with the first, on a chip that has
// nothing else to do; k_trellis_dc3_resolve then back-tracks row after row, taking for every row the walk whose hypothesis came
// true.  Same float sums in the same order as k_trellis_dc3: same files.  Scratch: 16 back-pointer bytes per (block, hypothesis).
// ---------------------------------------------------------------------------------------------
#define DC3_HYP 9
__global__ void __launch_bounds__(64)
k_trellis_dc3_fwd(MjhConst C, const MjhQuant *__restrict__ Q, const int16_t *__restrict__ coef_uq,
                  const MjhHuffTable *__restrict__ tabs, int slots_per_image, int4 dc_slot_of_comp,
                  const float *__restrict__ lambda_in, uint8_t *__restrict__ back9, int *__restrict__ jfin, int16_t *__restrict__ qspec, int rows_total)
{
  MJH_WAVE_GROUPS(16);
  const int img = blockIdx.y;
  const int lane = threadIdx.x;
  const int k = lane & 15;
  const int chain = blockIdx.x * 4 + (lane >> 4);          // (global block row of the image, hypothesis)
  if (chain >= rows_total * DC3_HYP) return;
  const int grow = chain / DC3_HYP, hyp = chain - grow * DC3_HYP;
  int comp = 0, br = grow;
  while (comp + 1 < C.ncomp && br >= C.c[comp].hib) { br -= C.c[comp].hib; comp++; }
  const MjhComp cc = C.c[comp];
  const int sub = br % cc.v;
  const int slot = comp == 0 ? dc_slot_of_comp.x : comp == 1 ? dc_slot_of_comp.y : comp == 2 ? dc_slot_of_comp.z : dc_slot_of_comp.w;
  const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slot;
  const int q0 = Q->q[cc.qtbl][0];
  const int dq = 8 * q0;
  const float rcp = Q->rcp8q[cc.qtbl][0];
  const float lt0 = Q->lambda_tbl[cc.qtbl][0];
  int ncand = (2 + 60 / q0) | 1;
  if (ncand > 9) ncand = 9;
  const int h = ncand / 2;
  if (hyp >= ncand || (sub == 0 && hyp > 0)) return;        // whole 16-lane groups leave together
  unsigned long long rsi = 0;
  for (int s = 0; s < 12; s++) rsi |= (unsigned long long)((T->ehufsi[s] + s) & 31) << (5 * s);
  const bool vlane = k < ncand;
  const int16_t *uq0 = coef_uq + (size_t)img * C.coefs_per_image + cc.coef_off;
  const float *lam = lambda_in + (size_t)img * C.total_real_blocks + cc.blk_off;
  const int row0 = br * cc.wib;
  uint8_t *bk = back9 + (((size_t)img * C.total_real_blocks + cc.blk_off + row0) * DC3_HYP) * 16;   // + (bi * 9 + hyp) * 16 + kk
  int last_dc = 0;
  if (sub > 0) {     // the hypothesis: the row above ended on candidate `hyp` of its last block
    const int xs = uq0[row0 - 1];
    const int x = xs < 0 ? -xs : xs;
    int cnd = udiv_exact(x + (dq >> 1), dq, rcp) - h + hyp;
    cnd = min(1023, max(-1023, cnd));
    last_dc = xs < 0 ? -cnd : cnd;
  }
  int prev_c0 = 0, prev_neg = 0;
  float prev_cost = 0.0f;
  auto fetch = [&](int b, unsigned &pk_o, float &lam_o) {
    const int xs = b < cc.wib ? (int)uq0[row0 + b] : 0;
    const float l = b < cc.wib ? lam[row0 + b] : 0.0f;
    const int x = xs < 0 ? -xs : xs;
    const int qv = udiv_exact(x + (dq >> 1), dq, rcp);
    pk_o = (unsigned)x | ((unsigned)qv << 16) | (xs < 0 ? 1u << 26 : 0u);
    lam_o = l * lt0;
  };
  unsigned pk_l;
  float lam_l;
  fetch(k, pk_l, lam_l);
  for (int g = 0; g < cc.wib; g += 16) {
    unsigned pk_n;
    float lam_n;
    fetch(g + 16 + k, pk_n, lam_n);
    const int steps = min(16, cc.wib - g);
    unsigned pk_s = (unsigned)grp_shfl((int)pk_l, 0, lane);
    float lam_s = grp_shfl_f(lam_l, 0, lane);
    for (int s = 0; s < steps; s++) {
      const int bi = g + s;
      const unsigned pk = pk_s;
      const float lambda_dc = lam_s;
      pk_s = (unsigned)grp_shfl((int)pk_l, (s + 1) & 15, lane);
      lam_s = grp_shfl_f(lam_l, (s + 1) & 15, lane);
      const int x = (int)(pk & 0xFFFFu), qv = (int)((pk >> 16) & 1023u), neg = (int)(pk >> 26);
      const int kk = neg ? ncand - 1 - k : k;
      const int c0 = neg ? -(qv + h) : qv - h;
      const int delta = mul24(qv - h + kk, dq) - x;
      const float dist = (float)mul24(delta, delta) * lambda_dc;
      float best;
      int bb = 0;
      if (bi == 0) {
        best = dc_rate(rsi, c0 + k - last_dc) + dist;
      } else {
        const int D = c0 - prev_c0;
        const float Rj = dc_rate(rsi, D + k - 7), R0 = dc_rate(rsi, D - 8);
        const float c_0 = add_bcast<0>(prev_cost, add_shl<7>(Rj, dist, true), true);
        const float c_1 = add_bcast<1>(prev_cost, add_shl<6>(Rj, dist));
        const float c_2 = add_bcast<2>(prev_cost, add_shl<5>(Rj, dist));
        const float c_3 = add_bcast<3>(prev_cost, add_shl<4>(Rj, dist));
        const float c_4 = add_bcast<4>(prev_cost, add_shl<3>(Rj, dist));
        const float c_5 = add_bcast<5>(prev_cost, add_shl<2>(Rj, dist));
        const float c_6 = add_bcast<6>(prev_cost, add_shl<1>(Rj, dist));
        const float c_7 = add_bcast<7>(prev_cost, Rj + dist);
        const float c_8 = add_bcast<8>(prev_cost, dpp_f<0x111>(R0, Rj) + dist);
        const float m = min9_f(c_0, c_1, c_2, c_3, c_4, c_5, c_6, c_7, c_8);
        unsigned e = (c_0 == m ? 1u : 0u) | (c_1 == m ? 2u : 0u) | (c_2 == m ? 4u : 0u) | (c_3 == m ? 8u : 0u) | (c_4 == m ? 16u : 0u) |
                     (c_5 == m ? 32u : 0u) | (c_6 == m ? 64u : 0u) | (c_7 == m ? 128u : 0u) | (c_8 == m ? 256u : 0u);
        bb = prev_neg ? ncand - 1 - (31 - __clz((int)e)) : __ffs((int)e) - 1;
        best = m;
      }
      prev_cost = vlane ? best : 3e38f;
      prev_c0 = c0;
      prev_neg = neg;
      if (vlane) bk[((size_t)bi * DC3_HYP + hyp) * 16 + kk] = (uint8_t)bb;
    }
    pk_l = pk_n; lam_l = lam_n;
  }
  {   // first minimum over the candidates of the last block, in candidate order (jcdctmgr.c:1309-1313)
    const float p0 = row_bcast_f<0>(prev_cost), p1 = row_bcast_f<1>(prev_cost), p2 = row_bcast_f<2>(prev_cost), p3 = row_bcast_f<3>(prev_cost),
                p4 = row_bcast_f<4>(prev_cost), p5 = row_bcast_f<5>(prev_cost), p6 = row_bcast_f<6>(prev_cost), p7 = row_bcast_f<7>(prev_cost),
                p8 = row_bcast_f<8>(prev_cost);
    const float m = min9_f(p0, p1, p2, p3, p4, p5, p6, p7, p8);
    const unsigned e = (p0 == m ? 1u : 0u) | (p1 == m ? 2u : 0u) | (p2 == m ? 4u : 0u) | (p3 == m ? 8u : 0u) | (p4 == m ? 16u : 0u) |
                       (p5 == m ? 32u : 0u) | (p6 == m ? 64u : 0u) | (p7 == m ? 128u : 0u) | (p8 == m ? 256u : 0u);
    int j = prev_neg ? ncand - 1 - (31 - __clz((int)e)) : __ffs((int)e) - 1;
    if (k == 0) jfin[((size_t)img * rows_total + grow) * DC3_HYP + hyp] = j;
    // ... and this walk's own back-track, 16 blocks per step, into the hypothesis' copy of the row (k_trellis_dc3_resolve keeps
    // the copy whose hypothesis held)
    __threadfence_block();
    int16_t *qs = qspec + ((size_t)img * C.total_real_blocks + cc.blk_off + row0) * DC3_HYP;
    auto fetch_back = [&](int top, int &xs_o, uint4 &w_o) {
      const int b = top - k;
      xs_o = 0; w_o = make_uint4(0, 0, 0, 0);
      if (top >= 0 && b >= 0) {
        xs_o = uq0[row0 + b];
        w_o = *reinterpret_cast<const uint4 *>(bk + ((size_t)b * DC3_HYP + hyp) * 16);
      }
    };
    int bx;
    uint4 w;
    fetch_back(cc.wib - 1, bx, w);
    for (int top = cc.wib - 1; top >= 0; top -= 16) {
      int bx_n;
      uint4 w_n;
      fetch_back(top - 16, bx_n, w_n);
      const int b = top - k;
      const int x = bx < 0 ? -bx : bx;
      const int qv = udiv_exact(x + (dq >> 1), dq, rcp);
      int myj = 0;
      const int steps = min(16, top + 1);
#define DC3_BACK(S)                                                                        \
      if (S < steps) {                                                                     \
        if (k == S) myj = j;                                                               \
        const unsigned word = j < 4 ? w.x : (j < 8 ? w.y : w.z);                           \
        const int nj = (int)((word >> (8 * (j & 3))) & 0xFF);                              \
        j = row_bcast<S>(nj);                                                              \
      }
      DC3_BACK(0) DC3_BACK(1) DC3_BACK(2) DC3_BACK(3) DC3_BACK(4) DC3_BACK(5) DC3_BACK(6) DC3_BACK(7)
      DC3_BACK(8) DC3_BACK(9) DC3_BACK(10) DC3_BACK(11) DC3_BACK(12) DC3_BACK(13) DC3_BACK(14) DC3_BACK(15)
#undef DC3_BACK
      if (b >= 0) {
        int cnd = qv - h + myj;
        cnd = min(1023, max(-1023, cnd));
        if (bx < 0) cnd = -cnd;
        qs[(size_t)b * DC3_HYP + hyp] = (int16_t)cnd;
      }
      bx = bx_n; w = w_n;
    }
  }
}

// row after row of an iMCU row: keep the copy of the walk whose hypothesis = the candidate the row above ended on (its last
// block's candidate index is what k_trellis_dc3_fwd left in jfin).  One wave per (component, iMCU row).
__global__ void __launch_bounds__(64)
k_trellis_dc3_resolve(MjhConst C, int16_t *__restrict__ coef_q, const int *__restrict__ jfin, const int16_t *__restrict__ qspec, int rows_total)
{
  const int img = blockIdx.y, lane = threadIdx.x;
  const int chain = blockIdx.x;
  const int comp = chain / C.mcu_rows, imcu = chain - comp * C.mcu_rows;
  const MjhComp cc = C.c[comp];
  int grow0 = 0;
  for (int c = 0; c < comp; c++) grow0 += C.c[c].hib;
  int16_t *qo0 = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;
  int hyp = 0;
  for (int sub = 0; sub < cc.v; sub++) {
    const int br = imcu * cc.v + sub;
    if (br >= cc.hib) break;
    const int row0 = br * cc.wib;
    const int16_t *qs = qspec + ((size_t)img * C.total_real_blocks + cc.blk_off + row0) * DC3_HYP;
    for (int b = lane; b < cc.wib; b += 64) qo0[row0 + b] = qs[(size_t)b * DC3_HYP + hyp];
    hyp = jfin[((size_t)img * rows_total + grow0 + br) * DC3_HYP + hyp];
  }
}

// =============================================================================================
// K7  Huffman bit packing of the interleaved baseline scan (row a12): encode_one_block
// jchuff.c:563-652, encode_mcu_huff :693-763, flush_bits :479-514 (pad with 1-bits),
// byte stuffing :354-387.  Three steps, all one lane per block:
//   k_enc_len   : bits of every block (dummy blocks included) -> len16[mcu-order position]
//   (scan)      : k_chunk_sums / k_scan_sums / k_offsets: exclusive prefix sum -> bit offset
//   k_enc_write : re-walk the block and OR its bits into the (zeroed) unstuffed bit stream
// then the 0xFF -> 0xFF00 stuffing is a second count/scan/scatter over the byte stream.
// =============================================================================================
struct EncTables {
  const MjhHuffTable *dc;
  const MjhHuffTable *ac;
};

__device__ __forceinline__ int mcu_position(const MjhConst &C, const MjhComp &cc, int r, int c)
{
  const int m = (r / cc.v) * C.mcus_per_row + c / cc.h;
  return m * C.blocks_per_mcu + cc.mcu_blk0 + (r % cc.v) * cc.h + (c % cc.h);
}

template <bool COMPACT>
__global__ void __launch_bounds__(256)
k_enc_len(MjhConst C, const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, const MjhHuffTable *__restrict__ tabs,
          int slots_per_image, int4 dc_slot_of_comp, int4 ac_slot_of_comp, uint16_t *__restrict__ len16, MjhImageMeta *__restrict__ meta)
{
  __shared__ unsigned char s_ac[256];
  __shared__ unsigned char s_dc[32];   // a DC difference of untrusted coefficient input can have up to 17 bits
  const int comp = blockIdx.y, img = blockIdx.z;
  const MjhComp cc = C.c[comp];
  const int tid = threadIdx.x;
  const int dslot = comp == 0 ? dc_slot_of_comp.x : comp == 1 ? dc_slot_of_comp.y : comp == 2 ? dc_slot_of_comp.z : dc_slot_of_comp.w;
  const int aslot = comp == 0 ? ac_slot_of_comp.x : comp == 1 ? ac_slot_of_comp.y : comp == 2 ? ac_slot_of_comp.z : ac_slot_of_comp.w;
  const MjhHuffTable *TD = tabs + (size_t)img * slots_per_image + dslot;
  const MjhHuffTable *TA = tabs + (size_t)img * slots_per_image + aslot;
  s_ac[tid] = TA->ehufsi[tid];
  if (tid < 32) s_dc[tid] = tid < 16 ? TD->ehufsi[tid] : 0;
  __syncthreads();
  const int t = blockIdx.x * 256 + tid;
  if (t >= cc.wpad * cc.hpad) return;
  const int r = t / cc.wpad, c = t - r * cc.wpad;
  const int16_t *q = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;
  const int dc = q[dc_source_block(cc, r, c)];
  int pr, pc, pred = 0;
  if (mcu_prev_block(C, cc, r, c, pr, pc)) pred = q[dc_source_block(cc, pr, pc)];
  const int df = dc - pred;
  int nb = bitlen((unsigned)(df < 0 ? -df : df));
  if (nb > (C.precision == 12 ? 15 : 11)) meta[img].bad_coef = 1u;   // MAX_COEF_BITS + 1 (jchuff.c:489)
  int bits = s_dc[nb] + nb;
  const bool real = r < cc.hib && c < cc.wib;
  if (COMPACT) {
    const int b = real ? r * cc.wib + c : 0;
    const unsigned long long m = nzmask[(size_t)img * C.total_real_blocks + cc.blk_off + b];
    const int zrl = s_ac[0xF0];
    int prev = 0;
    for_each_nonzero(q + b, (size_t)cc.kstride, m, real, [&](int pos, int v) {
      const int run = pos - prev - 1;
      prev = pos;
      const int nbv = bitlen((unsigned)(v < 0 ? -v : v));
      bits += (run >> 4) * zrl + s_ac[((run & 15) << 4) + nbv] + nbv;
    });
    if (!real || prev < 63) bits += s_ac[0];
    len16[(size_t)img * C.total_mcu_blocks + mcu_position(C, cc, r, c)] = (uint16_t)bits;
    return;
  }
  int x[64];
  {
    const int16_t *qb = q + (real ? r * cc.wib + c : 0);   // dummy blocks: load anything in bounds, ignore it
#pragma unroll
    for (int k = 1; k < 64; k++) x[k] = qb[(size_t)k * cc.kstride];
  }
  if (real) {
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
      const int v = x[k];
      if (v == 0) run++;
      else {
        bits += (run >> 4) * s_ac[0xF0];
        nb = bitlen((unsigned)(v < 0 ? -v : v));
        bits += s_ac[((run & 15) << 4) + nb] + nb;
        run = 0;
      }
    }
    if (run > 0) bits += s_ac[0];
  } else {
    bits += s_ac[0];
  }
  len16[(size_t)img * C.total_mcu_blocks + mcu_position(C, cc, r, c)] = (uint16_t)bits;
}

// ---- generic 32-bit exclusive scan over n items per image, items produced by a functor ----
// chunk = 2048 items (256 threads x 8)
#define SCAN_CHUNK 2048

template <class T>
__global__ void __launch_bounds__(256)
k_chunk_sums(const T *__restrict__ len16, int n_per_image, unsigned *__restrict__ sums, int chunks_per_image)
{
  __shared__ unsigned sh[4];
  const int img = blockIdx.y, chunk = blockIdx.x;
  const T *p = len16 + (size_t)img * n_per_image;
  unsigned s = 0;
  const int base = chunk * SCAN_CHUNK + threadIdx.x * 8;
#pragma unroll
  for (int i = 0; i < 8; i++) if (base + i < n_per_image) s += p[base + i];
  const unsigned tot = block_reduce_256(s, sh);
  if (threadIdx.x == 0) sums[(size_t)img * chunks_per_image + chunk] = tot;
}

// one workgroup per image: exclusive scan of the chunk sums in place; total -> totals[img]
__global__ void __launch_bounds__(256)
k_scan_sums(unsigned *__restrict__ sums, int chunks_per_image, unsigned *__restrict__ totals, const unsigned *__restrict__ stream_bits)
{
  __shared__ unsigned sh[4];
  const int img = blockIdx.x;
  unsigned *p = sums + (size_t)img * chunks_per_image;
  unsigned carry = 0;
  unsigned long long wide = 0;   // bit offsets are 32-bit: a scan beyond 2^32 bits (512 MB) is reported, not wrapped silently
  int nchunks = chunks_per_image;
  if (stream_bits) {   // byte-stuffing pass: only the chunks that hold entropy-coded words were produced
    const unsigned nwords = ((((stream_bits[img] + 7) >> 3) + 3) >> 2);
    nchunks = min(chunks_per_image, (int)((nwords + SCAN_CHUNK - 1) / SCAN_CHUNK));
  }
  for (int base = 0; base < nchunks; base += 256) {
    const int i = base + threadIdx.x;
    const unsigned v = i < nchunks ? p[i] : 0u;
    unsigned tot;
    const unsigned ex = block_excl_scan_256(v, sh, &tot);
    if (i < nchunks) p[i] = carry + ex;
    carry += tot;
    wide += tot;
  }
  if (threadIdx.x == 0) totals[img] = wide >= 0xFFF00000ull ? 0xFFFFFFFFu : carry;   // sentinel read by the host (MJH_ETOOSMALL)
}

template <class T>
__global__ void __launch_bounds__(256)
k_offsets(const T *__restrict__ len16, int n_per_image, const unsigned *__restrict__ sums,
          int chunks_per_image, unsigned *__restrict__ off32)
{
  __shared__ unsigned sh[4];
  const int img = blockIdx.y, chunk = blockIdx.x;
  const T *p = len16 + (size_t)img * n_per_image;
  unsigned *o = off32 + (size_t)img * n_per_image;
  const int base = chunk * SCAN_CHUNK + threadIdx.x * 8;
  unsigned v[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { v[i] = base + i < n_per_image ? p[base + i] : 0u; s += v[i]; }
  unsigned ex = block_excl_scan_256(s, sh, nullptr) + sums[(size_t)img * chunks_per_image + chunk];
#pragma unroll
  for (int i = 0; i < 8; i++) { if (base + i < n_per_image) o[base + i] = ex; ex += v[i]; }
}

// ---- restart intervals (emit_restart jchuff.c:668-686): the scan is cut into segments of
// restart_interval MCUs; every segment but the last is padded with 1-bits to a byte boundary and
// followed by an RSTn marker.  seg_x[s] = bits added after segment s; its exclusive prefix seg_E
// shifts the bit offsets of all later blocks.
__global__ void __launch_bounds__(256)
k_seg_extra(MjhConst C, const unsigned *__restrict__ off32, const unsigned *__restrict__ totals,
            unsigned *__restrict__ seg_x, int nseg)
{
  const int img = blockIdx.y;
  const int sidx = blockIdx.x * 256 + threadIdx.x;
  if (sidx >= nseg) return;
  const unsigned *o = off32 + (size_t)img * C.total_mcu_blocks;
  const long long per = (long long)C.restart_interval * C.blocks_per_mcu;
  const long long bs = sidx * per, be = bs + per;
  const unsigned start = o[bs];
  const unsigned end = be < C.total_mcu_blocks ? o[be] : totals[img];
  const unsigned L = end - start;
  seg_x[(size_t)img * nseg + sidx] = sidx < nseg - 1 ? ((8u - (L & 7u)) & 7u) + 16u : 0u;
}

// Restart markers inside the byte stream (their 0xFF is not data: never stuffed).  mpos[] is sorted; a lane looks at 32
// consecutive bytes, so ONE lower-bound search for the first byte of its span (made only when the span holds a 0xFF at
// all) and a cursor that moves forward replace a binary search per 0xFF byte.
struct MarkerCursor {
  const unsigned *mpos; int n, idx;
  __device__ __forceinline__ MarkerCursor(const unsigned *m, int nmark) : mpos(m), n(nmark), idx(-1) {}
  __device__ __forceinline__ bool at(unsigned bytepos)   // bytepos must not decrease between calls
  {
    if (idx < 0) {
      int lo = 0, hi = n;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (mpos[mid] < bytepos) lo = mid + 1; else hi = mid; }
      idx = lo;
    }
    while (idx < n && mpos[idx] < bytepos) idx++;
    return idx < n && mpos[idx] == bytepos;
  }
};

// ---------------------------------------------------------------------------------------------
// Bit writer: the workgroup's window.  The first form (k_enc_write, removed in round 5) ORed every finished 32-bit word of every block
// straight into HBM with a memory-side atomic (measured: 635 MB of "write traffic" per 64 4K frames for a 53 MB stream, 84 %
// of the wave time waiting).  Here the threads of a workgroup take 256 CONSECUTIVE positions of the scan (MCU order), so
// the workgroup owns one contiguous bit range [offset of its first block, offset of the next workgroup's first block):
// the range is assembled in LDS (ds_or, 16 KB window = 512 bits per block on average) and leaves as coalesced word stores;
// only the two boundary words, shared with the neighbouring workgroups, are ORed into memory.  A range that does not fit
// the window (possible in principle: 1665 bits per block worst case) takes the direct path, workgroup by workgroup.
// Same bits at the same offsets: the stream is identical.
// ---------------------------------------------------------------------------------------------
#define ENCW_WIN 4096
// (BitSink<LDSW>: mjh_device.h)

template <bool COMPACT, class W>
__device__ __forceinline__ void enc_write_block(const MjhConst &C, const MjhComp &cc, W &bw, const unsigned *s_ac, const unsigned *s_dc,
                                                const int16_t *__restrict__ q, const unsigned long long *__restrict__ nzmask_img, int r, int c)
{
  const int dc = q[dc_source_block(cc, r, c)];
  int pr, pc, pred = 0;
  if (mcu_prev_block(C, cc, r, c, pr, pc)) pred = q[dc_source_block(cc, pr, pc)];
  {
    const int df = dc - pred;
    const int a = df < 0 ? -df : df;
    const int nb = bitlen((unsigned)a);
    const unsigned e = s_dc[nb];
    bw.put(e & 0xFFFF, (int)(e >> 16));
    if (nb) bw.put((unsigned)(df < 0 ? df - 1 : df), nb);
  }
  const bool real = r < cc.hib && c < cc.wib;
  const int b = real ? r * cc.wib + c : 0;
  if (COMPACT) {
    const unsigned long long m = nzmask_img[cc.blk_off + b];
    int prev = 0;
    for_each_nonzero(q + b, (size_t)cc.kstride, m, real, [&](int pos, int v) {
      int run = pos - prev - 1;
      prev = pos;
      while (run > 15) { const unsigned e = s_ac[0xF0]; bw.put(e & 0xFFFF, (int)(e >> 16)); run -= 16; }
      const int a = v < 0 ? -v : v;
      const int nbv = bitlen((unsigned)a);
      const unsigned e = s_ac[(run << 4) + nbv];
      bw.put_sym(e, (unsigned)(v < 0 ? v - 1 : v), nbv);
    });
    if (!real || prev < 63) { const unsigned e = s_ac[0]; bw.put(e & 0xFFFF, (int)(e >> 16)); }
  } else {
    int x[64];
#pragma unroll
    for (int k = 1; k < 64; k++) x[k] = q[(size_t)k * cc.kstride + b];
    if (real) {
      int run = 0;
#pragma unroll
      for (int k = 1; k < 64; k++) {
        const int v = x[k];
        if (v == 0) run++;
        else {
          while (run > 15) { const unsigned e = s_ac[0xF0]; bw.put(e & 0xFFFF, (int)(e >> 16)); run -= 16; }
          const int a = v < 0 ? -v : v;
          const int nb = bitlen((unsigned)a);
          const unsigned e = s_ac[(run << 4) + nb];
          bw.put_sym(e, (unsigned)(v < 0 ? v - 1 : v), nb);
          run = 0;
        }
      }
      if (run > 0) { const unsigned e = s_ac[0]; bw.put(e & 0xFFFF, (int)(e >> 16)); }
    } else {
      const unsigned e = s_ac[0];
      bw.put(e & 0xFFFF, (int)(e >> 16));
    }
  }
}

template <bool COMPACT>
__global__ void __launch_bounds__(256)
k_enc_write_mcu(MjhConst C, const int16_t *__restrict__ coef_q, const unsigned long long *__restrict__ nzmask, const MjhHuffTable *__restrict__ tabs,
                int slots_per_image, int4 dc_slot_of_comp, int4 ac_slot_of_comp,
                const unsigned *__restrict__ off32, unsigned *__restrict__ stream, size_t stream_words_per_image,
                const unsigned *__restrict__ totals, const unsigned *__restrict__ seg_E, const unsigned *__restrict__ seg_totals,
                unsigned *__restrict__ mpos, int nseg)
{
  __shared__ unsigned s_ac[MJH_MAXC][256];   // size << 16 | code, per component
  __shared__ unsigned s_dc[MJH_MAXC][32];
  __shared__ unsigned s_win[ENCW_WIN];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int N = C.total_mcu_blocks, base = blockIdx.x * 256;
  if (totals[img] == 0xFFFFFFFFu) return;     // scan beyond the 32-bit offset range: reported by k_finish_bits
  for (int ci = 0; ci < C.ncomp; ci++) {
    const int dslot = ci == 0 ? dc_slot_of_comp.x : ci == 1 ? dc_slot_of_comp.y : ci == 2 ? dc_slot_of_comp.z : dc_slot_of_comp.w;
    const int aslot = ci == 0 ? ac_slot_of_comp.x : ci == 1 ? ac_slot_of_comp.y : ci == 2 ? ac_slot_of_comp.z : ac_slot_of_comp.w;
    const MjhHuffTable *TD = tabs + (size_t)img * slots_per_image + dslot;
    const MjhHuffTable *TA = tabs + (size_t)img * slots_per_image + aslot;
    s_ac[ci][tid] = ((unsigned)TA->ehufsi[tid] << 16) | TA->ehufco[tid];
    if (tid < 32) s_dc[ci][tid] = tid < 16 ? ((unsigned)TD->ehufsi[tid] << 16) | TD->ehufco[tid] : 0u;
  }
  // final bit offset of scan position p (restart intervals shift everything behind them by their pad + marker bits)
  auto final_off = [&](int p) -> unsigned {
    if (p >= N) return totals[img] + (nseg > 1 ? seg_totals[img] : 0u);
    const int mcu = p / C.blocks_per_mcu;
    return off32[(size_t)img * N + p] + (nseg > 1 ? seg_E[(size_t)img * nseg + mcu / C.restart_interval] : 0u);
  };
  const unsigned start = final_off(base), end = final_off(min(base + 256, N));
  const unsigned w0 = start >> 5, nw = ((end + 31u) >> 5) - w0;
  const bool window = nw <= (unsigned)ENCW_WIN;
  if (window) for (unsigned i = tid; i < nw; i += 256) s_win[i] = 0u;
  __syncthreads();
  unsigned *g = stream + (size_t)img * stream_words_per_image;
  const int p = base + tid;
  if (p < N) {
    const int mcu = p / C.blocks_per_mcu, bi = p - mcu * C.blocks_per_mcu;
    int comp = 0;
    for (int ci = 1; ci < C.ncomp; ci++) if (bi >= C.c[ci].mcu_blk0) comp = ci;
    const MjhComp cc = C.c[comp];
    const int local = bi - cc.mcu_blk0, yi = local / cc.h, xi = local - yi * cc.h;
    const int mrow = mcu / C.mcus_per_row, mcol = mcu - mrow * C.mcus_per_row;
    const int r = mrow * cc.v + yi, c = mcol * cc.h + xi;
    const int16_t *q = coef_q + (size_t)img * C.coefs_per_image + cc.coef_off;
    const unsigned long long *nzi = COMPACT ? nzmask + (size_t)img * C.total_real_blocks : nullptr;
    const unsigned fo = final_off(p);
    const int segi = C.restart_interval ? mcu / C.restart_interval : 0;
    const bool seg_end = C.restart_interval && segi < nseg - 1 && (mcu + 1) % C.restart_interval == 0 && bi == C.blocks_per_mcu - 1;
    auto emit = [&](auto &bw, unsigned wbase) {     // wbase: bit position of word 0 of the sink
      enc_write_block<COMPACT>(C, cc, bw, s_ac[comp], s_dc[comp], q, nzi, r, c);
      if (seg_end) {   // last block of a restart interval: pad with 1-bits, then RSTn (emit_restart jchuff.c:668-686)
        const unsigned bitpos = bw.bitpos() + wbase;
        const int pad = (int)((8u - (bitpos & 7u)) & 7u);
        if (pad) bw.put((1u << pad) - 1u, pad);
        mpos[(size_t)img * nseg + segi] = (bitpos + (unsigned)pad) >> 3;
        bw.put(0xFFD0u + (unsigned)(segi & 7), 16);
      }
      bw.flush();
    };
    if (window) { BitSink<true> bw; bw.init(s_win, fo - (w0 << 5)); emit(bw, w0 << 5); }
    else { BitSink<false> bw; bw.init(g, fo); emit(bw, 0u); }
  }
  __syncthreads();
  if (window)
    for (unsigned i = tid; i < nw; i += 256) {
      const unsigned v = s_win[i];
      if (v == 0u) continue;                        // (k_zero_stream has cleared the range)
      if (i == 0u || i == nw - 1u) atomicOr(&g[w0 + i], v);   // shared with the neighbouring workgroup
      else g[w0 + i] = v;
    }
}

// zero exactly the words the entropy coder is going to OR its bits into (the buffer itself is sized for the
// worst case of 1665 bits per block; clearing all of it would cost more HBM traffic than the whole encode)
__global__ void __launch_bounds__(256)
k_zero_stream(unsigned *__restrict__ stream, size_t stream_words_per_image, const unsigned *__restrict__ totals,
              const unsigned *__restrict__ seg_totals)
{
  const int img = blockIdx.y;
  if (totals[img] == 0xFFFFFFFFu) return;   // scan beyond the 32-bit offset range: reported by k_finish_bits, nothing to prepare
  const unsigned bits = totals[img] + (seg_totals ? seg_totals[img] : 0u);
  const unsigned nvec = ((bits >> 5) + 8) >> 2;   // uint4 units, a few words of slack for the trailing partial word
  uint4 *p = reinterpret_cast<uint4 *>(stream + (size_t)img * stream_words_per_image);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i <= nvec; i += gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}

// ---- header + stuffing + trailer -------------------------------------------------------------

// One workgroup per image: [prefix: SOI APP0 DQT SOF] [DHT...] [SOS] -> out, hdr_len.
// emit_multi_dht jcmarker.c:293-401 (max-compression profile: ONE DHT marker holding every table
// the scan needs, in component order) / emit_dht :257-290 (fastest profile: one marker per table).
__global__ void __launch_bounds__(64)
k_header(const uint8_t *__restrict__ prefix, int prefix_len, const uint8_t *__restrict__ sos, int sos_len,
         const MjhHuffTable *__restrict__ tabs, int slots_per_image, MjhDhtPlan dht, int ndht,
         int multi_dht, uint8_t *__restrict__ out, size_t out_stride, MjhImageMeta *__restrict__ meta, const unsigned *__restrict__ append_sizes)
{
  // append_sizes (the later scans of a sequential multi-scan file): the header goes over the EOI of the file so far
  const int img = blockIdx.x;
  const int lane = threadIdx.x;
  uint8_t *o = out + (size_t)img * out_stride;
  const int start = append_sizes ? (int)append_sizes[img] - 2 : 0;
  for (int i = lane; i < prefix_len; i += 64) o[start + i] = prefix[i];
  int pos = start + prefix_len;
  const int *slots = dht.slots, *ids = dht.ids;      // (up to six tables: three components with DC and AC tables of their own)
  if (multi_dht) {
    int length = 2;
    for (int i = 0; i < ndht; i++) length += (int)tabs[(size_t)img * slots_per_image + slots[i]].nsyms + 17;
    if (lane == 0) { o[pos] = 0xFF; o[pos + 1] = 0xC4; o[pos + 2] = (uint8_t)(length >> 8); o[pos + 3] = (uint8_t)length; }
    pos += 4;
  }
  for (int i = 0; i < ndht; i++) {
    const MjhHuffTable *T = tabs + (size_t)img * slots_per_image + slots[i];
    const int n = (int)T->nsyms;
    if (!multi_dht) {
      const int length = n + 2 + 1 + 16;
      if (lane == 0) { o[pos] = 0xFF; o[pos + 1] = 0xC4; o[pos + 2] = (uint8_t)(length >> 8); o[pos + 3] = (uint8_t)length; }
      pos += 4;
    }
    if (lane == 0) o[pos] = (uint8_t)ids[i];
    if (lane < 16) o[pos + 1 + lane] = T->bits[lane + 1];
    for (int j = lane; j < n; j += 64) o[pos + 17 + j] = T->huffval[j];
    pos += 17 + n;
  }
  for (int i = lane; i < sos_len; i += 64) o[pos + i] = sos[i];
  pos += sos_len;
  if (lane == 0) meta[img].hdr_len = (unsigned)pos;
}

// after the bit offsets are known: pad the last partial byte with 1-bits (jchuff.c:505-514)
__global__ void __launch_bounds__(64)
k_finish_bits(unsigned *__restrict__ totals, const unsigned *__restrict__ seg_totals, unsigned *__restrict__ stream,
              size_t stream_words_per_image, MjhImageMeta *__restrict__ meta, int nimg)
{
  const int img = blockIdx.x * 64 + threadIdx.x;
  if (img >= nimg) return;
  if (totals[img] == 0xFFFFFFFFu) {   // offsets wrapped: no file (the host turns the marker into MJH_ETOOSMALL)
    totals[img] = 0;
    meta[img].total_bits = 0xFFFFFFFFu;
    return;
  }
  const unsigned tb = totals[img] + (seg_totals ? seg_totals[img] : 0u);
  totals[img] = tb;   // from here on: total bits of the scan including restart padding and markers
  meta[img].total_bits = tb;
  if (tb & 7) {
    const unsigned padbits = 8 - (tb & 7);
    const unsigned bitpos = tb & 31;
    // the pad occupies bits [bitpos, bitpos+padbits) of big-endian word tb>>5
    const unsigned w = (((1u << padbits) - 1u) << (32 - bitpos - padbits));
    atomicOr(&stream[(size_t)img * stream_words_per_image + (tb >> 5)], __builtin_bswap32(w));
  }
}

// number of 0xFF bytes per 4-byte word -> chunk sums (chunk = 2048 words)
__device__ __forceinline__ unsigned ff_count(unsigned w)
{
  // bytes equal to 0xFF: (w & (w>>1) & ... ) trick via per-byte compare
  unsigned c = 0;
  c += ((w & 0xFFu) == 0xFFu);
  c += ((w & 0xFF00u) == 0xFF00u);
  c += ((w & 0xFF0000u) == 0xFF0000u);
  c += ((w & 0xFF000000u) == 0xFF000000u);
  return c;
}

__global__ void __launch_bounds__(256)
k_ff_chunk_sums(const unsigned *__restrict__ stream, size_t stream_words_per_image, const unsigned *__restrict__ totals,
                unsigned *__restrict__ sums, int chunks_per_image, const unsigned *__restrict__ mpos_all, int nseg)
{
  __shared__ unsigned sh[4];
  const int img = blockIdx.y;
  const unsigned nbytes = (totals[img] + 7) >> 3;
  const unsigned nwords = (nbytes + 3) >> 2;
  const unsigned *p = stream + (size_t)img * stream_words_per_image;
  const unsigned *mpos = nseg > 1 ? mpos_all + (size_t)img * nseg : nullptr;
  // the stream buffer is sized for the worst case; only the chunks that hold data are visited
  for (unsigned chunk = blockIdx.x; chunk * SCAN_CHUNK < nwords; chunk += gridDim.x) {
  unsigned s = 0;
  const unsigned base = chunk * SCAN_CHUNK + threadIdx.x * 8;
  MarkerCursor mc(mpos, nseg - 1);
#pragma unroll
  for (int i = 0; i < 8; i++) if (base + i < nwords) {  // bytes past nbytes are zero
    const unsigned w = p[base + i];
    unsigned cnt = ff_count(w);
    if (cnt && mpos) {   // the 0xFF of an RSTn marker is not data: never stuffed
      for (int b = 0; b < 4; b++)
        if (((w >> (8 * b)) & 0xFFu) == 0xFFu && mc.at((base + i) * 4 + b)) cnt--;
    }
    s += cnt;
  }
  const unsigned tot = block_reduce_256(s, sh);
  if (threadIdx.x == 0) sums[(size_t)img * chunks_per_image + chunk] = tot;
  }
}

__global__ void __launch_bounds__(256)
k_stuff_write(const unsigned *__restrict__ stream, size_t stream_words_per_image, const unsigned *__restrict__ totals,
              const unsigned *__restrict__ sums, int chunks_per_image, const unsigned *__restrict__ ff_totals,
              uint8_t *__restrict__ out, size_t out_stride, MjhImageMeta *__restrict__ meta, unsigned *__restrict__ sizes,
              const unsigned *__restrict__ mpos_all, int nseg)
{
  __shared__ unsigned sh[4];
  __shared__ unsigned s_out[STUFF_LDS_WORDS];
  const int img = blockIdx.y;
  const unsigned nbytes = (totals[img] + 7) >> 3;
  const unsigned nwords = (nbytes + 3) >> 2;
  const unsigned *p = stream + (size_t)img * stream_words_per_image;
  const unsigned hdr = meta[img].hdr_len;
  uint8_t *o = out + (size_t)img * out_stride + hdr;
  const unsigned *mpos = nseg > 1 ? mpos_all + (size_t)img * nseg : nullptr;
  for (unsigned chunk = blockIdx.x; chunk * SCAN_CHUNK < nwords; chunk += gridDim.x) {
  const unsigned base = chunk * SCAN_CHUNK + threadIdx.x * 8;
  unsigned w[8], s = 0;
  unsigned mk = 0;   // bit (4*i+b) set: byte b of word i is the 0xFF of a restart marker
  MarkerCursor mc(mpos, nseg - 1);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    w[i] = base + i < nwords ? p[base + i] : 0u;
    unsigned cnt = ff_count(w[i]);
    if (cnt && mpos) {
      for (int b = 0; b < 4; b++)
        if (((w[i] >> (8 * b)) & 0xFFu) == 0xFFu && mc.at((base + i) * 4 + b)) { cnt--; mk |= 1u << (4 * i + b); }
    }
    s += cnt;
  }
  unsigned tot;
  const unsigned before = sums[(size_t)img * chunks_per_image + chunk];      // stuffed bytes in front of the chunk
  const unsigned ex = block_excl_scan_256(s, sh, &tot) + before;
  const unsigned cfirst = chunk * SCAN_CHUNK * 4u, rend = min(cfirst + SCAN_CHUNK * 4u, nbytes);   // input bytes of the round
  const int nvalid = base * 4u < rend ? (int)min(32u, rend - base * 4u) : 0;
  stuff_store_round(o, cfirst + before, rend - cfirst + tot, base * 4u + ex, w, nvalid, mk, s_out);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned stuffed = nbytes + ff_totals[img];
    o[stuffed] = 0xFF;       // EOI, write_file_trailer jcmarker.c:791
    o[stuffed + 1] = 0xD9;
    meta[img].stuffed_len = stuffed;
    meta[img].file_len = hdr + stuffed + 2;
    sizes[img] = hdr + stuffed + 2;
  }
}

// ---- hand-over to the host (mjh_encode_host / mjh_collect): the files of a batch are packed back to back (16-byte
// aligned starts) straight into a pinned, device-mapped arena -- the kernel's stores travel over the host link, so no
// sizes have to reach the host before a copy can be queued.  table: [0] bytes used, [1] error flags (1: offset range /
// pool overflow, 2: internal), then {offset, size} per image (offset ~0: the file did not fit the arena).
__global__ void __launch_bounds__(256)
k_pack_results(const uint8_t *__restrict__ out, size_t out_stride, const unsigned *__restrict__ sizes,
               const MjhImageMeta *__restrict__ meta, const MjhProgCtl *__restrict__ ctl, int n,
               uint8_t *__restrict__ dst, size_t cap, unsigned long long *__restrict__ table)
{
  const int img = blockIdx.y;
  unsigned long long off = 0;
  for (int j = 0; j < img; j++) {
    const unsigned long long r = ((unsigned long long)sizes[j] + 15ull) & ~15ull;
    if (off + r <= cap) off += r;   // a file that does not fit takes no room
  }
  const unsigned sz = sizes[img];
  const unsigned long long r = ((unsigned long long)sz + 15ull) & ~15ull;
  const bool fits = off + r <= cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    table[2 + 2 * img] = fits ? off : ~0ull;
    table[3 + 2 * img] = sz;
    if (img == n - 1) table[0] = off + (fits ? r : 0ull);
    if (img == 0) {
      unsigned long long err = 0;
      for (int j = 0; j < n; j++) {
        if (meta && meta[j].total_bits == 0xFFFFFFFFu) err |= 1ull;
        if (meta && meta[j].bad_coef) err |= 4ull;
        if (ctl && ctl[j].error) err |= ctl[j].error == 1 ? 1ull : 2ull;
      }
      table[1] = err;
    }
  }
  if (!fits) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(out + (size_t)img * out_stride);
  uint4 *d = reinterpret_cast<uint4 *>(dst + off);
  const unsigned nvec = (sz + 15u) >> 4;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) d[i] = src[i];
}

// =============================================================================================
// host-callable launch wrappers (C++ linkage, used by mjh_encoder.cpp)
// =============================================================================================
#include "mjh_launch.h"

static inline dim3 g3(unsigned x, unsigned y, unsigned z) { return dim3(x, y, z); }

void mjh_launch_pack_results(const void *out, size_t out_stride, const unsigned *sizes, const void *meta, const void *prog_ctl, int n,
                             void *dst, size_t cap, void *table, hipStream_t s)
{
  hipLaunchKernelGGL(k_pack_results, dim3(16, n), dim3(256), 0, s, (const uint8_t *)out, out_stride, sizes, (const MjhImageMeta *)meta,
                     (const MjhProgCtl *)prog_ctl, n, (uint8_t *)dst, cap, (unsigned long long *)table);
}

void mjh_launch_import_coefs(const MjhConst &C, const MjhCoefSrc &S, void *coef_q, void *meta, int n, hipStream_t s)
{
  (void)hipMemsetAsync(meta, 0, (size_t)n * sizeof(MjhImageMeta), s);
  int m = 0;
  for (int i = 0; i < C.ncomp; i++) m = C.c[i].nblk > m ? C.c[i].nblk : m;
  dim3 grid((m + 63) / 64, C.ncomp, n);
  hipLaunchKernelGGL(k_import_coefs, grid, dim3(64), 0, s, C, S, (int16_t *)coef_q, (MjhImageMeta *)meta);
}

void mjh_launch_import_planes(const MjhConst &C, const MjhPlaneSrc &S, void *planes, int n, hipStream_t s)
{
  int pw = 0, ph = 0;
  for (int i = 0; i < C.ncomp; i++) { pw = C.c[i].pw > pw ? C.c[i].pw : pw; ph = C.c[i].ph > ph ? C.c[i].ph : ph; }
  dim3 grid((pw / 4 + 255) / 256, ph, n * C.ncomp);
  if (C.precision == 12) hipLaunchKernelGGL((k_import_planes<uint16_t>), grid, dim3(256), 0, s, C, S, (uint16_t *)planes);
  else hipLaunchKernelGGL((k_import_planes<uint8_t>), grid, dim3(256), 0, s, C, S, (uint8_t *)planes);
}

void mjh_launch_color(const MjhConst &C, const void *pix, size_t row_pitch, size_t img_stride, void *planes, int n, hipStream_t s)
{
  const int H0 = C.maxh, V0 = C.maxv;
  if (C.smoothing) {
    int pw = 0, ph = 0;
    for (int i = 0; i < C.ncomp; i++) { pw = C.c[i].pw > pw ? C.c[i].pw : pw; ph = C.c[i].ph > ph ? C.c[i].ph : ph; }
    dim3 grid((pw + 255) / 256, ph, n * C.ncomp);
    if (C.precision == 12) hipLaunchKernelGGL((k_color_smooth<uint16_t>), grid, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint16_t *)planes);
    else hipLaunchKernelGGL((k_color_smooth<uint8_t>), grid, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint8_t *)planes);
    return;
  }
  bool plain = true;       // luma at full resolution, chroma 1x1: what k_color / k_color_vec are written for
  for (int i = 0; i < C.ncomp; i++) plain = plain && (i == 0 ? (C.c[i].h == H0 && C.c[i].v == V0) : (C.c[i].h == 1 && C.c[i].v == 1));
  plain = plain && (H0 == 1 || H0 == 2 || H0 == 4) && (V0 == 1 || V0 == 2 || V0 == 4);      // (3x1 and the like: the generic kernel)
  if (!plain) {
    int pw = 0, ph = 0;
    for (int i = 0; i < C.ncomp; i++) { pw = C.c[i].pw > pw ? C.c[i].pw : pw; ph = C.c[i].ph > ph ? C.c[i].ph : ph; }
    dim3 grid((pw + 255) / 256, ph, n * C.ncomp);
    if (C.precision == 12) hipLaunchKernelGGL((k_color_generic<uint16_t>), grid, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint16_t *)planes);
    else hipLaunchKernelGGL((k_color_generic<uint8_t>), grid, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint8_t *)planes);
    return;
  }
  if (!C.no_ycc && C.precision == 8 && C.in_comps == 3 && C.ncomp == 3 && C.px_size == 3 && H0 == 2 && V0 <= 2 && (row_pitch & 7) == 0 &&
      (img_stride & 7) == 0 && ((uintptr_t)pix & 7) == 0 && C.off_g == 1 && (C.off_r == 0 || C.off_r == 2)) {
    dim3 gridv(((C.groups_x + 3) / 4 + 255) / 256, C.groups_y, n);
    if (V0 == 2) hipLaunchKernelGGL((k_color_vec<2>), gridv, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint8_t *)planes);
    else hipLaunchKernelGGL((k_color_vec<1>), gridv, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint8_t *)planes);
    return;
  }
  dim3 grid((C.groups_x + 255) / 256, C.groups_y, n);
#define LC(h, v) do { if (C.precision == 12) hipLaunchKernelGGL((k_color<h, v, uint16_t>), grid, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint16_t *)planes); \
                      else hipLaunchKernelGGL((k_color<h, v, uint8_t>), grid, dim3(256), 0, s, C, (const uint8_t *)pix, row_pitch, img_stride, (uint8_t *)planes); } while (0)
  if (H0 == 2 && V0 == 2) LC(2, 2);
  else if (H0 == 2 && V0 == 1) LC(2, 1);
  else if (H0 == 1 && V0 == 2) LC(1, 2);
  else if (H0 == 4 && V0 == 1) LC(4, 1);      // 4:1:1 (TJSAMP_411)
  else if (H0 == 1 && V0 == 4) LC(1, 4);      // 4:4:1 (TJSAMP_441)
  else if (H0 == 4 && V0 == 2) LC(4, 2);
  else if (H0 == 2 && V0 == 4) LC(2, 4);
  else LC(1, 1);
#undef LC
}

static int max_nblk(const MjhConst &C) { int m = 0; for (int i = 0; i < C.ncomp; i++) m = C.c[i].nblk > m ? C.c[i].nblk : m; return m; }
static int max_padblk(const MjhConst &C) { int m = 0; for (int i = 0; i < C.ncomp; i++) { int v = C.c[i].wpad * C.c[i].hpad; m = v > m ? v : m; } return m; }

void mjh_launch_dct(const MjhConst &C, const MjhQuant *Q, const void *planes, void *uq, void *q, float *lambda,
                    MjhHuffTable *stat_tabs, int spi, const int stat_slot[4], uint8_t *nq8, int n, hipStream_t s, int fastdiv, int ifast)
{
  const int nb = (stat_tabs && C.precision != 12) ? DCTQ_NB : 1;      // (= the kernel's NB: the STATS instantiations)
  dim3 grid(((max_nblk(C) + 63) / 64 + nb - 1) / nb, C.ncomp, n);
  const int4 sl = stat_tabs ? make_int4(stat_slot[0], stat_slot[1], stat_slot[2], stat_slot[3]) : make_int4(0, 0, 0, 0);
  if (ifast && C.precision == 12) hipLaunchKernelGGL(k_dct_quant_ifast12, grid, dim3(64), 0, s, C, Q, (const uint16_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, nq8);
  else if (ifast) hipLaunchKernelGGL(k_dct_quant_ifast, grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, nq8);   // (no fused statistics: the caller's business)
  else if (C.precision == 12) hipLaunchKernelGGL((k_dct_quant<uint16_t, false>), grid, dim3(64), 0, s, C, Q, (const uint16_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8);
  else if (stat_tabs && fastdiv) hipLaunchKernelGGL((k_dct_quant<uint8_t, true, true>), grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8);
  else if (stat_tabs) hipLaunchKernelGGL((k_dct_quant<uint8_t, true>), grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8);
  else if (fastdiv) hipLaunchKernelGGL((k_dct_quant<uint8_t, false, true>), grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8);
  else hipLaunchKernelGGL((k_dct_quant<uint8_t, false>), grid, dim3(64), 0, s, C, Q, (const uint8_t *)planes, (int16_t *)uq, (int16_t *)q, lambda, stat_tabs, spi, sl, nq8);
}

void mjh_launch_stats_ac(const MjhConst &C, const void *q, const unsigned long long *nzmask, MjhHuffTable *tabs, int spi, const int slot[4], int count_dummies, int n, hipStream_t s)
{
  dim3 grid((max_nblk(C) + 255) / 256, C.ncomp, n), gridc((max_nblk(C) + 256 * STATS_AC_ITER - 1) / (256 * STATS_AC_ITER), C.ncomp, n);
  if (nzmask) hipLaunchKernelGGL(k_stats_ac_compact, gridc, dim3(256), 0, s, C, (const int16_t *)q, nzmask, tabs, spi, make_int4(slot[0], slot[1], slot[2], slot[3]), count_dummies);
  else hipLaunchKernelGGL(k_stats_ac, grid, dim3(256), 0, s, C, (const int16_t *)q, tabs, spi, make_int4(slot[0], slot[1], slot[2], slot[3]), count_dummies);
}

void mjh_launch_stats_dc(const MjhConst &C, const void *q, MjhHuffTable *tabs, int spi, const int slot[4], int mcu_order, const int comp_restart[4], int n, hipStream_t s)
{
  const int4 sl = make_int4(slot[0], slot[1], slot[2], slot[3]);
  const int4 cr = make_int4(comp_restart[0], comp_restart[1], comp_restart[2], comp_restart[3]);
  const int per_wg = 256 * STATS_DC_ITER;
  if (mcu_order) {
    const int rpw = (C.mcu_rows + 127) / 128;       // at most 128 workgroups per (component, image)
    dim3 grid((C.mcu_rows + rpw - 1) / rpw, C.ncomp, n);
    hipLaunchKernelGGL(k_stats_dc_mcu, grid, dim3(256), 0, s, C, (const int16_t *)q, tabs, spi, sl, rpw);
  } else {
    dim3 grid((max_nblk(C) + per_wg - 1) / per_wg, C.ncomp, n);
    hipLaunchKernelGGL(k_stats_dc, grid, dim3(256), 0, s, C, (const int16_t *)q, tabs, spi, sl, cr);
  }
}

void mjh_launch_gen_tables(MjhHuffTable *tabs, int spi, const int *slots, int nslots, int n, hipStream_t s)
{
  int sl[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int i = 0; i < nslots && i < 8; i++) sl[i] = slots[i];
  hipLaunchKernelGGL(k_gen_tables, dim3(nslots, n), dim3(64), 0, s, tabs, spi, make_int4(sl[0], sl[1], sl[2], sl[3]), make_int4(sl[4], sl[5], sl[6], sl[7]));
}

void mjh_launch_gen_tables_list(MjhHuffTable *tabs, int spi, const int *d_slots, int nslots, int n, hipStream_t s)
{
  if (nslots > 0) hipLaunchKernelGGL(k_gen_tables_list, dim3(nslots, n), dim3(64), 0, s, tabs, spi, d_slots);
}

// exclusive prefix sum of 16-bit lengths, `npairs` independent arrays of n_per entries (the progressive path's
// parallel encode reuses the sequential coder's scan kernels)
void mjh_launch_scan16(const void *len16, int n_per, unsigned *sums, int chunks, unsigned *totals, unsigned *off32, int npairs, hipStream_t s)
{
  hipLaunchKernelGGL((k_chunk_sums<uint16_t>), dim3(chunks, npairs), dim3(256), 0, s, (const uint16_t *)len16, n_per, sums, chunks);
  hipLaunchKernelGGL(k_scan_sums, dim3(npairs), dim3(256), 0, s, sums, chunks, totals, (const unsigned *)nullptr);
  hipLaunchKernelGGL((k_offsets<uint16_t>), dim3(chunks, npairs), dim3(256), 0, s, (const uint16_t *)len16, n_per, sums, chunks, off32);
}

bool mjh_trellis_dc_speculative_ok(const MjhConst &C, int window_ok)
{ // the window kernel's conditions, and a component with more than one block row per iMCU row (else there is nothing to speculate on)
  return window_ok && C.delta_dc_weight <= 0.0f && C.maxv >= 2;
}

void mjh_launch_trellis_dc_speculative(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, const MjhHuffTable *tabs, int spi, const int dc_slot[4], const float *lambda,
                                       void *back9, int *jfin, void *qspec, int n, hipStream_t s)
{
  int rows = 0;
  for (int c = 0; c < C.ncomp; c++) rows += C.c[c].hib;
  hipLaunchKernelGGL(k_trellis_dc3_fwd, dim3((rows * DC3_HYP + 3) / 4, n), dim3(64), 0, s, C, Q, (const int16_t *)uq, tabs, spi,
                     make_int4(dc_slot[0], dc_slot[1], dc_slot[2], dc_slot[3]), lambda, (uint8_t *)back9, jfin, (int16_t *)qspec, rows);
  hipLaunchKernelGGL(k_trellis_dc3_resolve, dim3(C.ncomp * C.mcu_rows, n), dim3(64), 0, s, C, (int16_t *)q, (const int *)jfin, (const int16_t *)qspec, rows);
}

void mjh_launch_trellis_dc(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, const MjhHuffTable *tabs, int spi, const int dc_slot[4], const float *lambda, void *back, int n, hipStream_t s,
                           int window_ok, int chain0, int chain1)
{
  const int nchains = C.ncomp * C.mcu_rows;
  if (chain1 < 0 || chain1 > nchains) chain1 = nchains;      // (default: every chain)
  if (chain0 >= chain1) return;
  dim3 grid((chain1 - chain0 + 3) / 4, n);
  // the sliding-window kernel when every DC quantizer step 8q is >= 40 and the vertical-gradient term is off; the general DPP kernel otherwise
  if (window_ok && C.delta_dc_weight <= 0.0f)
    hipLaunchKernelGGL(k_trellis_dc3, grid, dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, tabs, spi, make_int4(dc_slot[0], dc_slot[1], dc_slot[2], dc_slot[3]), lambda, (uint8_t *)back, chain0, chain1);
  else hipLaunchKernelGGL(k_trellis_dc2, grid, dim3(64), 0, s, C, Q, (const int16_t *)uq, (int16_t *)q, tabs, spi, make_int4(dc_slot[0], dc_slot[1], dc_slot[2], dc_slot[3]), lambda, (uint8_t *)back, chain0, chain1);
}

void mjh_launch_encode(const MjhConst &C, const void *q, const unsigned long long *nzmask, const MjhHuffTable *tabs, int spi, const int dc_slot[4], const int ac_slot[4],
                       void *len16, void *off32, unsigned *sums, int chunks_per_image, unsigned *totals,
                       unsigned *stream, size_t stream_words_per_image, void *meta,
                       unsigned *seg_x, unsigned *seg_E, unsigned *seg_sums, unsigned *seg_totals, unsigned *mpos, int nseg,
                       int n, hipStream_t s)
{
  const int4 ds = make_int4(dc_slot[0], dc_slot[1], dc_slot[2], dc_slot[3]);
  const int4 as = make_int4(ac_slot[0], ac_slot[1], ac_slot[2], ac_slot[3]);
  dim3 grid((max_padblk(C) + 255) / 256, C.ncomp, n);
  if (nzmask) hipLaunchKernelGGL((k_enc_len<true>), grid, dim3(256), 0, s, C, (const int16_t *)q, nzmask, tabs, spi, ds, as, (uint16_t *)len16, (MjhImageMeta *)meta);
  else hipLaunchKernelGGL((k_enc_len<false>), grid, dim3(256), 0, s, C, (const int16_t *)q, nzmask, tabs, spi, ds, as, (uint16_t *)len16, (MjhImageMeta *)meta);
  hipLaunchKernelGGL((k_chunk_sums<uint16_t>), dim3(chunks_per_image, n), dim3(256), 0, s, (const uint16_t *)len16, C.total_mcu_blocks, sums, chunks_per_image);
  hipLaunchKernelGGL(k_scan_sums, dim3(n), dim3(256), 0, s, sums, chunks_per_image, totals, (const unsigned *)nullptr);
  hipLaunchKernelGGL((k_offsets<uint16_t>), dim3(chunks_per_image, n), dim3(256), 0, s, (const uint16_t *)len16, C.total_mcu_blocks, sums, chunks_per_image, (unsigned *)off32);
  const bool rst = C.restart_interval != 0 && nseg > 1;
  if (rst) {
    const int seg_chunks = (nseg + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_seg_extra, dim3((nseg + 255) / 256, n), dim3(256), 0, s, C, (const unsigned *)off32, totals, seg_x, nseg);
    hipLaunchKernelGGL((k_chunk_sums<unsigned>), dim3(seg_chunks, n), dim3(256), 0, s, (const unsigned *)seg_x, nseg, seg_sums, seg_chunks);
    hipLaunchKernelGGL(k_scan_sums, dim3(n), dim3(256), 0, s, seg_sums, seg_chunks, seg_totals, (const unsigned *)nullptr);
    hipLaunchKernelGGL((k_offsets<unsigned>), dim3(seg_chunks, n), dim3(256), 0, s, (const unsigned *)seg_x, nseg, seg_sums, seg_chunks, seg_E);
  }
  hipLaunchKernelGGL(k_zero_stream, dim3(64, n), dim3(256), 0, s, stream, stream_words_per_image, (const unsigned *)totals,
                     rst ? (const unsigned *)seg_totals : (const unsigned *)nullptr);
  dim3 gridm((C.total_mcu_blocks + 255) / 256, n);
  if (nzmask) hipLaunchKernelGGL((k_enc_write_mcu<true>), gridm, dim3(256), 0, s, C, (const int16_t *)q, nzmask, tabs, spi, ds, as, (const unsigned *)off32, stream, stream_words_per_image,
                                 (const unsigned *)totals, (const unsigned *)seg_E, (const unsigned *)seg_totals, mpos, rst ? nseg : 1);
  else hipLaunchKernelGGL((k_enc_write_mcu<false>), gridm, dim3(256), 0, s, C, (const int16_t *)q, nzmask, tabs, spi, ds, as, (const unsigned *)off32, stream, stream_words_per_image,
                          (const unsigned *)totals, (const unsigned *)seg_E, (const unsigned *)seg_totals, mpos, rst ? nseg : 1);
  hipLaunchKernelGGL(k_finish_bits, dim3((n + 63) / 64), dim3(64), 0, s, totals, rst ? (const unsigned *)seg_totals : (const unsigned *)nullptr, stream,
                     stream_words_per_image, (MjhImageMeta *)meta, n);
}

// One wave that keeps its hardware queue busy for `ticks` of the 100 MHz clock: mjh_streams_overlap (mjh_encoder.cpp) finds out with
// two of them whether two streams were dealt the same hardware queue
__global__ void __launch_bounds__(64) k_spin(unsigned long long ticks, unsigned *sink)
{
  const unsigned long long t0 = wall_clock64();
  unsigned n = 0;
  while (wall_clock64() - t0 < ticks) n++;
  if (sink && threadIdx.x == 0 && n == 0xFFFFFFFFu) *sink = n;
}
void mjh_launch_spin(unsigned long long ticks, hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, (unsigned *)nullptr); }

void mjh_launch_header(const void *prefix, int prefix_len, const void *sos, int sos_len, const MjhHuffTable *tabs, int spi,
                       const int dht_slots[8], const int dht_ids[8], int ndht, int multi_dht, void *out, size_t out_stride, void *meta, int n, hipStream_t s,
                       const unsigned *append_sizes)
{
  MjhDhtPlan plan;
  for (int i = 0; i < 8; i++) { plan.slots[i] = i < ndht ? dht_slots[i] : 0; plan.ids[i] = i < ndht ? dht_ids[i] : 0; }
  hipLaunchKernelGGL(k_header, dim3(n), dim3(64), 0, s, (const uint8_t *)prefix, prefix_len, (const uint8_t *)sos, sos_len, tabs, spi,
                     plan, ndht, multi_dht, (uint8_t *)out, out_stride, (MjhImageMeta *)meta, append_sizes);
}

void mjh_launch_stuff(const unsigned *stream, size_t stream_words_per_image, const unsigned *totals, unsigned *ffsums, int ff_chunks_per_image,
                      unsigned *ff_totals, void *out, size_t out_stride, void *meta, unsigned *sizes, const unsigned *mpos, int nseg, int n, hipStream_t s)
{
  // workgroups per image: the chunks are walked with a grid stride; a small batch of big images gets more of them
  int gx = 4096 / (n > 0 ? n : 1);
  if (gx < 128) gx = 128;
  if (gx > ff_chunks_per_image) gx = ff_chunks_per_image;
  hipLaunchKernelGGL(k_ff_chunk_sums, dim3(gx, n), dim3(256), 0, s, stream, stream_words_per_image, totals, ffsums, ff_chunks_per_image, mpos, nseg);
  hipLaunchKernelGGL(k_scan_sums, dim3(n), dim3(256), 0, s, ffsums, ff_chunks_per_image, ff_totals, totals);
  hipLaunchKernelGGL(k_stuff_write, dim3(gx, n), dim3(256), 0, s, stream, stream_words_per_image, totals, ffsums, ff_chunks_per_image,
                     ff_totals, (uint8_t *)out, out_stride, (MjhImageMeta *)meta, sizes, mpos, nseg);
}

