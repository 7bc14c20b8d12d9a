// mjh_encoder.cpp -- host side of libmozjpeg_hip.so: parameter capture, geometry, marker
// bytes, device buffers and the kernel schedule.  Everything the reference does on the host
// between its passes (pass scheduling jcmaster.c:612-1035, Huffman table construction, marker
// writing) is either a kernel here (tables, headers) or a fixed launch sequence, so an encode
// of a whole batch never synchronises with the host until the JPEG bytes are fetched.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <memory>
#include <chrono>
#include <vector>

#include "../../include/mozjpeg_hip.h"
#include "mjh_internal.h"
#include "mjh_launch.h"
#include "mjh_guard.h"
#include "mjh_numa.h"
#include "mjh_arith_table.h"

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MJH_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
// Every device allocation of this file goes through mjh_guard.cpp: exactly the bytes asked for, zero-filled -- or, with
// MJH_GUARD=1..3, canaries / an unmapped page next to the buffer (the tool that replaced round 3's 64 KB of blind slack)
#define mjh_dmalloc(pp, bytes) mjh_guard_alloc(reinterpret_cast<void **>(pp), (bytes), #pp, -1)

extern "C" const char *mjh_last_error(void) { return g_err; }
// MJH_GUARD modes: the canaries around every device buffer are compared whenever a batch is waited for
static int guard_verify()
{
  if (mjh_guard_mode() == 0) return MJH_OK;
  char msg[400];
  const int bad = mjh_guard_check(msg, sizeof(msg));
  return bad ? fail(MJH_EHIP, "MJH_GUARD: %d damaged canaries: %s", bad, msg) : MJH_OK;
}
extern "C" int mjh_debug_guard_check(void) { return guard_verify(); }
extern "C" int mjh_debug_guard_mode(void) { return mjh_guard_mode(); }
// (for mjh_pool.cpp, which is otherwise built on the public ABI: lets its argument checks leave a message too)
int mjh_internal_fail(int code, const char *msg) { return fail(code, "%s", msg); }
extern "C" const char *mjh_version(void) { return "mozjpeg_hip 0.3 (gfx950)"; }
extern "C" size_t mjh_params_size(void) { return sizeof(mjh_params); }
extern "C" int mjh_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
extern "C" int mjh_device_numa_node(int device) { return mjh_numa_node_of_device(device); }
extern "C" int mjh_bind_thread_to_device(int device) { return mjh_numa_bind_thread(device); }
extern "C" int mjh_device_placement(int device, char *buf, size_t n) { return buf && n ? mjh_numa_describe(device, buf, n) : 0; }

// zig-zag (jutils.c:59)
static const int kZZ[64] = {
  0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// ---- parameter helpers --------------------------------------------------------------------------
// Base quantization tables: index 0 = Annex K.1 (jcparam.c:76-99,:180-190), 3 = the max-compression default
// (jcparam.c:111-122 == :218-229), 0..8 = what `cjpeg -quant-table N` selects (generated header, data only)
#include "mjh_quant_presets.h"

extern "C" int mjh_params_set_quality(mjh_params *p, int quality, int force_baseline, int base_idx)
{
  if (!p) return fail(MJH_EINVAL, "null params");
  // jpeg_float_quality_scaling jcparam.c:340-357, truncated to int as jpeg_quality_scaling does
  float q = (float)quality;
  if (q <= 0.f) q = 1.f;
  if (q > 100.f) q = 100.f;
  q = q < 50.f ? 5000.f / q : 200.f - q * 2.f;
  const int scale = (int)q;
  if (base_idx < 0) base_idx = p->compress_profile == MJH_PROFILE_FASTEST ? 0 : 3;
  if (base_idx > 8) return fail(MJH_EINVAL, "base quant table index %d (0..8)", base_idx);
  const unsigned *bl = mjh_base_luma[base_idx];
  const unsigned *bc = mjh_base_chroma[base_idx];
  for (int t = 0; t < 2; t++) {
    const unsigned *b = t ? bc : bl;
    for (int i = 0; i < 64; i++) {  // jpeg_add_quant_table jcparam.c:55-64
      long v = ((long)b[i] * scale + 50L) / 100L;
      if (v <= 0) v = 1;
      if (v > 32767) v = 32767;
      if (force_baseline && v > 255) v = 255;
      p->quantval[t][i] = (uint16_t)v;
    }
  }
  return MJH_OK;
}

extern "C" int mjh_params_defaults(mjh_params *p, int width, int height, int input_components,
                                   int gray_output, int profile, int hsamp, int vsamp)
{
  if (!p) return fail(MJH_EINVAL, "null params");
  memset(p, 0, sizeof(*p));
  const bool maxc = profile != MJH_PROFILE_FASTEST;
  p->image_width = width;
  p->image_height = height;
  p->input_components = input_components;
  p->compress_profile = maxc ? MJH_PROFILE_MAX_COMPRESSION : MJH_PROFILE_FASTEST;
  if (input_components == 1 || gray_output) {  // jpeg_set_colorspace jcparam.c:597-602
    p->num_components = 1;
    p->component_id[0] = 1;
    p->h_samp_factor[0] = p->v_samp_factor[0] = 1;
  } else {                                      // :611-619
    p->num_components = 3;
    for (int i = 0; i < 3; i++) {
      p->component_id[i] = i + 1;
      p->h_samp_factor[i] = p->v_samp_factor[i] = 1;
      p->quant_tbl_no[i] = p->dc_tbl_no[i] = p->ac_tbl_no[i] = i > 0;
    }
    p->h_samp_factor[0] = hsamp;
    p->v_samp_factor[0] = vsamp;
  }
  for (int t = 0; t < 2; t++) { p->arith_dc_L[t] = 0; p->arith_dc_U[t] = 1; p->arith_ac_K[t] = 5; }   // jcparam.c:417-419
  p->optimize_coding = maxc;        // jcparam.c:436-444
  p->trellis_quant = maxc;          // :505
  p->trellis_quant_dc = 1;          // :516
  p->overshoot_deringing = maxc;    // :463
  p->lambda_log_scale1 = 14.75f;    // :507-508
  p->lambda_log_scale2 = 16.5f;
  p->trellis_num_loops = 1;         // :515
  p->write_JFIF_header = 1;
  return mjh_params_set_quality(p, 75, 1, -1);
}


// ---- scan scripts (jcparam.c:652-1004) -------------------------------------------------------------
static mjh_scan *fill_a_scan(mjh_scan *s, int ci, int Ss, int Se, int Ah, int Al)
{
  memset(s, 0, sizeof(*s));
  s->comps_in_scan = 1; s->component_index[0] = ci; s->Ss = Ss; s->Se = Se; s->Ah = Ah; s->Al = Al;
  return s + 1;
}
static mjh_scan *fill_dc_scans(mjh_scan *s, int first, int ncomps, int Ah, int Al)
{
  memset(s, 0, sizeof(*s));
  s->comps_in_scan = ncomps;
  for (int ci = 0; ci < ncomps; ci++) s->component_index[ci] = first + ci;
  s->Ss = s->Se = 0; s->Ah = Ah; s->Al = Al;
  return s + 1;
}

extern "C" int mjh_params_simple_progression(mjh_params *p)
{
  if (!p) return fail(MJH_EINVAL, "null params");
  mjh_scan *s = p->scan_info;
  const int nc = p->num_components;
  const bool maxc = p->compress_profile != MJH_PROFILE_FASTEST;
  p->optimize_scans = 0;
  if (nc == 3 && p->color_transform == MJH_COLOR_NONE) {   // all-purpose script for other colour spaces (jcparam.c:985-1003)
    s = fill_dc_scans(s, 0, nc, 0, maxc ? 0 : 1);
    for (int c = 0; c < nc; c++) s = fill_a_scan(s, c, 1, maxc ? 8 : 5, 0, 2);
    for (int c = 0; c < nc; c++) s = fill_a_scan(s, c, maxc ? 9 : 6, 63, 0, 2);
    for (int c = 0; c < nc; c++) s = fill_a_scan(s, c, 1, 63, 2, 1);
    if (!maxc) s = fill_dc_scans(s, 0, nc, 1, 0);
    for (int c = 0; c < nc; c++) s = fill_a_scan(s, c, 1, 63, 1, 0);
  } else if (nc == 3) {
    if (maxc) {   // jcparam.c:931-963
      if (p->dc_scan_opt_mode == 0) s = fill_dc_scans(s, 0, nc, 0, 0);                 // one DC scan for all components
      else if (p->dc_scan_opt_mode == 1) { s = fill_a_scan(s, 0, 0, 0, 0, 0); s = fill_a_scan(s, 1, 0, 0, 0, 0); s = fill_a_scan(s, 2, 0, 0, 0, 0); }
      else { s = fill_dc_scans(s, 0, 1, 0, 0); s = fill_dc_scans(s, 1, 2, 0, 0); }  // luma, then Cb+Cr interleaved
      s = fill_a_scan(s, 0, 1, 8, 0, 2); s = fill_a_scan(s, 1, 1, 8, 0, 0); s = fill_a_scan(s, 2, 1, 8, 0, 0);
      s = fill_a_scan(s, 0, 9, 63, 0, 2);
      s = fill_a_scan(s, 0, 1, 63, 2, 1); s = fill_a_scan(s, 0, 1, 63, 1, 0);
      s = fill_a_scan(s, 1, 9, 63, 0, 0); s = fill_a_scan(s, 2, 9, 63, 0, 0);
    } else {      // :964-982
      s = fill_dc_scans(s, 0, nc, 0, 1);
      s = fill_a_scan(s, 0, 1, 5, 0, 2); s = fill_a_scan(s, 2, 1, 63, 0, 1); s = fill_a_scan(s, 1, 1, 63, 0, 1);
      s = fill_a_scan(s, 0, 6, 63, 0, 2); s = fill_a_scan(s, 0, 1, 63, 2, 1);
      s = fill_dc_scans(s, 0, nc, 1, 0);
      s = fill_a_scan(s, 2, 1, 63, 1, 0); s = fill_a_scan(s, 1, 1, 63, 1, 0); s = fill_a_scan(s, 0, 1, 63, 1, 0);
    }
  } else if (nc == 1) {
    if (maxc) {   // :985-995
      s = fill_dc_scans(s, 0, 1, 0, 0);
      s = fill_a_scan(s, 0, 1, 8, 0, 2); s = fill_a_scan(s, 0, 9, 63, 0, 2);
      s = fill_a_scan(s, 0, 1, 63, 2, 1); s = fill_a_scan(s, 0, 1, 63, 1, 0);
    } else {      // :996-1003
      s = fill_dc_scans(s, 0, 1, 0, 1);
      s = fill_a_scan(s, 0, 1, 5, 0, 2); s = fill_a_scan(s, 0, 6, 63, 0, 2);
      s = fill_a_scan(s, 0, 1, 63, 2, 1);
      s = fill_dc_scans(s, 0, 1, 1, 0);
      s = fill_a_scan(s, 0, 1, 63, 1, 0);
    }
  } else return fail(MJH_EUNSUPPORTED, "scan scripts for %d components", nc);
  p->num_scans = (int)(s - p->scan_info);
  p->optimize_coding = 1;   // jcmaster.c:1091-1094
  return MJH_OK;
}

static int build_search_script(mjh_scan *out, int nc, int dc_scan_opt_mode)
{
  static const int fs[5] = { 2, 8, 5, 12, 18 };
  mjh_scan *s = out;
  s = fill_dc_scans(s, 0, dc_scan_opt_mode == 0 ? nc : 1, 0, 0);   // jcparam.c:791-794
  s = fill_a_scan(s, 0, 1, 8, 0, 0); s = fill_a_scan(s, 0, 9, 63, 0, 0);
  for (int Al = 0; Al < 3; Al++) {
    s = fill_a_scan(s, 0, 1, 63, Al + 1, Al); s = fill_a_scan(s, 0, 1, 8, 0, Al + 1); s = fill_a_scan(s, 0, 9, 63, 0, Al + 1);
  }
  s = fill_a_scan(s, 0, 1, 63, 0, 0);
  for (int i = 0; i < 5; i++) { s = fill_a_scan(s, 0, 1, fs[i], 0, 0); s = fill_a_scan(s, 0, fs[i] + 1, 63, 0, 0); }
  if (nc == 3) {
    s = fill_dc_scans(s, 1, 2, 0, 0);
    s = fill_a_scan(s, 1, 0, 0, 0, 0); s = fill_a_scan(s, 2, 0, 0, 0, 0);
    s = fill_a_scan(s, 1, 1, 8, 0, 0); s = fill_a_scan(s, 1, 9, 63, 0, 0);
    s = fill_a_scan(s, 2, 1, 8, 0, 0); s = fill_a_scan(s, 2, 9, 63, 0, 0);
    for (int Al = 0; Al < 2; Al++) {
      s = fill_a_scan(s, 1, 1, 63, Al + 1, Al); s = fill_a_scan(s, 2, 1, 63, Al + 1, Al);
      s = fill_a_scan(s, 1, 1, 8, 0, Al + 1); s = fill_a_scan(s, 1, 9, 63, 0, Al + 1);
      s = fill_a_scan(s, 2, 1, 8, 0, Al + 1); s = fill_a_scan(s, 2, 9, 63, 0, Al + 1);
    }
    s = fill_a_scan(s, 1, 1, 63, 0, 0); s = fill_a_scan(s, 2, 1, 63, 0, 0);
    for (int i = 0; i < 5; i++) {
      s = fill_a_scan(s, 1, 1, fs[i], 0, 0); s = fill_a_scan(s, 1, fs[i] + 1, 63, 0, 0);
      s = fill_a_scan(s, 2, 1, fs[i], 0, 0); s = fill_a_scan(s, 2, fs[i] + 1, 63, 0, 0);
    }
  }
  return (int)(s - out);
}

extern "C" int mjh_params_search_progression(mjh_params *p)
{
  if (!p) return fail(MJH_EINVAL, "null params");
  if (p->num_components != 3 && p->num_components != 1) return fail(MJH_EUNSUPPORTED, "scan search for %d components", p->num_components);
  if (p->num_components == 3 && p->color_transform == MJH_COLOR_NONE)   // jpeg_search_progression knows YCbCr and gray only (jcparam.c:749-757)
    return mjh_params_simple_progression(p);
  p->num_scans = build_search_script(p->scan_info, p->num_components, p->dc_scan_opt_mode);
  p->optimize_scans = 1;
  p->optimize_coding = 1;
  return MJH_OK;
}

// ---- encoder object -------------------------------------------------------------------------------
enum { SLOTS_BASE = 16, SLOT_FINAL = 8, SLOT_PROG = 16 };   // 0..7: per-component trellis-pass tables (DC,AC); 8..15: final DC t / AC t


struct mjh_encoder {
  mjh_params p;
  MjhConst C;
  int device = 0;
  int max_batch = 0;
  int last_n = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr, side_stream = nullptr;
  hipEvent_t copy_done = nullptr, ev_fork = nullptr, ev_join = nullptr, ev_side0 = nullptr, ev_side1 = nullptr;
  bool side_timed = false;
  // device buffers
  // mjh_encode_host: everything double-buffered (index = host_calls & 1) so that the H2D copy of batch n+1, the
  // kernels of batch n and the hand-over of the files of batch n-1 overlap
  uint8_t *d_pixb[2] = { nullptr, nullptr };     // device-side input pixels
  uint8_t *h_stage[2] = { nullptr, nullptr };    // pinned staging for callers whose pixels are in pageable memory
  uint8_t *h_res[2] = { nullptr, nullptr };      // pinned, device-mapped result arenas: the files of a batch packed back to back
  unsigned long long *h_tab[2] = { nullptr, nullptr };   // [0] bytes used, [1] error flags, then {offset, size} per image
  size_t res_cap = 0;
  hipEvent_t ev_h2d[2] = { nullptr, nullptr }, ev_pix_free[2] = { nullptr, nullptr }, ev_packed[2] = { nullptr, nullptr };
  hipStream_t d2h_stream = nullptr;
  unsigned host_calls = 0;
  size_t staged[2] = { 0, 0 };     // mjh_stage_commit: bytes of the staging buffer whose host->device copy is queued already
  int res_buf = -1;                // arena that holds the results of the last batch (-1: encoded through a *_device entry)
  int res_n[2] = { 0, 0 };         // images in each arena (0: nothing there)
  bool res_waited[2] = { false, false };
  uint8_t *d_plin = nullptr, *h_plin = nullptr;   // the same for mjh_encode_planes_host
  uint8_t *d_cfin = nullptr, *h_cfin = nullptr;   // and for mjh_encode_coefficients_host
  unsigned *d_prog_ffsums = nullptr;              // stuffed-byte counts of the shares of every scan (k_prog_stuff)
  unsigned *d_prog_mpos = nullptr;                // progressive + restart intervals: byte positions of the RSTn markers of every scan
  int mpos_per_image = 0;
  size_t pix_image_bytes = 0;
  uint8_t *d_planes = nullptr;
  int16_t *d_uq = nullptr, *d_q = nullptr, *d_q0 = nullptr;
  MjhQuant *d_quant = nullptr, *d_quant_init = nullptr;   // trellis_q_opt: d_quant holds one table set per image, d_quant_init the parameters' tables
  MjhHuffTable *d_tabs = nullptr, *d_tabs_init = nullptr;
  float *d_lambda = nullptr;
  uint8_t *d_back = nullptr;
  // SURVEY 8f row 4 options: per-block outputs of the AC trellis for trellis_eob_opt, 64-bit sums of trellis_q_opt
  void *d_eob_cost = nullptr; int *d_eob_has = nullptr; long long *d_qsums = nullptr;
  // compact coefficient records between the AC trellis and the sequential coder: non-zero position masks;
  // the values live in the AC planes of d_q, plane i+1 = i-th non-zero.  compact_last: the last batch's d_q is in that form
  unsigned long long *d_nzmask = nullptr; bool use_compact = false, compact_last = false;
  size_t small_batch = 400000;       // batches of fewer blocks run the AC trellis' first tier with one pass per tile
  uint8_t *d_nq8 = nullptr;          // per block: non-zero conventionally quantized AC coefficients (FDCT kernel) = tile-sort key of the AC trellis
  int copy_prio = 0;
  int fastdiv_all = 0;               // every table in use has q <= 255: the kernels divide by 8q with one multiply-high (MjhQuant.mdiv)
  int dc_mode = 0;
  int dc_late = 1;                   // large sequential batches: the DC chains of components >= dc_late (1: both chroma components, 2: Cr only) run behind the AC kernel, under the tail of small kernels (MJH_DC_LATE=0: all next to it)
  int dc_stats_side = 1;             // the final DC statistics run on the side stream behind the DC trellis (MJH_DC_STATS_SIDE=0: main stream)
  int dc_window_ok = 0;              // every component's DC quantizer step 8q >= 40: the DC trellis may use its sliding-window kernel
  int trellis_chunks = 0;            // image ranges of the tile-sorted first tier (MJH_TRELLIS_CHUNKS; 0 = by batch size): the general tiers of range c run next to the first tier of range c + 1
  hipEvent_t ev_chunk[4] = { nullptr, nullptr, nullptr, nullptr };
  int trellis_v3 = 4;                // passes per tile of the tile-sorted first tier (MJH_TRELLIS_V3; 0 = the general kernel)
  int dqt_off[4] = { -1, -1, -1, -1 };      // file offset of the first entry of every 8-bit DQT table
  int dqt_tabs[4] = { 0, 0, 0, 0 }, dqt_ntab = 0;   // quantization tables in DQT marker order (first use by a component)
  bool tbl_le1 = true;                      // every Huffman table number is 0 or 1 (a baseline-capable frame, jcmarker.c:699-710)
  int nbands = 1, freq_split = 8;
  unsigned *d_seg_x = nullptr, *d_seg_E = nullptr, *d_seg_sums = nullptr, *d_seg_totals = nullptr, *d_mpos = nullptr;
  int nseg = 1;
  int comp_restart[4] = { 0, 0, 0, 0 };
  int16_t *d_dense = nullptr; unsigned dense_cap = 0;       // raw coefficients of deferred blocks, 64 int16 per work-list slot
  unsigned *d_worklist = nullptr, *d_worklist2 = nullptr;   // deferred trellis blocks: [0] = count, [4+3i..6+3i] = (image, comp<<28|block, dense slot)
  int trellis_variant = 0;           // first-tier queue capacity of the AC trellis: 0 = 16, 1 = 20, 2 = 24, 3 = 32, 4 = 48 (all bit-identical)
  bool trellis_adapt = true;         // no MJH_TRELLIS_VARIANT given: follow the share of deferred blocks of the previous batches
  int defer_scale = 1;               // the counts of that pass come from one tile in defer_scale
  unsigned *h_defer = nullptr;       // pinned: work-list counters of an earlier trellis pass (count_heavy), read back asynchronously
  hipEvent_t ev_defer = nullptr; bool defer_pending = false; int defer_frames = 0;   // ... valid once ev_defer has completed; the frames they were counted over
  int spi = SLOTS_BASE;             // table slots per image (16 + 2 per progressive scan)
  // progressive mode
  bool progressive = false;
  int nscans = 0;                   // script scans; the per-component trellis statistics scans follow
  void *d_prog_scans = nullptr, *d_prog_ctl = nullptr;
  int *d_lists = nullptr;           // device copies of the scan / slot lists below
  std::vector<int> h_lists;
  // scans of one phase + the table slots they build; for the statistics the scans are split into AC-first scans
  // without restart intervals (parallel kernel) and the rest (sequential per-scan walk)
  struct PList { int scan_off, nscan, slot_off, nslot, par_off, npar, nacf, seq_off, nseq; bool any_refine; };
  void *d_prog_chunks = nullptr;     // chunk summaries of the parallel statistics / encode kernels
  int chunks_per_scan = 0;
  MjhProgPE pe{};                    // buffers of the parallel AC-first encode
  PList pl_trellis[2]{}, pl_phase[4]{}, pl_trellis_c[2][4]{};   // pl_trellis_c: per component (trellis_q_opt walks component-major)   // pl_trellis: the statistics scans of the trellis passes, one list per band
  int nphases = 0;
  unsigned *d_pool = nullptr; size_t pool_words = 0;
  uint8_t *d_outpool = nullptr; size_t outpool_bytes = 0;
  uint8_t *d_frame_hdr = nullptr; int frame_hdr_len = 0, file_hdr_len = 0;
  uint16_t *d_len16 = nullptr;
  unsigned *d_off32 = nullptr, *d_sums = nullptr, *d_totals = nullptr, *d_ffsums = nullptr, *d_fftotals = nullptr;
  unsigned *d_stream = nullptr;
  size_t stream_words = 0;         // per image
  int chunks = 0, ff_chunks = 0;
  uint8_t *d_out = nullptr;
  size_t out_stride = 0;
  unsigned *d_sizes = nullptr;
  void *d_meta = nullptr;
  uint8_t *d_prefix = nullptr, *d_sos = nullptr;
  int prefix_len = 0, sos_len = 0;
  // a SEQUENTIAL script of several scans (cjpeg -scans with whole-block scans, validate_script jcmaster.c:309-330): every scan is
  // coded through a view of the geometry that holds its components, with its own statistics / tables / restart interval;
  // its [DRI +] SOS bytes lie at sos_off of d_sos
  struct SeqScan { int ncomp; int comp[4]; int sos_off, sos_len; int dht_slots[8], dht_ids[8], ndht; int ri, nseg; };
  std::vector<SeqScan> seq_scans;
  int dht_slots[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dht_ids[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ndht = 0;
  bool debug_taps = false;
  int dc_chain_v = 1;           // ... and its vertical factor: the trellis passes' iMCU rows hold that many block rows (the DC chains span them)
  int sof_hv0 = 0;              // one component sampled other than 1x1: the SOF's sampling byte (the geometry is 1x1's, see check_supported)
  bool fdct_div_zero = false;   // a quantization step of 8192 / 16384 / 24576 with 8-bit samples: the reference's FDCT manager divides by zero (pixel / plane input is refused, coefficient input is fine)
  // profiling: 0 off, 1 every kernel, 2 only the dominant kernel (prof_focus).  Events accumulate over the
  // encode calls since the last read (prof_calls), every call records the same sequence of marks.
  int profiling = 0;
  const char *prof_focus = "trellis_ac";
  std::string prof_focus_name;      // mjh_set_profiling_focus: the caller's choice (normally the largest entry of a level-1 breakdown)
  int prof_calls = 0;
  size_t prof_per_call = 0;
  std::vector<hipEvent_t> side_events;   // 2 per call: the DC trellis on the side stream
  std::vector<std::string> prof_names;
  std::vector<const char *> prof_cnames;
  std::vector<float> prof_ms;
  std::vector<hipEvent_t> prof_events;
  std::vector<unsigned> h_sizes;
  bool sizes_valid = false;
  bool coef_input = false;         // last batch came in through mjh_encode_coefficients_*: d_meta[].bad_coef is meaningful
  hipStream_t last_stream = nullptr;   // the stream the last batch was queued on (mjh_encoder_sync waits for it)
  // Two batches in flight (mjh_encode_device on the encoder's own stream): consecutive calls alternate between this encoder's
  // buffers and streams and those of a TWIN (a second set, made at the first such call), so that the latency-bound tail of
  // batch k (final statistics, bit lengths, prefix sums, bit writing, stuffing) and the front end of batch k+1 (colour, FDCT)
  // share the chip, while the VALU-bound AC trellis of either batch has it for itself (events below).
  mjh_params p_created;               // the parameters the caller passed to mjh_encoder_create (p is what the encoder made of them)
  mjh_encoder *twin = nullptr;        // primary only
  mjh_encoder *owner = nullptr;       // twin only
  mjh_encoder *last = nullptr;        // primary only: who ran the most recent batch (nullptr = this encoder)
  int inflight = 2, inflight_mode = 1;   // MJH_INFLIGHT (1 = one batch at a time), MJH_INFLIGHT_MODE (0 = no ordering between the two sets, 1 = nothing next to a set's AC trellis kernel, 2 = only the colour kernel and the byte stuffing)
  unsigned dev_calls = 0;
  bool high_priority_streams = false; // twin: its streams come from the high-priority queue pool, never the primary's hardware queues
  hipEvent_t ev_done = nullptr, ev_tier1 = nullptr;   // end of this encoder's last pipeline / of its tile-sorted AC trellis kernel
  bool ev_done_set = false, ev_tier1_set = false;
  hipEvent_t ev_null_in = nullptr;
  // arithmetic coding (mjh_arith.hip): the scans to code (the script, or one synthetic whole-block scan for a sequential file),
  // phase lists in d_lists (pl_phase[].scan_off / nscan), the rate table of quantize_trellis_arith
  // DC trellis of one or two frames: block rows walked speculatively (k_trellis_dc3_fwd / _resolve): scratch, and MJH_DC_SPEC=0 turns it off
  uint8_t *d_back9 = nullptr; int *d_jfin = nullptr; int16_t *d_qspec = nullptr; int dc_spec = 1;
  bool arith = false;
  int arith_nscans = 0;
  void *d_arith_rates = nullptr;
  void *g_in[MJH_MAX_COMPS] = { nullptr, nullptr, nullptr, nullptr }; size_t g_in_bytes[MJH_MAX_COMPS] = { 0, 0, 0, 0 }; std::vector<void *> g_in_old;   // MJH_GUARD=2/3: fenced copies of the caller's device input
};

static long div_round_up(long a, long b) { return (a + b - 1) / b; }

// trellis_q_opt with the arithmetic coder.  The coder's trellis passes all select component 0 (no statistics pass sits between
// them and prepare_for_pass's trellis_pass case does not re-select the scan: jcmaster.c:686-702, :1001-1005); they are passes
// 0 .. T-1 with T = pass_number_scan_opt_base = (1 or 2) * num_components * trellis_num_loops + 1 (jcmaster.c:1135-1138, :1010), the
// sums are zeroed in front of every pass with number % M == 1 and the tables re-estimated behind every pass with
// (number + 1) % M == 0, M = (2 or 4) * num_components (:687-698, :1016-1030).  Passes with the same tables repeat each other and
// sums of k identical passes give the same quotient as the sums of one, so what the reference does amounts to: U = T / M times
// [trellis pass, estimate component 0's table from it], then one more pass with the last estimate if T > U * M (else the last
// estimate only reaches the DQT marker: a gray image with an odd number of loops).
static int arith_qopt_passes(const mjh_params *p) { return (p->use_scans_in_trellis ? 2 : 1) * p->num_components * (p->trellis_num_loops > 1 ? p->trellis_num_loops : 1) + 1; }
static int arith_qopt_updates(const mjh_params *p) { return arith_qopt_passes(p) / ((p->use_scans_in_trellis ? 4 : 2) * p->num_components); }

// The reference's one-marker DHT writer of the max-compression profile (emit_multi_dht jcmarker.c:293-401) sizes the marker in a
// loop that `continue`s past a component's AC table when its DC table was seen (or sent) before, and then writes every table not
// yet sent with the value count of THAT loop: a component whose DC table is an earlier component's while its AC table is new gets
// the AC table's 17 header bytes written with no values, outside the marker's length -- a corrupt file (djpeg: "17 extraneous bytes
// before marker 0xda"; found by tools/simt/fuzz_api.py, dc tables 1,1,1 with ac tables 0,0,1).  Such a table assignment is
// refused here rather than answered with a file of either kind.  comps: the components of one whole-block scan, in scan order.
// dseen / aseen: the tables seen in this scan or sent by an earlier one (optimal tables are made anew for every scan, the Annex K
// tables are sent once: jchuff.c finish_pass_gather / emit_dht's sent_table).
// Class of component c's DC table inside a progressive image: the progressive kernels keep two DC tables per scan, so the image's
// distinct DC table numbers are numbered in order of first use (0, 1; 2 and up: refused by check_supported).  Until round 6 the class
// was the low bit of the number, which refused 0 with 2 and 1 with 3.
static int dc_class(const mjh_params *p, int c)
{
  int seen[4], n = 0;
  for (int i = 0; i <= c; i++) {
    int k = 0;
    while (k < n && seen[k] != p->dc_tbl_no[i]) k++;
    if (k == n) seen[n++] = p->dc_tbl_no[i];
    if (i == c) return k;
  }
  return 0;
}

// see check_supported
static bool trellis_without_optimize_is_optimize(const mjh_params *p)
{
  return p->num_components == 1 && p->num_scans == 0 && !p->use_scans_in_trellis && !(p->trellis_q_opt && p->trellis_num_loops > 1);
}

static bool dht_writer_would_corrupt(const mjh_params *p, const int *comps, int k, bool dseen[4], bool aseen[4])
{
  for (int j = 0; j < k; j++) {
    const int d = p->dc_tbl_no[comps[j]] & 3, a = p->ac_tbl_no[comps[j]] & 3;
    if (dseen[d]) { if (!aseen[a]) return true; continue; }
    dseen[d] = true;
    aseen[a] = true;
  }
  return false;
}

static int check_supported(const mjh_params *p)
{
  if (p->image_width <= 0 || p->image_height <= 0 || p->image_width > 65500 || p->image_height > 65500)
    return fail(MJH_EINVAL, "bad image size %dx%d", p->image_width, p->image_height);
  if (p->data_precision != 0 && p->data_precision != 8 && p->data_precision != 12) return fail(MJH_EUNSUPPORTED, "data_precision %d", p->data_precision);
  if (p->smoothing_factor < 0 || p->smoothing_factor > 100) return fail(MJH_EINVAL, "smoothing_factor %d (0..100)", p->smoothing_factor);
  if (p->trellis_num_loops < 0 || p->trellis_num_loops > 16) return fail(MJH_EINVAL, "trellis_num_loops %d (0..16)", p->trellis_num_loops);
  if (p->trellis_freq_split < 0 || p->trellis_freq_split > 63) return fail(MJH_EINVAL, "trellis_freq_split %d (0..63)", p->trellis_freq_split);
  if (p->dc_scan_opt_mode < 0 || p->dc_scan_opt_mode > 2) return fail(MJH_EINVAL, "dc_scan_opt_mode %d (0..2)", p->dc_scan_opt_mode);
  if (!(p->trellis_delta_dc_weight == p->trellis_delta_dc_weight)) return fail(MJH_EINVAL, "trellis_delta_dc_weight is not a number");
  if (p->data_precision == 12 && p->trellis_quant)
    return fail(MJH_EUNSUPPORTED, "trellis quantization is 8-bit only in the reference (jccoefct.c:132-138: 12-bit + trellis aborts)");
  if (p->input_components != 1 && p->input_components != 3) return fail(MJH_EUNSUPPORTED, "input_components %d (RGB or gray only)", p->input_components);
  if (p->num_components != 1 && p->num_components != 3) return fail(MJH_EUNSUPPORTED, "num_components %d", p->num_components);
  if (p->num_components == 3 && p->input_components != 3) return fail(MJH_EUNSUPPORTED, "gray input cannot produce 3 components");
  if (p->color_transform != MJH_COLOR_YCC && p->color_transform != MJH_COLOR_NONE && p->color_transform != MJH_COLOR_YCC_IN) return fail(MJH_EINVAL, "color_transform %d", p->color_transform);
  if (p->color_transform == MJH_COLOR_NONE && p->num_components != 3) return fail(MJH_EUNSUPPORTED, "MJH_COLOR_NONE needs three components");
  if (p->color_transform == MJH_COLOR_YCC_IN && p->input_components != 3) return fail(MJH_EINVAL, "MJH_COLOR_YCC_IN needs three input samples per pixel");   // (one component out: grayscale_convert takes the Y samples, jccolor.c:448-466)
  if (p->num_components == 1) {
    // One component: its only scans are non-interleaved (per_scan_setup jcmaster.c:548-575: an MCU is one block, no dummy blocks) and
    // max_samp = its own factors (initial_setup :210-259), so the factors change nothing but the SOF byte -- cjpeg sets 2x1 on a
    // gray image for qualities 80..89 (set_quality_ratings rdswitch.c:566-570) -- EXCEPT through the trellis passes: they walk iMCU
    // rows of V block rows (compress_trellis_pass jccoefct.c:418-441: lastDC and the row above chain over the V rows), which is
    // what the DC trellis kernels do for a component with v > 1 anyway: they get a view of the geometry with that v (dc_chain_v).
    const int h = p->h_samp_factor[0], v = p->v_samp_factor[0];
    if (h < 1 || h > 4 || v < 1 || v > 4) return fail(MJH_EINVAL, "sampling factor %dx%d outside 1..4 (JERR_BAD_SAMPLING)", h, v);
  }
  if (p->num_components == 3) {
    // any sampling factors the reference takes (initial_setup jcmaster.c:210-259, jinit_downsampler jcsample.c:486-535): 1..4 each,
    // every component's factor divides the largest (no fractional downsampling), at most 10 blocks per MCU
    // (C_MAX_BLOCKS_IN_MCU, jcmaster.c:540-544).  4:4:4, 4:2:2, 4:4:0, 4:2:0, 4:1:1, 4:4:1 and the 4x2 / 2x4 ratios have their own
    // colour kernels; everything else (chroma other than 1x1, luma smaller than chroma) takes the generic one.
    int maxh = 1, maxv = 1, blocks = 0;
    for (int i = 0; i < 3; i++) {
      const int h = p->h_samp_factor[i], v = p->v_samp_factor[i];
      if (h < 1 || h > 4 || v < 1 || v > 4) return fail(MJH_EINVAL, "sampling factors %dx%d of component %d (1..4, jcmaster.c:216-218)", h, v, i);
      maxh = h > maxh ? h : maxh; maxv = v > maxv ? v : maxv;
      blocks += h * v;
    }
    for (int i = 0; i < 3; i++)
      if (maxh % p->h_samp_factor[i] || maxv % p->v_samp_factor[i]) return fail(MJH_EUNSUPPORTED, "fractional sampling ratio (the reference: JERR_FRACT_SAMPLE_NOTIMPL, jcsample.c:531)");
    if (blocks > 10) return fail(MJH_EINVAL, "%d blocks per MCU (at most 10, jcmaster.c:540-544)", blocks);
  }
  if (p->input_components == 3) {
    const int ps = p->input_pixel_size ? p->input_pixel_size : 3;
    if (ps != 3 && ps != 4) return fail(MJH_EINVAL, "input_pixel_size %d", ps);
    for (int i = 0; i < 3; i++) if (p->rgb_offset[i] < 0 || p->rgb_offset[i] >= ps) return fail(MJH_EINVAL, "rgb_offset out of range");
  } else if (p->input_pixel_size > 1) return fail(MJH_EINVAL, "grayscale input has 1 byte per pixel");
  for (int i = 0; i < p->num_components; i++) {
    if (p->quant_tbl_no[i] < 0 || p->quant_tbl_no[i] > 3 || p->dc_tbl_no[i] < 0 || p->dc_tbl_no[i] > 3 || p->ac_tbl_no[i] < 0 || p->ac_tbl_no[i] > 3)
      return fail(MJH_EINVAL, "table number out of range");
    for (int k = 0; k < 64; k++)
      if (p->quantval[p->quant_tbl_no[i]][k] == 0) return fail(MJH_EINVAL, "quantization table %d has a zero entry", p->quant_tbl_no[i]);
  }
  if (p->num_scans < 0 || p->num_scans > MJH_MAX_SCANS) return fail(p->num_scans < 0 ? MJH_EUNSUPPORTED : MJH_EINVAL, "num_scans %d", p->num_scans);
  if (p->num_scans > 0) {
    // progressive mode: the checks of validate_script (jcmaster.c:269-432) that matter here
    if (!p->optimize_coding && !p->arith_code) return fail(MJH_EUNSUPPORTED, "progressive mode forces optimize_coding (jcmaster.c:1091-1094)");

    // the progressive kernels keep two DC tables per scan (dc_class): any AC table numbers, and any two DC table numbers per image
    if (!p->arith_code)
      for (int i = 0; i < p->num_components; i++)
        if (dc_class(p, i) > 1)
          return fail(MJH_EUNSUPPORTED, "progressive mode: more than two different DC table numbers in one image (the kernels keep two DC tables per scan)");
    if (p->optimize_scans) {
      mjh_scan ref[MJH_MAX_SCANS];
      // scan 0 is whatever dc_scan_opt_mode was when the script was built (all components, or the luma alone,
      // jcparam.c:791-794) -- an application may change the mode afterwards, the final choice reads the current one
      const int n = build_search_script(ref, p->num_components, 0);
      const mjh_scan &s0 = p->scan_info[0];
      const bool s0_ok = s0.Ss == 0 && s0.Se == 0 && s0.Ah == 0 && s0.Al == 0 && s0.component_index[0] == 0 &&
                         (s0.comps_in_scan == 1 || (s0.comps_in_scan == p->num_components && memcmp(&ref[0], &s0, sizeof(mjh_scan)) == 0));
      if (n != p->num_scans || !s0_ok || memcmp(ref + 1, p->scan_info + 1, sizeof(mjh_scan) * (n - 1)) != 0)
        return fail(MJH_EUNSUPPORTED, "optimize_scans needs the jpeg_search_progression script");
    }
    bool dc_seen[MJH_MAX_COMPS] = { false, false, false, false };
    // the successive-approximation bookkeeping of validate_script (jcmaster.c:364-384): Al of the last scan that carried a
    // coefficient, -1 = not sent yet; a script with the scan search on is not validated at all (:286-292)
    int last_bitpos[MJH_MAX_COMPS][64];
    for (int c = 0; c < MJH_MAX_COMPS; c++) for (int k = 0; k < 64; k++) last_bitpos[c][k] = -1;
    for (int si = 0; si < p->num_scans; si++) {
      const mjh_scan &sc = p->scan_info[si];
      if (sc.comps_in_scan < 1 || sc.comps_in_scan > p->num_components) return fail(MJH_EINVAL, "scan %d: component count", si);
      for (int ci = 0; ci < sc.comps_in_scan; ci++) {
        const int c = sc.component_index[ci];
        if (c < 0 || c >= p->num_components || (ci > 0 && c <= sc.component_index[ci - 1])) return fail(MJH_EINVAL, "scan %d: component order", si);
      }
      if (sc.Ss < 0 || sc.Ss > 63 || sc.Se < sc.Ss || sc.Se > 63 || sc.Ah < 0 || sc.Ah > (p->data_precision == 12 ? 13 : 10) || sc.Al < 0 || sc.Al > (p->data_precision == 12 ? 13 : 10))
        return fail(MJH_EINVAL, "scan %d: bad progression parameters", si);
      if (sc.Ss == 0 && sc.Se != 0) return fail(MJH_EUNSUPPORTED, "scan %d: sequential multi-scan scripts are not supported", si);
      if (sc.Ss != 0 && sc.comps_in_scan != 1) return fail(MJH_EINVAL, "scan %d: AC scans are single-component", si);
      for (int ci = 0; ci < sc.comps_in_scan; ci++) {
        if (sc.Ss == 0) dc_seen[sc.component_index[ci]] = true;
        else if (!dc_seen[sc.component_index[ci]] && !p->optimize_scans) return fail(MJH_EINVAL, "scan %d: AC before DC", si);
        if (!p->optimize_scans)
          for (int k = sc.Ss; k <= sc.Se; k++) {
            int &lb = last_bitpos[sc.component_index[ci]][k];
            if (lb < 0 ? sc.Ah != 0 : (sc.Ah != lb || sc.Al != sc.Ah - 1))
              return fail(MJH_EINVAL, "scan %d: a first scan with Ah != 0, or a refinement that does not continue at the bit the last scan of the coefficient left (JERR_BAD_PROG_SCRIPT, jcmaster.c:371-381)", si);
            lb = sc.Al;
          }
      }
    }
    if (!p->optimize_scans)
      for (int c = 0; c < p->num_components; c++)
        if (!dc_seen[c]) return fail(MJH_EINVAL, "the script sends no DC data for component %d (JERR_MISSING_DATA, jcmaster.c:420-432)", c);
  }
  if (p->restart_interval > 65535u || p->restart_in_rows < 0) return fail(MJH_EINVAL, "bad restart interval");
  if (p->arith_code) {
    for (int i = 0; i < p->num_components; i++)
      if (p->dc_tbl_no[i] > 1 || p->ac_tbl_no[i] > 1) return fail(MJH_EUNSUPPORTED, "arithmetic coding: conditioning table numbers 0/1 only");
    for (int t = 0; t < 2; t++) {
      if (p->arith_dc_L[t] == 0 && p->arith_dc_U[t] == 0 && p->arith_ac_K[t] == 0) continue;   // a zeroed table: the defaults (mozjpeg_hip.h)
      if (p->arith_dc_L[t] < 0 || p->arith_dc_U[t] > 15 || p->arith_dc_L[t] > p->arith_dc_U[t] || p->arith_ac_K[t] < 1 || p->arith_ac_K[t] > 63)
        return fail(MJH_EINVAL, "arithmetic conditioning of table %d: L %d, U %d, K %d (0 <= L <= U <= 15, 1 <= K <= 63: T.81 B.2.4.3)", t, p->arith_dc_L[t], p->arith_dc_U[t], p->arith_ac_K[t]);
    }
  }
  // The trellis with optimize_coding switched off by hand: the reference's passes below pass_number_scan_opt_base select ONE
  // component each, pass_number / (2 loops), written for the (statistics, trellis) pairs of optimize_coding (jcmaster.c:451-466,
  // :1128-1139); every one of them gathers statistics and leaves optimal tables in the slots.  For ONE component that is the
  // schedule of optimize_coding itself -- main pass (statistics), trellis pass(es), output with the tables the last pass made:
  // the same bytes (trellis_without_optimize_is_optimize; checked against the reference on 50 parameter sets) -- so it is coded
  // that way.  With more components the last one is never quantized and coded with another component's tables: refused.
  if (p->trellis_quant && !p->optimize_coding && !p->arith_code && !trellis_without_optimize_is_optimize(p))
    return fail(MJH_EUNSUPPORTED, "trellis_quant requires optimize_coding (jcmaster.c:686-702 never selects a component otherwise)");
  if (p->huff_tables_given & ~0xFF) return fail(MJH_EINVAL, "huff_tables_given 0x%x", p->huff_tables_given);
  for (int k = 0; k < 8; k++) {
    if (!(p->huff_tables_given >> k & 1)) continue;
    // the checks of jpeg_make_c_derived_tbl (jchuff.c:262-316): at most 256 symbols, no code of all ones, no symbol twice, DC symbols 0..15
    int n = 0;
    unsigned code = 0;
    bool seen[256] = { false };
    for (int l = 1; l <= 16; l++) {
      if (n + p->huff_bits[k][l] > 256) return fail(MJH_EINVAL, "Huffman table %s %d: more than 256 symbols (JERR_BAD_HUFF_TABLE)", k & 1 ? "AC" : "DC", k >> 1);
      for (int i = 0; i < p->huff_bits[k][l]; i++, n++, code++) {
        const int sym = p->huff_vals[k][n];
        if (seen[sym] || (!(k & 1) && sym > 15)) return fail(MJH_EINVAL, "Huffman table %s %d: symbol 0x%02x twice or out of range (JERR_BAD_HUFF_TABLE)", k & 1 ? "AC" : "DC", k >> 1, sym);
        seen[sym] = true;
      }
      if (p->huff_bits[k][l] && code >= (1u << l)) return fail(MJH_EINVAL, "Huffman table %s %d: not a prefix code (JERR_BAD_HUFF_TABLE)", k & 1 ? "AC" : "DC", k >> 1);
      code <<= 1;
    }
  }
  if (!p->optimize_coding && !p->arith_code && p->data_precision != 12) {
    for (int i = 0; i < p->num_components; i++)
      if ((p->dc_tbl_no[i] > 1 && !(p->huff_tables_given >> (2 * p->dc_tbl_no[i]) & 1)) || (p->ac_tbl_no[i] > 1 && !(p->huff_tables_given >> (2 * p->ac_tbl_no[i] + 1) & 1)))
        return fail(MJH_EUNSUPPORTED, "no Huffman table in slot 2 / 3 (the Annex K tables exist for table numbers 0 and 1; others come through huff_tables_given; JERR_NO_HUFF_TABLE)");
  }
  if (p->dct_method != 0 && p->dct_method != 1) return fail(MJH_EUNSUPPORTED, "dct_method %d (0 = JDCT_ISLOW, 1 = JDCT_IFAST; the float DCT is outside the bit-exact path)", p->dct_method);
  if (p->trellis_stats_Ah < 0 || p->trellis_stats_Ah > 13 || p->trellis_stats_Al < 0 || p->trellis_stats_Al > 13) return fail(MJH_EINVAL, "trellis_stats_Ah / Al %d / %d", p->trellis_stats_Ah, p->trellis_stats_Al);
  return MJH_OK;
}

static void build_const(const mjh_params *p, MjhConst *C)
{
  memset(C, 0, sizeof(*C));
  C->W = p->image_width; C->H = p->image_height;
  C->in_comps = p->input_components; C->ncomp = p->num_components;
  C->px_size = p->input_pixel_size ? p->input_pixel_size : p->input_components;
  C->off_r = p->rgb_offset[0]; C->off_g = p->rgb_offset[1]; C->off_b = p->rgb_offset[2];
  if (C->off_r == 0 && C->off_g == 0 && C->off_b == 0) { C->off_g = 1; C->off_b = 2; }
  C->precision = p->data_precision == 12 ? 12 : 8;
  C->no_ycc = p->color_transform != MJH_COLOR_YCC;      // (MJH_COLOR_YCC_IN: the same null conversion, only the headers / scripts are YCbCr's)
  C->maxh = C->maxv = 1;
  for (int i = 0; i < C->ncomp; i++) {
    if (p->h_samp_factor[i] > C->maxh) C->maxh = p->h_samp_factor[i];
    if (p->v_samp_factor[i] > C->maxv) C->maxv = p->v_samp_factor[i];
  }
  C->mcus_per_row = (int)div_round_up(C->W, C->maxh * 8);   // per_scan_setup jcmaster.c:561-566
  C->mcu_rows = (int)div_round_up(C->H, C->maxv * 8);
  C->groups_x = C->mcus_per_row * 8;
  C->groups_y = C->mcu_rows * 8;
  C->real_groups_y = (int)div_round_up(C->H, C->maxv);
  long long plane_off = 0, coef_off = 0, blk_off = 0;
  int mcu_blk0 = 0;
  for (int i = 0; i < C->ncomp; i++) {
    MjhComp &c = C->c[i];
    c.h = p->h_samp_factor[i]; c.v = p->v_samp_factor[i];
    c.hexp = C->maxh / c.h; c.vexp = C->maxv / c.v;
    c.wib = (int)div_round_up((long)C->W * c.h, (long)C->maxh * 8);   // initial_setup jcmaster.c:237-247
    c.hib = (int)div_round_up((long)C->H * c.v, (long)C->maxv * 8);
    c.wpad = (int)(div_round_up(c.wib, c.h) * c.h);                   // jccoefct.c:587-601
    c.hpad = (int)(div_round_up(c.hib, c.v) * c.v);
    c.pw = c.wib * 8; c.ph = c.hib * 8;
    c.nblk = c.wib * c.hib;
    c.kstride = (c.nblk + 63) & ~63;
    c.qtbl = p->quant_tbl_no[i]; c.dctbl = p->dc_tbl_no[i] | (dc_class(p, i) << 8); c.actbl = p->ac_tbl_no[i];
    c.mcu_blk0 = mcu_blk0; mcu_blk0 += c.h * c.v;
    c.plane_off = plane_off; plane_off += (long long)c.pw * c.ph;
    c.coef_off = coef_off; coef_off += (long long)c.kstride * 64;
    c.blk_off = blk_off; blk_off += c.nblk;
  }
  C->blocks_per_mcu = mcu_blk0;
  C->total_mcu_blocks = C->mcus_per_row * C->mcu_rows * C->blocks_per_mcu;
  C->total_real_blocks = (int)blk_off;
  C->planes_per_image = plane_off;
  C->coefs_per_image = coef_off;
  C->deringing = p->overshoot_deringing;
  C->smoothing = p->smoothing_factor;
  C->trellis = p->trellis_quant;
  C->trellis_dc = p->trellis_quant_dc;
  C->delta_dc_weight = p->trellis_delta_dc_weight;
  // per_scan_setup jcmaster.c:595-600: restart_in_rows is converted per scan; here for the final
  // interleaved scan (the per-component statistics passes use their own MCUs_per_row, T10)
  C->restart_interval = (int)p->restart_interval;
  if (p->restart_in_rows > 0) {
    const long nominal = (long)p->restart_in_rows * C->mcus_per_row;
    C->restart_interval = (int)(nominal < 65535L ? nominal : 65535L);
  }
  for (int t = 0; t < 2; t++) {   // arithmetic conditioning (a zeroed table = the defaults 0 / 1 / 5, jcparam.c:417-419)
    const bool zeroed = p->arith_dc_L[t] == 0 && p->arith_dc_U[t] == 0 && p->arith_ac_K[t] == 0;
    C->ari_L[t] = zeroed ? 0 : p->arith_dc_L[t]; C->ari_U[t] = zeroed ? 1 : p->arith_dc_U[t]; C->ari_K[t] = zeroed ? 5 : p->arith_ac_K[t];
  }
  C->lambda_log_scale1 = p->lambda_log_scale1;
  C->lambda_log_scale2 = p->lambda_log_scale2;
  // pow() evaluated by the host libm, like the reference (jcdctmgr.c:1033-1037)
  if (p->lambda_log_scale2 > 0.0f) {
    C->pow_scale1 = pow(2.0, (double)p->lambda_log_scale1);
    C->pow_scale2 = pow(2.0, (double)p->lambda_log_scale2);
  } else {
    C->pow_scale1 = pow(2.0, (double)p->lambda_log_scale1 - 12.0);
    C->pow_scale2 = 0.0;
  }
}

// ---- marker bytes that do not depend on the image (jcmarker.c) -------------------------------------
static void put2(std::vector<uint8_t> &o, int v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

static void build_prefix(const mjh_params *p, std::vector<uint8_t> &o, bool *baseline_sof, int *file_hdr_len, int dqt_off[4] = nullptr)
{
  if (dqt_off) for (int t = 0; t < 4; t++) dqt_off[t] = -1;
  const bool multi = p->compress_profile != MJH_PROFILE_FASTEST;
  o.push_back(0xFF); o.push_back(0xD8);                     // SOI, write_file_header :649
  if (p->write_JFIF_header) {                               // emit_jfif_app0 :534-565 (version 1.01, density 1:1)
    o.push_back(0xFF); o.push_back(0xE0); put2(o, 16);
    const uint8_t jf[] = { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 };
    o.insert(o.end(), jf, jf + sizeof(jf));
  }
  if (p->color_transform == MJH_COLOR_NONE && p->num_components == 3) {   // emit_adobe_app14 :452-486: version 100, flags 0, transform 0 (RGB)
    const uint8_t ad[] = { 0xFF, 0xEE, 0, 14, 'A', 'd', 'o', 'b', 'e', 0, 100, 0, 0, 0, 0, 0 };
    o.insert(o.end(), ad, ad + sizeof(ad));
  }
  *file_hdr_len = (int)o.size();   // SOI + APP0 / APP14: what jpeg_start_compress writes
  int prec[MJH_MAX_COMPS], prec_any = 0;
  for (int ci = 0; ci < p->num_components; ci++) {
    prec[ci] = 0;
    for (int i = 0; i < 64; i++) if (p->quantval[p->quant_tbl_no[ci]][i] > 255) prec[ci] = 1;
    prec_any += prec[ci];
  }
  bool sent[4] = { false, false, false, false };
  if (multi) {                                              // emit_multi_dqt :189-254
    bool seen[4] = { false, false, false, false };
    int size = 2;
    for (int ci = 0; ci < p->num_components; ci++) {
      const int t = p->quant_tbl_no[ci];
      if (!seen[t]) { size += 64 * (prec[ci] + 1) + 1; seen[t] = true; }
    }
    o.push_back(0xFF); o.push_back(0xDB); put2(o, size);
  }
  for (int ci = 0; ci < p->num_components; ci++) {
    const int t = p->quant_tbl_no[ci];
    if (sent[t]) continue;
    if (!multi) {                                           // emit_dqt :140-186
      o.push_back(0xFF); o.push_back(0xDB);
      put2(o, prec[ci] ? 64 * 2 + 1 + 2 : 64 + 1 + 2);
    }
    o.push_back((uint8_t)(t + (prec[ci] << 4)));
    if (dqt_off && !prec[ci]) dqt_off[t] = (int)o.size();   // trellis_q_opt rewrites the 8-bit entries in place
    for (int i = 0; i < 64; i++) {
      const unsigned qv = p->quantval[t][kZZ[i]];
      if (prec[ci]) o.push_back((uint8_t)(qv >> 8));
      o.push_back((uint8_t)(qv & 0xFF));
    }
    sent[t] = true;
  }
  bool is_baseline = true;                                  // write_frame_header :699-734
  for (int ci = 0; ci < p->num_components; ci++)
    if (p->dc_tbl_no[ci] > 1 || p->ac_tbl_no[ci] > 1) is_baseline = false;
  if (prec_any || p->data_precision == 12) is_baseline = false;   // write_frame_header :699-703
  *baseline_sof = is_baseline;
  if (p->arith_code) { o.push_back(0xFF); o.push_back(p->num_scans > 0 ? 0xCA : 0xC9); }   // SOF10 / SOF9 (write_frame_header :720-725)
  else { o.push_back(0xFF); o.push_back(p->num_scans > 0 ? 0xC2 : (is_baseline ? 0xC0 : 0xC1)); }  // emit_sof :464-490 (SOF2 = progressive)
  put2(o, 3 * p->num_components + 2 + 5 + 1);
  o.push_back(p->data_precision == 12 ? 12 : 8);
  put2(o, p->image_height); put2(o, p->image_width);
  o.push_back((uint8_t)p->num_components);
  for (int ci = 0; ci < p->num_components; ci++) {
    o.push_back((uint8_t)p->component_id[ci]);
    o.push_back((uint8_t)((p->h_samp_factor[ci] << 4) + p->v_samp_factor[ci]));
    o.push_back((uint8_t)p->quant_tbl_no[ci]);
  }
}

static void build_sos(const mjh_params *p, int restart_interval, std::vector<uint8_t> &o)
{                                                           // emit_dri :452 (only when != 0, write_scan_header :778-781)
  if (restart_interval) { o.push_back(0xFF); o.push_back(0xDD); put2(o, 4); put2(o, restart_interval); }
                                                            // emit_sos :494-531, sequential scan of all components
  o.push_back(0xFF); o.push_back(0xDA);
  put2(o, 2 * p->num_components + 2 + 1 + 3);
  o.push_back((uint8_t)p->num_components);
  for (int ci = 0; ci < p->num_components; ci++) {
    o.push_back((uint8_t)p->component_id[ci]);
    o.push_back((uint8_t)((p->dc_tbl_no[ci] << 4) + p->ac_tbl_no[ci]));
  }
  o.push_back(0); o.push_back(63); o.push_back(0);
}

// Annex K.3 standard tables (jstdhuff.c:54-131), used when optimize_coding is off
static const uint8_t kStdDcLBits[17] = { 0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t kStdDcCBits[17] = { 0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t kStdDcVal[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t kStdAcLBits[17] = { 0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t kStdAcLVal[162] = {
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
  0xf9, 0xfa };
static const uint8_t kStdAcCBits[17] = { 0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t kStdAcCVal[162] = {
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
  0xf9, 0xfa };

// the Annex K.3 tables for the libjpeg drop-in's jpeg_set_defaults (std_huff_tables jstdhuff.c:31-131)
extern "C" int mjh_std_huffman_table(int is_ac, int tblno, const uint8_t **bits, const uint8_t **vals, int *nvals)
{
  if (!bits || !vals || !nvals || tblno < 0 || tblno > 1) return fail(MJH_EINVAL, "bad arguments");
  *bits = is_ac ? (tblno ? kStdAcCBits : kStdAcLBits) : (tblno ? kStdDcCBits : kStdDcLBits);
  *vals = is_ac ? (tblno ? kStdAcCVal : kStdAcLVal) : kStdDcVal;
  *nvals = is_ac ? 162 : 12;
  return MJH_OK;
}

static void fill_std_table(MjhHuffTable *T, const uint8_t *bits, const uint8_t *vals, int nvals)
{   // jpeg_make_c_derived_tbl jchuff.c:231-318 on the host (tables are constants here)
  memset(T, 0, sizeof(*T));
  memcpy(T->bits, bits, 17);
  memcpy(T->huffval, vals, nvals);
  T->nsyms = nvals;
  int p = 0;
  unsigned code = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l]; i++) {
      T->ehufsi[vals[p]] = (uint8_t)l;
      T->ehufco[vals[p]] = (uint16_t)code;
      code++; p++;
    }
    code <<= 1;
  }
}

static void free_all(mjh_encoder *e)
{
  if (!e) return;
  if (e->twin) { mjh_encoder *t = e->twin; e->twin = nullptr; free_all(t); }
  (void)hipSetDevice(e->device);
  for (hipEvent_t ev : { e->ev_done, e->ev_tier1 }) if (ev) (void)hipEventDestroy(ev);
  if (e->ev_null_in) (void)hipEventDestroy(e->ev_null_in);
  void *ptrs[] = { e->d_pixb[0], e->d_pixb[1], e->d_plin, e->d_cfin, e->d_prog_mpos, e->d_prog_ffsums, e->d_prog_chunks, e->pe.len16, e->pe.run16, e->pe.tail16, e->pe.be16, e->pe.off32, e->pe.sums, e->pe.totals, e->pe.T32, e->pe.tsums, e->pe.ttotals, e->pe.ne_bits, e->pe.ne2_bits, e->pe.e_bits, e->pe.info, e->pe.chist, e->pe.rmask, e->d_planes, e->d_uq, e->d_q, e->d_q0, e->d_quant, e->d_quant_init, e->d_tabs, e->d_tabs_init, e->d_lambda, e->d_back, e->d_eob_cost, e->d_eob_has, e->d_qsums, e->d_nzmask, e->d_nq8, e->d_dense, e->d_worklist, e->d_worklist2, e->d_prog_scans, e->d_prog_ctl, e->d_lists, e->d_pool, e->d_outpool, e->d_frame_hdr, e->d_seg_x, e->d_seg_E, e->d_seg_sums, e->d_seg_totals, e->d_mpos,
                   e->d_len16, e->d_off32, e->d_sums, e->d_totals, e->d_ffsums, e->d_fftotals, e->d_stream, e->d_out, e->d_sizes,
                   e->d_meta, e->d_prefix, e->d_sos, e->d_arith_rates, e->d_back9, e->d_jfin, e->d_qspec, e->g_in[0], e->g_in[1], e->g_in[2], e->g_in[3] };
  for (void *q : ptrs) if (q) (void)mjh_guard_free(q);
  for (void *q : e->g_in_old) (void)mjh_guard_free(q);
  if (e->ev_defer) (void)hipEventDestroy(e->ev_defer);
  if (e->h_defer) (void)hipHostFree(e->h_defer);
  for (int b = 0; b < 2; b++) {
    if (e->h_stage[b]) (void)hipHostFree(e->h_stage[b]);
    if (e->h_res[b]) (void)mjh_guard_host_free(e->h_res[b]);
    if (e->h_tab[b]) (void)mjh_guard_host_free(e->h_tab[b]);
    for (hipEvent_t ev : { e->ev_h2d[b], e->ev_pix_free[b], e->ev_packed[b] }) if (ev) (void)hipEventDestroy(ev);
  }
  if (e->d2h_stream) (void)hipStreamDestroy(e->d2h_stream);
  if (e->h_plin) (void)hipHostFree(e->h_plin);
  if (e->h_cfin) (void)hipHostFree(e->h_cfin);
  for (hipEvent_t ev : e->prof_events) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->side_events) (void)hipEventDestroy(ev);
  if (e->copy_done) (void)hipEventDestroy(e->copy_done);
  for (hipEvent_t ev : { e->ev_fork, e->ev_join, e->ev_side0, e->ev_side1, e->ev_chunk[0], e->ev_chunk[1], e->ev_chunk[2], e->ev_chunk[3] }) if (ev) (void)hipEventDestroy(ev);
  if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  delete e;
}

extern "C" void mjh_encoder_destroy(mjh_encoder *e) { free_all(e); }

#define HIPCHK_E(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { int rc_ = fail(MJH_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); free_all(e); return rc_; } } while (0)

// The geometry as ONE SCAN of a sequential script sees it (per_scan_setup jcmaster.c:548-626): its components in scan order; a
// single-component scan is non-interleaved -- an MCU is one block, the MCU rows are the component's block rows, there are no
// dummy blocks; a scan of several components keeps the frame's MCU grid with only its own blocks in an MCU.  The restart
// interval is the scan's own (restart_in_rows counts rows of ITS MCUs, :595-600).  MjhComp carries absolute offsets, so the
// view addresses the same buffers as the frame.
static MjhConst scan_view(const MjhConst &C, const mjh_params &p, const int *comps, int k)
{
  MjhConst V = C;
  V.ncomp = k;
  int mb = 0;
  for (int j = 0; j < k; j++) {
    V.c[j] = C.c[comps[j]];
    if (k == 1) {
      V.c[j].h = V.c[j].v = 1;
      V.c[j].wpad = V.c[j].wib; V.c[j].hpad = V.c[j].hib;
      V.mcus_per_row = V.c[j].wib; V.mcu_rows = V.c[j].hib;
    }
    V.c[j].mcu_blk0 = mb;
    mb += V.c[j].h * V.c[j].v;
  }
  V.blocks_per_mcu = mb;
  V.total_mcu_blocks = V.mcus_per_row * V.mcu_rows * mb;
  long ri = p.restart_interval;
  if (p.restart_in_rows > 0) { ri = (long)p.restart_in_rows * V.mcus_per_row; if (ri > 65535L) ri = 65535L; }
  V.restart_interval = (int)ri;
  return V;
}

// Do two streams run concurrently?  The HIP runtime deals streams over a few hardware queues (4 per priority level unless
// GPU_MAX_HW_QUEUES says otherwise), round-robin; two streams on one queue are served in turn, whatever the kernels.  A busy wave on
// each (150 us) takes 150 us when they overlap and 300 when they share a queue.  -1: could not tell (HIP error).
static int mjh_streams_overlap(hipStream_t a, hipStream_t b)
{
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
  int votes = 0;
  for (int rep = 0; rep < 3; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    mjh_launch_spin(15000ull, a);
    mjh_launch_spin(15000ull, b);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    votes += us < 240.0;
  }
  return votes >= 2;
}

static thread_local bool g_create_twin = false;
static thread_local hipStream_t g_twin_avoid[2] = { nullptr, nullptr };   // make_twin: the primary's main and side stream     // mjh_encoder_create is making the second buffer set of an encoder (make_twin)

extern "C" int mjh_encoder_create(const mjh_params *p, int max_batch, int device, mjh_encoder **out)
{
  if (!p || !out || max_batch < 1) return fail(MJH_EINVAL, "bad arguments");
  *out = nullptr;
  // A script whose scans are all whole-block scans (Ss = 0, Se = 63) is a sequential multi-scan file, not a progressive one
  // (validate_script jcmaster.c:309-330, :386-398): every component in exactly one scan, components in ascending order.  It is
  // taken out of the parameters here -- everything in front of the entropy stage is the sequential pipeline.
  mjh_params pn = *p;
  std::vector<mjh_scan> seq_script;
  if (p->num_scans > 0 && p->num_scans <= MJH_MAX_SCANS && !p->optimize_scans && p->scan_info[0].Ss == 0 && p->scan_info[0].Se == 63) {
    bool sent[MJH_MAX_COMPS] = { false, false, false, false };
    for (int si = 0; si < p->num_scans; si++) {
      const mjh_scan &sc = p->scan_info[si];
      if (sc.Ss != 0 || sc.Se != 63 || sc.Ah != 0 || sc.Al != 0) return fail(MJH_EINVAL, "scan %d: a sequential script holds whole-block scans only (Ss 0, Se 63, Ah 0, Al 0; JERR_BAD_PROG_SCRIPT)", si);
      if (sc.comps_in_scan < 1 || sc.comps_in_scan > p->num_components) return fail(MJH_EINVAL, "scan %d: component count", si);
      for (int ci = 0; ci < sc.comps_in_scan; ci++) {
        const int c = sc.component_index[ci];
        if (c < 0 || c >= p->num_components || (ci > 0 && c <= sc.component_index[ci - 1])) return fail(MJH_EINVAL, "scan %d: component order", si);
        if (sent[c]) return fail(MJH_EINVAL, "scan %d: component %d is coded twice (JERR_BAD_SCAN_SCRIPT)", si, c);
        sent[c] = true;
      }
      seq_script.push_back(sc);
    }
    for (int c = 0; c < p->num_components; c++) if (!sent[c]) return fail(MJH_EINVAL, "the script leaves component %d out (JERR_MISSING_DATA)", c);
    pn.num_scans = 0;
    if (seq_script.size() == 1) seq_script.clear();      // one scan of all components: the plain sequential file
    p = &pn;
  }
  if (p->arith_code && p->trellis_quant && p->trellis_q_opt && arith_qopt_updates(p) == 0) {   // no pass re-estimates the tables: no effect in the reference
    if (p != &pn) { pn = *p; p = &pn; }
    pn.trellis_q_opt = 0;
  }
  int rc = check_supported(p);
  if (rc) return rc;
  if (!p->arith_code && p->compress_profile != MJH_PROFILE_FASTEST && p->num_scans == 0) {     // (sequential Huffman files: the scans that carry DC and AC tables of several components)
    const int all[MJH_MAX_COMPS] = { 0, 1, 2, 3 };
    bool dseen[4] = { false, false, false, false }, aseen[4] = { false, false, false, false };
    bool bad = seq_script.empty() && dht_writer_would_corrupt(p, all, p->num_components, dseen, aseen);
    for (const mjh_scan &sc : seq_script) {
      if (p->optimize_coding || p->data_precision == 12) for (int t = 0; t < 4; t++) dseen[t] = aseen[t] = false;
      bad = bad || dht_writer_would_corrupt(p, sc.component_index, sc.comps_in_scan, dseen, aseen);
    }
    if (bad) return fail(MJH_EUNSUPPORTED, "dc_tbl_no / ac_tbl_no: a component reuses an earlier component's DC table with an AC table no earlier component had -- the reference's DHT writer (emit_multi_dht, jcmarker.c:293-401) writes a corrupt marker for this assignment");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MJH_EHIP, "no HIP device available: libmozjpeg_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(MJH_EINVAL, "device %d out of range (%d devices)", device, ndev);
  mjh_encoder *e = new mjh_encoder();
  e->p = *p;
  e->p_created = *p;
  e->high_priority_streams = g_create_twin;
  if (p->num_components == 1 && (p->h_samp_factor[0] != 1 || p->v_samp_factor[0] != 1)) {   // (check_supported: only the SOF byte differs)
    e->sof_hv0 = (p->h_samp_factor[0] << 4) + p->v_samp_factor[0];
    e->dc_chain_v = p->v_samp_factor[0];
    e->p.h_samp_factor[0] = e->p.v_samp_factor[0] = 1;
  }
  if (p->trellis_quant && !p->optimize_coding && !p->arith_code) e->p.optimize_coding = 1, e->p.huff_tables_given = 0;   // (check_supported: one component; the slots' tables are overwritten before anything reads them)
  if (p->data_precision == 12 && !p->huff_tables_given) e->p.optimize_coding = 1;   // standard tables are 8-bit only (jcparam.c:452-453, jcmaster.c:1102-1105: tables of the caller's own are kept)
  p = &e->p;     // (everything below reads the parameters as the encoder runs them: a 12-bit sequential scan script sends its scans' tables with every scan)
  e->device = device;
  e->max_batch = max_batch;
  build_const(p, &e->C);
  const MjhConst &C = e->C;
  e->progressive = p->num_scans > 0;
  e->nscans = p->num_scans;
  if (e->progressive) e->spi = SLOT_PROG + 2 * (p->num_scans + C.ncomp);
  e->nbands = p->trellis_quant && p->use_scans_in_trellis ? 2 : 1;          // jcmaster.c:451-460
  e->freq_split = p->trellis_freq_split > 0 ? p->trellis_freq_split : 8;   // jcparam.c:512
  HIPCHK_E(hipSetDevice(device));
  if (e->high_priority_streams) {
    // The twin's main and side stream must not share a hardware queue with the primary's, or the two batches take turns instead of
    // overlapping.  Streams of the same priority (another priority level has queues of its own, but its kernels are then dispatched
    // first and the sharing turns into alternation: measured, 4.68 against 4.43 ms) are made until two are found that overlap with
    // both of the primary's and with each other; the others are destroyed again.
    std::vector<hipStream_t> made;
    for (int tries = 0; tries < 12 && !e->side_stream; tries++) {
      hipStream_t c = nullptr;
      HIPCHK_E(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
      made.push_back(c);
      bool free_queue = true;
      for (hipStream_t other : { g_twin_avoid[0], g_twin_avoid[1], e->stream })
        if (other && mjh_streams_overlap(c, other) == 0) { free_queue = false; break; }
      if (!free_queue) continue;
      if (!e->stream) e->stream = c; else e->side_stream = c;
    }
    for (hipStream_t c : made) if (c != e->stream && c != e->side_stream) (void)hipStreamDestroy(c);
    if (!e->stream) HIPCHK_E(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));           // (fewer queues than streams: share)
    if (!e->side_stream) HIPCHK_E(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
  } else {
    HIPCHK_E(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIPCHK_E(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));     // (right behind the main stream: adjacent creations never share a hardware queue)
  }
  HIPCHK_E(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
  HIPCHK_E(hipEventCreateWithFlags(&e->ev_tier1, hipEventDisableTiming));
  if (const char *v = getenv("MJH_INFLIGHT")) { e->inflight = atoi(v); if (e->inflight < 1 || e->inflight > 2) e->inflight = 2; }
  if (const char *v = getenv("MJH_INFLIGHT_MODE")) { e->inflight_mode = atoi(v); if (e->inflight_mode < 0 || e->inflight_mode > 2) e->inflight_mode = 1; }
  {
    // (copy and hand-over streams at the greatest priority, i.e. in a hardware-queue pool of their own, were measured in round 3
    // with 16 libjpeg client threads: no gain with the coalescing shim, a loss without it -- default priority)
    e->copy_prio = 0;
    HIPCHK_E(hipStreamCreateWithPriority(&e->copy_stream, hipStreamNonBlocking, e->copy_prio));
  }
  HIPCHK_E(hipEventCreateWithFlags(&e->copy_done, hipEventDisableTiming));
  HIPCHK_E(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  HIPCHK_E(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  HIPCHK_E(hipEventCreate(&e->ev_side0));
  HIPCHK_E(hipEventCreate(&e->ev_side1));
  const size_t B = (size_t)max_batch;
  const size_t bps = C.precision == 12 ? 2 : 1;   // bytes per sample
  e->pix_image_bytes = (size_t)C.W * C.H * C.px_size * bps;
  HIPCHK_E(mjh_dmalloc((void **)&e->d_planes, B * C.planes_per_image * bps));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_uq, B * C.coefs_per_image * 2));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_q, B * C.coefs_per_image * 2));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_quant, sizeof(MjhQuant) * (p->trellis_quant && p->trellis_q_opt ? B : 1)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_quant_init, sizeof(MjhQuant)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_tabs, B * e->spi * sizeof(MjhHuffTable)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_tabs_init, B * e->spi * sizeof(MjhHuffTable)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_lambda, B * C.total_real_blocks * sizeof(float)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_back, B * (size_t)C.total_real_blocks * 16));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_worklist, 256 + B * (size_t)C.total_real_blocks * 12));   // (+ one 16-byte header per view)
  HIPCHK_E(mjh_dmalloc((void **)&e->d_worklist2, 256 + B * (size_t)C.total_real_blocks * 12));
  if (p->trellis_quant && p->trellis_eob_opt) {
    HIPCHK_E(mjh_dmalloc(&e->d_eob_cost, B * (size_t)C.total_real_blocks * 8));
    HIPCHK_E(mjh_dmalloc((void **)&e->d_eob_has, B * (size_t)C.total_real_blocks * 4));
  }
  if (p->trellis_quant && p->trellis_q_opt) HIPCHK_E(mjh_dmalloc((void **)&e->d_qsums, B * 4 * 64 * 2 * sizeof(long long)));
  {
    // sequential mode, one plain trellis round: the trellis hands compact records to the final statistics, the bit-length
    // and the bit-writing pass (the one-plane-per-position form serves the configurations below; both are bit-identical)
    const int nl = p->trellis_num_loops > 1 ? p->trellis_num_loops : 1;
    bool restart_scans = false;   // progressive: scans with restart intervals go through the sequential walk, which reads one plane per position
    if (e->progressive && (p->restart_interval || p->restart_in_rows)) restart_scans = true;
    e->use_compact = !p->arith_code && p->trellis_quant && nl == 1 && e->nbands == 1 && !p->trellis_eob_opt && !p->trellis_q_opt &&
                     (e->progressive ? !restart_scans : true);
    if (e->use_compact) HIPCHK_E(mjh_dmalloc((void **)&e->d_nzmask, B * (size_t)C.total_real_blocks * sizeof(unsigned long long)));
    if (e->use_compact && p->trellis_quant) HIPCHK_E(mjh_dmalloc((void **)&e->d_nq8, B * (size_t)C.total_real_blocks));
  }
  if (p->trellis_quant) {   // room for a quarter of all blocks (typically 1-2 % overflow); the rest would be read from the planes
    e->dense_cap = (unsigned)(B * (size_t)C.total_real_blocks / 4 + 1024);
    if (const char *dv = getenv("MJH_DENSE_CAP")) e->dense_cap = (unsigned)atoi(dv) + 1u;   // tests: deferred blocks beyond the dense copies come from the planes
    HIPCHK_E(mjh_dmalloc((void **)&e->d_dense, (size_t)e->dense_cap * 128));
  }
  {
    // starting point of the first tier's capacity (the measured record counts of every finished batch correct it): the coarser
    // the luma table, the fewer coefficients survive -- mean AC step >= 38 (q75 and below): 16 records, >= 22 (q85): 24,
    // >= 13 (q90): 32, finer: 48
    double m = 0.0;
    for (int k = 1; k < 64; k++) m += p->quantval[p->quant_tbl_no[0]][k];
    m /= 63.0;
    e->trellis_variant = m >= 38.0 ? 0 : m >= 22.0 ? 2 : m >= 13.0 ? 3 : 4;
  }
  if (const char *v = getenv("MJH_TRELLIS_VARIANT")) { e->trellis_variant = atoi(v); e->trellis_adapt = false; }
  HIPCHK_E(hipHostMalloc((void **)&e->h_defer, 64, hipHostMallocDefault));
  if (const char *v = getenv("MJH_TRELLIS_V3")) e->trellis_v3 = atoi(v);
  if (const char *v = getenv("MJH_SMALL_BATCH")) e->small_batch = (size_t)atoll(v);      // tests: the large-batch schedule on small images
  if (const char *v = getenv("MJH_TRELLIS_CHUNKS")) { e->trellis_chunks = atoi(v); if (e->trellis_chunks < 0 || e->trellis_chunks > 4) e->trellis_chunks = 0; }
  for (int i = 0; i < 4; i++) HIPCHK_E(hipEventCreateWithFlags(&e->ev_chunk[i], hipEventDisableTiming));
  if (const char *v = getenv("MJH_DC_LATE")) { e->dc_late = atoi(v); if (e->dc_late < 0 || e->dc_late > 2) e->dc_late = 1; }   // A/B knob
  e->dc_window_ok = 1;
  for (int i = 0; i < C.ncomp; i++) if (p->quantval[p->quant_tbl_no[i]][0] < 5) e->dc_window_ok = 0;
  if (const char *v = getenv("MJH_DC_SPEC")) e->dc_spec = atoi(v);
  HIPCHK_E(mjh_dmalloc((void **)&e->d_len16, B * (size_t)C.total_mcu_blocks * 2));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_off32, B * (size_t)C.total_mcu_blocks * 4));
  e->chunks = (C.total_mcu_blocks + 2047) / 2048;
  // worst case per block: DC 16+11, 63 x (16+10) = 1665 bits -> 53 words (12-bit: 16+15, 63 x (16+14) -> 61 words)
  size_t words = (size_t)C.total_mcu_blocks * (C.precision == 12 ? 61 : 53) + 64;
  if (C.restart_interval) words += (size_t)(C.mcus_per_row * C.mcu_rows) / C.restart_interval + 1;   // pad + RSTn per interval
  if (words > (size_t)1 << 27) words = (size_t)1 << 27;   // bit offsets are 32-bit
  e->stream_words = (words + 63) & ~(size_t)63;
  e->ff_chunks = (int)((e->stream_words + 2047) / 2048);
  HIPCHK_E(mjh_dmalloc((void **)&e->d_sums, B * e->chunks * sizeof(unsigned)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_ffsums, B * e->ff_chunks * sizeof(unsigned)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_totals, B * sizeof(unsigned)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_fftotals, B * sizeof(unsigned)));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_stream, B * e->stream_words * 4 + 4096));   // slack: a bit writer may touch two words past its last offset
  e->out_stride = ((size_t)2048 + 1024 * seq_script.size() + e->stream_words * 8 + 255) & ~(size_t)255;   // (every scan of a sequential script brings its own DHT + SOS)
  HIPCHK_E(mjh_dmalloc((void **)&e->d_out, B * e->out_stride));
  HIPCHK_E(mjh_dmalloc((void **)&e->d_sizes, B * sizeof(unsigned)));
  {
    const int nmcu = C.mcus_per_row * C.mcu_rows;
    e->nseg = C.restart_interval ? (nmcu + C.restart_interval - 1) / C.restart_interval : 1;
    int max_nseg = e->nseg;
    for (const mjh_scan &sc : seq_script) {     // a scan of a sequential script has its own MCU grid and restart interval
      const MjhConst V = scan_view(C, *p, sc.component_index, sc.comps_in_scan);
      const int nm = V.mcus_per_row * V.mcu_rows, ns1 = V.restart_interval ? (nm + V.restart_interval - 1) / V.restart_interval : 1;
      max_nseg = ns1 > max_nseg ? ns1 : max_nseg;
    }
    for (int i = 0; i < C.ncomp; i++) {     // restart interval of the per-component statistics passes, in blocks
      long ri = p->restart_interval;
      if (p->restart_in_rows > 0) { ri = (long)p->restart_in_rows * C.c[i].wib; if (ri > 65535L) ri = 65535L; }
      e->comp_restart[i] = (int)ri;
    }
    const size_t ns = (size_t)max_nseg;
    HIPCHK_E(mjh_dmalloc((void **)&e->d_seg_x, B * ns * 4));
    HIPCHK_E(mjh_dmalloc((void **)&e->d_seg_E, B * ns * 4));
    HIPCHK_E(mjh_dmalloc((void **)&e->d_mpos, B * ns * 4));
    HIPCHK_E(mjh_dmalloc((void **)&e->d_seg_sums, B * ((ns + 2047) / 2048) * 4));
    HIPCHK_E(mjh_dmalloc((void **)&e->d_seg_totals, B * 4));
  }
  HIPCHK_E(mjh_dmalloc(&e->d_meta, B * sizeof(MjhImageMeta)));
  e->h_sizes.resize(B);

  // quantizer constants
  MjhQuant hq;
  memset(&hq, 0, sizeof(hq));
  for (int t = 0; t < 4; t++)
    for (int k = 0; k < 64; k++) {
      const int q = p->quantval[t][kZZ[k]] ? p->quantval[t][kZZ[k]] : 1;
      hq.q[t][k] = (uint16_t)q;
      hq.dq8[t][k] = 8 * q;
      hq.rcp8q[t][k] = 1.0f / (float)(8 * q);
      hq.thr8[t][k] = (float)(8 * q - ((8 * q) >> 1));
      hq.lambda_tbl[t][k] = (float)(1.0 / (double)(q * q));   // jcdctmgr.c:1017-1021
      {   // the conventional quantizer's divisor: a UINT16 argument in the reference's 8-bit build (see MjhQuant)
        // JDCT_IFAST: quantval x the AA&N scale factors of the position (natural order) x 8, rounded at 11 bits (jcdctmgr.c:291-345)
        static const int aan[64] = { 16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
                                     21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906, 19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
                                     16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 12873, 17855, 16819, 15137, 12873, 10114, 6967, 3552,
                                     8867, 12299, 11585, 10426, 8867, 6967, 4799, 2446, 4520, 6270, 5906, 5315, 4520, 3552, 2446, 1247 };
        const int dc = p->dct_method == 1 ? (C.precision == 12 ? (int)(((long)q * aan[kZZ[k]] + 1024L) >> 11) : (int)((((long)q * aan[kZZ[k]] + 1024L) >> 11) & 0xFFFF))
                                          : C.precision == 12 ? 8 * q : (int)((8u * (unsigned)q) & 0xFFFFu);
        if (dc == 0)      // q = 8192, 16384, 24576: compute_reciprocal(0) divides by zero -- the reference dies
          for (int i = 0; i < C.ncomp; i++) if (p->quant_tbl_no[i] == t) e->fdct_div_zero = true;
        hq.dqc8[t][k] = dc ? dc : 8;
        hq.rcpc8q[t][k] = 1.0f / (float)(dc ? dc : 8);
      }
      if (q <= 255) {
        const unsigned d = 8u * (unsigned)q;
        int b = 0;
        while ((d >> b) != 0) b++;
        const int kk = 22 + b < 32 ? 22 + b : 32;
        hq.mdiv[t][k] = (uint32_t)((1ull << kk) / d + 1ull);
        hq.sdiv[t][k] = 32 - kk;
      }
    }
  e->fastdiv_all = 1;
  for (int t = 0; t < 4; t++) {
    hq.fastdiv[t] = 1;
    for (int k = 0; k < 64; k++) if (hq.q[t][k] > 255) hq.fastdiv[t] = 0;
  }
  for (int i = 0; i < C.ncomp; i++) if (!hq.fastdiv[p->quant_tbl_no[i]]) e->fastdiv_all = 0;
  HIPCHK_E(hipMemcpy(e->d_quant, &hq, sizeof(hq), hipMemcpyHostToDevice));
  HIPCHK_E(hipMemcpy(e->d_quant_init, &hq, sizeof(hq), hipMemcpyHostToDevice));

  // table template (zero counts; standard tables in the final slots when optimize_coding is off)
  {
    std::vector<MjhHuffTable> ht(B * e->spi);
    memset(ht.data(), 0, ht.size() * sizeof(MjhHuffTable));
    if (!p->optimize_coding || p->num_scans > 0) {   // progressive + trellis rates DC with the standard tables (T7)
      for (size_t b = 0; b < B; b++) {
        MjhHuffTable *T = &ht[b * e->spi + SLOT_FINAL];
        fill_std_table(&T[0], kStdDcLBits, kStdDcVal, 12);
        fill_std_table(&T[1], kStdAcLBits, kStdAcLVal, 162);
        fill_std_table(&T[2], kStdDcCBits, kStdDcVal, 12);
        fill_std_table(&T[3], kStdAcCBits, kStdAcCVal, 162);
        for (int k = 0; k < 8; k++)         // what the caller's table slots hold instead (mozjpeg_hip.h: huff_tables_given)
          if (p->huff_tables_given >> k & 1) {
            int nv = 0;
            for (int l = 1; l <= 16; l++) nv += p->huff_bits[k][l];
            fill_std_table(&T[k], p->huff_bits[k], p->huff_vals[k], nv);
          }
      }
    }
    HIPCHK_E(hipMemcpy(e->d_tabs_init, ht.data(), ht.size() * sizeof(MjhHuffTable), hipMemcpyHostToDevice));
  }
  // static marker bytes + DHT plan (emit_multi_dht jcmarker.c:365-398: component order, tables not yet sent)
  {
    std::vector<uint8_t> pre, sos;
    bool base;
    build_prefix(p, pre, &base, &e->file_hdr_len, e->dqt_off);
    if (e->sof_hv0) pre[pre.size() - 2] = (uint8_t)e->sof_hv0;     // (the prefix ends with the SOF's one component: id, HV, Tq)
    {
      bool seen[4] = { false, false, false, false };
      for (int ci = 0; ci < p->num_components; ci++) {
        const int t = p->quant_tbl_no[ci];
        if (!seen[t]) { e->dqt_tabs[e->dqt_ntab++] = t; seen[t] = true; }
        if (p->dc_tbl_no[ci] > 1 || p->ac_tbl_no[ci] > 1) e->tbl_le1 = false;
      }
    }
    build_sos(p, C.restart_interval, sos);
    e->prefix_len = (int)pre.size();
    e->sos_len = (int)sos.size();
    HIPCHK_E(mjh_dmalloc((void **)&e->d_prefix, pre.size()));
    HIPCHK_E(mjh_dmalloc((void **)&e->d_sos, sos.size()));
    HIPCHK_E(hipMemcpy(e->d_prefix, pre.data(), pre.size(), hipMemcpyHostToDevice));
    HIPCHK_E(hipMemcpy(e->d_sos, sos.data(), sos.size(), hipMemcpyHostToDevice));
    if (!seq_script.empty()) {
      // per scan: [DRI when the interval differs from the previous scan's, write_scan_header jcmarker.c:778-781] SOS, and the
      // tables its DHT carries: those of its components, DC then AC, each once -- with optimal tables every scan's are rebuilt and
      // sent again, the standard tables only until they have been sent (emit_multi_dht / emit_dht's sent_table, :257-401)
      std::vector<uint8_t> blob;
      bool dsent[4] = { false, false, false, false }, asent[4] = { false, false, false, false };
      int last_ri = 0;
      for (const mjh_scan &sc : seq_script) {
        mjh_encoder::SeqScan q;
        memset(&q, 0, sizeof(q));
        q.ncomp = sc.comps_in_scan;
        for (int j = 0; j < q.ncomp; j++) q.comp[j] = sc.component_index[j];
        const MjhConst V = scan_view(C, *p, q.comp, q.ncomp);
        const int nm = V.mcus_per_row * V.mcu_rows;
        q.ri = V.restart_interval;
        q.nseg = q.ri ? (nm + q.ri - 1) / q.ri : 1;
        q.sos_off = (int)blob.size();
        if (q.ri != last_ri) { blob.push_back(0xFF); blob.push_back(0xDD); put2(blob, 4); put2(blob, q.ri); last_ri = q.ri; }
        blob.push_back(0xFF); blob.push_back(0xDA);
        put2(blob, 2 * q.ncomp + 2 + 1 + 3);
        blob.push_back((uint8_t)q.ncomp);
        for (int j = 0; j < q.ncomp; j++) {
          blob.push_back((uint8_t)p->component_id[q.comp[j]]);
          blob.push_back((uint8_t)((p->dc_tbl_no[q.comp[j]] << 4) + p->ac_tbl_no[q.comp[j]]));
        }
        blob.push_back(0); blob.push_back(63); blob.push_back(0);
        q.sos_len = (int)blob.size() - q.sos_off;
        if (p->optimize_coding) for (int t = 0; t < 4; t++) dsent[t] = asent[t] = false;
        for (int j = 0; j < q.ncomp; j++) {
          const int d = p->dc_tbl_no[q.comp[j]], a = p->ac_tbl_no[q.comp[j]];
          if (!dsent[d] && q.ndht < 8) { q.dht_slots[q.ndht] = SLOT_FINAL + 2 * d; q.dht_ids[q.ndht] = d; q.ndht++; dsent[d] = true; }
          if (!asent[a] && q.ndht < 8) { q.dht_slots[q.ndht] = SLOT_FINAL + 2 * a + 1; q.dht_ids[q.ndht] = a + 0x10; q.ndht++; asent[a] = true; }
        }
        e->seq_scans.push_back(q);
      }
      (void)mjh_guard_free(e->d_sos);
      e->d_sos = nullptr;
      HIPCHK_E(mjh_dmalloc((void **)&e->d_sos, blob.size()));
      HIPCHK_E(hipMemcpy(e->d_sos, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    bool dc_sent[4] = { false, false, false, false }, ac_sent[4] = { false, false, false, false };
    e->ndht = 0;
    for (int ci = 0; ci < p->num_components; ci++) {
      const int d = p->dc_tbl_no[ci], a = p->ac_tbl_no[ci];
      // max-compression: the reference's loop `continue`s past the AC table when the DC table was
      // already sent (jcmarker.c:371-376); tables are shared pairwise here so the result is the same.
      if (!dc_sent[d] && e->ndht < 8) { e->dht_slots[e->ndht] = SLOT_FINAL + 2 * d; e->dht_ids[e->ndht] = d; e->ndht++; dc_sent[d] = true; }
      if (!ac_sent[a] && e->ndht < 8) { e->dht_slots[e->ndht] = SLOT_FINAL + 2 * a + 1; e->dht_ids[e->ndht] = a + 0x10; e->ndht++; ac_sent[a] = true; }
    }
  }
  if (p->arith_code) {
    e->arith = true;
    e->p.optimize_coding = 0;                                  // jcmaster.c:1088-1089
    // frame header (DQT + SOF9 / SOF10) = prefix minus SOI/APP0; it opens scan 0
    e->frame_hdr_len = e->prefix_len - e->file_hdr_len;
    HIPCHK_E(mjh_dmalloc((void **)&e->d_frame_hdr, (size_t)e->frame_hdr_len + 16));
    HIPCHK_E(hipMemcpy(e->d_frame_hdr, e->d_prefix + e->file_hdr_len, e->frame_hdr_len, hipMemcpyDeviceToDevice));
    const int ns = p->num_scans > 0 ? p->num_scans : seq_script.empty() ? 1 : (int)seq_script.size();
    e->arith_nscans = ns;
    std::vector<MjhProgScan> ps(ns);
    memset(ps.data(), 0, ps.size() * sizeof(MjhProgScan));
    const int nsl = 23, cfs = 42, lfs = 12;
    for (int si = 0; si < ns; si++) {
      MjhProgScan &d = ps[si];
      d.frame_header = si == 0;
      d.slot[0] = d.slot[1] = -1;
      if (p->num_scans == 0 && seq_script.empty()) {    // sequential file: one scan of whole blocks, every component (the table selectors are the components' own)
        d.ncomp = C.ncomp; d.Ss = 0; d.Se = 63;
        for (int c = 0; c < C.ncomp; c++) { d.comp[c] = c; d.comp_id[c] = p->component_id[c]; d.td[c] = p->dc_tbl_no[c]; d.ta[c] = p->ac_tbl_no[c]; }
        continue;
      }
      if (p->num_scans == 0) {    // a sequential script: whole blocks of the scan's components, in the scan's own MCU order
        const mjh_scan &ms = seq_script[si];
        d.ncomp = ms.comps_in_scan; d.Ss = 0; d.Se = 63;
        for (int ci = 0; ci < ms.comps_in_scan; ci++) { const int c = ms.component_index[ci]; d.comp[ci] = c; d.comp_id[ci] = p->component_id[c]; d.td[ci] = p->dc_tbl_no[c]; d.ta[ci] = p->ac_tbl_no[c]; }
        continue;
      }
      const mjh_scan &ms = p->scan_info[si];
      d.ncomp = ms.comps_in_scan;
      d.Ss = ms.Ss; d.Se = ms.Se; d.Ah = ms.Ah; d.Al = ms.Al;
      if (p->optimize_scans) {   // jcmaster.c:487-497, :799-818 (as for the Huffman coder)
        if (si >= lfs && si < nsl) d.al_sel = 1;
        if (si >= cfs) d.al_sel = 2;
        if (si >= 6 && si <= 8) d.cond = 1;
        if (si >= 9 && si <= 11) d.cond = 2;
      }
      for (int ci = 0; ci < ms.comps_in_scan; ci++) {
        const int c = ms.component_index[ci];
        d.comp[ci] = c;
        d.comp_id[ci] = p->component_id[c];
        d.td[ci] = (ms.Ss == 0 && ms.Ah == 0) ? p->dc_tbl_no[c] : 0;   // emit_sos jcmarker.c:519-523
        d.ta[ci] = ms.Se ? p->ac_tbl_no[c] : 0;
      }
    }
    {
      int last_ri = 0;   // write_file_header resets last_restart_interval to 0 (jcmarker.c:660)
      for (int si = 0; si < ns; si++) {
        MjhProgScan &d = ps[si];
        const long per_row = d.ncomp > 1 ? C.mcus_per_row : C.c[d.comp[0]].wib;
        long ri = p->restart_interval;
        if (p->restart_in_rows > 0) { ri = (long)p->restart_in_rows * per_row; if (ri > 65535L) ri = 65535L; }
        d.ri = (int)ri;
        d.emit_dri = d.ri != last_ri;
        last_ri = d.ri;
      }
    }
    HIPCHK_E(mjh_dmalloc(&e->d_prog_scans, ps.size() * sizeof(MjhProgScan)));
    HIPCHK_E(hipMemcpy(e->d_prog_scans, ps.data(), ps.size() * sizeof(MjhProgScan), hipMemcpyHostToDevice));
    HIPCHK_E(mjh_dmalloc(&e->d_prog_ctl, B * sizeof(MjhProgCtl)));
    // phase lists: everything at once, or the scan search's four phases (see the Huffman path below)
    std::vector<int> ph[4];
    e->nphases = 1;
    for (int si = 0; si < ns; si++) {
      int k = 0;
      if (p->optimize_scans && p->num_scans > 0) {
        e->nphases = 4;
        if ((si >= lfs && si < nsl) || si >= cfs) k = 3;
        else if (si >= 6 && si <= 8) k = 1;
        else if (si >= 9 && si <= 11) k = 2;
      }
      ph[k].push_back(si);
    }
    for (int k = 0; k < 4; k++) {
      e->pl_phase[k] = mjh_encoder::PList{};
      e->pl_phase[k].scan_off = (int)e->h_lists.size();
      e->pl_phase[k].nscan = (int)ph[k].size();
      e->h_lists.insert(e->h_lists.end(), ph[k].begin(), ph[k].end());
    }
    HIPCHK_E(mjh_dmalloc((void **)&e->d_lists, e->h_lists.size() * sizeof(int) + 16));
    HIPCHK_E(hipMemcpy(e->d_lists, e->h_lists.data(), e->h_lists.size() * sizeof(int), hipMemcpyHostToDevice));
    {
      // jget_arith_rates (jcarith.c:944-976): the estimated bits of a decision depend on the bin's state byte only; evaluated
      // with the host libm in double like the reference (log), rounded to float where the reference assigns to float
      std::vector<float> rt(512);
      for (int st = 0; st < 256; st++) {
        const int idx = (st & 0x7F) < 114 ? (st & 0x7F) : 0, mps = st >> 7;
        const float prob_lps = (float)((double)mjh_ari_qe[idx] / 46340.95);
        const float prob_0 = mps ? prob_lps : (float)(1.0 - (double)prob_lps);
        const float prob_1 = (float)(1.0 - (double)prob_0);
        rt[2 * st] = (float)(-log((double)prob_0) / log(2.0));
        rt[2 * st + 1] = (float)(-log((double)prob_1) / log(2.0));
      }
      HIPCHK_E(mjh_dmalloc(&e->d_arith_rates, rt.size() * sizeof(float)));
      HIPCHK_E(hipMemcpy(e->d_arith_rates, rt.data(), rt.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  } else if (e->progressive) {
    // frame header (DQT + SOF2) = prefix minus SOI/APP0; it opens scan 0's buffer (jcmaster.c:680-681)
    e->frame_hdr_len = e->prefix_len - e->file_hdr_len;
    e->d_frame_hdr = nullptr;
    HIPCHK_E(mjh_dmalloc((void **)&e->d_frame_hdr, (size_t)e->frame_hdr_len + 16));
    HIPCHK_E(hipMemcpy(e->d_frame_hdr, e->d_prefix + e->file_hdr_len, e->frame_hdr_len, hipMemcpyDeviceToDevice));
    // scan descriptors: the script, then one AC-first statistics scan per component for the trellis passes
    std::vector<MjhProgScan> ps(p->num_scans + C.ncomp * e->nbands);
    memset(ps.data(), 0, ps.size() * sizeof(MjhProgScan));
    const int nsl = 23, cfs = 42, lfs = 12;
    for (int si = 0; si < p->num_scans; si++) {
      const mjh_scan &ms = p->scan_info[si];
      MjhProgScan &d = ps[si];
      d.ncomp = ms.comps_in_scan;
      d.Ss = ms.Ss; d.Se = ms.Se; d.Ah = ms.Ah; d.Al = ms.Al;
      d.slot[0] = d.slot[1] = -1;
      d.frame_header = si == 0;
      if (p->optimize_scans) {   // jcmaster.c:487-497
        if (si >= lfs && si < nsl) d.al_sel = 1;
        if (si >= cfs) d.al_sel = 2;
        if (si >= 6 && si <= 8) d.cond = 1;     // luma candidates of level Al = 2 / 3: coded only where the level below improved
        if (si >= 9 && si <= 11) d.cond = 2;
      }
      for (int ci = 0; ci < ms.comps_in_scan; ci++) {
        const int c = ms.component_index[ci];
        d.comp[ci] = c;
        d.comp_id[ci] = p->component_id[c];
        d.td[ci] = (ms.Ss == 0 && ms.Ah == 0) ? p->dc_tbl_no[c] : 0;   // emit_sos jcmarker.c:519-523
        d.ta[ci] = ms.Se ? p->ac_tbl_no[c] : 0;
        if (ms.Ss == 0) {
          if (ms.Ah == 0) {
            const int t = p->dc_tbl_no[c], cls = dc_class(p, c);     // (check_supported: at most two distinct table numbers per image)
            if (d.slot[cls] < 0) {
              d.slot[cls] = SLOT_PROG + 2 * si + cls;
              d.dht_slot[d.ndht] = d.slot[cls]; d.dht_id[d.ndht] = t; d.ndht++;
            }
          }
        } else {
          d.slot[0] = SLOT_PROG + 2 * si;
          d.dht_slot[0] = d.slot[0]; d.dht_id[0] = 0x10 + p->ac_tbl_no[c]; d.ndht = 1;
        }
      }
    }
    for (int band = 0; band < e->nbands; band++)
      for (int c = 0; c < C.ncomp; c++) {
        MjhProgScan &d = ps[p->num_scans + band * C.ncomp + c];
        d.ncomp = 1; d.comp[0] = c; d.Ah = p->trellis_stats_Ah; d.Al = p->trellis_stats_Al;   // jcmaster.c:451-466, T15: whatever the object's last coded scan left
        d.Ss = e->nbands == 1 ? 1 : band == 0 ? 1 : e->freq_split + 1;
        d.Se = e->nbands == 1 ? 63 : band == 0 ? e->freq_split : 63;
        if (d.Se < d.Ss) { d.Ss = 1; d.Se = 63; }         // empty band: its passes are skipped, the descriptor only has to be valid
        d.slot[0] = 2 * c + 1; d.slot[1] = -1; d.seed = 1;
      }
    // restart intervals are a per-scan quantity (per_scan_setup jcmaster.c:595-600, T10): `restart_in_rows` rows of the
    // SCAN's MCUs (a single-component scan's MCU is one block, its row is width_in_blocks blocks), capped at 65535
    {
      int last_ri = 0, total = 0;   // write_file_header resets last_restart_interval to 0 (jcmarker.c:660)
      for (size_t si = 0; si < ps.size(); si++) {
        MjhProgScan &d = ps[si];
        const MjhComp &c0 = C.c[d.comp[0]];
        const long units = d.ncomp > 1 ? (long)C.mcus_per_row * C.mcu_rows : (long)c0.nblk;
        const long per_row = d.ncomp > 1 ? C.mcus_per_row : c0.wib;
        long ri = p->restart_interval;
        if (p->restart_in_rows > 0) { ri = (long)p->restart_in_rows * per_row; if (ri > 65535L) ri = 65535L; }
        d.ri = (int)ri;
        d.nrst = ri > 0 && units > 0 ? (int)((units - 1) / ri) : 0;
        d.mpos_off = total;
        total += d.nrst;
        if ((int)si < p->num_scans) { d.emit_dri = d.ri != last_ri; last_ri = d.ri; }   // the trellis statistics scans write no header
      }
      e->mpos_per_image = total;
      HIPCHK_E(mjh_dmalloc((void **)&e->d_prog_mpos, (B * (size_t)total + 16) * sizeof(unsigned)));
    }
    HIPCHK_E(mjh_dmalloc((void **)&e->d_prog_ffsums, B * ps.size() * 8 * sizeof(unsigned)));   // PROG_STUFF_SPLIT shares per (scan, image)
    HIPCHK_E(mjh_dmalloc(&e->d_prog_scans, ps.size() * sizeof(MjhProgScan)));
    HIPCHK_E(hipMemcpy(e->d_prog_scans, ps.data(), ps.size() * sizeof(MjhProgScan), hipMemcpyHostToDevice));
    HIPCHK_E(mjh_dmalloc(&e->d_prog_ctl, B * sizeof(MjhProgCtl)));
    // scan lists per phase (+ the table slots each phase has to build)
    auto add_list = [&](const std::vector<int> &scn) {
      mjh_encoder::PList pl;
      pl.scan_off = (int)e->h_lists.size(); pl.nscan = (int)scn.size();
      e->h_lists.insert(e->h_lists.end(), scn.begin(), scn.end());
      pl.slot_off = (int)e->h_lists.size(); pl.nslot = 0;
      for (int si : scn)
        for (int t = 0; t < 2; t++)
          if (ps[si].slot[t] >= 0 && (ps[si].Ss != 0 || ps[si].Ah == 0)) { e->h_lists.push_back(ps[si].slot[t]); pl.nslot++; }
      // scans without a restart interval go through the parallel chain (mjh_prog.hip), the others are walked in order
      // (the first-pass AC scans first: the chain runs them in kernels of their own, mjh_launch_prog_encode)
      pl.par_off = (int)e->h_lists.size(); pl.npar = 0; pl.nacf = 0; pl.any_refine = false;
      for (int si : scn) if (ps[si].ri == 0 && ps[si].Ss != 0 && ps[si].Ah == 0) { e->h_lists.push_back(si); pl.npar++; pl.nacf++; }
      for (int si : scn) if (ps[si].ri == 0 && !(ps[si].Ss != 0 && ps[si].Ah == 0)) { e->h_lists.push_back(si); pl.npar++; if (ps[si].Ss != 0 && ps[si].Ah != 0) pl.any_refine = true; }
      pl.seq_off = (int)e->h_lists.size(); pl.nseq = 0;
      for (int si : scn) if (ps[si].ri != 0) { e->h_lists.push_back(si); pl.nseq++; }
      return pl;
    };
    std::vector<int> tr[2], a, a2, a3, b;
    for (int band = 0; band < e->nbands; band++)
      for (int c = 0; c < C.ncomp; c++) tr[band].push_back(p->num_scans + band * C.ncomp + c);
    if (p->optimize_scans) {
      // phase A: everything the Al decisions need, in three sub-phases -- the reference stops coding luma candidates at the
      // first successive-approximation level that does not pay, so levels 2 and 3 (scans 6-8, 9-11) wait for the verdict on
      // the level below and are skipped, image by image, where it was negative; phase B: the frequency-split candidates
      for (int si = 0; si < p->num_scans; si++) {
        if ((si >= lfs && si < nsl) || si >= cfs) b.push_back(si);
        else if (si >= 6 && si <= 8) a2.push_back(si);
        else if (si >= 9 && si <= 11) a3.push_back(si);
        else a.push_back(si);
      }
      e->nphases = 4;
    } else {
      for (int si = 0; si < p->num_scans; si++) a.push_back(si);
      e->nphases = 1;
    }
    e->pl_trellis[0] = add_list(tr[0]);
    e->pl_trellis[1] = add_list(tr[1]);
    for (int band = 0; band < e->nbands; band++)
      for (int c = 0; c < C.ncomp; c++) e->pl_trellis_c[band][c] = add_list(std::vector<int>{ p->num_scans + band * C.ncomp + c });
    if (p->optimize_scans) { e->pl_phase[0] = add_list(a); e->pl_phase[1] = add_list(a2); e->pl_phase[2] = add_list(a3); e->pl_phase[3] = add_list(b); }
    else e->pl_phase[0] = add_list(a);
    {
      int mx = 0, maxlist = 1;
      for (int c = 0; c < C.ncomp; c++) mx = C.c[c].nblk > mx ? C.c[c].nblk : mx;
      e->chunks_per_scan = (mx + MJH_PSTAT_BLOCKS - 1) / MJH_PSTAT_BLOCKS;
      for (const mjh_encoder::PList *pl : { &e->pl_trellis[0], &e->pl_trellis[1], &e->pl_phase[0], &e->pl_phase[1], &e->pl_phase[2], &e->pl_phase[3] }) maxlist = pl->npar > maxlist ? pl->npar : maxlist;
      HIPCHK_E(mjh_dmalloc(&e->d_prog_chunks, B * (size_t)maxlist * e->chunks_per_scan * sizeof(MjhProgChunk)));
      // parallel encode of the AC-first scans: block lengths / runs / offsets per (scan, image) pair
      const int maxpar = maxlist;
      e->pe.chunks_per_scan = e->chunks_per_scan;
      e->pe.nblk_pad = e->chunks_per_scan * MJH_PSTAT_BLOCKS;
      e->pe.chunks = (MjhProgChunk *)e->d_prog_chunks;
      if (maxpar > 0) {
        const size_t pairs = B * (size_t)maxpar, ent = pairs * (size_t)e->pe.nblk_pad, nch = pairs * (size_t)e->chunks_per_scan;
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.len16, ent * 2));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.run16, ent * 2));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.tail16, ent * 2));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.be16, ent * 2));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.off32, ent * 4));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.T32, ent * 4));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.sums, nch * 4));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.tsums, nch * 4));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.totals, pairs * 4));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.ttotals, pairs * 4));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.ne_bits, nch * (MJH_PSTAT_BLOCKS / 64) * 8));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.e_bits, nch * (MJH_PSTAT_BLOCKS / 64) * 8));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.ne2_bits, nch * (MJH_PSTAT_BLOCKS / 64) * 8));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.info, pairs * sizeof(MjhProgPair)));
        HIPCHK_E(mjh_dmalloc((void **)&e->pe.chist, nch * 256 * 4));   // symbol counts per chunk: 1 KB per (scan, image, chunk of 2048 blocks)
        int maxoth = 0;   // scans of a list that are not first-pass AC scans (refinement and DC scans): the refinement masks of their blocks
        for (const mjh_encoder::PList *pl : { &e->pl_phase[0], &e->pl_phase[1], &e->pl_phase[2], &e->pl_phase[3] }) maxoth = pl->npar - pl->nacf > maxoth ? pl->npar - pl->nacf : maxoth;
        if (maxoth > 0 && e->use_compact) HIPCHK_E(mjh_dmalloc((void **)&e->pe.rmask, B * (size_t)maxoth * 3 * e->pe.nblk_pad * 8));
      }
    }
    HIPCHK_E(mjh_dmalloc((void **)&e->d_lists, e->h_lists.size() * sizeof(int) + 16));
    HIPCHK_E(hipMemcpy(e->d_lists, e->h_lists.data(), e->h_lists.size() * sizeof(int), hipMemcpyHostToDevice));
    // every candidate scan of the search keeps its own bit stream: the bands are coded ~11 times over
    e->pool_words = e->stream_words * (p->optimize_scans ? 8 : 2);
    if (e->pool_words > ((size_t)1 << 27)) e->pool_words = (size_t)1 << 27;   // 32-bit bit offsets
    e->outpool_bytes = (size_t)1280 * (p->num_scans + 1) + 8 * e->pool_words;
    HIPCHK_E(mjh_dmalloc((void **)&e->d_pool, B * e->pool_words * 4 + 4096));   // slack: a bit writer may touch two words past its last offset
    HIPCHK_E(mjh_dmalloc((void **)&e->d_outpool, B * e->outpool_bytes));
  }
  HIPCHK_E(hipDeviceSynchronize());
  *out = e;
  return MJH_OK;
}

// ---- the kernel schedule ---------------------------------------------------------------------------
struct Prof {
  mjh_encoder *e;
  hipStream_t s;
  size_t next = 0;
  bool first_call = true, enabled = false, prev_focus = false;
  void mark(const char *name)
  {
    if (mjh_guard_serial()) {   // MJH_GUARD + MJH_GUARD_LOG: one step at a time, named in the log before it is queued
      (void)hipStreamSynchronize(s);
      (void)hipStreamSynchronize(e->side_stream);
      mjh_guard_note(name);
    }
    if (!enabled) return;
    const bool focus = name && strcmp(name, e->prof_focus) == 0;
    const bool closing = prev_focus && !focus;          // the mark that ends the focus kernel's interval
    prev_focus = focus;
    if (e->profiling == 2 && !focus && !closing) return;
    if (next >= e->prof_events.size()) { hipEvent_t ev; (void)hipEventCreate(&ev); e->prof_events.push_back(ev); }
    (void)hipEventRecord(e->prof_events[next++], s);
    if (first_call && name && !(e->profiling == 2 && !focus)) e->prof_names.push_back(name);
  }
  void finish()
  {
    if (!enabled) return;
    if (first_call) e->prof_per_call = next;
    e->prof_calls++;
  }
};

// input_read (optional) is recorded once the kernels that read the caller's input have been queued; before_output
// (optional) is waited for before the first kernel that overwrites the output files of the previous batch
// One component sampled with V > 1: its trellis passes walk iMCU rows of V block rows (compress_trellis_pass jccoefct.c:418-441) although
// nothing else in the file knows about V -- the kernels that chain the DC trellis get the geometry with that v and its iMCU row count.
static MjhConst dc_chain_view(const mjh_encoder *e, const MjhConst &CV)
{
  MjhConst D = CV;
  if (e->dc_chain_v > 1 && CV.ncomp == 1) { D.c[0].v = e->dc_chain_v; D.mcu_rows = (CV.c[0].hib + e->dc_chain_v - 1) / e->dc_chain_v; }
  return D;
}

static int run_pipeline(mjh_encoder *e, const void *d_pixels, size_t row_pitch, size_t image_stride, int n, hipStream_t s,
                        const MjhPlaneSrc *plane_src = nullptr, const MjhCoefSrc *coef_src = nullptr,
                        hipEvent_t input_read = nullptr, hipEvent_t before_output = nullptr)
{
  const MjhConst &C = e->C;
  const mjh_params &p = e->p;
  const int spi = e->spi;
  e->sizes_valid = false;
  e->last_n = n;
  e->res_buf = -1;
  e->coef_input = coef_src != nullptr;
  e->last_stream = s;
  if (!e->owner) e->last = nullptr;      // (mjh_encode_device names the twin afterwards when it ran the batch)
  // two batches in flight: the other buffer set's AC trellis kernel has the chip for itself -- nothing of this batch starts before it has ended
  mjh_encoder *const peer = e->owner ? e->owner : e->twin;
  const int if_mode = peer ? (e->owner ? e->owner : e)->inflight_mode : 0;      // 1: nothing next to the other set's AC trellis; 2: only the memory- / latency-bound kernels
  const bool ordered = if_mode >= 1;
  if (if_mode == 1 && peer->ev_tier1_set) HIPCHK(hipStreamWaitEvent(s, peer->ev_tier1, 0));
  e->ev_tier1_set = false;
  const bool peer_done_set = peer && peer->ev_done_set;
  if (if_mode == 2) e->ev_done_set = false;
  Prof pr{ e, s };
  if (e->profiling && e->prof_calls < 256) {
    pr.enabled = true;
    pr.first_call = e->prof_calls == 0;
    if (pr.first_call) e->prof_names.clear();
    pr.next = (size_t)e->prof_calls * e->prof_per_call;
  }
  HIPCHK(hipMemcpyAsync(e->d_tabs, e->d_tabs_init, (size_t)n * spi * sizeof(MjhHuffTable), hipMemcpyDeviceToDevice, s));
  if (!coef_src && e->fdct_div_zero)
    return fail(MJH_EINVAL, "a quantization step of 8192, 16384 or 24576 with 8-bit samples: the reference's FDCT manager divides by zero there (compute_reciprocal, jcdctmgr.c:182-203 with `quantval << 3` as its UINT16 argument)");
  if (coef_src) {    // jpeg_write_coefficients: the caller's quantized blocks go straight to the entropy-coding passes
    pr.mark("import_coefs");
    mjh_launch_import_coefs(C, *coef_src, e->d_q, e->d_meta, n, s);
  } else {
    if (plane_src) {   // jpeg_write_raw_data: the caller's component planes replace colour conversion + downsampling
      pr.mark("import_planes");
      mjh_launch_import_planes(C, *plane_src, e->d_planes, n, s);
    } else {
      pr.mark("color");
      mjh_launch_color(C, d_pixels, row_pitch, image_stride, e->d_planes, n, s);
    }
    if (input_read) HIPCHK(hipEventRecord(input_read, s));
  }
  int tr_dc[4], tr_ac[4], fin_dc[4], fin_ac[4], zero4[4] = { 0, 0, 0, 0 };
  for (int i = 0; i < 4; i++) {
    tr_dc[i] = 2 * i; tr_ac[i] = 2 * i + 1;
    fin_dc[i] = SLOT_FINAL + 2 * (i < C.ncomp ? p.dc_tbl_no[i] : 0);
    fin_ac[i] = SLOT_FINAL + 2 * (i < C.ncomp ? p.ac_tbl_no[i] : 0) + 1;
  }
  // Sequential mode with trellis quantization: the AC statistics of the conventionally quantized blocks are gathered by
  // the FDCT kernel itself (no separate pass over the 63 planes), and the quantized AC planes it would write are never
  // read (the trellis recomputes them), so they are not stored; optionally the statistics of the final coefficients
  // are gathered by the trellis back-track.
  const bool fuse_seq = !e->progressive && p.trellis_quant && !coef_src && !e->arith;
  // The pre-trellis AC statistics are gathered inside the FDCT kernel (debug taps expose the pre-trellis quantized planes, so
  // they keep the unfused schedule).  Measured: the fused FDCT kernel costs what the separate statistics pass cost (7.19 vs
  // 7.23 ms per 64 4K frames) but moves 3.2 GB less.  The FINAL statistics inside either trellis kernel cost more than their
  // own pass over the compact records (rounds 2 and 3: 3.83 vs 3.36 + 0.38 ms; 2.77 vs 2.36 + 0.33 ms) and were removed in round 5.
  // band-limited passes (use_scans_in_trellis) keep the other band's quantized planes, and the block-row
  // pass of trellis_eob_opt changes coefficients behind the per-block DP: neither fusion applies then
  const bool ext_eob = p.trellis_quant && p.trellis_eob_opt, ext_qopt = p.trellis_quant && p.trellis_q_opt;
  const bool compact = e->use_compact && !coef_src;   // (coefficient input runs no trellis: planes stay one per position)
  unsigned long long *const nzm = compact ? e->d_nzmask : nullptr;
  e->compact_last = compact;
  const int nbands = p.trellis_quant ? e->nbands : 1;
  const bool fuse_pre = fuse_seq && !e->debug_taps && nbands == 1 && !ext_qopt && p.dct_method == 0;   // (q_opt: one component at a time, each from the stored planes)
  // The first tier's queue capacity of the AC trellis trades LDS occupancy (16 records: 15 waves per CU, 48: 5) against the
  // share of blocks that have to be redone by the general big-capacity tier.  The first tier counts, whatever its own
  // capacity, how many blocks of the batch have more than 16 / 24 / 32 records (count_heavy); the counts of an earlier
  // batch (copied back asynchronously together with the number of frames they belong to, looked at only once the
  // event behind the copy has completed, never waited for) give the capacity directly: the smallest one that leaves less
  // than ~6 % of the blocks to the general tier.  The counts do not depend on the capacity in use, so a steady workload
  // settles on ONE plan; a move down needs the share to fall to half the ceiling (a workload sitting on a ceiling does not
  // alternate).  Every capacity gives the same bytes: this is a performance choice only.
  auto adapt_first_tier = [&]() -> int {
    if (!e->trellis_adapt || !e->defer_pending) return MJH_OK;
    const hipError_t qs = hipEventQuery(e->ev_defer);
    if (qs == hipErrorNotReady) { (void)hipGetLastError(); return MJH_OK; }
    HIPCHK(qs);
    e->defer_pending = false;
    const double blocks = ((double)e->defer_frames * (double)C.total_real_blocks + 1.0) / (double)e->defer_scale;
    const double s16 = e->h_defer[1] / blocks, s24 = e->h_defer[2] / blocks, s32 = e->h_defer[3] / blocks;
    const int cur = e->trellis_variant == 1 ? 2 : e->trellis_variant;
    auto level_for = [&](double slack) { return s16 < 0.06 * slack ? 0 : s24 < 0.06 * slack ? 2 : s32 < 0.10 * slack ? 3 : 4; };
    const int up = level_for(1.0), down = level_for(0.5);
    if (up > cur) e->trellis_variant = up;
    else if (down < cur) e->trellis_variant = down;
    return MJH_OK;
  };
  if (ext_qopt) {
    // trellis_q_opt re-estimates d_quant per image during the encode: the FDCT and the conventional quantization of THIS
    // encode start from the parameters' tables again (the reference's tables of a new jpeg_start_compress), not from what
    // the previous call of this encoder left
    for (int i = 0; i < n; i++)
      HIPCHK(hipMemcpyAsync(e->d_quant + i, e->d_quant_init, sizeof(MjhQuant), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemsetAsync(e->d_qsums, 0, (size_t)n * 4 * 64 * 2 * sizeof(long long), s));
  }
  if (!coef_src) {
    if (if_mode == 2 && peer->ev_tier1_set) HIPCHK(hipStreamWaitEvent(s, peer->ev_tier1, 0));     // (the colour kernel is bound by HBM: it may run next to the trellis)
    pr.mark("dct_quant");
    mjh_launch_dct(C, e->d_quant, e->d_planes, e->d_uq, e->d_q, e->d_lambda, fuse_pre ? e->d_tabs : nullptr, spi, tr_ac, e->d_nq8, n, s, e->fastdiv_all, p.dct_method == 1);
  }

  if (e->arith) {
    // Arithmetic coding: [trellis pass of component 0 with the coder's own rate model] -> size the scans (scan search: phase by
    // phase, with the Huffman path's selection kernels) -> lay the chosen scans out -> code them into the file.  A script of
    // one scan without search is coded in one pass.
    if (before_output) { HIPCHK(hipStreamWaitEvent(s, before_output, 0)); before_output = nullptr; }
    mjh_launch_prog_reset(e->d_prog_ctl, e->arith_nscans, n, s);
    if (p.trellis_quant && !coef_src) {
      if (e->debug_taps) {
        if (!e->d_q0) HIPCHK(mjh_dmalloc((void **)&e->d_q0, (size_t)e->max_batch * C.coefs_per_image * 2));
        HIPCHK(hipMemcpyAsync(e->d_q0, e->d_q, (size_t)n * C.coefs_per_image * 2, hipMemcpyDeviceToDevice, s));
      }
      const int split = p.trellis_freq_split > 0 ? p.trellis_freq_split : 8;
      const int updates = ext_qopt ? arith_qopt_updates(&p) : 0;
      const int runs = ext_qopt ? updates + (arith_qopt_passes(&p) > updates * (p.use_scans_in_trellis ? 4 : 2) * C.ncomp ? 1 : 0) : 1;   // (see arith_qopt_passes)
      MjhConst C0 = C;       // component 0 alone: the sums of trellis_q_opt
      C0.ncomp = 1;
      for (int r = 0; r < runs; r++) {
        pr.mark("trellis_arith");
        mjh_launch_trellis_arith(dc_chain_view(e, C), e->d_quant, ext_qopt ? 1 : 0, e->d_uq, e->d_q, e->d_lambda, e->d_arith_rates, e->d_back, 1, p.use_scans_in_trellis ? split : 63,
                                 p.trellis_quant_dc, p.trellis_delta_dc_weight, e->comp_restart[0], e->progressive ? 1 : 0, n, s);
        if (r < updates) {
          pr.mark("trellis_q_opt(sums)");
          mjh_launch_qopt_accumulate(C0, e->d_uq, e->d_q, e->d_qsums, n, s);
          pr.mark("trellis_q_opt(tables)");
          mjh_launch_qopt_update(e->d_qsums, e->d_quant, n, s);
        }
      }
    }
    const int whole = e->progressive ? 0 : 1;
    if (e->arith_nscans == 1) {
      pr.mark("arith_encode");
      mjh_launch_arith_scans(C, e->d_prog_scans, e->d_lists + e->pl_phase[0].scan_off, 1, e->d_prog_ctl, e->d_q, e->d_frame_hdr, e->frame_hdr_len,
                             e->d_prefix, e->file_hdr_len, e->d_out, e->out_stride, e->d_sizes, whole, 2, n, s);
    } else {
      static const char *const kSize[4] = { "arith_size(A)", "arith_size(A2)", "arith_size(A3)", "arith_size(B)" };
      for (int ph = 0; ph < e->nphases; ph++) {
        const mjh_encoder::PList &pl = e->pl_phase[ph];
        pr.mark(kSize[e->nphases == 4 ? ph : 0]);
        if (pl.nscan)
          mjh_launch_arith_scans(C, e->d_prog_scans, e->d_lists + pl.scan_off, pl.nscan, e->d_prog_ctl, e->d_q, e->d_frame_hdr, e->frame_hdr_len,
                                 e->d_prefix, e->file_hdr_len, e->d_out, e->out_stride, e->d_sizes, whole, 0, n, s);
        if (p.optimize_scans) { pr.mark("prog_select"); mjh_launch_prog_select(e->d_prog_ctl, C.ncomp, ph, p.dc_scan_opt_mode, n, s); }
      }
      pr.mark("arith_layout");
      mjh_launch_arith_layout(e->d_prog_ctl, e->d_prefix, e->file_hdr_len, e->d_out, e->out_stride, e->d_sizes, n, s);
      pr.mark("arith_encode");
      mjh_launch_arith_scans(C, e->d_prog_scans, e->d_lists, e->arith_nscans, e->d_prog_ctl, e->d_q, e->d_frame_hdr, e->frame_hdr_len,
                             e->d_prefix, e->file_hdr_len, e->d_out, e->out_stride, e->d_sizes, whole, 1, n, s);
    }
    if (ext_qopt && !coef_src) {   // the final tables into the DQT marker(s) (jcmaster.c:1014-1030; SOF9 / SOF10 whatever their precision)
      pr.mark("trellis_q_opt(DQT)");
      mjh_launch_qopt_fix(e->d_quant, e->d_out, e->out_stride, e->d_sizes, e->file_hdr_len, e->prefix_len - (10 + 3 * C.ncomp), e->dqt_tabs, e->dqt_ntab,
                          p.compress_profile != MJH_PROFILE_FASTEST, 0, n, s);
    }
    pr.mark(nullptr);
    pr.finish();
    HIPCHK(hipGetLastError());
    return MJH_OK;
  }
  if (e->progressive) {
    if (before_output) { HIPCHK(hipStreamWaitEvent(s, before_output, 0)); before_output = nullptr; }   // the hand-over also reads the scan control block
    mjh_launch_prog_reset(e->d_prog_ctl, e->nscans, n, s);   // (the scan pool is zeroed phase by phase, only what k_prog_alloc hands out)
  }
  // trellis_num_loops (statistics, trellis) rounds (jcmaster.c:451-466): every round gathers the statistics of the
  // current quantized coefficients and re-runs the trellis from the unquantized ones (components are independent,
  // so doing all of them per round equals the reference's component-major order); with use_scans_in_trellis a round
  // is two such pass pairs, AC bands 1..split and split+1..63.
  // One (statistics, trellis) pass pair: for all components (CV = C) or, with trellis_q_opt, for ONE component through a
  // one-component view of the geometry (MjhComp carries absolute offsets, so the view addresses the same buffers).
  const int nloops = p.trellis_quant && !e->arith ? (p.trellis_num_loops > 1 ? p.trellis_num_loops : 1) : 0;   // (the arithmetic coder has its own trellis pass below)
  bool join_late = false;            // the side stream (late DC chains + final DC statistics) is joined in front of the final tables
  bool final_dc_counted = false;     // ... and the side stream the DC statistics, right behind the DC trellis (under the AC kernel)
  auto trellis_pass = [&](const MjhConst &CV, const int *sl_dc_seq, const int *sl_dc_prog, const int *sl_ac, const int *crst,
                          const mjh_encoder::PList *plt, int Ss, int Se, bool first_pass, bool last_loop, int qstride) -> int {
    if (!first_pass)   // fresh (zero) statistics for this pass
      HIPCHK(hipMemcpyAsync(e->d_tabs, e->d_tabs_init, (size_t)n * spi * sizeof(MjhHuffTable), hipMemcpyDeviceToDevice, s));
    const int *sl_dc = sl_dc_seq;
    if (!e->progressive) {
      // passes 0,2,4 of SURVEY 3.3 (statistics of the conventionally quantized component; the sequential coder's gather
      // counts whole blocks whatever the band, jchuff.c:812-915): AC part fused into the FDCT kernel in the first pass,
      // a pass over the previous result afterwards ...
      if (!first_pass || !fuse_pre) {
        pr.mark("stats_ac(pre-trellis)");
        mjh_launch_stats_ac(CV, e->d_q, nullptr, e->d_tabs, spi, sl_ac, 0, n, s);
      }
      pr.mark("stats_dc(pre-trellis)");
      mjh_launch_stats_dc(CV, e->d_q, e->d_tabs, spi, sl_dc, 0, crst, n, s);
      int slots[8], ns = 0;
      for (int i = 0; i < CV.ncomp; i++) { slots[ns++] = sl_dc[i]; slots[ns++] = sl_ac[i]; }
      pr.mark("gen_tables(trellis)");
      mjh_launch_gen_tables(e->d_tabs, spi, slots, ns, n, s);
    } else {
      // progressive: the trellis passes gather AC-first statistics (Ss..Se of the band, Al=0, seeded counts,
      // jcphuff.c:257-264); the DC rate table stays the STANDARD table
      pr.mark("prog_stats(pre-trellis)");
      if (plt->nseq)
        mjh_launch_prog_stats(C, e->d_prog_scans, e->d_lists + plt->seq_off, plt->nseq, e->d_prog_ctl, e->d_q, e->d_tabs, spi, e->d_prog_mpos, e->mpos_per_image, n, s);
      mjh_launch_prog_stats_par(C, e->d_prog_scans, e->d_lists + plt->par_off, plt->npar, e->d_prog_ctl, e->d_q, e->d_tabs, spi,
                                e->pe, plt->any_refine, nullptr, n, s, plt->nacf);   // (the conventionally quantized planes: one per position)
      pr.mark("gen_tables(trellis)");
      mjh_launch_gen_tables_list(e->d_tabs, spi, e->d_lists + plt->slot_off, plt->nslot, n, s);
      sl_dc = sl_dc_prog;
    }
    if (e->debug_taps && first_pass) {
      if (!e->d_q0) HIPCHK(mjh_dmalloc((void **)&e->d_q0, (size_t)e->max_batch * C.coefs_per_image * 2));
      HIPCHK(hipMemcpyAsync(e->d_q0, e->d_q, (size_t)n * C.coefs_per_image * 2, hipMemcpyDeviceToDevice, s));
    }
    // ... passes 1,3,5: trellis quantization with those tables.  The DC Viterbi (a few hundred
    // latency-bound chains) and the AC DP (every block) touch disjoint coefficient planes, so the
    // DC kernel runs on a side stream underneath the AC kernel.
    e->side_timed = false;
    bool dc_late = false;
    const int dc_mode = e->dc_mode;   // experiments (MJH_DC_MODE): 1 = DC trellis on the main stream, before the AC kernel
    if (p.trellis_quant_dc && dc_mode == 1) {
      pr.mark("trellis_dc(serial)");
      mjh_launch_trellis_dc(dc_chain_view(e, CV), e->d_quant, e->d_uq, e->d_q, e->d_tabs, spi, sl_dc, e->d_lambda, e->d_back, n, s, e->dc_window_ok);
    } else if (p.trellis_quant_dc) {
      HIPCHK(hipEventRecord(e->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
      if (pr.enabled && e->profiling == 1) {
        while (e->side_events.size() < 2 * (size_t)(e->prof_calls + 1)) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); e->side_events.push_back(ev); }
        HIPCHK(hipEventRecord(e->side_events[2 * e->prof_calls], e->side_stream));
      }
      // one or two frames: the DC trellis is the critical path of the whole encode then (a luma chain = the block rows of an iMCU
      // row one after the other): every block row is walked for every value the row above can end on, in parallel, and a second
      // kernel back-tracks the walks whose hypothesis held (mjh_kernels.hip, K6 speculative)
      const size_t spec_bytes = 2 * (size_t)C.total_real_blocks * 9 * 16;
      const bool spec = e->dc_spec && n <= 2 && qstride == 0 && spec_bytes <= ((size_t)256 << 20) && mjh_trellis_dc_speculative_ok(CV, e->dc_window_ok);
      if (spec && !e->d_back9) {
        int rows = 0;
        for (int c = 0; c < C.ncomp; c++) rows += C.c[c].hib;
        HIPCHK(mjh_dmalloc((void **)&e->d_back9, spec_bytes));
        HIPCHK(mjh_dmalloc((void **)&e->d_jfin, 2 * (size_t)rows * 9 * sizeof(int)));
        HIPCHK(mjh_dmalloc((void **)&e->d_qspec, 2 * (size_t)C.total_real_blocks * 9 * sizeof(int16_t)));
      }
      // The AC and the DC trellis share one VALU budget while they run side by side, and behind the big AC kernel the main
      // stream has ~0.4 ms of small, latency-bound kernels (the general tiers, the final AC statistics) that leave the chip
      // mostly idle.  So, for a large sequential batch, only the LUMA chains (two thirds of the DC work) start next to the AC
      // kernel; the chroma chains and the final DC statistics start when it has finished and run under that tail, and the
      // main stream joins the side stream in front of the final tables instead of behind the trellis.
      const bool final_dc_here = !e->progressive && p.optimize_coding && last_loop && nbands == 1 && qstride == 0 && !ext_eob && e->dc_stats_side && e->seq_scans.empty();
      // (the late chains hang on the tile-sorted AC kernel's event: without that kernel -- MJH_TRELLIS_V3=0, a capacity beyond its tiers --
      // every chain starts here, the older schedule)
      const bool sorted_tier = nzm && e->d_nq8 && e->trellis_v3 > 0 && e->trellis_variant <= 4;
      dc_late = e->dc_late > 0 && sorted_tier && !spec && final_dc_here && nloops == 1 && CV.ncomp == 3 && !e->debug_taps && (size_t)n * C.total_real_blocks >= e->small_batch;
      if (spec) mjh_launch_trellis_dc_speculative(CV, e->d_quant, e->d_uq, e->d_q, e->d_tabs, spi, sl_dc, e->d_lambda, e->d_back9, e->d_jfin, e->d_qspec, n, e->side_stream);
      else
      mjh_launch_trellis_dc(dc_chain_view(e, CV), e->d_quant, e->d_uq, e->d_q, e->d_tabs, spi, sl_dc, e->d_lambda, e->d_back, n, e->side_stream, e->dc_window_ok,
                            0, dc_late ? e->dc_late * CV.mcu_rows : -1);   // (the DC entries never change: image 0's tables serve all)
      if (pr.enabled && e->profiling == 1) { HIPCHK(hipEventRecord(e->side_events[2 * e->prof_calls + 1], e->side_stream)); e->side_timed = true; }
      if (final_dc_here && !dc_late) {
        // the final DC statistics need nothing but the DC trellis's result: counted here, they cost no time of their own
        mjh_launch_stats_dc(C, e->d_q, e->d_tabs, spi, fin_dc, 1, zero4, n, e->side_stream);
        final_dc_counted = true;
      }
      if (!dc_late) HIPCHK(hipEventRecord(e->ev_join, e->side_stream));
    }
    const bool extended = nbands > 1 || ext_eob || qstride != 0;
    if (!extended) { const int rc = adapt_first_tier(); if (rc != MJH_OK) return rc; }
    const bool v3 = nzm && e->d_nq8 && e->trellis_v3 > 0 && e->trellis_variant <= 4 && !extended;     // the tile-sorted first tier (plain compact pass)
    // ... and this batch's AC trellis waits for the tail of the other set's batch (queued before this call)
    const bool excl = ordered && v3 && !e->progressive;
    if (excl && peer_done_set) HIPCHK(hipStreamWaitEvent(s, peer->ev_done, 0));
    // Image ranges of the first tier (mjh_launch_trellis_ac): measured per 64 4K frames, two ranges shorten the trellis interval
    // from 1.72 to 1.61 ms at the same step time; 128 1080p frames: 0.97 -> 0.92 ms; 64 1080p frames lose 6 % (each range's
    // launch drains before the next starts), a progressive batch gains nothing (profiles/r06f_chunks.md): two ranges from six
    // million blocks of a sequential batch on, one otherwise
    const int trellis_ranges = !v3 || e->debug_taps || (size_t)n * C.total_real_blocks < e->small_batch ? 1
                               : e->trellis_chunks > 0 ? e->trellis_chunks
                               : !e->progressive && (size_t)n * C.total_real_blocks >= (size_t)6000000 ? 2 : 1;
    // the record counts the first tier's capacity is chosen by: from one tile in eight of a large batch (scaled back below)
    const int count_mask = v3 && (size_t)n * C.total_real_blocks >= (size_t)2000000 ? 7 : 0;
    pr.mark("trellis_ac");
    mjh_launch_trellis_ac(CV, e->d_quant, e->d_uq, e->d_q, e->d_tabs, spi, sl_ac, e->d_lambda, e->d_worklist, e->d_worklist2, e->d_dense, e->dense_cap,
                          e->trellis_variant,
                          Ss, Se, ext_eob ? e->d_eob_cost : nullptr, ext_eob ? e->d_eob_has : nullptr, nzm, qstride, n, s,
                          e->d_nq8, v3 ? ((size_t)n * C.total_real_blocks < e->small_batch ? 1 : e->trellis_v3) : 0,   // (a small batch: one pass per tile -- four times the workgroups, a quarter of their length: latency matters more than the sorting)
                          e->fastdiv_all, dc_late ? e->ev_side0 : nullptr, excl ? e->ev_tier1 : nullptr,
                          trellis_ranges, e->side_stream, e->ev_chunk, count_mask);
    if (excl) e->ev_tier1_set = true;
    if (dc_late) {
      if (!v3) return fail(MJH_EINVAL, "internal: the late DC chains need the tile-sorted trellis' event");
      HIPCHK(hipStreamWaitEvent(e->side_stream, e->ev_side0, 0));
      mjh_launch_trellis_dc(CV, e->d_quant, e->d_uq, e->d_q, e->d_tabs, spi, sl_dc, e->d_lambda, e->d_back, n, e->side_stream, e->dc_window_ok, e->dc_late * CV.mcu_rows, -1);
      mjh_launch_stats_dc(C, e->d_q, e->d_tabs, spi, fin_dc, 1, zero4, n, e->side_stream);
      final_dc_counted = true;
      HIPCHK(hipEventRecord(e->ev_join, e->side_stream));
      join_late = true;
    }
    if (e->trellis_adapt && !extended && first_pass && !e->defer_pending) {   // (one read-back in flight at a time; its frame count travels with it)
      if (!e->ev_defer) HIPCHK(hipEventCreateWithFlags(&e->ev_defer, hipEventDisableTiming));
      e->defer_frames = n;
      e->defer_scale = count_mask + 1;
      HIPCHK(hipMemcpyAsync(&e->h_defer[0], e->d_worklist, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
      HIPCHK(hipEventRecord(e->ev_defer, s));
      e->defer_pending = true;
    }
    if (ext_eob) {   // jcdctmgr.c:1224-1297: end-of-band runs along every block row, with the band's AC rate table
      pr.mark("trellis_eob_runs");
      mjh_launch_trellis_eob_chain(CV, e->d_q, e->d_tabs, spi, sl_ac, e->d_eob_cost, e->d_eob_has, Ss, Se, n, s);
    }
    if (ext_qopt) {  // :1299-1306
      pr.mark("trellis_q_opt(sums)");
      mjh_launch_qopt_accumulate(CV, e->d_uq, e->d_q, e->d_qsums, n, s);
    }
    if (p.trellis_quant_dc && dc_mode != 1 && !dc_late) {
      pr.mark("join(trellis_dc)");
      HIPCHK(hipStreamWaitEvent(s, e->ev_join, 0));
    }
    return MJH_OK;
  };
  auto band_limits = [&](int band, int &Ss, int &Se) {
    Ss = nbands == 1 ? 1 : band == 0 ? 1 : e->freq_split + 1;
    Se = nbands == 1 ? 63 : band == 0 ? e->freq_split : 63;
  };
  if (!ext_qopt) {
    bool first_pass = true;
    for (int loop = 0; loop < nloops; loop++)
      for (int band = 0; band < nbands; band++) {
        int Ss, Se;
        band_limits(band, Ss, Se);
        if (Se < Ss) continue;   // quantize_trellis returns at once (jcdctmgr.c:979-980); the statistics of that pass feed nothing
        const int rc = trellis_pass(C, tr_dc, fin_dc, tr_ac, e->comp_restart, &e->pl_trellis[band], Ss, Se, first_pass, loop == nloops - 1, 0);
        if (rc != MJH_OK) return rc;
        first_pass = false;
      }
  } else {
    // trellis_q_opt: the reference walks its passes component-major (component, round, band) and re-estimates the
    // quantization tables after every group of num_components (component, round) units (prepare_for_pass jcmaster.c:687-698,
    // finish_pass_master :1014-1030), so with more than one round a later component is quantized with tables estimated from
    // earlier ones: the same order here, one component at a time, every image with its own table set.
    // (every image's table set was reset to the parameters' tables in front of the FDCT launch)
    bool first_pass = true;
    for (int u = 0; u < C.ncomp * nloops; u++) {
      const int ci = u / nloops, loop = u % nloops;
      MjhConst CV = C;
      CV.ncomp = 1;
      CV.c[0] = C.c[ci];
      const int v_dc_seq[4] = { tr_dc[ci], 0, 0, 0 }, v_dc_prog[4] = { fin_dc[ci], 0, 0, 0 }, v_ac[4] = { tr_ac[ci], 0, 0, 0 };
      const int v_rst[4] = { e->comp_restart[ci], 0, 0, 0 };
      for (int band = 0; band < nbands; band++) {
        int Ss, Se;
        band_limits(band, Ss, Se);
        if (Se < Ss) continue;
        const int rc = trellis_pass(CV, v_dc_seq, v_dc_prog, v_ac, v_rst, &e->pl_trellis_c[band][ci], Ss, Se, first_pass, loop == nloops - 1, 1);
        if (rc != MJH_OK) return rc;
        first_pass = false;
      }
      if ((u + 1) % C.ncomp == 0) { pr.mark("trellis_q_opt(tables)"); mjh_launch_qopt_update(e->d_qsums, e->d_quant, n, s); }
    }
  }
  if (e->progressive) {
    // every candidate scan of a phase: statistics -> optimal tables -> exact size -> headers, bits, stuffing
    for (int ph = 0; ph < e->nphases; ph++) {
      const mjh_encoder::PList &pl = e->pl_phase[ph];
      static const char *const kStats[4] = { "prog_stats(A)", "prog_stats(A2)", "prog_stats(A3)", "prog_stats(B)" };
      static const char *const kTabs[4] = { "gen_tables(A)", "gen_tables(A2)", "gen_tables(A3)", "gen_tables(B)" };
      static const char *const kEnc[4] = { "prog_encode(A)", "prog_encode(A2)", "prog_encode(A3)", "prog_encode(B)" };
      const int pn = p.optimize_scans ? ph : 0;
      pr.mark(kStats[pn]);
      // the sequential walks (refinement / DC / restart scans: a few long workgroups) and the parallel AC-first
      // statistics touch different table slots: the latter run on the side stream underneath the former
      const bool both = pl.nseq > 0 && pl.npar > 0;
      hipStream_t ps = s;
      if (both) {
        HIPCHK(hipEventRecord(e->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
        ps = e->side_stream;
      }
      mjh_launch_prog_stats_par(C, e->d_prog_scans, e->d_lists + pl.par_off, pl.npar, e->d_prog_ctl, e->d_q, e->d_tabs, spi,
                                e->pe, pl.any_refine, nzm, n, ps, pl.nacf);
      if (both) HIPCHK(hipEventRecord(e->ev_join, e->side_stream));
      if (pl.nseq)
        mjh_launch_prog_stats(C, e->d_prog_scans, e->d_lists + pl.seq_off, pl.nseq, e->d_prog_ctl, e->d_q, e->d_tabs, spi, e->d_prog_mpos, e->mpos_per_image, n, s);
      if (both) HIPCHK(hipStreamWaitEvent(s, e->ev_join, 0));
      pr.mark(kTabs[pn]);
      mjh_launch_gen_tables_list(e->d_tabs, spi, e->d_lists + pl.slot_off, pl.nslot, n, s);
      pr.mark(kEnc[pn]);
      mjh_launch_prog_encode(C, e->d_prog_scans, e->d_lists + pl.scan_off, pl.nscan, e->d_lists + pl.seq_off, pl.nseq, e->d_lists + pl.par_off, pl.npar,
                             e->pe, e->d_prog_ctl, e->d_q, e->d_tabs, spi, e->d_pool, e->pool_words,
                             e->d_frame_hdr, e->frame_hdr_len, p.compress_profile != MJH_PROFILE_FASTEST, e->d_outpool, e->outpool_bytes,
                             e->d_prog_mpos, e->mpos_per_image, e->d_prog_ffsums, nzm, n, s, e->side_stream, e->ev_fork, e->ev_join, pl.nacf);
      if (p.optimize_scans) { pr.mark("prog_select"); mjh_launch_prog_select(e->d_prog_ctl, C.ncomp, ph, p.dc_scan_opt_mode, n, s); }
    }
    if (before_output) HIPCHK(hipStreamWaitEvent(s, before_output, 0));
    pr.mark("prog_concat");
    mjh_launch_prog_concat(e->d_prog_ctl, e->d_prefix, e->file_hdr_len, e->d_outpool, e->outpool_bytes, e->d_out, e->out_stride, e->d_sizes, n, s);
    if (ext_qopt) {   // jcmaster.c:1014-1030 + the precision rule of jcmarker.c:189-254
      pr.mark("trellis_q_opt(DQT)");
      mjh_launch_qopt_fix(e->d_quant, e->d_out, e->out_stride, e->d_sizes, e->file_hdr_len, e->prefix_len - (10 + 3 * C.ncomp), e->dqt_tabs, e->dqt_ntab,
                          p.compress_profile != MJH_PROFILE_FASTEST, !e->progressive && C.precision == 8 && e->tbl_le1, n, s);
    }
    pr.mark(nullptr);
    pr.finish();
    HIPCHK(hipGetLastError());
    return MJH_OK;
  }
  if (!e->seq_scans.empty() && !e->arith) {
    // A sequential script of several scans: per scan, through its view of the geometry, the statistics of ITS MCU order -> its
    // tables -> its header behind the file so far -> its entropy-coded data (jcmaster.c:1090-1101: two passes per scan with
    // optimal tables).  The scans reuse the table slots and the buffers of the one-scan path, one after the other.
    if (before_output) HIPCHK(hipStreamWaitEvent(s, before_output, 0));
    for (size_t si = 0; si < e->seq_scans.size(); si++) {
      const mjh_encoder::SeqScan &q = e->seq_scans[si];
      const MjhConst V = scan_view(C, p, q.comp, q.ncomp);
      int v_dc[4] = { 0, 0, 0, 0 }, v_ac[4] = { 0, 0, 0, 0 };
      for (int j = 0; j < q.ncomp; j++) { v_dc[j] = fin_dc[q.comp[j]]; v_ac[j] = fin_ac[q.comp[j]]; }
      if (p.optimize_coding) {
        HIPCHK(hipMemcpyAsync(e->d_tabs, e->d_tabs_init, (size_t)n * spi * sizeof(MjhHuffTable), hipMemcpyDeviceToDevice, s));   // fresh counts
        pr.mark("stats_ac(final)");
        mjh_launch_stats_ac(V, e->d_q, nzm, e->d_tabs, spi, v_ac, 1, n, s);
        pr.mark("stats_dc(final)");
        mjh_launch_stats_dc(V, e->d_q, e->d_tabs, spi, v_dc, 1, zero4, n, s);
        pr.mark("gen_tables(final)");
        int slots[8], ns = 0;
        for (int j = 0; j < q.ncomp; j++) {
          bool have_d = false, have_a = false;
          for (int t = 0; t < ns; t++) { have_d = have_d || slots[t] == v_dc[j]; have_a = have_a || slots[t] == v_ac[j]; }
          if (!have_d) slots[ns++] = v_dc[j];
          if (!have_a) slots[ns++] = v_ac[j];
        }
        mjh_launch_gen_tables(e->d_tabs, spi, slots, ns, n, s);
      }
      pr.mark("header");
      mjh_launch_header(si == 0 ? e->d_prefix : nullptr, si == 0 ? e->prefix_len : 0, e->d_sos + q.sos_off, q.sos_len, e->d_tabs, spi, q.dht_slots, q.dht_ids, q.ndht,
                        p.compress_profile != MJH_PROFILE_FASTEST, e->d_out, e->out_stride, e->d_meta, n, s, si == 0 ? nullptr : e->d_sizes);
      pr.mark("huff_encode");
      mjh_launch_encode(V, e->d_q, nzm, e->d_tabs, spi, v_dc, v_ac, e->d_len16, e->d_off32, e->d_sums, e->chunks, e->d_totals,
                        e->d_stream, e->stream_words, e->d_meta, e->d_seg_x, e->d_seg_E, e->d_seg_sums, e->d_seg_totals, e->d_mpos, q.nseg, n, s);
      pr.mark("byte_stuff");
      mjh_launch_stuff(e->d_stream, e->stream_words, e->d_totals, e->d_ffsums, e->ff_chunks, e->d_fftotals, e->d_out, e->out_stride,
                       e->d_meta, e->d_sizes, e->d_mpos, q.nseg, n, s);
    }
  } else {
    if (p.optimize_coding) {
      // pass 6: statistics of the interleaved scan (dummy blocks included) -> final tables
      pr.mark("stats_ac(final)");
      mjh_launch_stats_ac(C, e->d_q, nzm, e->d_tabs, spi, fin_ac, 1, n, s);
      if (!final_dc_counted) {
        pr.mark("stats_dc(final)");
        mjh_launch_stats_dc(C, e->d_q, e->d_tabs, spi, fin_dc, 1, zero4, n, s);
      }
      if (join_late) {
        pr.mark("join(trellis_dc)");
        HIPCHK(hipStreamWaitEvent(s, e->ev_join, 0));
      }
      pr.mark("gen_tables(final)");
      mjh_launch_gen_tables(e->d_tabs, spi, e->dht_slots, e->ndht, n, s);
    }
    // pass 7: headers + entropy-coded data
    if (before_output) HIPCHK(hipStreamWaitEvent(s, before_output, 0));
    pr.mark("header");
    mjh_launch_header(e->d_prefix, e->prefix_len, e->d_sos, e->sos_len, e->d_tabs, spi, e->dht_slots, e->dht_ids, e->ndht,
                      p.compress_profile != MJH_PROFILE_FASTEST, e->d_out, e->out_stride, e->d_meta, n, s);
    pr.mark("huff_encode");
    mjh_launch_encode(C, e->d_q, nzm, e->d_tabs, spi, fin_dc, fin_ac, e->d_len16, e->d_off32, e->d_sums, e->chunks, e->d_totals,
                      e->d_stream, e->stream_words, e->d_meta, e->d_seg_x, e->d_seg_E, e->d_seg_sums, e->d_seg_totals, e->d_mpos, e->nseg, n, s);
    if (if_mode == 2) { HIPCHK(hipEventRecord(e->ev_done, s)); e->ev_done_set = true; }      // (the bit writer is the tail's last VALU-bound kernel)
    pr.mark("byte_stuff");
    mjh_launch_stuff(e->d_stream, e->stream_words, e->d_totals, e->d_ffsums, e->ff_chunks, e->d_fftotals, e->d_out, e->out_stride,
                     e->d_meta, e->d_sizes, e->d_mpos, e->nseg, n, s);
  }
  if (ext_qopt) {   // jcmaster.c:1014-1030 + the precision rule of jcmarker.c:189-254
    pr.mark("trellis_q_opt(DQT)");
    mjh_launch_qopt_fix(e->d_quant, e->d_out, e->out_stride, e->d_sizes, e->file_hdr_len, e->prefix_len - (10 + 3 * C.ncomp), e->dqt_tabs, e->dqt_ntab,
                        p.compress_profile != MJH_PROFILE_FASTEST, !e->progressive && C.precision == 8 && e->tbl_le1, n, s);
  }
  pr.mark(nullptr);
  pr.finish();
  if (peer && !(if_mode == 2 && e->ev_done_set)) { HIPCHK(hipEventRecord(e->ev_done, s)); e->ev_done_set = true; }
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

extern "C" const mjh_params *mjh_encoder_params(const mjh_encoder *e) { return e ? &e->p : nullptr; }

// the encoder whose buffers hold the most recent batch: the twin, when it ran the last device-resident call
static mjh_encoder *cur(mjh_encoder *e) { return e && e->last ? e->last : e; }

// 1: every device-resident batch runs on the encoder's own buffer set, one after the other; 2 (the default, MJH_INFLIGHT): two sets take turns
extern "C" int mjh_set_inflight(mjh_encoder *e, int batches)
{
  if (!e || batches < 1 || batches > 2) return fail(MJH_EINVAL, "batches in flight: 1 or 2");
  e->inflight = batches;
  return MJH_OK;
}

// the second buffer set of an encoder with two batches in flight: a complete encoder of its own on streams of the other queue pool
static int make_twin(mjh_encoder *e)
{
  mjh_encoder *t = nullptr;
  g_create_twin = true;
  g_twin_avoid[0] = e->stream; g_twin_avoid[1] = e->side_stream;
  const int rc = mjh_encoder_create(&e->p_created, e->max_batch, e->device, &t);
  g_create_twin = false;
  if (rc) return rc;
  t->owner = e;
  t->inflight = 1;
  t->profiling = e->profiling; t->prof_focus_name = e->prof_focus_name; t->prof_focus = e->prof_focus;
  e->twin = t;
  return MJH_OK;
}

// An entry other than mjh_encode_host behind a mjh_encode_host call: that batch's files may still be on their way out of
// the (single) output buffers on the D2H stream -- nothing of the new batch may run before they have left
static int wait_pending_pack(mjh_encoder *e, hipStream_t s)
{
  if (e->res_buf >= 0 && !e->res_waited[e->res_buf]) HIPCHK(hipStreamWaitEvent(s, e->ev_packed[e->res_buf], 0));
  return MJH_OK;
}

// MJH_GUARD=2/3: a device entry point works on a fenced copy of the caller's input that holds exactly the bytes the entry
// may read (last image, last row, last sample), so that an over-read of the CALLER's memory faults like any other
static int guard_input(mjh_encoder *e, int slot, const void **p, size_t bytes)
{
  if (mjh_guard_mode() < 2 || !*p || !bytes) return MJH_OK;
  static const bool off = getenv("MJH_GUARD_INPUT") && atoi(getenv("MJH_GUARD_INPUT")) == 0;
  if (off) return MJH_OK;
  HIPCHK(hipDeviceSynchronize());
  // a copy buffer is kept for the next call of the same size, and outgrown ones live until the encoder goes: un-mapping a
  // buffer and mapping the next one in the same call sequence gave copies that differed from their source on this runtime
  // (profiles/r04a_guard_notes.md), so nothing is un-mapped while the encoder works
  if (e->g_in[slot] && e->g_in_bytes[slot] != bytes) { e->g_in_old.push_back(e->g_in[slot]); e->g_in[slot] = nullptr; }
  if (!e->g_in[slot]) HIPCHK(mjh_dmalloc(&e->g_in[slot], bytes));
  e->g_in_bytes[slot] = bytes;
  HIPCHK(hipMemcpy(e->g_in[slot], *p, bytes, hipMemcpyDeviceToDevice));
  HIPCHK(hipDeviceSynchronize());   // (a device-to-device copy returns before it is done, and the encoder's streams do not wait for the null stream)
  if (getenv("MJH_GUARD_VERIFY_COPY") && bytes <= ((size_t)1 << 28)) {
    std::vector<uint8_t> a(bytes), b(bytes);
    HIPCHK(hipMemcpy(a.data(), *p, bytes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), e->g_in[slot], bytes, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < bytes; i++) if (a[i] != b[i]) { if (!bad) first = i; bad++; }
    fprintf(stderr, "guard_input: %zu bytes, %zu differ (first at %zu) src %p copy %p\n", bytes, bad, first, *p, e->g_in[slot]);
  }
  *p = e->g_in[slot];
  return MJH_OK;
}

extern "C" int mjh_encode_device(mjh_encoder *e, const void *d_pixels, size_t row_pitch, size_t image_stride, int n, void *stream)
{
  if (!e || !d_pixels || n < 1 || n > e->max_batch) return fail(MJH_EINVAL, "bad arguments (n=%d, max_batch=%d)", n, e ? e->max_batch : 0);
  {
    const size_t row_bytes = (size_t)e->C.W * e->C.px_size * (e->C.precision == 12 ? 2 : 1);
    if (row_pitch < row_bytes || (n > 1 && image_stride < row_pitch * (size_t)(e->C.H - 1) + row_bytes))
      return fail(MJH_EINVAL, "row_pitch %zu / image_stride %zu too small for %dx%d images of %zu-byte rows", row_pitch, image_stride, e->C.W, e->C.H, row_bytes);
  }
  HIPCHK(hipSetDevice(e->device));
  {
    const size_t row_bytes = (size_t)e->C.W * e->C.px_size * (e->C.precision == 12 ? 2 : 1);
    const int rg = guard_input(e, 0, &d_pixels, (size_t)(n - 1) * image_stride + (size_t)(e->C.H - 1) * row_pitch + row_bytes);
    if (rg) return rg;
  }
  hipStream_t s = stream ? (hipStream_t)stream : e->stream;
  if (stream == (void *)1) {
    // "behind the null stream": the encoder's own (non-blocking) stream does not synchronise with the legacy default stream
    // by itself -- an event recorded there now orders the encode behind everything queued on it so far
    if (!e->ev_null_in) HIPCHK(hipEventCreateWithFlags(&e->ev_null_in, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->ev_null_in, nullptr));
    s = e->stream;
    HIPCHK(hipStreamWaitEvent(s, e->ev_null_in, 0));
  }
  { const int rcw = wait_pending_pack(e, s); if (rcw) return rcw; }
  // Two batches in flight: on the encoder's own stream consecutive calls alternate between the two buffer sets (the results of
  // call k stay where they are until call k + 2).  Debug taps, the memory checker and a caller's stream keep to one set.
  mjh_encoder *t = e;
  if (e->inflight > 1 && !stream && !e->debug_taps && mjh_guard_mode() == 0 && !e->owner) {
    if (!e->twin && make_twin(e) != MJH_OK) e->inflight = 1;      // (no room for a second set: one batch at a time)
    if (e->twin && (e->dev_calls++ & 1u)) t = e->twin;
    if (t != e) s = t->stream;
  }
  const int rc = run_pipeline(t, d_pixels, row_pitch, image_stride, n, s);
  e->last = t == e ? nullptr : t;
  return rc;
}

// ---- host entry: pixels in host memory -> JPEG files in host memory ----------------------------------
// Staging copies of callers whose pixels live in pageable memory are spread over a few worker threads (one core
// moves ~5-10 GB/s, the host link wants ~50); callers that hand over pinned memory (mjh_host_alloc /
// mjh_host_register) skip the staging copy altogether.
namespace {
class CopyPool {
 public:
  // never destroyed: the detached workers wait on its condition variable for the life of the process (a static
  // object's destructor would block in pthread_cond_destroy at exit)
  static CopyPool &get() { static CopyPool *p = new CopyPool; return *p; }
  int threads() const { return nthreads_; }
  // run fn(i) for i in [0, njobs) on the pool (the caller works too); returns when all are done
  void run(int njobs, const std::function<void(int)> &fn)
  {
    if (njobs <= 0) return;
    if (nthreads_ <= 1 || njobs == 1) { for (int i = 0; i < njobs; i++) fn(i); return; }
    std::unique_lock<std::mutex> big(submit_);   // one batch of jobs at a time
    // a batch is an object of its own: a worker that still holds the previous batch only ever sees THAT batch's
    // (exhausted) counters, whatever the job counts of consecutive batches are
    auto job = std::make_shared<Job>();
    job->fn = &fn; job->njobs = njobs;
    {
      std::lock_guard<std::mutex> lk(m_);
      cur_ = job;
      gen_++;
    }
    cv_.notify_all();
    work(*job);
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return job->done == job->njobs; });
    cur_.reset();      // (fn dies with the caller's frame: nobody may start a job of this batch any more -- next is exhausted)
  }
 private:
  struct Job {
    const std::function<void(int)> *fn = nullptr;
    int njobs = 0, done = 0;          // done: under m_
    std::atomic<int> next{ 0 };
  };
  CopyPool()
  {
    int n = 0;
    if (const char *v = getenv("MJH_HOST_THREADS")) n = atoi(v);
    if (n <= 0) { n = (int)std::thread::hardware_concurrency() / 2; if (n > 8) n = 8; }
    if (n < 1) n = 1;
    nthreads_ = n;
    for (int i = 1; i < n; i++) std::thread([this] { loop(); }).detach();
  }
  void work(Job &job)
  {
    for (;;) {
      const int i = job.next.fetch_add(1);
      if (i >= job.njobs) break;
      (*job.fn)(i);
      std::lock_guard<std::mutex> lk(m_);
      if (++job.done == job.njobs) cv_done_.notify_all();
    }
  }
  void loop()
  {
    unsigned long seen = 0;
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        job = cur_;
      }
      if (job) work(*job);
    }
  }
  std::mutex m_, submit_;
  std::condition_variable cv_, cv_done_;
  std::shared_ptr<Job> cur_;
  int nthreads_ = 1;
  unsigned long gen_ = 0;
};
}  // namespace

extern "C" void *mjh_host_alloc(size_t bytes)
{
  void *p = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  // (pinned on the NUMA node of the calling thread's current device: the frames are read by that device's DMA engines)
  if (mjh_numa_host_alloc(&p, bytes ? bytes : 1, hipHostMallocDefault, dev) != hipSuccess) { (void)hipGetLastError(); fail(MJH_ENOMEM, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
extern "C" void mjh_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int mjh_host_register(void *p, size_t bytes)
{
  if (!p || !bytes) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return MJH_OK;
}
extern "C" int mjh_host_unregister(void *p)
{
  if (!p) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipHostUnregister(p));
  return MJH_OK;
}

static bool is_pinned(const void *p)
{
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof(a));
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

static int host_buffers(mjh_encoder *e)
{
  if (e->d_pixb[0]) return MJH_OK;
  const size_t in_bytes = (size_t)e->max_batch * e->pix_image_bytes;
  // results: a JPEG file is normally a small fraction of its input; files that do not fit the arena are fetched from
  // the device buffer one by one (slow path, flagged in the table)
  e->res_cap = (size_t)e->max_batch * (e->pix_image_bytes / 2 + 65536);
  HIPCHK(hipStreamCreateWithPriority(&e->d2h_stream, hipStreamNonBlocking, e->copy_prio));
  for (int b = 0; b < 2; b++) {
    HIPCHK(mjh_dmalloc((void **)&e->d_pixb[b], in_bytes));
    HIPCHK(mjh_guard_host_alloc((void **)&e->h_res[b], e->res_cap, hipHostMallocMapped, "h_res"));
    HIPCHK(mjh_guard_host_alloc((void **)&e->h_tab[b], (2 + 2 * (size_t)e->max_batch) * sizeof(unsigned long long), hipHostMallocMapped, "h_tab"));
    HIPCHK(hipEventCreateWithFlags(&e->ev_h2d[b], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_pix_free[b], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_packed[b], hipEventDisableTiming));
  }
  return MJH_OK;
}

// after a batch has been queued on e->stream: pack its files into the pinned arena `b` on the D2H stream
static int queue_pack(mjh_encoder *e, int b, int n)
{
  // The files leave for the host on a stream of their own so that batch k's hand-over runs under batch k + 1's kernels.  One
  // image at a time (a synchronous libjpeg client) has nothing to overlap with: there the hand-over stays on the main stream
  // and saves the cross-stream hop (an event wait between hardware queues costs tens of microseconds of a ~1.4 ms image;
  // measured A/B on one client thread, 4K: 720 / 721 images/s against 673 / 697 with the hop, profiles/r05n_dropin_handover_ab.md).
  hipStream_t ps = n == 1 ? e->stream : e->d2h_stream;
  if (ps != e->stream) {
    HIPCHK(hipEventRecord(e->ev_join, e->stream));
    HIPCHK(hipStreamWaitEvent(ps, e->ev_join, 0));
  }
  void *d_res = nullptr, *d_tab = nullptr;
  HIPCHK(hipHostGetDevicePointer(&d_res, e->h_res[b], 0));
  HIPCHK(hipHostGetDevicePointer(&d_tab, e->h_tab[b], 0));
  mjh_launch_pack_results(e->d_out, e->out_stride, e->d_sizes, (e->progressive || e->arith) ? nullptr : e->d_meta, (e->progressive || e->arith) ? e->d_prog_ctl : nullptr,
                          n, d_res, e->res_cap, d_tab, ps);
  HIPCHK(hipEventRecord(e->ev_packed[b], ps));
  e->res_buf = b;
  e->res_n[b] = n;
  e->res_waited[b] = false;
  return MJH_OK;
}

extern "C" int mjh_encode_host(mjh_encoder *e, const void *pixels, size_t row_pitch, size_t image_stride, int n)
{
  if (!e || !pixels || n < 1 || n > e->max_batch) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipSetDevice(e->device));
  const size_t row_bytes = (size_t)e->C.W * e->C.px_size * (e->C.precision == 12 ? 2 : 1);
  if (row_pitch < row_bytes) return fail(MJH_EINVAL, "row_pitch %zu is smaller than a row (%zu bytes)", row_pitch, row_bytes);
  if (n > 1 && image_stride < row_pitch * (size_t)(e->C.H - 1) + row_bytes)
    return fail(MJH_EINVAL, "image_stride %zu is smaller than one image (%zu bytes): images would overlap", image_stride, row_pitch * (size_t)(e->C.H - 1) + row_bytes);
  int rc = host_buffers(e);
  if (rc) return rc;
  const int b = (int)(e->host_calls & 1u);
  const int H = e->C.H;
  // (mjh_stage_commit: a prefix of the encoder's own packed staging buffer is on its way already)
  const size_t done = (pixels == e->h_stage[b] && row_pitch == row_bytes && image_stride == e->pix_image_bytes) ? e->staged[b] : 0;
  if (e->staged[b] != 0 && done == 0) {
    // the committed prefix belongs to a batch that never came: forget it (its copies are waited for), refuse this call
    HIPCHK(hipStreamSynchronize(e->copy_stream));
    e->staged[b] = 0;
    return fail(MJH_EINVAL, "mjh_stage_commit was used: the batch has to come from the staging buffer, packed");
  }
  e->host_calls++;
  const bool own_staging = pixels == e->h_stage[0] || pixels == e->h_stage[1];
  const bool direct = own_staging || is_pinned(pixels);
  // d_pixb[b] was last read by the colour kernel of the call before the previous one
  HIPCHK(hipStreamWaitEvent(e->copy_stream, e->ev_pix_free[b], 0));
  e->staged[b] = 0;
  if (direct) {
    // pinned source: the DMA engine reads the caller's memory; it must stay untouched until mjh_wait_input / mjh_collect
    for (int i = 0; i < n; i++) {
      const uint8_t *src = (const uint8_t *)pixels + (size_t)i * image_stride;
      uint8_t *dst = e->d_pixb[b] + (size_t)i * e->pix_image_bytes;
      const size_t lo = (size_t)i * e->pix_image_bytes, hi = lo + e->pix_image_bytes;
      if (done >= hi) continue;
      const size_t skip = done > lo ? done - lo : 0;
      if (row_pitch == row_bytes) HIPCHK(hipMemcpyAsync(dst + skip, src + skip, e->pix_image_bytes - skip, hipMemcpyHostToDevice, e->copy_stream));
      else HIPCHK(hipMemcpy2DAsync(dst, row_bytes, src, row_pitch, row_bytes, (size_t)H, hipMemcpyHostToDevice, e->copy_stream));
    }
  } else {
    if (!e->h_stage[b]) HIPCHK(mjh_numa_host_alloc((void **)&e->h_stage[b], (size_t)e->max_batch * e->pix_image_bytes, hipHostMallocDefault, e->device));
    HIPCHK(hipEventSynchronize(e->ev_h2d[b]));   // the staging buffer's previous H2D copy (two calls ago)
    CopyPool &pool = CopyPool::get();
    const int parts = pool.threads() > 1 ? pool.threads() * 2 : 1;   // row bands per image
    for (int i = 0; i < n; i++) {
      const uint8_t *src = (const uint8_t *)pixels + (size_t)i * image_stride;
      uint8_t *dst = e->h_stage[b] + (size_t)i * e->pix_image_bytes;
      pool.run(parts, [&](int part) {
        const int y0 = (int)((long long)H * part / parts), y1 = (int)((long long)H * (part + 1) / parts);
        if (row_pitch == row_bytes) memcpy(dst + (size_t)y0 * row_bytes, src + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes);
        else for (int y = y0; y < y1; y++) memcpy(dst + (size_t)y * row_bytes, src + (size_t)y * row_pitch, row_bytes);
      });
      // image i travels while image i+1 is being staged
      HIPCHK(hipMemcpyAsync(e->d_pixb[b] + (size_t)i * e->pix_image_bytes, dst, e->pix_image_bytes, hipMemcpyHostToDevice, e->copy_stream));
    }
  }
  HIPCHK(hipEventRecord(e->ev_h2d[b], e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->ev_h2d[b], 0));
  // the output buffers are single: the files of the previous batch must have left for the host before this batch's
  // header / stuffing kernels overwrite them (they are the last kernels of the schedule)
  rc = run_pipeline(e, e->d_pixb[b], row_bytes, e->pix_image_bytes, n, e->stream, nullptr, nullptr, e->ev_pix_free[b],
                    e->host_calls > 1 ? e->ev_packed[b ^ 1] : nullptr);
  if (rc) return rc;
  return queue_pack(e, b, n);
}

// One batch out of images that n OTHER encoders of the same parameters and device have staged (each through its own
// mjh_host_staging buffer, possibly mjh_stage_commit'ted): what is left of every member's host->device copy is queued on the
// member's copy stream, the images are gathered device-to-device into this encoder's input buffer, and the batch runs like a
// mjh_encode_host batch (results through mjh_collect / mjh_get_jpeg of THIS encoder, image i = members[i]).  The members'
// staging buffers flip as after a mjh_encode_host call of their own.  The caller guarantees that nobody else uses the member
// encoders during the call (the libjpeg shim: their client threads are blocked in jpeg_finish_compress, waiting for this batch).
extern "C" int mjh_encode_gather(mjh_encoder *e, mjh_encoder *const *members, int n)
{
  if (!e || !members || n < 1 || n > e->max_batch) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipSetDevice(e->device));
  int rc = host_buffers(e);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    const mjh_encoder *m = members[i];
    if (!m || m->device != e->device || m->pix_image_bytes != e->pix_image_bytes || memcmp(&m->p, &e->p, sizeof(mjh_params)) != 0)
      return fail(MJH_EINVAL, "member %d does not match the batch encoder (parameters / device)", i);
    if (!m->d_pixb[m->host_calls & 1u] || !m->h_stage[m->host_calls & 1u]) return fail(MJH_EINVAL, "member %d has nothing staged", i);
  }
  const int b = (int)(e->host_calls++ & 1u);
  HIPCHK(hipStreamWaitEvent(e->copy_stream, e->ev_pix_free[b], 0));   // this encoder's input buffer: its reader of two calls ago
  for (int i = 0; i < n; i++) {
    mjh_encoder *m = members[i];
    const int mb = (int)(m->host_calls & 1u);
    if (m->staged[mb] < m->pix_image_bytes) {
      if (m->staged[mb] == 0) HIPCHK(hipStreamWaitEvent(m->copy_stream, m->ev_pix_free[mb], 0));
      HIPCHK(hipMemcpyAsync(m->d_pixb[mb] + m->staged[mb], m->h_stage[mb] + m->staged[mb], m->pix_image_bytes - m->staged[mb], hipMemcpyHostToDevice, m->copy_stream));
    }
    m->staged[mb] = 0;
    HIPCHK(hipEventRecord(m->ev_h2d[mb], m->copy_stream));
    m->host_calls++;                                       // its next image goes to the other buffer
    m->res_buf = -1; m->sizes_valid = false; m->last_n = 0;
    HIPCHK(hipStreamWaitEvent(e->copy_stream, m->ev_h2d[mb], 0));
    HIPCHK(hipMemcpyAsync(e->d_pixb[b] + (size_t)i * e->pix_image_bytes, m->d_pixb[mb], e->pix_image_bytes, hipMemcpyDeviceToDevice, e->copy_stream));
    HIPCHK(hipEventRecord(m->ev_pix_free[mb], e->copy_stream));   // the member's device buffer is free once the gather copy has read it
  }
  HIPCHK(hipEventRecord(e->ev_h2d[b], e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->ev_h2d[b], 0));
  const size_t row_bytes = (size_t)e->C.W * e->C.px_size * (e->C.precision == 12 ? 2 : 1);
  rc = run_pipeline(e, e->d_pixb[b], row_bytes, e->pix_image_bytes, n, e->stream, nullptr, nullptr, e->ev_pix_free[b],
                    e->host_calls > 1 ? e->ev_packed[b ^ 1] : nullptr);
  if (rc) return rc;
  return queue_pack(e, b, n);
}

extern "C" int mjh_host_staging(mjh_encoder *e, void **buffer, size_t *bytes)
{
  if (!e || !buffer) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipSetDevice(e->device));
  int rc = host_buffers(e);
  if (rc) return rc;
  const int b = (int)(e->host_calls & 1u);   // the buffer the NEXT mjh_encode_host call uses
  if (!e->h_stage[b]) HIPCHK(mjh_numa_host_alloc((void **)&e->h_stage[b], (size_t)e->max_batch * e->pix_image_bytes, hipHostMallocDefault, e->device));
  HIPCHK(hipEventSynchronize(e->ev_h2d[b]));
  if (e->staged[b]) {
    // an image was abandoned after part of it had been committed (jpeg_abort_compress behind >= 256 scanlines): its queued
    // copies still read the buffer that is handed out again now, and the next image starts from byte 0
    HIPCHK(hipStreamSynchronize(e->copy_stream));
    e->staged[b] = 0;
  }
  *buffer = e->h_stage[b];
  if (bytes) *bytes = (size_t)e->max_batch * e->pix_image_bytes;
  return MJH_OK;
}

// The first `bytes` of the staging buffer handed out by mjh_host_staging (packed images, whole rows) are final: their host->device
// copy is queued now, so that the PCIe transfer runs while the caller is still producing the rest (a libjpeg client writes its
// scanlines one call at a time).  The mjh_encode_host call for that buffer then copies only what is left.
extern "C" int mjh_stage_commit(mjh_encoder *e, size_t bytes)
{
  if (!e) return fail(MJH_EINVAL, "null encoder");
  const int b = (int)(e->host_calls & 1u);
  if (!e->d_pixb[b] || !e->h_stage[b]) return fail(MJH_EINVAL, "mjh_stage_commit before mjh_host_staging");
  const size_t cap = (size_t)e->max_batch * e->pix_image_bytes;
  if (bytes > cap) bytes = cap;
  if (bytes <= e->staged[b]) return MJH_OK;
  HIPCHK(hipSetDevice(e->device));
  if (e->staged[b] == 0) HIPCHK(hipStreamWaitEvent(e->copy_stream, e->ev_pix_free[b], 0));   // the device buffer's previous reader (two calls ago)
  HIPCHK(hipMemcpyAsync(e->d_pixb[b] + e->staged[b], e->h_stage[b] + e->staged[b], bytes - e->staged[b], hipMemcpyHostToDevice, e->copy_stream));
  e->staged[b] = bytes;
  return MJH_OK;
}

extern "C" int mjh_wait_input(mjh_encoder *e)
{
  if (!e) return fail(MJH_EINVAL, "null encoder");
  if (e->host_calls == 0) return MJH_OK;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipEventSynchronize(e->ev_h2d[(e->host_calls - 1) & 1u]));
  return MJH_OK;
}

static int wait_results(mjh_encoder *e, int b)
{
  if (b < 0 || e->res_n[b] == 0) return fail(MJH_EINVAL, "no batch of this age was encoded through mjh_encode_host");
  if (!e->res_waited[b]) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventSynchronize(e->ev_packed[b]));
    e->res_waited[b] = true;
    { const int rg = guard_verify(); if (rg) return rg; }
  }
  const unsigned long long err = e->h_tab[b][1];
  if (err & 1) return fail(MJH_ETOOSMALL, "entropy-coded data of an image exceeds the 32-bit bit-offset range of one scan / the bit-stream pool");
  if (err & 2) return fail(MJH_EHIP, "internal: scan size prediction mismatch");
  return MJH_OK;
}

extern "C" int mjh_collect(mjh_encoder *e, int age, const void **base, const mjh_result **results, int *count)
{
  if (!e || age < 0 || age > 1) return fail(MJH_EINVAL, "bad arguments (age is 0 or 1)");
  if (e->res_buf < 0 && age == 0) return fail(MJH_EINVAL, "the last batch was not encoded through mjh_encode_host");
  const int b = age == 0 ? e->res_buf : (int)((e->host_calls - 2) & 1u);
  if (age == 1 && e->host_calls < 2) return fail(MJH_EINVAL, "there is no batch before the last one");
  int rc = wait_results(e, b);
  if (rc) return rc;
  const unsigned long long *t = e->h_tab[b];
  for (int i = 0; i < e->res_n[b]; i++)
    if (t[2 + 2 * i] == ~0ull) return fail(MJH_ETOOSMALL, "file %d does not fit the pinned result arena: fetch it with mjh_get_jpeg", i);
  if (base) *base = e->h_res[b];
  if (results) *results = reinterpret_cast<const mjh_result *>(t + 2);
  if (count) *count = e->res_n[b];
  return MJH_OK;
}

static int check_coef_args(mjh_encoder *e, const void *const coefs[], const size_t blocks_per_row[], int n)
{
  if (!e || !coefs || !blocks_per_row || n < 1 || n > e->max_batch)
    return fail(MJH_EINVAL, "bad arguments (n=%d, max_batch=%d)", n, e ? e->max_batch : 0);
  if (e->p.trellis_quant)
    return fail(MJH_EINVAL, "trellis quantization needs the unquantized DCT output: create the encoder with trellis_quant = 0 "
                            "for coefficient input (jpeg_copy_critical_parameters does the same, jctrans.c:102)");
  for (int c = 0; c < e->C.ncomp; c++)
    if (!coefs[c] || ((uintptr_t)coefs[c] & 3) || blocks_per_row[c] < (size_t)e->C.c[c].wib)
      return fail(MJH_EINVAL, "bad coefficient array %d (NULL, not 4-byte aligned, or fewer than %d blocks per row)", c, e->C.c[c].wib);
  return MJH_OK;
}

extern "C" int mjh_encode_coefficients_device(mjh_encoder *e, const void *const d_coefs[MJH_MAX_COMPS], const size_t blocks_per_row[MJH_MAX_COMPS],
                                              const size_t image_stride[MJH_MAX_COMPS], int n, void *stream)
{
  int rc = check_coef_args(e, d_coefs, blocks_per_row, n);
  if (rc) return rc;
  if (n > 1 && !image_stride) return fail(MJH_EINVAL, "image_stride is required for n > 1");
  HIPCHK(hipSetDevice(e->device));
  MjhCoefSrc cs;
  memset(&cs, 0, sizeof(cs));
  for (int c = 0; c < e->C.ncomp; c++) {
    cs.base[c] = d_coefs[c]; cs.blocks_per_row[c] = (long long)blocks_per_row[c];
    cs.stride[c] = image_stride ? (long long)image_stride[c] : 0;
    const int rg = guard_input(e, c, &cs.base[c], (size_t)(n - 1) * (size_t)cs.stride[c] + ((size_t)(e->C.c[c].hib - 1) * blocks_per_row[c] + (size_t)e->C.c[c].wib) * 128);
    if (rg) return rg;
  }
  { const int rcw = wait_pending_pack(e, stream ? (hipStream_t)stream : e->stream); if (rcw) return rcw; }
  return run_pipeline(e, nullptr, 0, 0, n, stream ? (hipStream_t)stream : e->stream, nullptr, &cs);
}

extern "C" int mjh_encode_coefficients_host(mjh_encoder *e, const void *const coefs[MJH_MAX_COMPS], const size_t blocks_per_row[MJH_MAX_COMPS],
                                            const size_t image_stride[MJH_MAX_COMPS], int n)
{
  int rc = check_coef_args(e, coefs, blocks_per_row, n);
  if (rc) return rc;
  if (n > 1 && !image_stride) return fail(MJH_EINVAL, "image_stride is required for n > 1");
  HIPCHK(hipSetDevice(e->device));
  // real blocks only, rows packed to width_in_blocks
  size_t off[MJH_MAXC], per_image = 0;
  for (int c = 0; c < e->C.ncomp; c++) { off[c] = per_image; per_image += (size_t)e->C.c[c].nblk * 128; }
  HIPCHK(hipStreamSynchronize(e->stream));   // the staging buffers may still feed the previous batch
  if (!e->d_cfin) {
    HIPCHK(mjh_dmalloc((void **)&e->d_cfin, (size_t)e->max_batch * per_image));
    HIPCHK(mjh_numa_host_alloc((void **)&e->h_cfin, (size_t)e->max_batch * per_image, hipHostMallocDefault, e->device));
  }
  for (int i = 0; i < n; i++) {
    for (int c = 0; c < e->C.ncomp; c++) {
      const uint8_t *src = (const uint8_t *)coefs[c] + (image_stride ? (size_t)i * image_stride[c] : 0);
      uint8_t *dst = e->h_cfin + (size_t)i * per_image + off[c];
      const size_t row = (size_t)e->C.c[c].wib * 128;
      for (int r = 0; r < e->C.c[c].hib; r++) memcpy(dst + (size_t)r * row, src + (size_t)r * blocks_per_row[c] * 128, row);
    }
    HIPCHK(hipMemcpyAsync(e->d_cfin + (size_t)i * per_image, e->h_cfin + (size_t)i * per_image, per_image, hipMemcpyHostToDevice, e->copy_stream));
  }
  HIPCHK(hipEventRecord(e->copy_done, e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->copy_done, 0));
  MjhCoefSrc cs;
  memset(&cs, 0, sizeof(cs));
  for (int c = 0; c < e->C.ncomp; c++) { cs.base[c] = e->d_cfin + off[c]; cs.blocks_per_row[c] = e->C.c[c].wib; cs.stride[c] = (long long)per_image; }
  { const int rcw = wait_pending_pack(e, e->stream); if (rcw) return rcw; }
  return run_pipeline(e, nullptr, 0, 0, n, e->stream, nullptr, &cs);
}

static int check_plane_args(mjh_encoder *e, const void *const planes[], const size_t row_pitch[], const int plane_width[],
                            const int plane_height[], int n)
{
  if (!e || !planes || !row_pitch || !plane_width || !plane_height || n < 1 || n > e->max_batch)
    return fail(MJH_EINVAL, "bad arguments (n=%d, max_batch=%d)", n, e ? e->max_batch : 0);
  const size_t ss = e->C.precision == 12 ? 2 : 1;
  for (int c = 0; c < e->C.ncomp; c++)
    if (!planes[c] || plane_width[c] < 1 || plane_height[c] < 1 || row_pitch[c] < (size_t)plane_width[c] * ss)
      return fail(MJH_EINVAL, "bad plane %d (pointer, size %dx%d or pitch %zu)", c, plane_width[c], plane_height[c], row_pitch[c]);
  return MJH_OK;
}

extern "C" int mjh_encode_planes_device(mjh_encoder *e, const void *const d_planes[MJH_MAX_COMPS], const size_t row_pitch[MJH_MAX_COMPS],
                                        const size_t image_stride[MJH_MAX_COMPS], const int plane_width[MJH_MAX_COMPS],
                                        const int plane_height[MJH_MAX_COMPS], int n, void *stream)
{
  int rc = check_plane_args(e, d_planes, row_pitch, plane_width, plane_height, n);
  if (rc) return rc;
  if (n > 1 && !image_stride) return fail(MJH_EINVAL, "image_stride is required for n > 1");
  HIPCHK(hipSetDevice(e->device));
  MjhPlaneSrc ps;
  memset(&ps, 0, sizeof(ps));
  for (int c = 0; c < e->C.ncomp; c++) {
    ps.base[c] = d_planes[c]; ps.pitch[c] = (long long)row_pitch[c]; ps.stride[c] = image_stride ? (long long)image_stride[c] : 0;
    ps.w[c] = plane_width[c]; ps.h[c] = plane_height[c];
    const int rg = guard_input(e, c, &ps.base[c], (size_t)(n - 1) * (size_t)ps.stride[c] + (size_t)(ps.h[c] - 1) * row_pitch[c] + (size_t)ps.w[c] * (e->C.precision == 12 ? 2 : 1));
    if (rg) return rg;
  }
  { const int rcw = wait_pending_pack(e, stream ? (hipStream_t)stream : e->stream); if (rcw) return rcw; }
  return run_pipeline(e, nullptr, 0, 0, n, stream ? (hipStream_t)stream : e->stream, &ps);
}

extern "C" int mjh_encode_planes_host(mjh_encoder *e, const void *const planes[MJH_MAX_COMPS], const size_t row_pitch[MJH_MAX_COMPS],
                                      const size_t image_stride[MJH_MAX_COMPS], const int plane_width[MJH_MAX_COMPS],
                                      const int plane_height[MJH_MAX_COMPS], int n)
{
  int rc = check_plane_args(e, planes, row_pitch, plane_width, plane_height, n);
  if (rc) return rc;
  if (n > 1 && !image_stride) return fail(MJH_EINVAL, "image_stride is required for n > 1");
  HIPCHK(hipSetDevice(e->device));
  const size_t ss = e->C.precision == 12 ? 2 : 1;
  // only the part of each plane the encoder reads (<= width_in_blocks*8 x height_in_blocks*8) is staged, tightly packed
  MjhPlaneSrc ps;
  memset(&ps, 0, sizeof(ps));
  size_t off[MJH_MAXC], per_image = 0;
  for (int c = 0; c < e->C.ncomp; c++) {
    ps.w[c] = plane_width[c] < e->C.c[c].pw ? plane_width[c] : e->C.c[c].pw;
    ps.h[c] = plane_height[c] < e->C.c[c].ph ? plane_height[c] : e->C.c[c].ph;
    ps.pitch[c] = (long long)((size_t)ps.w[c] * ss);
    off[c] = per_image;
    per_image += ((size_t)ps.w[c] * ps.h[c] * ss + 15) & ~(size_t)15;
  }
  const size_t cap = (size_t)e->C.planes_per_image * ss + 16 * MJH_MAXC;   // >= per_image by construction
  HIPCHK(hipStreamSynchronize(e->stream));   // the staging buffers may still feed the previous batch
  if (!e->d_plin) {
    HIPCHK(mjh_dmalloc((void **)&e->d_plin, (size_t)e->max_batch * cap));
    HIPCHK(mjh_numa_host_alloc((void **)&e->h_plin, (size_t)e->max_batch * cap, hipHostMallocDefault, e->device));
  }
  for (int i = 0; i < n; i++) {
    for (int c = 0; c < e->C.ncomp; c++) {
      const uint8_t *src = (const uint8_t *)planes[c] + (image_stride ? (size_t)i * image_stride[c] : 0);
      uint8_t *dst = e->h_plin + (size_t)i * per_image + off[c];
      for (int y = 0; y < ps.h[c]; y++) memcpy(dst + (size_t)y * ps.pitch[c], src + (size_t)y * row_pitch[c], (size_t)ps.pitch[c]);
    }
    HIPCHK(hipMemcpyAsync(e->d_plin + (size_t)i * per_image, e->h_plin + (size_t)i * per_image, per_image, hipMemcpyHostToDevice, e->copy_stream));
  }
  HIPCHK(hipEventRecord(e->copy_done, e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->copy_done, 0));
  for (int c = 0; c < e->C.ncomp; c++) { ps.base[c] = e->d_plin + off[c]; ps.stride[c] = (long long)per_image; }
  { const int rcw = wait_pending_pack(e, e->stream); if (rcw) return rcw; }
  return run_pipeline(e, nullptr, 0, 0, n, e->stream, &ps);
}

extern "C" int mjh_encoder_sync(mjh_encoder *e)
{
  if (!e) return fail(MJH_EINVAL, "null encoder");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->last_stream ? e->last_stream : e->stream));
  if (e->twin) HIPCHK(hipStreamSynchronize(e->twin->last_stream ? e->twin->last_stream : e->twin->stream));
  return guard_verify();
}

static int fetch_sizes(mjh_encoder *e)
{
  if (e->sizes_valid) return MJH_OK;
  if (e->res_buf >= 0) {   // encoded through mjh_encode_host: sizes and error flags came over with the files
    int rc = wait_results(e, e->res_buf);
    if (rc) return rc;
    for (int i = 0; i < e->last_n; i++) e->h_sizes[i] = (unsigned)e->h_tab[e->res_buf][3 + 2 * i];
    e->sizes_valid = true;
    return MJH_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  { const int rg = guard_verify(); if (rg) return rg; }
  HIPCHK(hipMemcpy(e->h_sizes.data(), e->d_sizes, (size_t)e->last_n * sizeof(unsigned), hipMemcpyDeviceToHost));
  if (e->progressive || e->arith) {
    std::vector<MjhProgCtl> ctl(e->last_n);
    HIPCHK(hipMemcpy(ctl.data(), e->d_prog_ctl, (size_t)e->last_n * sizeof(MjhProgCtl), hipMemcpyDeviceToHost));
    for (int i = 0; i < e->last_n; i++)
      if (ctl[i].error) return fail(ctl[i].error == 1 ? MJH_ETOOSMALL : MJH_EHIP, "progressive encode of image %d failed (%s)", i,
                                    ctl[i].error == 1 ? "scan buffers exceed the bit-stream pool" : "internal: scan size prediction mismatch");
  }
  else {
    std::vector<MjhImageMeta> mt(e->last_n);
    HIPCHK(hipMemcpy(mt.data(), e->d_meta, (size_t)e->last_n * sizeof(MjhImageMeta), hipMemcpyDeviceToHost));
    for (int i = 0; i < e->last_n; i++)
      if (mt[i].total_bits == 0xFFFFFFFFu) return fail(MJH_ETOOSMALL, "entropy-coded data of image %d exceeds the 32-bit (2^32 bits = 512 MB) offset range of one scan", i);

  }
  if (e->coef_input) {   // untrusted coefficients: the import / length kernels flag values no Huffman symbol exists for
    std::vector<MjhImageMeta> mt(e->last_n);
    HIPCHK(hipMemcpy(mt.data(), e->d_meta, (size_t)e->last_n * sizeof(MjhImageMeta), hipMemcpyDeviceToHost));
    for (int i = 0; i < e->last_n; i++)
      if (mt[i].bad_coef) return fail(MJH_EINVAL, "image %d holds a DCT coefficient out of range (JERR_BAD_DCT_COEF, jchuff.c:489,596,624)", i);
  }
  e->sizes_valid = true;
  return MJH_OK;
}

extern "C" int mjh_get_jpeg_size(mjh_encoder *e, int i, size_t *size)
{
  e = cur(e);
  if (!e || !size || i < 0 || i >= e->last_n) return fail(MJH_EINVAL, "bad image index");
  int rc = fetch_sizes(e);
  if (rc) return rc;
  *size = e->h_sizes[i];
  return MJH_OK;
}

extern "C" int mjh_get_jpeg(mjh_encoder *e, int i, void *dst, size_t cap, size_t *size)
{
  e = cur(e);
  if (!e || !dst || i < 0 || i >= e->last_n) return fail(MJH_EINVAL, "bad image index");
  int rc = fetch_sizes(e);
  if (rc) return rc;
  const size_t n = e->h_sizes[i];
  if (size) *size = n;
  if (n > cap) return fail(MJH_ETOOSMALL, "output buffer too small: need %zu bytes", n);
  if (e->res_buf >= 0 && e->h_tab[e->res_buf][2 + 2 * i] != ~0ull) {
    memcpy(dst, e->h_res[e->res_buf] + e->h_tab[e->res_buf][2 + 2 * i], n);
    return MJH_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpy(dst, e->d_out + (size_t)i * e->out_stride, n, hipMemcpyDeviceToHost));
  return MJH_OK;
}

extern "C" int mjh_get_output_device(mjh_encoder *e, void **d_base, size_t *stride, void **d_sizes)
{
  e = cur(e);
  if (!e) return fail(MJH_EINVAL, "null encoder");
  if (d_base) *d_base = e->d_out;
  if (stride) *stride = e->out_stride;
  if (d_sizes) *d_sizes = e->d_sizes;
  return MJH_OK;
}

extern "C" int mjh_set_debug_taps(mjh_encoder *e, int on) { if (!e) return fail(MJH_EINVAL, "null encoder"); e->debug_taps = on != 0; return MJH_OK; }
extern "C" int mjh_set_profiling(mjh_encoder *e, int on)
{
  if (!e) return fail(MJH_EINVAL, "null encoder");
  if (on < 0 || on > 2) return fail(MJH_EINVAL, "profiling level must be 0, 1 or 2");
  if (e->twin) { e->twin->prof_focus_name = e->prof_focus_name; (void)mjh_set_profiling(e->twin, on); }
  e->profiling = on;
  e->prof_calls = 0;
  if (e->prof_focus_name.empty()) e->prof_focus = e->p.trellis_quant && !e->progressive ? "trellis_ac" : "dct_quant";
  else e->prof_focus = e->prof_focus_name.c_str();
  return MJH_OK;
}

// which interval of the schedule profiling level 2 brackets: a name out of mjh_get_kernel_times (e.g. the largest entry of a
// level-1 pass over the same workload); NULL or "" = the built-in choice.  Takes effect with the next mjh_set_profiling.
extern "C" int mjh_set_profiling_focus(mjh_encoder *e, const char *name)
{
  if (!e) return fail(MJH_EINVAL, "null encoder");
  e->prof_focus_name = name ? name : "";
  return MJH_OK;
}

static int kernel_times_one(mjh_encoder *e);

extern "C" int mjh_get_kernel_times(mjh_encoder *e, const char *const **names, const float **ms, int *count)
{
  if (!e || !count) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  const int calls_e = e->prof_calls;
  const int rc = kernel_times_one(e);
  if (rc) return rc;
  if (e->twin && e->twin->prof_calls > 0) {     // two buffer sets took turns: the averages of both, weighted by their calls
    const int calls_t = e->twin->prof_calls;
    const int rt = kernel_times_one(e->twin);
    if (rt) return rt;
    if (calls_e == 0) { e->prof_cnames = e->twin->prof_cnames; e->prof_ms = e->twin->prof_ms; }
    else
      for (size_t i = 0; i < e->prof_cnames.size(); i++)
        for (size_t j = 0; j < e->twin->prof_cnames.size(); j++)
          if (!strcmp(e->prof_cnames[i], e->twin->prof_cnames[j]))
            e->prof_ms[i] = (e->prof_ms[i] * (float)calls_e + e->twin->prof_ms[j] * (float)calls_t) / (float)(calls_e + calls_t);
  }
  if (names) *names = e->prof_cnames.data();
  if (ms) *ms = e->prof_ms.data();
  *count = (int)e->prof_cnames.size();
  return MJH_OK;
}

static int kernel_times_one(mjh_encoder *e)
{
  const size_t n = e->prof_names.size();
  const int calls = e->prof_calls;
  e->prof_ms.assign(n, 0.f);
  e->prof_cnames.resize(n);
  for (size_t i = 0; i < n; i++) e->prof_cnames[i] = e->prof_names[i].c_str();
  for (int c = 0; c < calls; c++)
    for (size_t i = 0; i < n; i++) {
      const size_t k = (size_t)c * e->prof_per_call + i;
      if (k + 1 >= e->prof_events.size()) continue;
      float t = 0.f;
      HIPCHK(hipEventElapsedTime(&t, e->prof_events[k], e->prof_events[k + 1]));
      e->prof_ms[i] += t / (float)calls;
    }
  if (e->side_timed && e->profiling == 1 && calls > 0) {   // the DC trellis runs concurrently on the side stream: its own event pairs
    float sum = 0.f;
    for (int c = 0; c < calls && 2 * (size_t)c + 1 < e->side_events.size(); c++) {
      float t = 0.f;
      HIPCHK(hipEventElapsedTime(&t, e->side_events[2 * c], e->side_events[2 * c + 1]));
      sum += t;
    }
    static const char *kDc = "trellis_dc(side stream, overlaps trellis_ac)";
    e->prof_cnames.push_back(kDc);
    e->prof_ms.push_back(sum / (float)calls);
  }
  e->prof_calls = 0;   // the next encode call starts a new accumulation
  return MJH_OK;
}

extern "C" int mjh_component_geometry(const mjh_encoder *e, int c, int *wib, int *hib, int *pw, int *ph)
{
  if (!e || c < 0 || c >= e->C.ncomp) return fail(MJH_EINVAL, "bad component");
  if (wib) *wib = e->C.c[c].wib;
  if (hib) *hib = e->C.c[c].hib;
  if (pw) *pw = e->C.c[c].pw;
  if (ph) *ph = e->C.c[c].ph;
  return MJH_OK;
}

// The Huffman table one entry of the scan script was coded with in image `image` of the last batch (progressive encoders; a scan
// search codes every candidate, also the ones its file leaves out).  For the libjpeg drop-in, whose compress object keeps in its
// table slots what the LAST CODED scan put there (jpeg_gen_optimal_table writes into cinfo->dc / ac_huff_tbl_ptrs, jchuff.c:1092-1105).
extern "C" int mjh_get_scan_table(mjh_encoder *e, int image, int scan, int tblno, uint8_t bits[17], uint8_t vals[256])
{
  e = cur(e);
  if (!e || !bits || !vals || image < 0 || image >= e->last_n) return fail(MJH_EINVAL, "bad arguments");
  if (!e->progressive || e->arith || scan < 0 || scan >= e->nscans || tblno < 0 || tblno > 3) return fail(MJH_EINVAL, "no such scan table");
  const mjh_scan &sc = e->p.scan_info[scan];
  {   // the table's class inside the image (dc_class): the class of the first component that names it
    int cls = -1;
    for (int i = 0; i < e->p.num_components && cls < 0; i++) if (e->p.dc_tbl_no[i] == tblno) cls = dc_class(&e->p, i);
    if (sc.Ss == 0 && cls < 0) return fail(MJH_EINVAL, "no component uses DC table %d", tblno);
    tblno = cls < 0 ? 0 : cls;
  }
  if (sc.Ss == 0 && sc.Ah != 0) return fail(MJH_EINVAL, "a DC refinement scan has no table");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  MjhHuffTable t;
  HIPCHK(hipMemcpy(&t, e->d_tabs + (size_t)image * e->spi + SLOT_PROG + 2 * scan + (sc.Ss == 0 ? tblno : 0), sizeof(t), hipMemcpyDeviceToHost));
  memcpy(bits, t.bits, 17);
  memcpy(vals, t.huffval, 256);
  return MJH_OK;
}

extern "C" int mjh_read_tap(mjh_encoder *e, int what, int image, int comp, void *dst, size_t cap, size_t *size)
{
  e = cur(e);
  if (!e || !dst || image < 0 || image >= e->last_n) return fail(MJH_EINVAL, "bad arguments");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  const MjhConst &C = e->C;
  if (what == MJH_TAP_HUFF_BITS || what == MJH_TAP_HUFF_VALS) {
    std::vector<MjhHuffTable> t(4);
    const size_t need = what == MJH_TAP_HUFF_BITS ? 4 * 17 : 4 * 256;
    if (cap < need) return fail(MJH_ETOOSMALL, "need %zu bytes", need);
    HIPCHK(hipMemcpy(t.data(), e->d_tabs + (size_t)image * e->spi + SLOT_FINAL, 4 * sizeof(MjhHuffTable), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) {
      if (what == MJH_TAP_HUFF_BITS) memcpy((uint8_t *)dst + i * 17, t[i].bits, 17);
      else memcpy((uint8_t *)dst + i * 256, t[i].huffval, 256);
    }
    if (size) *size = need;
    return MJH_OK;
  }
  if (what == MJH_TAP_PROG_SCAN_US) {
    if (!e->progressive) return fail(MJH_EINVAL, "not a progressive encoder");
    const size_t need = sizeof(((MjhProgCtl *)0)->scan_us);
    if (cap < need) return fail(MJH_ETOOSMALL, "need %zu bytes", need);
    HIPCHK(hipMemcpy(dst, (const uint8_t *)e->d_prog_ctl + (size_t)image * sizeof(MjhProgCtl) + offsetof(MjhProgCtl, scan_us), need, hipMemcpyDeviceToHost));
    if (size) *size = need;
    return MJH_OK;
  }
  if (comp < 0 || comp >= C.ncomp) return fail(MJH_EINVAL, "bad component");
  const MjhComp &cc = C.c[comp];
  if (what == MJH_TAP_PLANE) {
    const size_t bps = C.precision == 12 ? 2 : 1;
    const size_t need = (size_t)cc.pw * cc.ph * bps;
    if (cap < need) return fail(MJH_ETOOSMALL, "need %zu bytes", need);
    HIPCHK(hipMemcpy(dst, e->d_planes + ((size_t)image * C.planes_per_image + cc.plane_off) * bps, need, hipMemcpyDeviceToHost));
    if (size) *size = need;
    return MJH_OK;
  }
  const int16_t *src = what == MJH_TAP_COEF_UQ ? e->d_uq : what == MJH_TAP_COEF_Q ? e->d_q : what == MJH_TAP_COEF_Q0 ? e->d_q0 : nullptr;
  if (!src) return fail(MJH_EINVAL, "tap %d not available (enable debug taps before encoding)", what);
  const size_t need = (size_t)64 * cc.nblk * 2;
  if (cap < need) return fail(MJH_ETOOSMALL, "need %zu bytes", need);
  if (what == MJH_TAP_COEF_Q && e->compact_last) {
    // the batch left compact records (plane i+1 = i-th non-zero of the block, d_nzmask = its positions): expand to [64][nblk]
    std::vector<int16_t> planes((size_t)64 * cc.nblk);
    std::vector<unsigned long long> mask(cc.nblk);
    HIPCHK(hipMemcpy2D(planes.data(), (size_t)cc.nblk * 2, src + (size_t)image * C.coefs_per_image + cc.coef_off, (size_t)cc.kstride * 2,
                       (size_t)cc.nblk * 2, 64, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mask.data(), e->d_nzmask + (size_t)image * C.total_real_blocks + cc.blk_off, (size_t)cc.nblk * 8, hipMemcpyDeviceToHost));
    int16_t *out = (int16_t *)dst;
    for (int b = 0; b < cc.nblk; b++) {
      out[b] = planes[b];   // DC plane
      unsigned long long m = mask[b];
      int i = 0;
      for (int k = 1; k < 64; k++) out[(size_t)k * cc.nblk + b] = ((m >> k) & 1ull) ? planes[(size_t)(++i) * cc.nblk + b] : (int16_t)0;
    }
    if (size) *size = need;
    return MJH_OK;
  }
  // strip the kstride padding: [64][nblk]
  HIPCHK(hipMemcpy2D(dst, (size_t)cc.nblk * 2, src + (size_t)image * C.coefs_per_image + cc.coef_off, (size_t)cc.kstride * 2,
                     (size_t)cc.nblk * 2, 64, hipMemcpyDeviceToHost));
  if (size) *size = need;
  return MJH_OK;
}
