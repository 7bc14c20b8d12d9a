/*
 * mozjpeg_hip.h -- C ABI of libmozjpeg_hip.so, the MI355X-native JPEG encode hot path.
 *
 * Two layers are exported (plain pointers and sizes only; no C++/torch types):
 *
 *  (1) The batch encoder (mjh_*): the native interface of the GPU pipeline.  One encoder =
 *      one parameter set + one GPU + device-resident working buffers for up to max_batch
 *      equally sized images.  It replaces, for a whole image at a time, the reference's
 *      pixel->bytes path:
 *        jpeg_start_compress      jcapistd.c:44   (parameter capture  -> mjh_encoder_create)
 *        jpeg_write_scanlines     jcapistd.c:90   (pixel hand-over    -> mjh_encode_*)
 *        jpeg_finish_compress     jcapimin.c:176  (passes 1..N + bytes-> mjh_encode_*, mjh_get_jpeg)
 *      and the parameter helpers a caller needs to fill mjh_params:
 *        jpeg_set_defaults        jcparam.c:386   -> mjh_params_defaults
 *        jpeg_set_quality         jcparam.c:360   -> mjh_params_set_quality
 *        jpeg_simple_progression  jcparam.c:859   -> (later rounds)
 *
 *  (2) The libjpeg drop-in symbols (declared in mozjpeg_hip_jpeglib.h, built into
 *      libmozjpeg_hip_jpeg62.so): jpeg_start_compress / jpeg_write_scanlines /
 *      jpeg_finish_compress with the reference's exact signatures (jpeglib.h:1065-1076),
 *      implemented on top of layer (1).
 *
 * Error convention: functions return 0 on success or a negative MJH_E* code;
 * mjh_last_error() returns a thread-local message.  Nothing here falls back to a CPU
 * implementation: an unsupported configuration is an error (MJH_EUNSUPPORTED).
 */
#ifndef MOZJPEG_HIP_H
#define MOZJPEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJH_MAX_COMPS 4
#define MJH_MAX_SCANS 64

#define MJH_OK            0
#define MJH_EINVAL       -1   /* bad argument */
#define MJH_EUNSUPPORTED -2   /* configuration outside the GPU hot path (no CPU fallback) */
#define MJH_EHIP         -3   /* HIP runtime error (message has the hipError string) */
#define MJH_ENOMEM       -4
#define MJH_ETOOSMALL    -5   /* output buffer too small */

/* profile values = the reference's JINT_COMPRESS_PROFILE GUIDs (jpeglib.h:353-356) */
#define MJH_PROFILE_MAX_COMPRESSION 0x5D083AAD
#define MJH_PROFILE_FASTEST         0x2AEA5CB4

/* one entry of a progressive scan script (jpeg_scan_info, jpeglib.h:196-201) */
typedef struct {
  int comps_in_scan;
  int component_index[MJH_MAX_COMPS];
  int Ss, Se, Ah, Al;
} mjh_scan;

/* Everything the hot path reads from a jpeg_compress_struct at jpeg_start_compress
 * (SURVEY 8b "Inputs read from cinfo"; field names follow jpeglib.h:399-488 and the
 * extension parameters of jpeglib.h:321-356). */
typedef struct {
  int image_width, image_height;
  int input_components;            /* 3 = interleaved RGB, 1 = grayscale */
  int num_components;              /* 3 = YCbCr output, 1 = grayscale output */
  int h_samp_factor[MJH_MAX_COMPS], v_samp_factor[MJH_MAX_COMPS];
  int quant_tbl_no[MJH_MAX_COMPS], dc_tbl_no[MJH_MAX_COMPS], ac_tbl_no[MJH_MAX_COMPS];
  int component_id[MJH_MAX_COMPS];
  uint16_t quantval[4][64];        /* natural order, = quant_tbl_ptrs[i]->quantval */
  int compress_profile;            /* MJH_PROFILE_* : marker layout (jcmarker.c:189,293) */
  int optimize_coding;
  int trellis_quant, trellis_quant_dc, overshoot_deringing;
  float lambda_log_scale1, lambda_log_scale2;
  unsigned restart_interval;
  int restart_in_rows;
  int num_scans;                   /* 0 = one sequential scan (baseline) */
  mjh_scan scan_info[MJH_MAX_SCANS];
  int optimize_scans;
  int write_JFIF_header;
  /* layout of one input pixel, the extended RGB colour spaces of jpeglib.h:262-290 / jccolext.c
   * (TurboJPEG's TJPF_*): bytes per pixel (0 = input_components) and the byte offsets of R, G, B
   * (all 0 = R,G,B at 0,1,2).  input_components stays 3 for every RGB-family layout. */
  int input_pixel_size;
  int rgb_offset[3];
  /* 0 or 8: 8-bit samples (one byte each); 12: 12-bit samples stored as uint16 (jpeg12_write_scanlines,
   * J12SAMPLE).  row_pitch / image_stride stay in BYTES.  Trellis quantization has no 12-bit reference
   * behaviour (jccoefct.c:132-138) and is rejected. */
  int data_precision;
  /* JINT_TRELLIS_NUM_LOOPS (jpeglib.h:345, default 1, 0 is read as 1): (statistics, trellis) pass pairs per component,
   * each trellis pass restarting from the unquantized coefficients with the tables of the previous result
   * (jcmaster.c:451-466, :1128-1138) */
  int trellis_num_loops;
  /* cinfo->smoothing_factor (jpeglib.h:459, cjpeg -smooth N), 0..100: input smoothing inside the full-size and the
   * 2x2 downsamplers (jcsample.c:306-455); the other sampling ratios have no smoothing variant in the reference */
  int smoothing_factor;
  /* colour transform between input pixels and JPEG components: MJH_COLOR_YCC (0) = RGB -> YCbCr / gray (rgb_ycc_convert,
   * rgb_gray_convert, grayscale_convert of jccolor.c), MJH_COLOR_NONE = the three input samples become the three
   * components unconverted (null_convert jccolor.c:479, JCS_RGB output = `cjpeg -rgb`; an Adobe APP14 marker with
   * transform 0 replaces the JFIF APP0, component ids are whatever component_id[] says, 'R' 'G' 'B' for cjpeg),
   * MJH_COLOR_YCC_IN = input pixels that are YCbCr already (what an application sets with in_color_space = JCS_YCbCr) */
  int color_transform;
  /* JINT_DC_SCAN_OPT_MODE (jpeglib.h:349, cjpeg -dc-scan-opt N; library default 0, jcparam.c:495): 0 = one DC scan for
   * all components, 1 = one DC scan per component, 2 = luma alone, then chroma interleaved or separate -- with the scan
   * search whichever is smaller (jcmaster.c:836-838, :904-913).  mjh_params_simple_progression /
   * mjh_params_search_progression read it when they build the script (jcparam.c:791-794, :934-947); the encoder
   * reads it for the scan search's final choice of the chroma DC scans. */
  int dc_scan_opt_mode;
  /* JFLOAT_TRELLIS_DELTA_DC_WEIGHT (jpeglib.h:337, cjpeg -trellis-dc-ver-weight W; default 0): weight of the
   * vertical-gradient error against the block above (same iMCU row) in the DC trellis (jcdctmgr.c:1069-1084) */
  float trellis_delta_dc_weight;
  /* JBOOLEAN_USE_SCANS_IN_TRELLIS + JINT_TRELLIS_FREQ_SPLIT (jpeglib.h:326,344; split 0 is read as the default 8):
   * two (statistics, trellis) pass pairs per component, AC bands 1..split and split+1..63 (jcmaster.c:451-460) */
  int use_scans_in_trellis, trellis_freq_split;
  /* JBOOLEAN_TRELLIS_EOB_OPT (jpeglib.h:325): end-of-band runs over all-zero blocks chosen by a second dynamic
   * programme along each block row (jcdctmgr.c:1224-1297) */
  int trellis_eob_opt;
  /* JBOOLEAN_TRELLIS_Q_OPT (jpeglib.h:327): quantization tables re-estimated from the trellis result
   * (sums jcdctmgr.c:1299-1306, update jcmaster.c:1014-1030) */
  int trellis_q_opt;
  /* cinfo->arith_code (jpeglib.h:420, cjpeg -arithmetic): arithmetic entropy coding (jcarith.c) instead of Huffman -- SOF9 /
   * SOF10 frames, DAC markers, no Huffman tables; with trellis_quant the coder's own rate model (quantize_trellis_arith
   * jcdctmgr.c:1334-1667).  An adaptive coder is one dependent chain per scan: this mode is there for completeness, not for
   * throughput.  With trellis_q_opt the reference's pass arithmetic decides whether component 0's table is ever re-estimated
   * (jcmaster.c:687-698, :1016-1030, :1135-1138): reproduced for any number of loops. */
  int arith_code;
  /* cinfo->arith_dc_L / arith_dc_U / arith_ac_K (jpeglib.h:447-449) of conditioning tables 0 and 1: the DC category thresholds
   * (jcarith.c:442-445, :757-760) and the AC position Kx that switches the magnitude bins (:533, :802), written into the DAC
   * marker (emit_dac jcmarker.c:404-448) and read by the coder's trellis (jget_arith_rates jcarith.c:949-951).
   * mjh_params_defaults sets the library defaults 0 / 1 / 5 (jcparam.c:417-419); a table whose three values are all 0 (a
   * zeroed struct of an older caller; Kx = 0 is not a valid conditioning) is read as those defaults.  0 <= L <= U <= 15, 1 <= K <= 63. */
  int arith_dc_L[2], arith_dc_U[2], arith_ac_K[2];
  /* cinfo->Ah / cinfo->Al as the compress object holds them when the image STARTS.  The trellis passes of a progressive image
   * gather their statistics through jcphuff.c with whatever these fields hold -- select_scan_parameters sets Ss / Se for those
   * passes and nothing else (jcmaster.c:451-466): 0 / 0 in a new object, the last coded scan's values when the
   * object has compressed an image before (refinement statistics after a script that ends with a refinement scan, first-pass
   * statistics at the best chroma Al after a scan search).  Read only with num_scans > 0 and trellis_quant. */
  int trellis_stats_Ah, trellis_stats_Al;
  /* cinfo->dc_huff_tbl_ptrs[t] / ac_huff_tbl_ptrs[t] as the object holds them when the image starts, for the two uses the
   * reference has for them: (1) optimize_coding off: the tables the scans are coded with and the DHT markers carry
   * (start_pass_huff -> jpeg_make_c_derived_tbl jchuff.c:190-196, :231-318); (2) a progressive image with the DC trellis: the
   * rates of the DC candidates, because no pass in front of the trellis makes a DC table (compress_trellis_pass
   * jccoefct.c:388-389).  Bit 2 t + is_ac of huff_tables_given says slot t's DC / AC table is given in
   * huff_bits[2 t + is_ac][0..16] (counts per code length, [0] unused) and huff_vals[2 t + is_ac][] (symbols in code order);
   * a slot not given holds the Annex K.3 table for t = 0, 1 (what jpeg_set_defaults installs) and nothing for t = 2, 3.
   * A table must be a legal code (the checks of jpeg_make_c_derived_tbl: JERR_BAD_HUFF_TABLE), DC symbols 0..15; a symbol the
   * image needs and the table lacks is coded with no bits, as jchuff.c does. */
  int huff_tables_given;
  uint8_t huff_bits[8][17];
  uint8_t huff_vals[8][256];
  /* cinfo->dct_method (jpeglib.h:456): 0 = JDCT_ISLOW (jfdctint.c), 1 = JDCT_IFAST (jfdctfst.c: AA&N with 8-bit constants,
   * its scale factors folded into the divisors, jcdctmgr.c:291-345) -- what the legacy TurboJPEG calls select below quality 96
   * (turbojpeg.c:522-527) and `cjpeg -dct fast`.  8- and 12-bit samples. */
  int dct_method;
} mjh_params;

#define MJH_COLOR_YCC  0
#define MJH_COLOR_NONE 1
#define MJH_COLOR_YCC_IN 2   /* the input samples ARE Y, Cb, Cr (in_color_space = JCS_YCbCr, jpeg_color_space = JCS_YCbCr: null_convert
                              * jccolor.c:479 via jinit_color_converter :687-692): unconverted like MJH_COLOR_NONE, but the file is an
                              * ordinary YCbCr one -- JFIF APP0, no Adobe marker, YCbCr's progressive scripts; with num_components = 1 the Y
                              * samples become a grayscale file (grayscale_convert :448-466) */

typedef struct mjh_encoder mjh_encoder;

/* ---- parameter helpers (host only) ---------------------------------------------------- */
/* jpeg_set_defaults (jcparam.c:386-519) for an RGB->YCbCr (or ->gray) encode of the given size
 * and profile, plus the sampling factors of component 0 (others 1x1). */
int mjh_params_defaults(mjh_params *p, int width, int height, int input_components,
                        int gray_output, int compress_profile, int hsamp, int vsamp);
/* jpeg_set_quality (jcparam.c:360-380) with the base table selected by base_quant_tbl_idx
 * (-1 = the profile's default: 3 for max compression, 0 for fastest; jcparam.c:509-510; 0..8 = cjpeg -quant-table N). */
int mjh_params_set_quality(mjh_params *p, int quality, int force_baseline, int base_quant_tbl_idx);

/* The Annex K.3 Huffman tables jpeg_set_defaults installs (std_huff_tables jstdhuff.c:31-131): bits[0..16] and the
 * symbol list of DC / AC table 0 (luminance) or 1 (chrominance).  Constants owned by the library. */
int mjh_std_huffman_table(int is_ac, int tblno, const uint8_t **bits, const uint8_t **vals, int *nvals);

/* jpeg_simple_progression (jcparam.c:859-1004): the profile's fixed script (9 scans for YCbCr in
 * the max-compression profile, 10 in the fastest profile); clears optimize_scans. */
int mjh_params_simple_progression(mjh_params *p);
/* jpeg_search_progression (jcparam.c:733-852): the 64 (YCbCr) / 23 (gray) candidate scans of the
 * scan search; sets optimize_scans (what jpeg_set_defaults selects in the max-compression profile). */
int mjh_params_search_progression(mjh_params *p);

/* ---- encoder lifetime ----------------------------------------------------------------- */
/* Creates the device-resident state for up to max_batch images of p's geometry on HIP device
 * `device`.  Returns MJH_EUNSUPPORTED for configurations the GPU path does not cover. */
int mjh_encoder_create(const mjh_params *p, int max_batch, int device, mjh_encoder **out);
/* number of visible HIP devices (0 if there is none: every mjh_encoder_create then fails with MJH_EHIP) */
int mjh_device_count(void);
/* Host-side placement of a device (one compress object per thread, libjpeg.txt:2198-2200): the NUMA node the
 * device's PCIe root belongs to (-1: unknown, one-node host, or MJH_NUMA=0), and "run the calling thread on that node's
 * CPUs" (returns the node, or -1 when nothing was changed).  The encoder pins its own staging / result buffers on that
 * node; a client that fills mjh_host_staging buffers or owns pinned frames should do so from a bound thread.
 * mjh_device_placement writes a one-line description (for logs / bench lines) and returns its length. */
int mjh_device_numa_node(int device);
int mjh_bind_thread_to_device(int device);
int mjh_device_placement(int device, char *buf, size_t n);
void mjh_encoder_destroy(mjh_encoder *e);
/* the parameters the encoder was created with (owned by the encoder) */
const mjh_params *mjh_encoder_params(const mjh_encoder *e);

/* ---- one process, several GPUs --------------------------------------------- */
/* A pool owns one encoder per device (devices == NULL: every visible device) and drives each from its own host thread.
 * mjh_pool_encode_host deals the n images of a batch round-robin (image i -> device i mod N, nothing is exchanged
 * between devices), pipelines every device's share through mjh_encode_host / mjh_collect in steps of
 * max_batch_per_device, and returns the finished files in the caller's image order: (*jpegs)[i] is (*sizes)[i] bytes,
 * in host memory owned by the pool, valid until the next call.  Pixels may be pageable or pinned. */
typedef struct mjh_pool mjh_pool;
int mjh_pool_create(const mjh_params *p, int max_batch_per_device, const int *devices, int ndevices, mjh_pool **out);
void mjh_pool_destroy(mjh_pool *pool);
int mjh_pool_device_count(const mjh_pool *pool);
int mjh_pool_encode_host(mjh_pool *pool, const void *pixels, size_t row_pitch, size_t image_stride, int n,
                         const uint8_t *const **jpegs, const size_t **sizes);
const char *mjh_pool_last_error(const mjh_pool *pool);

/* ---- encode ---------------------------------------------------------------------------- */
/* Encode n images that are ALREADY in device memory (interleaved samples, row_pitch bytes per
 * row, image_stride bytes between images).  `stream` is a hipStream_t passed as void*
 * (NULL = the encoder's own stream, which is NOT ordered behind the null stream; (void *)1 =
 * the encoder's own stream, made to wait for everything queued on the null stream so far -- for
 * pixels produced on the legacy default stream).  With MJH_SPLIT=2 in the environment when the
 * encoder is created (sequential mode) a batch larger than half of max_batch runs as two image
 * ranges concurrently on streams of the encoder, forked from and joined into `stream`: about 5 %
 * more throughput on large batches.  Asynchronous: results are valid after
 * mjh_encoder_sync() or any later synchronising call. */
int mjh_encode_device(mjh_encoder *e, const void *d_pixels, size_t row_pitch, size_t image_stride,
                      int n, void *stream);
/* Batches in flight on the encoder's own stream (stream == NULL above): 2 (the default; MJH_INFLIGHT in the environment) -- consecutive
 * mjh_encode_device calls alternate between two complete sets of working buffers and streams inside the encoder, so that the
 * front end of one batch (colour conversion, FDCT) runs next to the tail of the other (final statistics, bit lengths, prefix sums,
 * bit writing, byte stuffing), while the VALU-bound AC trellis of either keeps the chip to itself; the results of
 * call k stay valid until call k + 2, and every accessor (mjh_get_jpeg, mjh_get_output_device, taps, ...) refers to the most recent
 * call.  1 -- one batch at a time on one buffer set (half the device memory).  A caller's stream, debug taps and the memory checker
 * always run one batch at a time.  The files are the same either way. */
int mjh_set_inflight(mjh_encoder *e, int batches);
/* Same with host pixels -- the whole-image form of jpeg_write_scanlines (jcapistd.c:90) + jpeg_finish_compress
 * (jcapimin.c:176) for a batch.  ASYNCHRONOUS and double-buffered: the call queues the host->device copy, the kernel
 * schedule and the hand-over of the finished files to pinned host memory, and returns; the copy of batch k+1 overlaps
 * the kernels of batch k and the hand-over of batch k-1.  Pixels in pinned memory (mjh_host_alloc,
 * mjh_host_register, or the encoder's own buffer from mjh_host_staging) are read by the DMA engine where they lie and
 * must stay untouched until mjh_wait_input or mjh_collect returns; pixels in ordinary (pageable) memory are first
 * copied to a pinned staging buffer by a few worker threads (MJH_HOST_THREADS, default min(8, cores/2)) and may be
 * reused as soon as the call returns.  Results: mjh_collect (zero-copy) or mjh_get_jpeg[_size]. */
int mjh_encode_host(mjh_encoder *e, const void *pixels, size_t row_pitch, size_t image_stride, int n);
/* One finished file of a batch inside the pinned result arena. */
typedef struct { uint64_t offset, size; } mjh_result;
/* Waits for a batch queued by mjh_encode_host -- age 0: the most recent call, age 1: the call before it (so that a
 * loop can queue batch k+1 before it picks up batch k) -- and returns its files where the device put them: file i is
 * results[i].size bytes at (const uint8_t *)base + results[i].offset, in pinned host memory owned by the encoder.
 * The memory of a batch is reused by the SECOND mjh_encode_host call after the one that queued it. */
int mjh_collect(mjh_encoder *e, int age, const void **base, const mjh_result **results, int *count);
/* Blocks until the pixels handed to the most recent mjh_encode_host call have been read (pinned sources only matter). */
int mjh_wait_input(mjh_encoder *e);
/* The encoder's own pinned staging buffer for the NEXT mjh_encode_host call (max_batch images, tightly packed rows):
 * a caller that produces pixels row by row (the libjpeg drop-in's jpeg_write_scanlines) writes them here and passes
 * the pointer to mjh_encode_host, which then copies nothing on the host. */
int mjh_host_staging(mjh_encoder *e, void **buffer, size_t *bytes);
/* The first `bytes` of that staging buffer (packed images, whole rows) are final: their host->device copy is queued now and
 * runs while the caller produces the rest -- what jpeg_write_scanlines does with a client's rows (jcapistd.c:90-135 hands
 * them on strip by strip as well).  The mjh_encode_host call for the buffer copies only the remainder. */
int mjh_stage_commit(mjh_encoder *e, size_t bytes);
/* One batch out of the images n OTHER encoders (same parameters, same device) have staged that way, one image each:
 * image i of the batch = what members[i] holds.  For callers that get their images one at a time from several threads
 * (the libjpeg shim coalesces concurrent jpeg_finish_compress calls): the device then runs ONE schedule for the lot
 * instead of n single-image ones.  Results through mjh_collect / mjh_get_jpeg of `e`; nobody else may use the member
 * encoders during the call. */
int mjh_encode_gather(mjh_encoder *e, mjh_encoder *const *members, int n);
/* Pinned host memory for zero-copy hand-over (hipHostMalloc / hipHostRegister underneath; no HIP types in the ABI). */
void *mjh_host_alloc(size_t bytes);
void mjh_host_free(void *p);
int mjh_host_register(void *p, size_t bytes);
int mjh_host_unregister(void *p);
/* Component planes instead of pixels: jpeg_write_raw_data (jcapistd.c:145) for whole images, the entry
 * tj3CompressFromYUVPlanes8 (turbojpeg.c:1222) uses.  Colour conversion and downsampling are skipped; the
 * encoder must have been created with the sampling factors the planes were made for (input_components and
 * the pixel layout fields are ignored by these calls).  Plane c of image i starts at planes[c] + i *
 * image_stride[c] (image_stride may be NULL for n == 1), its rows are row_pitch[c] BYTES apart and
 * plane_width[c] x plane_height[c] samples of it are valid.  The encoder reads width_in_blocks*8 x
 * height_in_blocks*8 samples per component (mjh_component_geometry); where the plane is smaller its last
 * sample / row is replicated, which is exactly what tj3CompressFromYUVPlanes8 does before it calls
 * jpeg_write_raw_data (turbojpeg.c:1295-1316), so TurboJPEG-layout planes (tj3YUVPlaneWidth/Height) can be
 * passed as they are. */
int mjh_encode_planes_device(mjh_encoder *e, const void *const d_planes[MJH_MAX_COMPS], const size_t row_pitch[MJH_MAX_COMPS],
                             const size_t image_stride[MJH_MAX_COMPS], const int plane_width[MJH_MAX_COMPS],
                             const int plane_height[MJH_MAX_COMPS], int n, void *stream);
int mjh_encode_planes_host(mjh_encoder *e, const void *const planes[MJH_MAX_COMPS], const size_t row_pitch[MJH_MAX_COMPS],
                           const size_t image_stride[MJH_MAX_COMPS], const int plane_width[MJH_MAX_COMPS],
                           const int plane_height[MJH_MAX_COMPS], int n);
/* Quantized DCT coefficients instead of pixels: jpeg_write_coefficients (jctrans.c:44) for whole images -- the
 * lossless re-encode jpegtran performs ("jpegrescan": optimal tables, progressive scan search).  Only the
 * entropy-coding passes run; quantval[] of the parameters is written to the DQT marker unchanged.  The encoder
 * must have been created with trellis_quant = 0 (there is no unquantized data; jpeg_copy_critical_parameters
 * switches it off as well, jctrans.c:102) -- otherwise MJH_EINVAL.  coefs[c] of image i starts at coefs[c] +
 * i * image_stride[c] bytes (image_stride may be NULL for n == 1) and holds height_in_blocks rows of
 * blocks_per_row[c] (>= width_in_blocks) blocks of 64 int16 in natural order (JBLOCKARRAY layout, 4-byte
 * aligned); dummy blocks are generated, not read (compress_output jctrans.c:322-373). */
int mjh_encode_coefficients_device(mjh_encoder *e, const void *const d_coefs[MJH_MAX_COMPS], const size_t blocks_per_row[MJH_MAX_COMPS],
                                   const size_t image_stride[MJH_MAX_COMPS], int n, void *stream);
int mjh_encode_coefficients_host(mjh_encoder *e, const void *const coefs[MJH_MAX_COMPS], const size_t blocks_per_row[MJH_MAX_COMPS],
                                 const size_t image_stride[MJH_MAX_COMPS], int n);
int mjh_encoder_sync(mjh_encoder *e);

/* Size in bytes of JPEG i of the last batch (synchronises). */
int mjh_get_jpeg_size(mjh_encoder *e, int i, size_t *size);
/* Copy JPEG i of the last batch to host memory (synchronises). */
int mjh_get_jpeg(mjh_encoder *e, int i, void *dst, size_t cap, size_t *size);
/* Device-side view of the outputs of the last batch: file i starts at base + i*stride and is
 * sizes[i] bytes long (sizes is a device pointer to uint32). */
int mjh_get_output_device(mjh_encoder *e, void **d_base, size_t *stride, void **d_sizes);

/* The Huffman table entry `scan` of the scan script was coded with in image `image` of the last batch: bits[0..16] (counts per
 * code length) and the symbols in code order.  Progressive encoders only; tblno selects the DC table of a DC scan (the number its
 * components carry in dc_tbl_no) and is ignored for AC scans.  A scan search codes all its candidates, also those the file
 * leaves out: the libjpeg drop-in keeps the object's table slots as the reference's last coded scans leave them
 * (jpeg_gen_optimal_table writes into cinfo->dc_huff_tbl_ptrs / ac_huff_tbl_ptrs, jchuff.c:1092-1105).  Synchronises. */
int mjh_get_scan_table(mjh_encoder *e, int image, int scan, int tblno, uint8_t bits[17], uint8_t vals[256]);

/* ---- introspection for parity tests and profiling ------------------------------------- */
enum {
  MJH_TAP_PLANE = 1,     /* uint8  [ph][pw] downsampled samples of one component            */
  MJH_TAP_COEF_UQ = 2,   /* int16  [64 zig-zag][nblk] raw DCT (x8) coefficients             */
  MJH_TAP_COEF_Q = 3,    /* int16  [64 zig-zag][nblk] quantized (after trellis if enabled)  */
  MJH_TAP_COEF_Q0 = 4,   /* int16  quantized before trellis (kept only when debug taps on)  */
  MJH_TAP_HUFF_BITS = 5, /* uint8  [4 slots: DC0,AC0,DC1,AC1][17] final tables               */
  MJH_TAP_HUFF_VALS = 6, /* uint8  [4][256]                                                  */
  MJH_TAP_PROG_SCAN_US = 7 /* uint32 [2][64] progressive: microseconds the statistics [0] / encode [1]
                              workgroup of each scan-script entry ran (0 = not run)              */
};
int mjh_set_debug_taps(mjh_encoder *e, int on);
int mjh_read_tap(mjh_encoder *e, int what, int image, int component, void *dst, size_t cap, size_t *size);
/* geometry of component c: blocks across/down (real blocks) and plane size */
int mjh_component_geometry(const mjh_encoder *e, int c, int *width_in_blocks, int *height_in_blocks,
                           int *plane_width, int *plane_height);

/* Per-kernel HIP-event timing.  level 0 = off; 1 = every kernel of the schedule (the events serialise
 * back-to-back launches, so whole-step throughput drops by some percent); 2 = only the dominant kernel
 * (the AC trellis when trellis quantization is on, else the DCT/quantize kernel): two events per encode
 * call.  Times accumulate over the mjh_encode_* calls (at most 256) since the level was set or since the
 * last read; mjh_get_kernel_times synchronises, returns the AVERAGE milliseconds per call and starts a new
 * accumulation (names/ms arrays are owned by the encoder; *count entries). */
int mjh_set_profiling(mjh_encoder *e, int level);
/* Which interval level 2 brackets: a name returned by mjh_get_kernel_times (normally the largest entry of a level-1
 * pass over the same workload, so that "dominant" is measured, not assumed); NULL or "" = the built-in choice.  Takes
 * effect with the next mjh_set_profiling call.  When a batch runs as concurrent image ranges (mjh_encode_device), a
 * kernel's time per call is the sum of its launches over the ranges. */
int mjh_set_profiling_focus(mjh_encoder *e, const char *name);
int mjh_get_kernel_times(mjh_encoder *e, const char *const **names, const float **ms, int *count);

/* Memory checking (debugging aid, environment MJH_GUARD read once per process):
 * 0 = off (device buffers are plain hipMalloc blocks of exactly the size needed), 1 = canaries around every device
 * buffer, 2 / 3 = every device buffer (and a private copy of the caller's device input) ends / starts at an unmapped
 * page, so that a kernel that strays past it faults on the spot.  In the modes 1-3 the canaries are compared whenever
 * a batch is waited for (the call fails with MJH_EHIP and names the buffer); mjh_debug_guard_check does it on demand. */
int mjh_debug_guard_mode(void);
int mjh_debug_guard_check(void);
/* the checker's own test (tools/guard_probe.py): touches one byte at `offset` relative to the end of a 1000-byte buffer */
int mjh_debug_guard_selftest(long offset, int write);

const char *mjh_last_error(void);
const char *mjh_version(void);
/* sizeof(mjh_params) of THIS library.  mjh_params has grown at its end between versions (0.3: arith_code); a caller built
 * against an older header would pass a shorter struct.  Bindings compare their own sizeof with this before the first
 * mjh_encoder_create (the Python binding and both shims do) and fill the struct through mjh_params_defaults, which zeroes it. */
size_t mjh_params_size(void);

#ifdef __cplusplus
}
#endif
#endif /* MOZJPEG_HIP_H */
