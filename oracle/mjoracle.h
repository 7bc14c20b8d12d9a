/*
 * mjoracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the mozjpeg encode hot path,
 * written from the reference's behaviour as a whole-image, planar encoder.  It exists so the
 * HIP path can be checked stage by stage (planes, raw DCT, quantized, trellised
 * coefficients, histograms, Huffman tables, final bytes) and byte for byte.
 *
 * PARITY PIN: this restatement is itself pinned against (i) the reference's own golden
 * MD5_JPEG_420_ISLOW = 9a68f56bc76e466aa7e52f415d0f4a5f (CMakeLists.txt:1391) and the other
 * cjpeg -revert bittest constants usable on this path, and (ii) the REAL reference compiled
 * from /root/reference into oracle/_ref (oracle/Makefile), byte for byte, on the fixture set
 * under tests/golden/ (tests/test_oracle_goldens.py, tests/golden/make_goldens.py).
 * The reference has no test that pins trellis / deringing / scan search; for
 * those modes the compiled reference is the only authority.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * The product (mozjpeg_amd/) never links, imports or executes anything in oracle/.
 */
#ifndef MJORACLE_H
#define MJORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJO_MAX_COMPS 4
#define MJO_MAX_SCANS 64

typedef struct {
  int comps_in_scan;
  int component_index[MJO_MAX_COMPS];
  int Ss, Se, Ah, Al;
} mjo_scan;

/* Everything jpeg_start_compress would read from cinfo (SURVEY 8b "Inputs read from cinfo"). */
typedef struct {
  int width, height;
  int input_components;           /* 3: interleaved RGB, 1: gray */
  int num_components;             /* 3: YCbCr, 1: gray (RGB->gray if input is RGB) */
  int h_samp[MJO_MAX_COMPS], v_samp[MJO_MAX_COMPS];
  int quant_tbl_no[MJO_MAX_COMPS], dc_tbl_no[MJO_MAX_COMPS], ac_tbl_no[MJO_MAX_COMPS];
  int component_id[MJO_MAX_COMPS];
  uint16_t qtbl[4][64];           /* natural (row-major) order, like JQUANT_TBL.quantval */
  int fastest_profile;            /* 1: JCP_FASTEST marker layout (one DQT/DHT marker per table) */
  int optimize_coding;
  int trellis_quant, trellis_quant_dc, overshoot_deringing;
  float lambda_log_scale1, lambda_log_scale2;
  int restart_interval, restart_in_rows;
  int num_scans;                  /* 0: single sequential scan */
  mjo_scan scans[MJO_MAX_SCANS];
  int optimize_scans;
  int write_jfif;
  int data_precision;             /* 0 or 8: 8-bit samples (uint8); 12: 12-bit samples (uint16), no trellis */
  int trellis_num_loops;          /* JINT_TRELLIS_NUM_LOOPS (jcparam.c:515 default 1; 0 is read as 1): trellis passes per component */
  int smoothing_factor;           /* cinfo->smoothing_factor 0..100 (cjpeg -smooth N): input smoothing in the downsampler, jcsample.c:306-455 */
  int trellis_q_opt;              /* JBOOLEAN_TRELLIS_Q_OPT: quantization tables re-estimated from the trellis result (jcmaster.c:1014-1030).
                                   *  */
  /* the remaining trellis options of SURVEY 8f row 4 */
  int trellis_eob_opt;            /* JBOOLEAN_TRELLIS_EOB_OPT: EOB runs over all-zero blocks optimised along a block row, jcdctmgr.c:1224-1297 */
  int use_scans_in_trellis;       /* JBOOLEAN_USE_SCANS_IN_TRELLIS: two trellis passes per component, bands 1..split / split+1..63, jcmaster.c:453-460 */
  int trellis_freq_split;         /* JINT_TRELLIS_FREQ_SPLIT (0 is read as the default 8, jcparam.c:512) */
  int rgb_output;                 /* jpeg_color_space JCS_RGB (cjpeg -rgb): null_convert jccolor.c:479, Adobe APP14 instead of JFIF APP0,
                                   * all-purpose progressive script; set it through mjo_set_rgb_output */
  float trellis_delta_dc_weight;  /* JFLOAT_TRELLIS_DELTA_DC_WEIGHT (cjpeg -trellis-dc-ver-weight): the DC trellis mixes the vertical-gradient
                                   * error against the block above of the same iMCU row into the candidate distortion, jcdctmgr.c:1069-1084 */
  int dc_scan_opt_mode;           /* JINT_DC_SCAN_OPT_MODE (cjpeg -dc-scan-opt N): 0 one DC scan for all components, 1 one per component,
                                   * 2 luma alone + chroma by the scan search's choice (jcparam.c:791,887-940, jcmaster.c:836-838,905-913);
                                   * set it through mjo_set_dc_scan_opt_mode (it rebuilds the script) */
  int arith_code;                 /* cinfo->arith_code (cjpeg -arithmetic): QM-coder instead of Huffman, SOF9 / SOF10, DAC markers;
                                   * with trellis_quant the rate model of quantize_trellis_arith */
  int arith_dc_L[4], arith_dc_U[4], arith_ac_K[4];   /* cinfo->arith_dc_L / arith_dc_U / arith_ac_K of conditioning tables 0 / 1 (jpeglib.h:447-449;
                                   * defaults 0 / 1 / 5, jcparam.c:417-419): jcarith.c:442-445,533,757-760,802,949-951, jcmarker.c:440-444 */
  int ycc_input;                  /* in_color_space = JCS_YCbCr with jpeg_color_space = JCS_YCbCr: the three input samples are Y, Cb, Cr already
                                   * (jinit_color_converter jccolor.c:687-692 -> null_convert :479); everything else is an ordinary YCbCr file */
  int dct_method;                 /* cinfo->dct_method: 0 JDCT_ISLOW, 1 JDCT_IFAST (cjpeg -dct fast; jfdctfst.c, divisors jcdctmgr.c:291-345,
                                   * the trellis' copy of the raw coefficients rescaled :731-750) */
} mjo_params;
/* jpeg_set_colorspace(cinfo, JCS_RGB) (jcparam.c:611-619): three 1x1 components 'R' 'G' 'B', tables 0, no JFIF marker */
void mjo_set_rgb_output(mjo_params *p);
/* jpeg_c_set_int_param(JINT_DC_SCAN_OPT_MODE) followed by the script builder the parameters already selected
 * (jpeg_simple_progression or jpeg_search_progression; nothing to rebuild for a sequential file) */
void mjo_set_dc_scan_opt_mode(mjo_params *p, int mode);

/* jpeg_set_defaults + jpeg_set_quality + colorspace defaults, as cjpeg would leave them:
 * profile_fastest=0 is the max-compression profile (jcparam.c:386-519).
 * subsampling h x v applies to component 0 (the others are 1x1). */
void mjo_default_params(mjo_params *p, int width, int height, int input_components,
                        int gray_output, int quality, int force_baseline, int profile_fastest,
                        int hsamp, int vsamp, int base_quant_tbl_idx);
/* the two scan scripts of jcparam.c */
void mjo_simple_progression(mjo_params *p);   /* jpeg_simple_progression, optimize_scans off */
void mjo_search_progression(mjo_params *p);   /* jpeg_search_progression (64 scans, YCbCr) */

/* geometry helpers (jcmaster.c:237-259) */
typedef struct {
  int wib, hib;        /* width/height in blocks (real blocks) */
  int wpad, hpad;      /* rounded up to the sampling factors (dummy blocks included) */
  int pw, ph;          /* sample plane size: wib*8 x hib*8 */
} mjo_geom;
void mjo_geometry(const mjo_params *p, mjo_geom g[MJO_MAX_COMPS], int *mcus_per_row, int *mcu_rows);

/* Stage taps.  All buffers are caller-allocated. */
/* a1-a3: colour conversion + downsampling + edge replication -> planes[c] (g[c].pw x g[c].ph) */
void mjo_color_downsample(const mjo_params *p, const uint8_t *pixels, size_t row_stride, uint8_t *planes[MJO_MAX_COMPS]);
/* a4-a8: deringing + FDCT + quantize -> coef_uq/coef_q [hpad*wpad][64] natural order, dummies built */
void mjo_forward(const mjo_params *p, uint8_t *const planes[MJO_MAX_COMPS],
                 int16_t *coef_uq[MJO_MAX_COMPS], int16_t *coef_q[MJO_MAX_COMPS]);
/* a11 */
void mjo_gen_optimal_table(long freq[257], uint8_t bits[17], uint8_t huffval[256]);
/* one block: used by unit tests */
void mjo_fdct_islow(int data[64]);
void mjo_deringing(int data[64], int q0);

/* Whole encode.  Returns number of bytes written to out (0 on error / insufficient cap).
 * If taps != NULL the post-trellis quantized coefficients etc. are copied out. */
typedef struct {
  int16_t *coef_uq[MJO_MAX_COMPS];    /* optional, [hpad*wpad*64] */
  int16_t *coef_q0[MJO_MAX_COMPS];    /* optional, quantized before trellis */
  int16_t *coef_q[MJO_MAX_COMPS];     /* optional, final (after trellis) */
  uint8_t *planes[MJO_MAX_COMPS];     /* optional */
  /* final tables: [tbl][17] bits, [tbl][256] vals for the LAST sequential scan */
  uint8_t dc_bits[4][17], dc_vals[4][256], ac_bits[4][17], ac_vals[4][256];
} mjo_taps;

size_t mjo_encode(const mjo_params *p, const uint8_t *pixels, size_t row_stride,
                  uint8_t *out, size_t cap, mjo_taps *taps);

/* Same encode from caller-supplied component planes instead of pixels: jpeg_write_raw_data
 * (jcapistd.c:145) as tj3CompressFromYUVPlanes8 drives it (turbojpeg.c:1222-1335).  Plane ci is
 * src_w[ci] x src_h[ci] samples (uint8, or uint16 for 12-bit), rows src_stride[ci] BYTES apart; if it
 * is smaller than width_in_blocks*8 x height_in_blocks*8 its last sample / row is replicated. */
size_t mjo_encode_planes(const mjo_params *p, const uint8_t *const src[MJO_MAX_COMPS],
                         const size_t src_stride[MJO_MAX_COMPS], const int src_w[MJO_MAX_COMPS],
                         const int src_h[MJO_MAX_COMPS], uint8_t *out, size_t cap, mjo_taps *taps);

/* Entropy-code existing quantized coefficients: jpeg_write_coefficients (jctrans.c:44), what jpegtran does.
 * coef[ci] = height_in_blocks rows of blocks_per_row[ci] (>= width_in_blocks) blocks of 64 int16 in natural
 * order.  p->qtbl only goes into the DQT marker; p->trellis_quant must be 0 (jctrans.c:102). */
size_t mjo_encode_coefficients(const mjo_params *p, const int16_t *const coef[MJO_MAX_COMPS],
                               const size_t blocks_per_row[MJO_MAX_COMPS], uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
