"""Fixture images and switch sets shared by the golden generator and the tests."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def images():
    import oracle_lib as O
    big = O.synthetic_frame(640, 480, 1234)
    rng = np.random.default_rng(7)
    edge = np.zeros((64, 80, 3), np.uint8)
    edge[:, :40] = 255
    edge[20:40, 50:70] = (255, 0, 0)
    return {
        "testorig": O.read_ppm(os.path.join(HERE, "testorig.ppm")),   # the reference's own test image
        "syn250x187": big[100:287, 50:300].copy(),
        "syn96x64": big[0:64, 0:96].copy(),
        "syn17x33": big[:33, :17].copy(),
        "syn1x1": big[:1, :1].copy(),
        "noise121x75": rng.integers(0, 256, (75, 121, 3), dtype=np.uint8),
        "edge80x64": edge,
        "syn640x480": big,
    }


# testimages/test.scan of the reference (a scan script its CTest suite feeds to `cjpeg -scans`): (components, Ss, Se, Ah, Al)
TEST_SCAN = [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 9, 0, 0), ((0,), 10, 41, 0, 2), ((0,), 10, 41, 2, 1), ((0,), 10, 41, 1, 0), ((0,), 42, 63, 0, 0),
             ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)]

# switch sets (cjpeg vocabulary, see oracle_lib.make_params).  "gpu": covered by the HIP path today.
CASES = [
    ("revert", dict(revert=True), True),
    ("revert_opt", dict(revert=True, optimize=True), True),
    ("base_notrellis_noover", dict(baseline=True, notrellis=True, noovershoot=True), True),
    ("base_notrellis", dict(baseline=True, notrellis=True), True),
    ("base_noover", dict(baseline=True, noovershoot=True), True),
    ("base_notrellis_dc", dict(baseline=True, notrellis_dc=True), True),
    ("base", dict(baseline=True), True),
    ("base_q90_444", dict(baseline=True, quality=90, sample=(1, 1)), True),
    ("base_q30", dict(baseline=True, quality=30), True),
    ("base_422", dict(baseline=True, sample=(2, 1)), True),
    ("revert_440", dict(revert=True, sample=(1, 2)), True),
    ("revert_gray", dict(revert=True, gray=True), True),
    ("base_gray", dict(baseline=True, gray=True), True),
    ("base_qtbl0", dict(baseline=True, quant_table=0), True),
    ("base_restart1", dict(baseline=True, restart=1), True),
    ("base_restart5b", dict(baseline=True, restart="5b"), True),
    ("fastcrush", dict(fastcrush=True), True),
    ("default_progressive", dict(), True),
    ("q85_420_progressive", dict(quality=85), True),
    ("revert_progressive", dict(revert=True, progressive=True), True),
    ("q5_16bit_tables", dict(quality=5, fastcrush=True), True),
    # quality 1: ten steps of table 0 are 8192 or more, and the reference's 8-bit FDCT manager hands `quantval << 3` to
    # compute_reciprocal as a UINT16 (jcdctmgr.c:182, :278-282) -- q = 8450 divides by 67600 mod 65536 = 2064 there, while the DQT
    # marker and the trellis see 8450.  Part of the reference's files (the noise fixture has coefficients that show it).
    ("q1_wrapped_divisors", dict(quality=1), True),
    ("q1_wrapped_divisors_gray_fastcrush_notrellis", dict(quality=1, gray=True, fastcrush=True, notrellis=True), True),
    ("q1_wrapped_divisors_revert_422", dict(quality=1, revert=True, progressive=True, sample=(2, 1)), True),
    # restart intervals inside progressive scans (emit_restart jcphuff.c:438; per-scan interval T10, DRI per scan)
    ("prog_search_restart1", dict(restart=1), True),
    ("fastcrush_restart2", dict(fastcrush=True, restart=2), True),
    ("revert_prog_restart3b", dict(revert=True, progressive=True, restart="3b"), True),
    ("fastcrush_444_restart1", dict(fastcrush=True, restart=1, sample=(1, 1)), True),
    ("gray_prog_restart1", dict(gray=True, restart=1), True),
    ("prog_notrellis_restart7b", dict(notrellis=True, restart="7b"), True),
    # more than one (statistics, trellis) round per component (JINT_TRELLIS_NUM_LOOPS)
    ("base_trellis_loops2", dict(baseline=True, trellis_loops=2), True),
    ("default_progressive_trellis_loops2", dict(trellis_loops=2), True),
    ("base_444_trellis_loops3", dict(baseline=True, trellis_loops=3, sample=(1, 1), quality=90), True),
    # input smoothing (cjpeg -smooth N, jcsample.c:306-455; context-row preprocessing jcprepct.c:200-262)
    ("revert_opt_smooth1", dict(revert=True, optimize=True, smooth=1), True),     # testorig: MD5_JPEG_420S_IFAST_OPT, CMakeLists.txt:1367
    ("base_smooth30", dict(baseline=True, smooth=30), True),
    ("base_444_smooth100", dict(baseline=True, smooth=100, sample=(1, 1)), True),
    ("revert_422_smooth50", dict(revert=True, smooth=50, sample=(2, 1)), True),
    ("revert_440_smooth10", dict(revert=True, smooth=10, sample=(1, 2)), True),
    ("gray_progressive_smooth20", dict(smooth=20, gray=True), True),
    # 4:1 luma sampling ratios (int_downsample jcsample.c:151; TurboJPEG's TJSAMP_411 / TJSAMP_441), up to 10 blocks per MCU
    ("revert_411", dict(revert=True, sample=(4, 1)), True),
    ("base_441", dict(baseline=True, sample=(1, 4)), True),
    ("base_4x2_restart1", dict(baseline=True, sample=(4, 2), restart=1), True),
    ("default_progressive_2x4", dict(sample=(2, 4)), True),
    ("base_411_smooth20", dict(baseline=True, sample=(4, 1), smooth=20), True),
    # JCS_RGB output (cjpeg -rgb): null_convert jccolor.c:479, Adobe APP14, component ids 'R' 'G' 'B', all-purpose progressive script
    ("rgb_revert", dict(revert=True, rgb=True), True),                      # + -icc test1.icc = the reference's MD5_JPEG_RGB_ISLOW (drop-in test)
    ("rgb_base", dict(baseline=True, rgb=True), True),
    ("rgb_default_progressive", dict(rgb=True), True),
    # cjpeg -trellis-dc-ver-weight W (JFLOAT_TRELLIS_DELTA_DC_WEIGHT, jcdctmgr.c:1069-1084): needs a block above inside the iMCU row
    ("base_dc_ver_weight1", dict(baseline=True, dc_ver_weight=1.0), True),
    ("base_440_dc_ver_weight0p7", dict(baseline=True, dc_ver_weight=0.7, sample=(1, 2)), True),
    ("q60_progressive_dc_ver_weight2", dict(dc_ver_weight=2.0, quality=60), True),
    ("base_2x4_dc_ver_weight0p3", dict(baseline=True, dc_ver_weight=0.3, sample=(2, 4)), True),
    # cjpeg -dc-scan-opt N (JINT_DC_SCAN_OPT_MODE, jcparam.c:791-794,:934-947, jcmaster.c:836-838,:904-913)
    ("dc_scan_opt1", dict(dc_scan_opt=1), True),
    ("dc_scan_opt2", dict(dc_scan_opt=2), True),
    ("fastcrush_dc_scan_opt1", dict(fastcrush=True, dc_scan_opt=1), True),
    ("fastcrush_dc_scan_opt2", dict(fastcrush=True, dc_scan_opt=2), True),
    ("q30_444_dc_scan_opt2_restart1", dict(dc_scan_opt=2, quality=30, sample=(1, 1), restart=1), True),
    ("gray_dc_scan_opt1", dict(dc_scan_opt=1, gray=True), True),
    # trellis_q_opt (JBOOLEAN_TRELLIS_Q_OPT, sums jcdctmgr.c:1299-1306, table update jcmaster.c:1014-1030); with more than one
    # trellis round the reference re-estimates the tables between groups of its component-major passes: same order on the device
    ("base_trellis_q_opt", dict(baseline=True, trellis_q_opt=True), True),
    ("default_progressive_trellis_q_opt", dict(trellis_q_opt=True), True),
    ("base_422_trellis_q_opt_loops3", dict(baseline=True, trellis_q_opt=True, trellis_loops=3, sample=(2, 1)), True),
    # trellis_eob_opt (jcdctmgr.c:1224-1297) and use_scans_in_trellis (jcmaster.c:451-460)
    ("default_progressive_eob_opt", dict(trellis_eob_opt=True), True),
    ("fastcrush_scans_in_trellis_eob_opt", dict(fastcrush=True, use_scans_in_trellis=True, trellis_eob_opt=True), True),
    ("base_scans_in_trellis", dict(baseline=True, use_scans_in_trellis=True), True),
    ("progressive_all_trellis_options", dict(use_scans_in_trellis=True, trellis_freq_split=5, trellis_eob_opt=True, trellis_q_opt=True,
                                             trellis_loops=2), True),
    ("progressive_all_trellis_options_1loop", dict(use_scans_in_trellis=True, trellis_freq_split=5, trellis_eob_opt=True, trellis_q_opt=True,
                                                   dc_ver_weight=0.5), True),
    ("base_444_eob_opt_q90", dict(baseline=True, trellis_eob_opt=True, quality=90, sample=(1, 1)), True),
    ("base_scans_in_trellis_loops2_split20", dict(baseline=True, use_scans_in_trellis=True, trellis_loops=2, trellis_freq_split=20), True),
    ("base_q_opt_scans_split63", dict(baseline=True, trellis_q_opt=True, use_scans_in_trellis=True, trellis_freq_split=63), True),   # empty second band
    # trellis_q_opt on 16-bit tables: the DQT is written at the precision of the FINAL values (jcmarker.c:189-254 behind
    # jcmaster.c:1014-1030): both tables stay 16-bit / one 8-bit and one 16-bit table
    ("q3_16bit_trellis_q_opt", dict(quality=3, fastcrush=True, trellis_q_opt=True), True),
    ("q20_table0_mixed_precision_q_opt", dict(quality=20, fastcrush=True, quant_table=0, trellis_q_opt=True), True),
    ("q40_progressive_eob_opt_restart1", dict(trellis_eob_opt=True, quality=40, restart=1), True),
    # sampling factors beyond "luma HxV, chroma 1x1" (cjpeg -sample HxV,HxV,HxV; initial_setup jcmaster.c:210-259): chroma that is
    # itself subsampled unevenly, luma smaller than chroma, a 3:1 ratio (int_downsample jcsample.c:151), all components 2x1
    ("base_samp_22_21_11", dict(baseline=True, sample=((2, 2), (2, 1), (1, 1))), True),
    ("default_samp_21_11_12", dict(sample=((2, 1), (1, 1), (1, 2))), True),
    ("revert_samp_12_22_11", dict(revert=True, sample=((1, 2), (2, 2), (1, 1))), True),
    ("base_samp_31_11_11", dict(baseline=True, sample=((3, 1), (1, 1), (1, 1))), True),
    ("fastcrush_samp_21_21_21_restart1", dict(fastcrush=True, restart=1, sample=((2, 1), (2, 1), (2, 1))), True),
    # scan scripts given by the application (cjpeg -scans; refenc -scanspec): sequential files of several whole-block scans
    # (validate_script jcmaster.c:309-330: every scan has its own MCU order, statistics pass, tables and restart interval) and a
    # progressive script that is neither jpeg_simple_progression's nor the search's
    ("script_seq_y_cbcr", dict(scans=[((0,), 0, 63, 0, 0), ((1, 2), 0, 63, 0, 0)]), True),
    ("script_seq_each_restart1", dict(restart=1, scans=[((0,), 0, 63, 0, 0), ((1,), 0, 63, 0, 0), ((2,), 0, 63, 0, 0)]), True),
    ("script_seq_ycb_cr_revert_422", dict(revert=True, sample=(2, 1), scans=[((0, 1), 0, 63, 0, 0), ((2,), 0, 63, 0, 0)]), True),
    ("script_seq_arith_y_cbcr_restart1", dict(arithmetic=True, restart=1, scans=[((0,), 0, 63, 0, 0), ((1, 2), 0, 63, 0, 0)]), True),
    ("script_prog_custom", dict(scans=[((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 63, 0, 1), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0),
                                       ((0, 1, 2), 0, 0, 1, 0), ((0,), 1, 63, 1, 0)]), True),
    # arithmetic entropy coding (cjpeg -arithmetic): jcarith.c (sequential SOF9, progressive SOF10, restarts,
    # DAC markers), with the coder's own trellis rate model (quantize_trellis_arith jcdctmgr.c:1334-1667) where trellis is on
    ("arith_revert", dict(arithmetic=True, revert=True), True),
    ("arith_base_notrellis", dict(arithmetic=True, baseline=True, notrellis=True), True),
    ("arith_base", dict(arithmetic=True, baseline=True), True),
    ("arith_base_notrellis_dc", dict(arithmetic=True, baseline=True, notrellis_dc=True), True),
    ("arith_base_q92_444", dict(arithmetic=True, baseline=True, quality=92, sample=(1, 1)), True),
    ("arith_base_restart1", dict(arithmetic=True, baseline=True, restart=1), True),
    ("arith_base_restart5b", dict(arithmetic=True, baseline=True, restart="5b"), True),
    ("arith_base_gray", dict(arithmetic=True, baseline=True, gray=True), True),
    ("arith_base_dc_ver_weight1", dict(arithmetic=True, baseline=True, dc_ver_weight=1.0), True),
    ("arith_revert_progressive", dict(arithmetic=True, revert=True, progressive=True), True),
    ("arith_fastcrush_notrellis", dict(arithmetic=True, fastcrush=True, notrellis=True), True),
    ("arith_fastcrush", dict(arithmetic=True, fastcrush=True), True),
    ("arith_fastcrush_restart2", dict(arithmetic=True, fastcrush=True, restart=2), True),
    ("arith_default_progressive", dict(arithmetic=True), True),
    ("arith_q40_422_progressive", dict(arithmetic=True, quality=40, sample=(2, 1)), True),
    # trellis_q_opt with the arithmetic coder: T = (1 or 2) * components * loops + 1 trellis passes of component 0, tables
    # re-estimated behind every pass with (pass + 1) % ((2 or 4) * components) == 0 (jcmaster.c:687-698, :1016-1030, :1135-1138):
    # never with three components and one loop (the reference writes the file it writes without the option) ...
    ("arith_base_q_opt", dict(arithmetic=True, baseline=True, trellis_q_opt=True), True),
    ("arith_default_progressive_q_opt_scans_in_trellis", dict(arithmetic=True, trellis_q_opt=True, use_scans_in_trellis=True), True),
    # ... once for a gray image and one loop (the estimate only reaches the DQT marker: no trellis pass follows it), once or
    # more with further loops (later passes quantize component 0 with the estimated table); a 16-bit table that turns 8-bit
    ("arith_gray_q_opt", dict(arithmetic=True, baseline=True, gray=True, trellis_q_opt=True), True),
    ("arith_base_q_opt_loops2", dict(arithmetic=True, baseline=True, trellis_q_opt=True, trellis_loops=2), True),
    ("arith_gray_q_opt_loops4_restart2", dict(arithmetic=True, baseline=True, gray=True, trellis_q_opt=True, trellis_loops=4, restart=2), True),
    ("arith_default_progressive_q_opt_loops4", dict(arithmetic=True, trellis_q_opt=True, trellis_loops=4), True),
    ("arith_fastcrush_q_opt_loops3_scans_in_trellis_444", dict(arithmetic=True, fastcrush=True, trellis_q_opt=True, trellis_loops=3, use_scans_in_trellis=True, sample=(1, 1)), True),
    ("arith_q3_16bit_q_opt_loops2", dict(arithmetic=True, quality=3, trellis_q_opt=True, trellis_loops=2), True),
    # non-default conditioning (cinfo->arith_dc_L / arith_dc_U / arith_ac_K, API-only fields; refenc -arith-cond): DC category
    # thresholds and the AC position Kx, per table, in the coder, its trellis rate model and the DAC marker
    ("arith_base_cond", dict(arithmetic=True, baseline=True, arith_cond=((1, 3, 9), (0, 2, 20))), True),
    ("arith_fastcrush_cond", dict(arithmetic=True, fastcrush=True, arith_cond=((2, 5, 1), (1, 1, 63))), True),
    ("arith_default_progressive_cond", dict(arithmetic=True, quality=85, arith_cond=((0, 15, 12), (3, 4, 2))), True),
    # ONE component sampled other than 1x1 (SOF byte only: its scans are non-interleaved, per_scan_setup jcmaster.c:548-575).  cjpeg
    # does this to every gray image at qualities 80..89 (set_quality_ratings rdswitch.c:566-570 sets 2x1 on component 0) --
    # tools/simt/fuzz_cjpeg.py found the encoder refusing `cjpeg -quality 85 -grayscale`.  V > 1 only without the trellis
    # (compress_trellis_pass chains the DC trellis over the V block rows of an iMCU row: the DC trellis kernels get a view of the
    # geometry with that V -- the gray_*x2* / *x4* cases with the trellis on).
    ("gray_2x1_q85_progressive", dict(gray=True, quality=85, gray_sample=(2, 1)), True),
    ("gray_2x1_q85_base", dict(gray=True, baseline=True, quality=85, gray_sample=(2, 1)), True),
    ("gray_2x1_revert_restart1", dict(gray=True, revert=True, restart=1, gray_sample=(2, 1)), True),
    ("gray_4x1_base_smooth30_restart3b", dict(gray=True, baseline=True, smooth=30, restart="3b", gray_sample=(4, 1)), True),
    ("gray_2x1_arith_fastcrush", dict(gray=True, arithmetic=True, fastcrush=True, gray_sample=(2, 1)), True),
    ("gray_2x2_base_notrellis", dict(gray=True, baseline=True, notrellis=True, gray_sample=(2, 2)), True),
    ("gray_1x2_revert_progressive", dict(gray=True, revert=True, progressive=True, gray_sample=(1, 2)), True),
    ("gray_2x2_base_trellis", dict(gray=True, baseline=True, gray_sample=(2, 2)), True),
    ("gray_1x2_progressive_dc_ver_weight2", dict(gray=True, dc_ver_weight=2.0, gray_sample=(1, 2)), True),
    ("gray_1x4_fastcrush_restart1_eob_opt_loops2", dict(gray=True, fastcrush=True, restart=1, trellis_eob_opt=True, trellis_loops=2, gray_sample=(1, 4)), True),
    ("gray_2x2_arith_base_q_opt", dict(gray=True, baseline=True, arithmetic=True, trellis_q_opt=True, gray_sample=(2, 2)), True),
    # input samples that are Y, Cb, Cr already (in_color_space = JCS_YCbCr -> null_convert, jccolor.c:687-692, :479): MJH_COLOR_YCC_IN;
    # the fixtures' bytes are simply read as YCbCr
    ("yccin_base", dict(baseline=True, yccin=True), True),
    ("yccin_progressive_422", dict(yccin=True, sample=(2, 1)), True),
    ("yccin_revert_restart1", dict(revert=True, yccin=True, restart=1), True),
    ("yccin_gray_progressive", dict(yccin=True, gray=True), True),        # YCbCr samples into a grayscale file: the Y samples (grayscale_convert jccolor.c:448-466)
    # JDCT_IFAST (cjpeg -dct fast; what the legacy tjCompress2 selects below quality 96): jfdctfst.c, divisors jcdctmgr.c:291-345, and
    # under the max-compression profile the trellis on the unscaled coefficients (:731-750).  The first three are the reference's own
    # bit tests on testorig (CMakeLists.txt:1459, :1498 with testimages/test.scan, :1561)
    ("ifast_revert_422_opt", dict(revert=True, sample=(2, 1), dct="fast", optimize=True), True),
    ("ifast_revert_q100_testscan", dict(revert=True, quality=100, dct="fast", scans=TEST_SCAN), True),
    ("ifast_revert_3x2_prog", dict(revert=True, sample=(3, 2), dct="fast", progressive=True), True),
    ("ifast_revert", dict(revert=True, dct="fast"), True),                       # tjCompress2(..., TJSAMP_420, 75, 0)
    ("ifast_revert_q30_444", dict(revert=True, dct="fast", quality=30, sample=(1, 1)), True),
    ("ifast_revert_gray_restart1", dict(revert=True, dct="fast", gray=True, restart=1), True),
    ("ifast_base", dict(baseline=True, dct="fast"), True),                       # cjpeg -baseline -dct fast: trellis + deringing on AA&N coefficients
    ("ifast_default_progressive", dict(dct="fast"), True),
    ("ifast_q1_wrapped_divisors", dict(quality=1, dct="fast", baseline=True), True),
    ("ifast_base_q92_444_notrellis_dc", dict(baseline=True, dct="fast", quality=92, sample=(1, 1), notrellis_dc=True), True),
    ("ifast_arith_base", dict(arithmetic=True, baseline=True, dct="fast"), True),
    ("ifast_base_trellis_q_opt", dict(baseline=True, dct="fast", trellis_q_opt=True), True),
    # table numbers of the application's own (API-only): any two DC table numbers in a progressive image (round 6; until then two
    # numbers with the same low bit were refused), any AC table numbers
    ("prog_dctbl_022", dict(dc_tbl=(0, 2, 2)), True),
    ("prog_dctbl_133_actbl_203", dict(dc_tbl=(1, 3, 3), ac_tbl=(2, 0, 3), fastcrush=True), True),
    ("prog_dctbl_313_revert", dict(dc_tbl=(3, 1, 3), revert=True, progressive=True), True),
    ("prog_dctbl_200_restart1", dict(dc_tbl=(2, 0, 0), fastcrush=True, restart=1), True),
    ("revert_opt_dctbl_232", dict(dc_tbl=(2, 3, 2), ac_tbl=(1, 3, 0), revert=True, optimize=True), True),
    # the trellis with optimize_coding switched off by hand (API-only), one component: the reference's passes are the schedule of
    # optimize_coding, its file the same bytes (round 6; colour images stay refused: the reference's own djpeg rejects its files)
    ("base_gray_no_optimize", dict(baseline=True, gray=True, no_optimize=True), True),
    ("base_gray_q90_loops3_no_optimize", dict(baseline=True, gray=True, quality=90, trellis_loops=3, no_optimize=True), True),
    ("base_gray_restart1_eob_opt_no_optimize", dict(baseline=True, gray=True, restart=1, trellis_eob_opt=True, no_optimize=True), True),
    ("base_gray_ifast_q_opt_no_optimize", dict(baseline=True, gray=True, dct="fast", trellis_q_opt=True, no_optimize=True), True),
]



def images12():
    """12-bit fixtures (uint16 samples 0..4095)"""
    import oracle_lib as O
    big = O.synthetic_frame12(640, 480, 1234)
    rng = np.random.default_rng(9)
    return {
        "syn12_250x187": big[100:287, 50:300].copy(),
        "syn12_17x33": big[:33, :17].copy(),
        "syn12_1x1": big[:1, :1].copy(),
        "noise12_121x75": rng.integers(0, 4096, (75, 121, 3)).astype(np.uint16),
    }


# 12-bit: trellis has no reference behaviour (jccoefct.c:132-138 aborts) => -notrellis everywhere
CASES12 = [
    ("p12_base_q90_444", dict(precision=12, baseline=True, notrellis=True, quality=90, sample=(1, 1)), True),
    ("p12_base_420", dict(precision=12, baseline=True, notrellis=True), True),
    ("p12_base_q90_444_noover_restart1", dict(precision=12, baseline=True, notrellis=True, noovershoot=True, quality=90, sample=(1, 1), restart=1), True),
    ("p12_default_progressive", dict(precision=12, notrellis=True), True),
    ("p12_fastcrush", dict(precision=12, notrellis=True, fastcrush=True), True),
    ("p12_revert_q90", dict(precision=12, revert=True, quality=90), True),
    ("p12_base_gray", dict(precision=12, baseline=True, notrellis=True, gray=True), True),
    # 12-bit sequential scan scripts: optimal tables are forced (the standard ones are 8-bit), so EVERY scan carries its tables
    # (tools/simt/fuzz_more.py found the encoder sending them only once)
    ("p12_script_seq_ycr_cb", dict(precision=12, notrellis=True, revert=True, scans=[((0, 2), 0, 63, 0, 0), ((1,), 0, 63, 0, 0)]), True),
    ("p12_script_seq_y_cb_cr_restart1", dict(precision=12, notrellis=True, baseline=True, restart=1, scans=[((0,), 0, 63, 0, 0), ((1,), 0, 63, 0, 0), ((2,), 0, 63, 0, 0)]), True),
    # 12-bit samples through the arithmetic coder (magnitude categories up to 15 bits)
    ("p12_arith_base_q90_444", dict(precision=12, arithmetic=True, baseline=True, notrellis=True, quality=90, sample=(1, 1)), True),
    ("p12_arith_progressive", dict(precision=12, arithmetic=True, notrellis=True), True),
    # JDCT_IFAST on 12-bit samples (jfdctfst.c; the divisors stay full-width DCTELEMs, jcdctmgr.c:337-341)
    ("p12_ifast_revert", dict(precision=12, revert=True, dct="fast"), True),
    ("p12_ifast_base_q90_444", dict(precision=12, baseline=True, notrellis=True, quality=90, sample=(1, 1), dct="fast"), True),
    ("p12_ifast_fastcrush_q1", dict(precision=12, notrellis=True, fastcrush=True, quality=1, dct="fast"), True),
]

# constants the REFERENCE itself pins for this path (CMakeLists.txt:1347-1420), cjpeg -revert ... testorig.ppm
REFERENCE_PINNED = {
    ("testorig", "revert"): "9a68f56bc76e466aa7e52f415d0f4a5f",        # MD5_JPEG_420_ISLOW   :1391
    ("testorig", "revert_440"): "538bc02bd4b4658fd85de6ece6cbeda6",    # MD5_JPEG_440_ISLOW   :1354
    ("testorig", "revert_gray"): "72b51f894b8f4a10b3ee3066770aa38d",   # MD5_JPEG_GRAY_ISLOW  :1362
    ("testorig", "revert_opt_smooth1"): "388708217ac46273ca33086b22827ed8",   # MD5_JPEG_420S_IFAST_OPT :1367 (-sample 2x2 -smooth 1 -dct int -opt)
    ("testorig", "ifast_revert_422_opt"): "2540287b79d913f91665e660303ab2c8",         # MD5_JPEG_422_IFAST_OPT :1352 (-revert -sample 2x1 -dct fast -opt)
    ("testorig", "ifast_revert_q100_testscan"): "0ba15f9dab81a703505f835f9dbbac6d",   # MD5_JPEG_420_IFAST_Q100_PROG :1359 (-revert -sample 2x2 -quality 100 -dct fast -scans test.scan)
    ("testorig", "ifast_revert_3x2_prog"): "1ee5d2c1a77f2da495f993c8c7cceca5",        # MD5_JPEG_3x2_IFAST_PROG :1386 (-revert -sample 3x2 -dct fast -prog)
}


# Planar (raw data) input: jpeg_write_raw_data as tj3CompressFromYUVPlanes8 drives it.
# (width, height, switches); the planes come from oracle_lib.synthetic_planes(params, seed=7).
PLANE_CASES = [
    ("yuv_227x149_base", 227, 149, dict(baseline=True)),
    ("yuv_64x48_revert", 64, 48, dict(revert=True)),                       # TurboJPEG's configuration
    ("yuv_130x75_base_422", 130, 75, dict(baseline=True, sample=(2, 1))),
    ("yuv_97x61_q85_progressive", 97, 61, dict(quality=85)),
    ("yuv_50x33_base_gray", 50, 33, dict(baseline=True, sample=(1, 1), gray=True)),
    ("yuv_101x77_fastcrush_440", 101, 77, dict(fastcrush=True, sample=(1, 2))),
    ("yuv_1x1_base", 1, 1, dict(baseline=True)),
    ("yuv_321x243_base_444_restart1", 321, 243, dict(baseline=True, sample=(1, 1), restart=1)),
    ("yuv_640x480_revert_opt", 640, 480, dict(revert=True, optimize=True)),
]


# Transcoding (jpeg_write_coefficients): (name, fixture image, switches of the encode that made the
# source file, jpegtran switches, the same switches in make_params vocabulary).  The source file is produced by the
# oracle (itself pinned above); the golden is the REAL jpegtran's output for that file.
TRANSCODE_CASES = [
    ("tr_base_to_rescan", "syn250x187", dict(baseline=True), ["-progressive"], dict()),
    ("tr_base_to_revert", "syn250x187", dict(baseline=True), ["-revert"], dict(revert=True)),
    ("tr_base_to_revert_opt", "syn250x187", dict(baseline=True), ["-revert", "-optimize"], dict(revert=True, optimize=True)),
    ("tr_revert422_to_rescan", "syn250x187", dict(revert=True, sample=(2, 1)), ["-progressive"], dict()),
    ("tr_revert422_to_revert_prog", "syn250x187", dict(revert=True, sample=(2, 1)), ["-revert", "-progressive"], dict(revert=True, progressive=True)),
    ("tr_gray_to_rescan", "syn96x64", dict(baseline=True, gray=True), ["-progressive"], dict()),
    ("tr_444_to_fastcrush", "noise121x75", dict(quality=90, sample=(1, 1), fastcrush=True), ["-fastcrush", "-progressive"], dict(fastcrush=True)),
    ("tr_base_to_revert_restart2", "testorig", dict(baseline=True), ["-revert", "-restart", "2"], dict(revert=True, restart=2)),
    ("tr_base_to_revert_opt_restart3b", "testorig", dict(baseline=True), ["-revert", "-optimize", "-restart", "3B"], dict(revert=True, optimize=True, restart="3b")),
    ("tr_testorig_default_to_rescan", "testorig", dict(), ["-progressive"], dict()),
    ("tr_1x1_to_rescan", "syn1x1", dict(baseline=True), ["-progressive"], dict()),
    ("tr_640x480_to_rescan", "syn640x480", dict(baseline=True), ["-progressive"], dict()),
    # jpegtran -arithmetic: Huffman-coded source re-coded with the arithmetic coder (and the scan search sizing its candidates with it)
    ("tr_base_to_arith_rescan", "syn250x187", dict(baseline=True), ["-arithmetic", "-progressive"], dict(arithmetic=True)),
    ("tr_base_to_arith_revert", "syn250x187", dict(baseline=True), ["-arithmetic", "-revert"], dict(arithmetic=True, revert=True)),
    ("tr_revert422_to_arith_revert_restart2", "testorig", dict(revert=True, sample=(2, 1)), ["-arithmetic", "-revert", "-restart", "2"], dict(arithmetic=True, revert=True, restart=2)),
]
