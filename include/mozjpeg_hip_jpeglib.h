/*
 * mozjpeg_hip_jpeglib.h -- the libjpeg drop-in boundary of the MI355X hot path.
 *
 * libmozjpeg_hip_jpeg62.so (mozjpeg_amd/csrc/jpeg_shim.c) exports, with the reference's exact
 * signatures, the libjpeg entry points that bracket the encode hot path:
 *
 *   void       jpeg_start_compress (j_compress_ptr cinfo, boolean write_all_tables);
 *                  replaces jcapistd.c:44   (declared jpeglib.h:1065)
 *   JDIMENSION jpeg_write_scanlines(j_compress_ptr cinfo, JSAMPARRAY scanlines, JDIMENSION num_lines);
 *                  replaces jcapistd.c:90   (declared jpeglib.h:1067)
 *   JDIMENSION jpeg12_write_scanlines(j_compress_ptr cinfo, J12SAMPARRAY scanlines, JDIMENSION num_lines);
 *                  the 12-bit twin (declared jpeglib.h:1070): rows of 16-bit samples, data_precision 12
 *   void       jpeg_finish_compress(j_compress_ptr cinfo);
 *                  replaces jcapimin.c:176  (declared jpeglib.h:1076)
 *   JDIMENSION jpeg_write_raw_data(j_compress_ptr cinfo, JSAMPIMAGE data, JDIMENSION num_lines);
 *   JDIMENSION jpeg12_write_raw_data(j_compress_ptr cinfo, J12SAMPIMAGE data, JDIMENSION num_lines);
 *                  replace jcapistd.c:145   (declared jpeglib.h:1084,:1086): component planes, one iMCU row
 *                  per call, when cinfo->raw_data_in is set -- what tj3CompressFromYUVPlanes8 uses
 *                  (turbojpeg.c:1222).  jpeg_start_compress also fills the geometry fields callers read back
 *                  (comp_info[].width_in_blocks/height_in_blocks, max_h/v_samp_factor, total_iMCU_rows).
 *   void       jpeg_write_coefficients(j_compress_ptr cinfo, jvirt_barray_ptr *coef_arrays);
 *                  replaces jctrans.c:44    (declared jpeglib.h:1178): lossless re-encode of existing
 *                  quantized coefficients (jpegtran); the virtual arrays are read at jpeg_finish_compress,
 *                  so transforms executed in between are honoured.
 *
 *   void       jpeg_abort_compress(j_compress_ptr), jpeg_abort(j_common_ptr),
 *              jpeg_destroy_compress(j_compress_ptr), jpeg_destroy(j_common_ptr);
 *                  replace jcapimin.c:118-135 / jcomapi.c:30-106: release what the entry points above attached
 *                  to the object (staging image, cached encoder), then chain to the next definition in link order.
 *
 * Two libraries carry these symbols:
 *  - libmozjpeg_hip_jpeg62.so (INTERPOSING): only the symbols above; everything else of the libjpeg API
 *    (jpeg_CreateCompress, jpeg_set_defaults, jpeg_set_quality, jpeg_c_set_*_param, jpeg_mem_dest, jpeg_stdio_dest,
 *    jpeg_std_error, ...) keeps being served by the host's libjpeg.so.62; the shim is placed in front of it (link
 *    order or LD_PRELOAD), so an UNCHANGED client such as `cjpeg` runs the GPU path.
 *  - mozjpeg_amd/standalone/libjpeg.so.62 (STAND-ALONE, jpeg_shim.c + jpeg_api.c built with -DMJH_STANDALONE): the
 *    whole compress side of the API from our own sources (object life cycle jcapimin.c:34-135, error manager jerror.c,
 *    pool memory manager jmemmgr.c, jpeg_set_defaults / jpeg_set_quality / jpeg_simple_progression / colour spaces
 *    jcparam.c, extension accessors jcext.c, destinations jdatadst.c, markers / ICC / tables, jpeg_copy_critical_parameters
 *    jctrans.c:97); an unchanged `cjpeg` runs against it alone.  See INTEGRATION.md 1 / 1b.
 *
 * Contract kept from the reference:
 *  - call order / global_state: CSTATE_START -> SCANNING | RAW_OK | WRCOEFS -> START, wrong order = ERREXIT1(JERR_BAD_STATE)
 *  - scanline memory is only read during the call (rows are copied into the staging buffer)
 *  - output only through cinfo->dest (init_destination / empty_output_buffer / term_destination)
 *  - SOI + JFIF APP0 are emitted by jpeg_start_compress so that jpeg_write_marker / ICC / COM
 *    markers written by the application land where the reference would put them
 *  - errors never return: ERREXIT -> cinfo->err->error_exit.  A configuration the GPU path does
 *    not cover is an ERROR (message on stderr + JERR_NOT_COMPILED), not a silent CPU fallback;
 *    setting MOZJPEG_HIP_PASSTHROUGH=1 in the environment turns it into an explicit, logged
 *    hand-over to the next jpeg_start_compress in link order instead.
 *
 * This header intentionally declares nothing new: the prototypes are the ones in the tree's own
 * <jpeglib.h>, against which the shim is compiled (struct jpeg_compress_struct is ABI:
 * jcapimin.c:41-45 checks its size and JPEG_LIB_VERSION).
 */
#ifndef MOZJPEG_HIP_JPEGLIB_H
#define MOZJPEG_HIP_JPEGLIB_H
#define MOZJPEG_HIP_SHIM_SYMBOLS "jpeg_start_compress jpeg_write_scanlines jpeg12_write_scanlines jpeg_finish_compress jpeg_write_raw_data jpeg12_write_raw_data jpeg_write_coefficients jpeg_abort_compress jpeg_abort jpeg_destroy_compress jpeg_destroy"
#endif
