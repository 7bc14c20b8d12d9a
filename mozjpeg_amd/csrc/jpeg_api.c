/*
 * jpeg_api.c -- the rest of the libjpeg COMPRESS API around the MI355X hot path: together with
 * jpeg_shim.c (-DMJH_STANDALONE) this file is a libjpeg.so.62 an unchanged client such as cjpeg can run against without
 * any code of the reference in the process.  What is restated here is host-side bookkeeping only -- object life cycle,
 * parameter setters, memory / error / destination managers, marker helpers; the pixel -> bytes path is the GPU's.
 *
 * Compiled against the libjpeg headers of the tree it replaces (struct jpeg_compress_struct, jpeg_memory_mgr,
 * jpeg_error_mgr ... are ABI; the message texts come from that tree's jerror.h the way its own jerror.c gets them).
 * Each entry point cites the reference function whose behaviour it keeps.
 *
 * Not provided (this is the compress half): the decompressor, lossless mode, arithmetic coding, backing-store files
 * of the memory manager (virtual arrays always live in memory).  The decompress symbols an unchanged cjpeg binary
 * references (it can read JPEG input files) exist as stubs that raise JERR_NOT_COMPILED.
 */
#define JPEG_INTERNALS
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jinclude.h"
#include "jpeglib.h"   /* JPEG_INTERNALS: jpegint.h + jerror.h */
#include "jversion.h"  /* JVERSION / JCOPYRIGHT_SHORT: two of the message texts */

#include "mozjpeg_hip.h"
#include "jpeg_shim.h"
#include "mjh_quant_presets.h"

/* =====================================================================================================================
 * error manager -- jerror.c:70-243
 * ===================================================================================================================== */
#define JMESSAGE(code, string)  string,
const char * const jpeg_std_message_table[] = {
#include "jerror.h"
  NULL
};

static void api_output_message(j_common_ptr cinfo)
{ /* output_message jerror.c:97-114 */
  char buffer[JMSG_LENGTH_MAX];
  (*cinfo->err->format_message) (cinfo, buffer);
  fprintf(stderr, "%s\n", buffer);
}

static void api_error_exit(j_common_ptr cinfo)
{ /* error_exit jerror.c:70-80: message, clean up, exit */
  (*cinfo->err->output_message) (cinfo);
  jpeg_destroy(cinfo);
  exit(EXIT_FAILURE);
}

static void api_emit_message(j_common_ptr cinfo, int msg_level)
{ /* emit_message jerror.c:126-149: first warning always, later ones only at trace level >= 3; traces by level */
  struct jpeg_error_mgr *err = cinfo->err;
  if (msg_level < 0) {
    if (err->num_warnings == 0 || err->trace_level >= 3) (*err->output_message) (cinfo);
    err->num_warnings++;
  } else if (err->trace_level >= msg_level)
    (*err->output_message) (cinfo);
}

static void api_format_message(j_common_ptr cinfo, char *buffer)
{ /* format_message jerror.c:158-205 */
  struct jpeg_error_mgr *err = cinfo->err;
  const int code = err->msg_code;
  const char *text = NULL, *p;
  int is_string = 0;
  if (code > 0 && code <= err->last_jpeg_message) text = err->jpeg_message_table[code];
  else if (err->addon_message_table != NULL && code >= err->first_addon_message && code <= err->last_addon_message)
    text = err->addon_message_table[code - err->first_addon_message];
  if (text == NULL) { err->msg_parm.i[0] = code; text = err->jpeg_message_table[0]; }
  for (p = text; *p; p++)
    if (*p == '%') { is_string = p[1] == 's'; break; }
  if (is_string) snprintf(buffer, JMSG_LENGTH_MAX, text, err->msg_parm.s);
  else snprintf(buffer, JMSG_LENGTH_MAX, text, err->msg_parm.i[0], err->msg_parm.i[1], err->msg_parm.i[2], err->msg_parm.i[3],
                err->msg_parm.i[4], err->msg_parm.i[5], err->msg_parm.i[6], err->msg_parm.i[7]);
}

static void api_reset_error_mgr(j_common_ptr cinfo)
{ /* reset_error_mgr jerror.c:216-222 */
  cinfo->err->num_warnings = 0;
  cinfo->err->msg_code = 0;
}

struct jpeg_error_mgr *jpeg_std_error(struct jpeg_error_mgr *err)
{ /* jerror.c:231-251 */
  memset(err, 0, sizeof(*err));
  err->error_exit = api_error_exit;
  err->emit_message = api_emit_message;
  err->output_message = api_output_message;
  err->format_message = api_format_message;
  err->reset_error_mgr = api_reset_error_mgr;
  err->jpeg_message_table = jpeg_std_message_table;
  err->last_jpeg_message = (int)JMSG_LASTMSGCODE - 1;
  return err;
}

/* =====================================================================================================================
 * small utilities -- jutils.c
 * ===================================================================================================================== */
const int jpeg_natural_order[DCTSIZE2 + 16] = {   /* jutils.c:59-79: zig-zag -> natural, 16 extra entries for safety */
  0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63
};
long jdiv_round_up(long a, long b) { return (a + b - 1L) / b; }
long jround_up(long a, long b) { a += b - 1L; return a - (a % b); }
void jcopy_sample_rows(JSAMPARRAY input_array, int source_row, JSAMPARRAY output_array, int dest_row, int num_rows, JDIMENSION num_cols)
{
  int r;
  for (r = 0; r < num_rows; r++) memcpy(output_array[dest_row + r], input_array[source_row + r], (size_t)num_cols * sizeof(JSAMPLE));
}
void jcopy_block_row(JBLOCKROW input_row, JBLOCKROW output_row, JDIMENSION num_blocks)
{
  memcpy(output_row, input_row, (size_t)num_blocks * sizeof(JBLOCK));
}
void jzero_far(void *target, size_t bytestozero) { memset(target, 0, bytestozero); }

/* =====================================================================================================================
 * memory manager -- jmemmgr.c (pools, sample / block arrays, virtual arrays held entirely in memory)
 * ===================================================================================================================== */
typedef struct api_chunk { struct api_chunk *next; size_t size; double align_; } api_chunk;   /* header of every allocation */

/* The control blocks keep the reference's field order (jmemmgr.c:146-181): clients that mix this library with another
 * libjpeg in one process (a decompressor in front of jpegtran's transforms) hand such arrays across. */
struct jvirt_sarray_control {
  JSAMPARRAY mem_buffer;
  JDIMENSION rows_in_array, samplesperrow, maxaccess, rows_in_mem, rowsperchunk, cur_start_row, first_undef_row;
  boolean pre_zero, dirty, b_s_open;
  jvirt_sarray_ptr next;
};
struct jvirt_barray_control {
  JBLOCKARRAY mem_buffer;
  JDIMENSION rows_in_array, blocksperrow, maxaccess, rows_in_mem, rowsperchunk, cur_start_row, first_undef_row;
  boolean pre_zero, dirty, b_s_open;
  jvirt_barray_ptr next;
};

typedef struct {
  struct jpeg_memory_mgr pub;
  api_chunk *pools[JPOOL_NUMPOOLS];
  jvirt_sarray_ptr virt_sarray_list;
  jvirt_barray_ptr virt_barray_list;
} api_mem;

static size_t sample_size(j_common_ptr cinfo)
{ /* rows of 12- and 16-bit objects hold 2-byte samples (alloc_sarray jmemmgr.c:441-452) */
  const int prec = cinfo->is_decompressor ? ((j_decompress_ptr)cinfo)->data_precision : ((j_compress_ptr)cinfo)->data_precision;
  return prec > 8 ? 2 : 1;
}

static void *pool_alloc(j_common_ptr cinfo, int pool_id, size_t size)
{
  api_mem *m = (api_mem *)cinfo->mem;
  api_chunk *c;
  if (pool_id < 0 || pool_id >= JPOOL_NUMPOOLS) ERREXIT1(cinfo, JERR_BAD_POOL_ID, pool_id);
  if (size > (size_t)1000000000) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 7);   /* MAX_ALLOC_CHUNK jmemmgr.c / jmemnobs.c */
  c = (api_chunk *)malloc(sizeof(api_chunk) + size + 32);
  if (c == NULL) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 1);
  c->size = size;
  c->next = m->pools[pool_id];
  m->pools[pool_id] = c;
  {   /* 32-byte aligned payload like the reference's ALIGN_SIZE */
    unsigned char *p = (unsigned char *)(c + 1);
    p += (32 - ((size_t)p & 31)) & 31;
    return p;
  }
}
static void *api_alloc_small(j_common_ptr cinfo, int pool_id, size_t size) { return pool_alloc(cinfo, pool_id, size); }
static void *api_alloc_large(j_common_ptr cinfo, int pool_id, size_t size) { return pool_alloc(cinfo, pool_id, size); }

static JSAMPARRAY api_alloc_sarray(j_common_ptr cinfo, int pool_id, JDIMENSION samplesperrow, JDIMENSION numrows)
{ /* alloc_sarray jmemmgr.c:415-520: row pointers + rows, rows padded to the alignment unit */
  const size_t ss = sample_size(cinfo);
  size_t rowbytes;
  JSAMPARRAY rows;
  unsigned char *data;
  JDIMENSION r;
  if (samplesperrow == 0) ERREXIT1(cinfo, JERR_WIDTH_OVERFLOW, 0);
  rowbytes = (((size_t)samplesperrow * ss) + 63) & ~(size_t)63;
  if (numrows && rowbytes > (size_t)1000000000 / numrows) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 3);
  rows = (JSAMPARRAY)pool_alloc(cinfo, pool_id, (size_t)numrows * sizeof(JSAMPROW));
  data = (unsigned char *)pool_alloc(cinfo, pool_id, rowbytes * numrows);
  for (r = 0; r < numrows; r++) rows[r] = (JSAMPROW)(data + (size_t)r * rowbytes);
  return rows;
}

static JBLOCKARRAY api_alloc_barray(j_common_ptr cinfo, int pool_id, JDIMENSION blocksperrow, JDIMENSION numrows)
{ /* alloc_barray jmemmgr.c:527-573 */
  JBLOCKARRAY rows;
  JBLOCKROW data;
  JDIMENSION r;
  if (blocksperrow == 0) ERREXIT1(cinfo, JERR_WIDTH_OVERFLOW, 0);
  if (numrows && (size_t)blocksperrow * sizeof(JBLOCK) > (size_t)1000000000 / numrows) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 6);
  rows = (JBLOCKARRAY)pool_alloc(cinfo, pool_id, (size_t)numrows * sizeof(JBLOCKROW));
  data = (JBLOCKROW)pool_alloc(cinfo, pool_id, (size_t)blocksperrow * numrows * sizeof(JBLOCK));
  for (r = 0; r < numrows; r++) rows[r] = data + (size_t)r * blocksperrow;
  return rows;
}

static jvirt_sarray_ptr api_request_virt_sarray(j_common_ptr cinfo, int pool_id, boolean pre_zero, JDIMENSION samplesperrow,
                                                JDIMENSION numrows, JDIMENSION maxaccess)
{ /* request_virt_sarray jmemmgr.c:611-639: only JPOOL_IMAGE, realised later */
  api_mem *m = (api_mem *)cinfo->mem;
  jvirt_sarray_ptr v;
  if (pool_id != JPOOL_IMAGE) ERREXIT1(cinfo, JERR_BAD_POOL_ID, pool_id);
  v = (jvirt_sarray_ptr)pool_alloc(cinfo, pool_id, sizeof(*v));
  memset(v, 0, sizeof(*v));
  v->rows_in_array = numrows; v->samplesperrow = samplesperrow; v->maxaccess = maxaccess; v->pre_zero = pre_zero;
  v->next = m->virt_sarray_list; m->virt_sarray_list = v;
  return v;
}

static jvirt_barray_ptr api_request_virt_barray(j_common_ptr cinfo, int pool_id, boolean pre_zero, JDIMENSION blocksperrow,
                                                JDIMENSION numrows, JDIMENSION maxaccess)
{ /* request_virt_barray jmemmgr.c:642-670 */
  api_mem *m = (api_mem *)cinfo->mem;
  jvirt_barray_ptr v;
  if (pool_id != JPOOL_IMAGE) ERREXIT1(cinfo, JERR_BAD_POOL_ID, pool_id);
  v = (jvirt_barray_ptr)pool_alloc(cinfo, pool_id, sizeof(*v));
  memset(v, 0, sizeof(*v));
  v->rows_in_array = numrows; v->blocksperrow = blocksperrow; v->maxaccess = maxaccess; v->pre_zero = pre_zero;
  v->next = m->virt_barray_list; m->virt_barray_list = v;
  return v;
}

static void api_realize_virt_arrays(j_common_ptr cinfo)
{ /* realize_virt_arrays jmemmgr.c:673-776 with unlimited memory: every array gets its full height in memory */
  api_mem *m = (api_mem *)cinfo->mem;
  jvirt_sarray_ptr s;
  jvirt_barray_ptr b;
  for (s = m->virt_sarray_list; s; s = s->next)
    if (s->mem_buffer == NULL) {
      s->rows_in_mem = s->rows_in_array;
      s->mem_buffer = api_alloc_sarray(cinfo, JPOOL_IMAGE, s->samplesperrow, s->rows_in_mem);
      s->rowsperchunk = s->rows_in_mem; s->cur_start_row = 0; s->first_undef_row = 0; s->dirty = FALSE;
    }
  for (b = m->virt_barray_list; b; b = b->next)
    if (b->mem_buffer == NULL) {
      b->rows_in_mem = b->rows_in_array;
      b->mem_buffer = api_alloc_barray(cinfo, JPOOL_IMAGE, b->blocksperrow, b->rows_in_mem);
      b->rowsperchunk = b->rows_in_mem; b->cur_start_row = 0; b->first_undef_row = 0; b->dirty = FALSE;
    }
}

static JSAMPARRAY api_access_virt_sarray(j_common_ptr cinfo, jvirt_sarray_ptr p, JDIMENSION start_row, JDIMENSION num_rows, boolean writable)
{ /* access_virt_sarray jmemmgr.c:851-933, in-memory case: range checks + the undefined-rows protocol */
  const JDIMENSION end_row = start_row + num_rows;
  if (end_row > p->rows_in_array || num_rows > p->maxaccess || p->mem_buffer == NULL) ERREXIT(cinfo, JERR_BAD_VIRTUAL_ACCESS);
  if (start_row < p->cur_start_row || end_row > p->cur_start_row + p->rows_in_mem) ERREXIT(cinfo, JERR_VIRTUAL_BUG);   /* array of another library that is partly swapped out */
  if (p->first_undef_row < end_row) {
    JDIMENSION undef_row = p->first_undef_row;
    if (p->first_undef_row < start_row) { if (writable) ERREXIT(cinfo, JERR_BAD_VIRTUAL_ACCESS); undef_row = start_row; }
    if (writable) p->first_undef_row = end_row;
    if (p->pre_zero) {
      const size_t bytes = (size_t)p->samplesperrow * sample_size(cinfo);
      for (; undef_row < end_row; undef_row++) memset(p->mem_buffer[undef_row - p->cur_start_row], 0, bytes);
    } else if (!writable) ERREXIT(cinfo, JERR_BAD_VIRTUAL_ACCESS);
  }
  if (writable) p->dirty = TRUE;
  return p->mem_buffer + (start_row - p->cur_start_row);
}

static JBLOCKARRAY api_access_virt_barray(j_common_ptr cinfo, jvirt_barray_ptr p, JDIMENSION start_row, JDIMENSION num_rows, boolean writable)
{ /* access_virt_barray jmemmgr.c:936-1018, in-memory case */
  const JDIMENSION end_row = start_row + num_rows;
  if (end_row > p->rows_in_array || num_rows > p->maxaccess || p->mem_buffer == NULL) ERREXIT(cinfo, JERR_BAD_VIRTUAL_ACCESS);
  if (start_row < p->cur_start_row || end_row > p->cur_start_row + p->rows_in_mem) ERREXIT(cinfo, JERR_VIRTUAL_BUG);
  if (p->first_undef_row < end_row) {
    JDIMENSION undef_row = p->first_undef_row;
    if (p->first_undef_row < start_row) { if (writable) ERREXIT(cinfo, JERR_BAD_VIRTUAL_ACCESS); undef_row = start_row; }
    if (writable) p->first_undef_row = end_row;
    if (p->pre_zero) {
      const size_t bytes = (size_t)p->blocksperrow * sizeof(JBLOCK);
      for (; undef_row < end_row; undef_row++) memset(p->mem_buffer[undef_row - p->cur_start_row], 0, bytes);
    } else if (!writable) ERREXIT(cinfo, JERR_BAD_VIRTUAL_ACCESS);
  }
  if (writable) p->dirty = TRUE;
  return p->mem_buffer + (start_row - p->cur_start_row);
}

static void api_free_pool(j_common_ptr cinfo, int pool_id)
{ /* free_pool jmemmgr.c:1024-1093 */
  api_mem *m = (api_mem *)cinfo->mem;
  api_chunk *c, *n;
  if (pool_id < 0 || pool_id >= JPOOL_NUMPOOLS) ERREXIT1(cinfo, JERR_BAD_POOL_ID, pool_id);
  if (pool_id == JPOOL_IMAGE) { m->virt_sarray_list = NULL; m->virt_barray_list = NULL; }
  for (c = m->pools[pool_id]; c; c = n) { n = c->next; free(c); }
  m->pools[pool_id] = NULL;
}

static void api_self_destruct(j_common_ptr cinfo)
{ /* self_destruct jmemmgr.c:1100-1118 */
  int pool;
  for (pool = JPOOL_NUMPOOLS - 1; pool >= JPOOL_PERMANENT; pool--) api_free_pool(cinfo, pool);
  free(cinfo->mem);
  cinfo->mem = NULL;
}

void jinit_memory_mgr(j_common_ptr cinfo)
{ /* jmemmgr.c:1125-1219 */
  api_mem *m;
  cinfo->mem = NULL;
  m = (api_mem *)calloc(1, sizeof(*m));
  if (m == NULL) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 0);
  m->pub.alloc_small = api_alloc_small;
  m->pub.alloc_large = api_alloc_large;
  m->pub.alloc_sarray = api_alloc_sarray;
  m->pub.alloc_barray = api_alloc_barray;
  m->pub.request_virt_sarray = api_request_virt_sarray;
  m->pub.request_virt_barray = api_request_virt_barray;
  m->pub.realize_virt_arrays = api_realize_virt_arrays;
  m->pub.access_virt_sarray = api_access_virt_sarray;
  m->pub.access_virt_barray = api_access_virt_barray;
  m->pub.free_pool = api_free_pool;
  m->pub.self_destruct = api_self_destruct;
  m->pub.max_alloc_chunk = 1000000000L;
  m->pub.max_memory_to_use = 0;
  {
    const char *v = getenv("JPEGMEM");   /* jmemmgr.c:1203-1218: honoured as a number, never acted on (no backing store) */
    if (v != NULL) { long mx = 0; char ch = 'x'; if (sscanf(v, "%ld%c", &mx, &ch) > 0) { if (ch == 'm' || ch == 'M') mx *= 1000L; m->pub.max_memory_to_use = mx * 1000L; } }
  }
  cinfo->mem = &m->pub;
}

/* =====================================================================================================================
 * object life cycle -- jcapimin.c:34-135, jcomapi.c
 * ===================================================================================================================== */
void jpeg_abort(j_common_ptr cinfo)
{ /* jcomapi.c:30-55 */
  mjh_shim_drop(cinfo);
  if (cinfo->mem == NULL) return;
  (*cinfo->mem->free_pool) (cinfo, JPOOL_IMAGE);
  if (cinfo->is_decompressor) { cinfo->global_state = DSTATE_START; ((j_decompress_ptr)cinfo)->marker_list = NULL; }
  else cinfo->global_state = CSTATE_START;
}

void jpeg_destroy(j_common_ptr cinfo)
{ /* jcomapi.c:70-80 */
  mjh_shim_drop(cinfo);
  if (cinfo->mem != NULL) (*cinfo->mem->self_destruct) (cinfo);
  cinfo->mem = NULL;
  cinfo->global_state = 0;
}

void jpeg_abort_compress(j_compress_ptr cinfo) { jpeg_abort((j_common_ptr)cinfo); }       /* jcapimin.c:129-133 */
void jpeg_destroy_compress(j_compress_ptr cinfo) { jpeg_destroy((j_common_ptr)cinfo); }   /* jcapimin.c:117-121 */

JQUANT_TBL *jpeg_alloc_quant_table(j_common_ptr cinfo)
{ /* jcomapi.c:88-96 */
  JQUANT_TBL *t = (JQUANT_TBL *)(*cinfo->mem->alloc_small) (cinfo, JPOOL_PERMANENT, sizeof(JQUANT_TBL));
  t->sent_table = FALSE;
  return t;
}

JHUFF_TBL *jpeg_alloc_huff_table(j_common_ptr cinfo)
{ /* jcomapi.c:99-107 */
  JHUFF_TBL *t = (JHUFF_TBL *)(*cinfo->mem->alloc_small) (cinfo, JPOOL_PERMANENT, sizeof(JHUFF_TBL));
  t->sent_table = FALSE;
  return t;
}

void jpeg_CreateCompress(j_compress_ptr cinfo, int version, size_t structsize)
{ /* jcapimin.c:34-110 */
  int i;
  cinfo->mem = NULL;
  if (version != JPEG_LIB_VERSION) ERREXIT2(cinfo, JERR_BAD_LIB_VERSION, JPEG_LIB_VERSION, version);
  if (structsize != sizeof(struct jpeg_compress_struct)) ERREXIT2(cinfo, JERR_BAD_STRUCT_SIZE, (int)sizeof(struct jpeg_compress_struct), (int)structsize);
  {
    struct jpeg_error_mgr *err = cinfo->err;
    void *client_data = cinfo->client_data;
    memset(cinfo, 0, sizeof(struct jpeg_compress_struct));
    cinfo->err = err;
    cinfo->client_data = client_data;
  }
  cinfo->is_decompressor = FALSE;
  jinit_memory_mgr((j_common_ptr)cinfo);
  for (i = 0; i < NUM_QUANT_TBLS; i++) {
    cinfo->quant_tbl_ptrs[i] = NULL;
#if JPEG_LIB_VERSION >= 70
    cinfo->q_scale_factor[i] = 100;
#endif
  }
  cinfo->input_gamma = 1.0;
  cinfo->data_precision = BITS_IN_JSAMPLE;
  cinfo->global_state = CSTATE_START;
  /* the extension parameters live in the master structure, so it exists from the start (jcapimin.c:99-109) */
  cinfo->master = (struct jpeg_comp_master *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_PERMANENT, sizeof(struct jpeg_comp_master));
  memset(cinfo->master, 0, sizeof(struct jpeg_comp_master));
  cinfo->master->compress_profile = JCP_MAX_COMPRESSION;
}

void jpeg_suppress_tables(j_compress_ptr cinfo, boolean suppress)
{ /* jcapimin.c:148-167 */
  int i;
  for (i = 0; i < NUM_QUANT_TBLS; i++) if (cinfo->quant_tbl_ptrs[i] != NULL) cinfo->quant_tbl_ptrs[i]->sent_table = suppress;
  for (i = 0; i < NUM_HUFF_TBLS; i++) {
    if (cinfo->dc_huff_tbl_ptrs[i] != NULL) cinfo->dc_huff_tbl_ptrs[i]->sent_table = suppress;
    if (cinfo->ac_huff_tbl_ptrs[i] != NULL) cinfo->ac_huff_tbl_ptrs[i]->sent_table = suppress;
  }
}

/* =====================================================================================================================
 * parameters -- jcparam.c, jcext.c
 * ===================================================================================================================== */
#define NEED_START(cinfo) do { if ((cinfo)->global_state != CSTATE_START) ERREXIT1(cinfo, JERR_BAD_STATE, (cinfo)->global_state); } while (0)

void jpeg_add_quant_table(j_compress_ptr cinfo, int which_tbl, const unsigned int *basic_table, int scale_factor, boolean force_baseline)
{ /* jcparam.c:30-68 */
  JQUANT_TBL **slot;
  int i;
  NEED_START(cinfo);
  if (which_tbl < 0 || which_tbl >= NUM_QUANT_TBLS) ERREXIT1(cinfo, JERR_DQT_INDEX, which_tbl);
  slot = &cinfo->quant_tbl_ptrs[which_tbl];
  if (*slot == NULL) *slot = jpeg_alloc_quant_table((j_common_ptr)cinfo);
  for (i = 0; i < DCTSIZE2; i++) {
    long v = ((long)basic_table[i] * scale_factor + 50L) / 100L;
    if (v <= 0L) v = 1L;
    if (v > 32767L) v = 32767L;
    if (force_baseline && v > 255L) v = 255L;
    (*slot)->quantval[i] = (UINT16)v;
  }
  (*slot)->sent_table = FALSE;
}

void jpeg_set_linear_quality(j_compress_ptr cinfo, int scale_factor, boolean force_baseline)
{ /* jcparam.c:311-325: the base tables of the current JINT_BASE_QUANT_TBL_IDX */
  const int idx = cinfo->master->quant_tbl_master_idx;
  jpeg_add_quant_table(cinfo, 0, mjh_base_luma[idx], scale_factor, force_baseline);
  jpeg_add_quant_table(cinfo, 1, mjh_base_chroma[idx], scale_factor, force_baseline);
}

float jpeg_float_quality_scaling(float quality)
{ /* jcparam.c:334-357 */
  if (quality <= 0.f) quality = 1.f;
  if (quality > 100.f) quality = 100.f;
  return quality < 50.f ? 5000.f / quality : 200.f - quality * 2.f;
}
int jpeg_quality_scaling(int quality) { return (int)jpeg_float_quality_scaling((float)quality); }   /* jcparam.c:328-332 */

void jpeg_set_quality(j_compress_ptr cinfo, int quality, boolean force_baseline)
{ /* jcparam.c:360-374 */
  jpeg_set_linear_quality(cinfo, jpeg_quality_scaling(quality), force_baseline);
}

static void install_std_table(j_compress_ptr cinfo, JHUFF_TBL **slot, int is_ac, int tblno)
{ /* add_huff_table jstdhuff.c:20-47 */
  const uint8_t *bits, *vals;
  int n;
  if (*slot == NULL) *slot = jpeg_alloc_huff_table((j_common_ptr)cinfo);   /* (jstdhuff.c:26-29: only a DEcompressor leaves an existing table alone -- a compressor gets the Annex K table back, unsent) */
  mjh_std_huffman_table(is_ac, tblno, &bits, &vals, &n);
  memcpy((*slot)->bits, bits, 17);
  memset((*slot)->huffval, 0, sizeof((*slot)->huffval));
  memcpy((*slot)->huffval, vals, (size_t)n);
  (*slot)->sent_table = FALSE;
}

void jpeg_set_colorspace(j_compress_ptr cinfo, J_COLOR_SPACE colorspace)
{ /* jcparam.c:571-650 */
  static const struct { int n; struct { int id, h, v, q, d, a; } c[4]; } T[] = {
    /* JCS_GRAYSCALE */ { 1, { { 1, 1, 1, 0, 0, 0 } } },
    /* JCS_RGB       */ { 3, { { 0x52, 1, 1, 0, 0, 0 }, { 0x47, 1, 1, 0, 0, 0 }, { 0x42, 1, 1, 0, 0, 0 } } },
    /* JCS_YCbCr     */ { 3, { { 1, 2, 2, 0, 0, 0 }, { 2, 1, 1, 1, 1, 1 }, { 3, 1, 1, 1, 1, 1 } } },
    /* JCS_CMYK      */ { 4, { { 0x43, 1, 1, 0, 0, 0 }, { 0x4D, 1, 1, 0, 0, 0 }, { 0x59, 1, 1, 0, 0, 0 }, { 0x4B, 1, 1, 0, 0, 0 } } },
    /* JCS_YCCK      */ { 4, { { 1, 2, 2, 0, 0, 0 }, { 2, 1, 1, 1, 1, 1 }, { 3, 1, 1, 1, 1, 1 }, { 4, 2, 2, 0, 0, 0 } } },
  };
  int ci, t = -1;
  NEED_START(cinfo);
  cinfo->jpeg_color_space = colorspace;
  cinfo->write_JFIF_header = FALSE;
  cinfo->write_Adobe_marker = FALSE;
  switch (colorspace) {
  case JCS_GRAYSCALE: cinfo->write_JFIF_header = TRUE; t = 0; break;
  case JCS_RGB: cinfo->write_Adobe_marker = TRUE; t = 1; break;
  case JCS_YCbCr: cinfo->write_JFIF_header = TRUE; t = 2; break;
  case JCS_CMYK: cinfo->write_Adobe_marker = TRUE; t = 3; break;
  case JCS_YCCK: cinfo->write_Adobe_marker = TRUE; t = 4; break;
  case JCS_UNKNOWN:
    cinfo->num_components = cinfo->input_components;
    if (cinfo->num_components < 1 || cinfo->num_components > MAX_COMPONENTS) ERREXIT2(cinfo, JERR_COMPONENT_COUNT, cinfo->num_components, MAX_COMPONENTS);
    for (ci = 0; ci < cinfo->num_components; ci++) {
      jpeg_component_info *c = &cinfo->comp_info[ci];
      c->component_id = ci; c->h_samp_factor = c->v_samp_factor = 1; c->quant_tbl_no = c->dc_tbl_no = c->ac_tbl_no = 0;
    }
    return;
  default:
    ERREXIT(cinfo, JERR_BAD_J_COLORSPACE);
  }
  cinfo->num_components = T[t].n;
  for (ci = 0; ci < T[t].n; ci++) {
    jpeg_component_info *c = &cinfo->comp_info[ci];
    c->component_id = T[t].c[ci].id; c->h_samp_factor = T[t].c[ci].h; c->v_samp_factor = T[t].c[ci].v;
    c->quant_tbl_no = T[t].c[ci].q; c->dc_tbl_no = T[t].c[ci].d; c->ac_tbl_no = T[t].c[ci].a;
  }
}

void jpeg_default_colorspace(j_compress_ptr cinfo)
{ /* jcparam.c:525-565 */
  switch (cinfo->in_color_space) {
  case JCS_GRAYSCALE: jpeg_set_colorspace(cinfo, JCS_GRAYSCALE); break;
  case JCS_RGB: case JCS_EXT_RGB: case JCS_EXT_RGBX: case JCS_EXT_BGR: case JCS_EXT_BGRX: case JCS_EXT_XBGR: case JCS_EXT_XRGB:
  case JCS_EXT_RGBA: case JCS_EXT_BGRA: case JCS_EXT_ABGR: case JCS_EXT_ARGB:
    jpeg_set_colorspace(cinfo, cinfo->master->lossless ? JCS_RGB : JCS_YCbCr); break;
  case JCS_YCbCr: jpeg_set_colorspace(cinfo, JCS_YCbCr); break;
  case JCS_CMYK: jpeg_set_colorspace(cinfo, JCS_CMYK); break;
  case JCS_YCCK: jpeg_set_colorspace(cinfo, JCS_YCCK); break;
  case JCS_UNKNOWN: jpeg_set_colorspace(cinfo, JCS_UNKNOWN); break;
  default: ERREXIT(cinfo, JERR_BAD_IN_COLORSPACE);
  }
}

/* ---- progressive scripts: jcparam.c:652-1004 ---- */
static jpeg_scan_info *one_scan(jpeg_scan_info *s, int ci, int ncomps, int Ss, int Se, int Ah, int Al)
{ /* fill_a_scan / fill_a_scan_pair / an interleaved DC scan: components ci .. ci+ncomps-1 */
  int k;
  s->comps_in_scan = ncomps;
  for (k = 0; k < ncomps; k++) s->component_index[k] = ci + k;
  s->Ss = Ss; s->Se = Se; s->Ah = Ah; s->Al = Al;
  return s + 1;
}
static jpeg_scan_info *each_comp(jpeg_scan_info *s, int ncomps, int Ss, int Se, int Ah, int Al)
{ /* fill_scans */
  int ci;
  for (ci = 0; ci < ncomps; ci++) s = one_scan(s, ci, 1, Ss, Se, Ah, Al);
  return s;
}
static jpeg_scan_info *dc_scans(jpeg_scan_info *s, int ncomps, int Ah, int Al)
{ /* fill_dc_scans: interleaved if it fits */
  return ncomps <= MAX_COMPS_IN_SCAN ? one_scan(s, 0, ncomps, 0, 0, Ah, Al) : each_comp(s, ncomps, 0, 0, Ah, Al);
}

static jpeg_scan_info *script_space(j_compress_ptr cinfo, int nscans, int floor_size)
{ /* permanent-pool script buffer, reused between calls (jcparam.c:766-773, :909-915) */
  if (cinfo->script_space == NULL || cinfo->script_space_size < nscans) {
    cinfo->script_space_size = nscans > floor_size ? nscans : floor_size;
    cinfo->script_space = (jpeg_scan_info *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_PERMANENT,
                                                                       (size_t)cinfo->script_space_size * sizeof(jpeg_scan_info));
  }
  cinfo->scan_info = cinfo->script_space;
  cinfo->num_scans = nscans;
  return cinfo->script_space;
}

static boolean search_progression(j_compress_ptr cinfo)
{ /* jpeg_search_progression jcparam.c:733-852: the candidate scans of the scan search */
  static const int fs[5] = { 2, 8, 5, 12, 18 };
  const int ncomps = cinfo->num_components;
  struct jpeg_comp_master *m = cinfo->master;
  jpeg_scan_info *s;
  int Al, i, nscans;
  NEED_START(cinfo);
  if (ncomps == 3 && cinfo->jpeg_color_space == JCS_YCbCr) nscans = 64;
  else if (ncomps == 1) nscans = 23;
  else { m->num_scans_luma = 0; return FALSE; }
  s = script_space(cinfo, nscans, 64);
  m->Al_max_luma = 3; m->num_scans_luma_dc = 1; m->num_frequency_splits = 5;
  m->num_scans_luma = m->num_scans_luma_dc + (3 * m->Al_max_luma + 2) + (2 * m->num_frequency_splits + 1);
  s = m->dc_scan_opt_mode == 0 ? dc_scans(s, ncomps, 0, 0) : dc_scans(s, 1, 0, 0);
  s = one_scan(s, 0, 1, 1, 8, 0, 0); s = one_scan(s, 0, 1, 9, 63, 0, 0);
  for (Al = 0; Al < m->Al_max_luma; Al++) {
    s = one_scan(s, 0, 1, 1, 63, Al + 1, Al); s = one_scan(s, 0, 1, 1, 8, 0, Al + 1); s = one_scan(s, 0, 1, 9, 63, 0, Al + 1);
  }
  s = one_scan(s, 0, 1, 1, 63, 0, 0);
  for (i = 0; i < m->num_frequency_splits; i++) { s = one_scan(s, 0, 1, 1, fs[i], 0, 0); s = one_scan(s, 0, 1, fs[i] + 1, 63, 0, 0); }
  if (ncomps == 1) { m->Al_max_chroma = 0; m->num_scans_chroma_dc = 0; return TRUE; }
  m->Al_max_chroma = 2; m->num_scans_chroma_dc = 3;
  s = one_scan(s, 1, 2, 0, 0, 0, 0);                                     /* chroma DC combined, then separate */
  s = one_scan(s, 1, 1, 0, 0, 0, 0); s = one_scan(s, 2, 1, 0, 0, 0, 0);
  s = one_scan(s, 1, 1, 1, 8, 0, 0); s = one_scan(s, 1, 1, 9, 63, 0, 0);
  s = one_scan(s, 2, 1, 1, 8, 0, 0); s = one_scan(s, 2, 1, 9, 63, 0, 0);
  for (Al = 0; Al < m->Al_max_chroma; Al++) {
    s = one_scan(s, 1, 1, 1, 63, Al + 1, Al); s = one_scan(s, 2, 1, 1, 63, Al + 1, Al);
    s = one_scan(s, 1, 1, 1, 8, 0, Al + 1); s = one_scan(s, 1, 1, 9, 63, 0, Al + 1);
    s = one_scan(s, 2, 1, 1, 8, 0, Al + 1); s = one_scan(s, 2, 1, 9, 63, 0, Al + 1);
  }
  s = one_scan(s, 1, 1, 1, 63, 0, 0); s = one_scan(s, 2, 1, 1, 63, 0, 0);
  for (i = 0; i < m->num_frequency_splits; i++) {
    s = one_scan(s, 1, 1, 1, fs[i], 0, 0); s = one_scan(s, 1, 1, fs[i] + 1, 63, 0, 0);
    s = one_scan(s, 2, 1, 1, fs[i], 0, 0); s = one_scan(s, 2, 1, fs[i] + 1, 63, 0, 0);
  }
  return TRUE;
}

void jpeg_simple_progression(j_compress_ptr cinfo)
{ /* jcparam.c:859-1004 */
  struct jpeg_comp_master *m = cinfo->master;
  const boolean maxc = m->compress_profile == JCP_MAX_COMPRESSION;
  jpeg_scan_info *s;
  int ncomps, nscans;
  if (m->optimize_scans && search_progression(cinfo)) return;
  NEED_START(cinfo);
  if (m->lossless) { m->lossless = FALSE; jpeg_default_colorspace(cinfo); }
  ncomps = cinfo->num_components;
  if (ncomps == 3 && cinfo->jpeg_color_space == JCS_YCbCr) {
    nscans = !maxc ? 10 : m->dc_scan_opt_mode == 0 ? 9 : m->dc_scan_opt_mode == 1 ? 11 : 10;
    s = script_space(cinfo, nscans, 10);
    if (maxc) {
      if (m->dc_scan_opt_mode == 0) s = dc_scans(s, ncomps, 0, 0);
      else if (m->dc_scan_opt_mode == 1) s = each_comp(s, 3, 0, 0, 0, 0);
      else { s = dc_scans(s, 1, 0, 0); s = one_scan(s, 1, 2, 0, 0, 0, 0); }
      s = one_scan(s, 0, 1, 1, 8, 0, 2); s = one_scan(s, 1, 1, 1, 8, 0, 0); s = one_scan(s, 2, 1, 1, 8, 0, 0);
      s = one_scan(s, 0, 1, 9, 63, 0, 2);
      s = one_scan(s, 0, 1, 1, 63, 2, 1); s = one_scan(s, 0, 1, 1, 63, 1, 0);
      s = one_scan(s, 1, 1, 9, 63, 0, 0); s = one_scan(s, 2, 1, 9, 63, 0, 0);
    } else {
      s = dc_scans(s, ncomps, 0, 1);
      s = one_scan(s, 0, 1, 1, 5, 0, 2); s = one_scan(s, 2, 1, 1, 63, 0, 1); s = one_scan(s, 1, 1, 1, 63, 0, 1);
      s = one_scan(s, 0, 1, 6, 63, 0, 2); s = one_scan(s, 0, 1, 1, 63, 2, 1);
      s = dc_scans(s, ncomps, 1, 0);
      s = one_scan(s, 2, 1, 1, 63, 1, 0); s = one_scan(s, 1, 1, 1, 63, 1, 0); s = one_scan(s, 0, 1, 1, 63, 1, 0);
    }
  } else {   /* all-purpose script for the other colour spaces */
    const int dcn = ncomps > MAX_COMPS_IN_SCAN ? ncomps : 1;
    nscans = maxc ? dcn + 4 * ncomps : 2 * dcn + 4 * ncomps;
    s = script_space(cinfo, nscans, 10);
    if (maxc) {
      s = dc_scans(s, ncomps, 0, 0);
      s = each_comp(s, ncomps, 1, 8, 0, 2); s = each_comp(s, ncomps, 9, 63, 0, 2);
      s = each_comp(s, ncomps, 1, 63, 2, 1); s = each_comp(s, ncomps, 1, 63, 1, 0);
    } else {
      s = dc_scans(s, ncomps, 0, 1);
      s = each_comp(s, ncomps, 1, 5, 0, 2); s = each_comp(s, ncomps, 6, 63, 0, 2);
      s = each_comp(s, ncomps, 1, 63, 2, 1);
      s = dc_scans(s, ncomps, 1, 0);
      s = each_comp(s, ncomps, 1, 63, 1, 0);
    }
  }
  (void)s;
}

void jpeg_set_defaults(j_compress_ptr cinfo)
{ /* jcparam.c:386-519, in its order (the quality tables are made BEFORE the profile's base-table index is set) */
  struct jpeg_comp_master *m = cinfo->master;
  const boolean maxc = m->compress_profile == JCP_MAX_COMPRESSION;
  int i;
  NEED_START(cinfo);
  if (cinfo->comp_info == NULL)
    cinfo->comp_info = (jpeg_component_info *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_PERMANENT, MAX_COMPONENTS * sizeof(jpeg_component_info));
#if JPEG_LIB_VERSION >= 70
  cinfo->scale_num = 1; cinfo->scale_denom = 1;
#endif
  jpeg_set_quality(cinfo, 75, TRUE);
  install_std_table(cinfo, &cinfo->dc_huff_tbl_ptrs[0], 0, 0);
  install_std_table(cinfo, &cinfo->ac_huff_tbl_ptrs[0], 1, 0);
  install_std_table(cinfo, &cinfo->dc_huff_tbl_ptrs[1], 0, 1);
  install_std_table(cinfo, &cinfo->ac_huff_tbl_ptrs[1], 1, 1);
  for (i = 0; i < NUM_ARITH_TBLS; i++) { cinfo->arith_dc_L[i] = 0; cinfo->arith_dc_U[i] = 1; cinfo->arith_ac_K[i] = 5; }
  cinfo->scan_info = NULL;
  cinfo->num_scans = 0;
  m->lossless = FALSE;
  cinfo->raw_data_in = FALSE;
  cinfo->arith_code = FALSE;
  cinfo->optimize_coding = maxc;
  if (cinfo->data_precision == 12) cinfo->optimize_coding = TRUE;
  cinfo->CCIR601_sampling = FALSE;
#if JPEG_LIB_VERSION >= 70
  cinfo->do_fancy_downsampling = TRUE;
#endif
  m->overshoot_deringing = maxc;
  cinfo->smoothing_factor = 0;
  cinfo->dct_method = JDCT_DEFAULT;
  cinfo->restart_interval = 0;
  cinfo->restart_in_rows = 0;
  cinfo->JFIF_major_version = 1;
  cinfo->JFIF_minor_version = 1;
  cinfo->density_unit = 0;
  cinfo->X_density = 1;
  cinfo->Y_density = 1;
  jpeg_default_colorspace(cinfo);
  m->dc_scan_opt_mode = 0;
  if (maxc) { m->optimize_scans = TRUE; jpeg_simple_progression(cinfo); }
  else m->optimize_scans = FALSE;
  m->trellis_quant = maxc;
  m->lambda_log_scale1 = 14.75f;
  m->lambda_log_scale2 = 16.5f;
  m->quant_tbl_master_idx = maxc ? 3 : 0;
  m->use_lambda_weight_tbl = TRUE;
  m->use_scans_in_trellis = FALSE;
  m->trellis_freq_split = 8;
  m->trellis_num_loops = 1;
  m->trellis_q_opt = FALSE;
  m->trellis_quant_dc = TRUE;
  m->trellis_delta_dc_weight = 0.0f;
}

void jpeg_enable_lossless(j_compress_ptr cinfo, int predictor_selection_value, int point_transform)
{ /* jcparam.c:1013-1040: accepted, refused at jpeg_start_compress (the GPU path has no lossless coder) */
  NEED_START(cinfo);
  cinfo->master->lossless = TRUE;
  cinfo->Ss = predictor_selection_value; cinfo->Se = 0; cinfo->Ah = 0; cinfo->Al = point_transform;
  if (cinfo->Ss < 1 || cinfo->Ss > 7 || cinfo->Al < 0 || cinfo->Al >= cinfo->data_precision)
    ERREXIT4(cinfo, JERR_BAD_PROGRESSION, cinfo->Ss, cinfo->Se, cinfo->Ah, cinfo->Al);
}

/* ---- extension parameters: jcext.c ---- */
static boolean *bool_param(j_compress_ptr cinfo, J_BOOLEAN_PARAM param)
{
  struct jpeg_comp_master *m = cinfo->master;
  switch (param) {
  case JBOOLEAN_OPTIMIZE_SCANS: return &m->optimize_scans;
  case JBOOLEAN_TRELLIS_QUANT: return &m->trellis_quant;
  case JBOOLEAN_TRELLIS_QUANT_DC: return &m->trellis_quant_dc;
  case JBOOLEAN_TRELLIS_EOB_OPT: return &m->trellis_eob_opt;
  case JBOOLEAN_USE_LAMBDA_WEIGHT_TBL: return &m->use_lambda_weight_tbl;
  case JBOOLEAN_USE_SCANS_IN_TRELLIS: return &m->use_scans_in_trellis;
  case JBOOLEAN_TRELLIS_Q_OPT: return &m->trellis_q_opt;
  case JBOOLEAN_OVERSHOOT_DERINGING: return &m->overshoot_deringing;
  }
  return NULL;
}
boolean jpeg_c_bool_param_supported(const j_compress_ptr cinfo, J_BOOLEAN_PARAM param) { return bool_param(cinfo, param) != NULL; }
void jpeg_c_set_bool_param(j_compress_ptr cinfo, J_BOOLEAN_PARAM param, boolean value)
{
  boolean *b = bool_param(cinfo, param);
  if (b == NULL) ERREXIT(cinfo, JERR_BAD_PARAM);
  *b = value;
}
boolean jpeg_c_get_bool_param(const j_compress_ptr cinfo, J_BOOLEAN_PARAM param)
{
  boolean *b = bool_param(cinfo, param);
  if (b == NULL) ERREXIT(cinfo, JERR_BAD_PARAM);
  return b ? *b : FALSE;
}

static float *float_param(j_compress_ptr cinfo, J_FLOAT_PARAM param)
{
  switch (param) {
  case JFLOAT_LAMBDA_LOG_SCALE1: return &cinfo->master->lambda_log_scale1;
  case JFLOAT_LAMBDA_LOG_SCALE2: return &cinfo->master->lambda_log_scale2;
  case JFLOAT_TRELLIS_DELTA_DC_WEIGHT: return &cinfo->master->trellis_delta_dc_weight;
  }
  return NULL;
}
boolean jpeg_c_float_param_supported(const j_compress_ptr cinfo, J_FLOAT_PARAM param) { return float_param(cinfo, param) != NULL; }
void jpeg_c_set_float_param(j_compress_ptr cinfo, J_FLOAT_PARAM param, float value)
{
  float *f = float_param(cinfo, param);
  if (f == NULL) ERREXIT(cinfo, JERR_BAD_PARAM);
  *f = value;
}
float jpeg_c_get_float_param(const j_compress_ptr cinfo, J_FLOAT_PARAM param)
{
  float *f = float_param(cinfo, param);
  if (f == NULL) ERREXIT(cinfo, JERR_BAD_PARAM);
  return f ? *f : -1.0f;
}

static int *int_param(j_compress_ptr cinfo, J_INT_PARAM param)
{
  struct jpeg_comp_master *m = cinfo->master;
  switch (param) {
  case JINT_COMPRESS_PROFILE: return &m->compress_profile;
  case JINT_TRELLIS_FREQ_SPLIT: return &m->trellis_freq_split;
  case JINT_TRELLIS_NUM_LOOPS: return &m->trellis_num_loops;
  case JINT_BASE_QUANT_TBL_IDX: return &m->quant_tbl_master_idx;
  case JINT_DC_SCAN_OPT_MODE: return &m->dc_scan_opt_mode;
  }
  return NULL;
}
boolean jpeg_c_int_param_supported(const j_compress_ptr cinfo, J_INT_PARAM param) { return int_param(cinfo, param) != NULL; }
void jpeg_c_set_int_param(j_compress_ptr cinfo, J_INT_PARAM param, int value)
{ /* jcext.c:160-190: the profile takes its two GUIDs only, the base-table index silently ignores values outside 0..8 */
  int *v = int_param(cinfo, param);
  if (v == NULL) ERREXIT(cinfo, JERR_BAD_PARAM);
  if (param == JINT_COMPRESS_PROFILE && value != JCP_MAX_COMPRESSION && value != JCP_FASTEST) ERREXIT(cinfo, JERR_BAD_PARAM_VALUE);
  if (param == JINT_BASE_QUANT_TBL_IDX && (value < 0 || value > 8)) return;
  *v = value;
}
int jpeg_c_get_int_param(const j_compress_ptr cinfo, J_INT_PARAM param)
{
  int *v = int_param(cinfo, param);
  if (v == NULL) ERREXIT(cinfo, JERR_BAD_PARAM);
  return v ? *v : -1;
}

/* =====================================================================================================================
 * destination managers -- jdatadst.c
 * ===================================================================================================================== */
#define OUTPUT_BUF_SIZE 4096

typedef struct { struct jpeg_destination_mgr pub; FILE *outfile; JOCTET *buffer; } stdio_dest;

static void stdio_init(j_compress_ptr cinfo)
{
  stdio_dest *d = (stdio_dest *)cinfo->dest;
  d->buffer = (JOCTET *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_IMAGE, OUTPUT_BUF_SIZE);
  d->pub.next_output_byte = d->buffer;
  d->pub.free_in_buffer = OUTPUT_BUF_SIZE;
}
static boolean stdio_empty(j_compress_ptr cinfo)
{
  stdio_dest *d = (stdio_dest *)cinfo->dest;
  if (fwrite(d->buffer, 1, OUTPUT_BUF_SIZE, d->outfile) != (size_t)OUTPUT_BUF_SIZE) ERREXIT(cinfo, JERR_FILE_WRITE);
  d->pub.next_output_byte = d->buffer;
  d->pub.free_in_buffer = OUTPUT_BUF_SIZE;
  return TRUE;
}
static void stdio_term(j_compress_ptr cinfo)
{
  stdio_dest *d = (stdio_dest *)cinfo->dest;
  const size_t n = OUTPUT_BUF_SIZE - d->pub.free_in_buffer;
  if (n > 0 && fwrite(d->buffer, 1, n, d->outfile) != n) ERREXIT(cinfo, JERR_FILE_WRITE);
  fflush(d->outfile);
  if (ferror(d->outfile)) ERREXIT(cinfo, JERR_FILE_WRITE);
}

void jpeg_stdio_dest(j_compress_ptr cinfo, FILE *outfile)
{ /* jdatadst.c:194-222 */
  stdio_dest *d;
  if (cinfo->dest == NULL)
    cinfo->dest = (struct jpeg_destination_mgr *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_PERMANENT, sizeof(stdio_dest));
  else if (cinfo->dest->init_destination != stdio_init) ERREXIT(cinfo, JERR_BUFFER_SIZE);   /* a destination of another kind is in place */
  d = (stdio_dest *)cinfo->dest;
  d->pub.init_destination = stdio_init;
  d->pub.empty_output_buffer = stdio_empty;
  d->pub.term_destination = stdio_term;
  d->outfile = outfile;
}

typedef struct {
  struct jpeg_destination_mgr pub;
  unsigned char **outbuffer;
  unsigned long *outsize;
  unsigned char *newbuffer;   /* buffer this manager allocated (the caller frees the final one) */
  JOCTET *buffer;
  size_t bufsize;
} mem_dest;

static void mem_init(j_compress_ptr cinfo) { (void)cinfo; }
static boolean mem_empty(j_compress_ptr cinfo)
{ /* empty_mem_output_buffer jdatadst.c:118-146: double the buffer */
  mem_dest *d = (mem_dest *)cinfo->dest;
  const size_t nextsize = d->bufsize * 2;
  JOCTET *next = (JOCTET *)malloc(nextsize);
  if (next == NULL) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 10);
  memcpy(next, d->buffer, d->bufsize);
  free(d->newbuffer);
  d->newbuffer = next;
  d->pub.next_output_byte = next + d->bufsize;
  d->pub.free_in_buffer = d->bufsize;
  d->buffer = next;
  d->bufsize = nextsize;
  return TRUE;
}
static void mem_term(j_compress_ptr cinfo)
{
  mem_dest *d = (mem_dest *)cinfo->dest;
  *d->outbuffer = d->buffer;
  *d->outsize = (unsigned long)(d->bufsize - d->pub.free_in_buffer);
}

void jpeg_mem_dest(j_compress_ptr cinfo, unsigned char **outbuffer, unsigned long *outsize)
{ /* jdatadst.c:238-285 */
  mem_dest *d;
  if (outbuffer == NULL || outsize == NULL) ERREXIT(cinfo, JERR_BUFFER_SIZE);
  if (cinfo->dest == NULL)
    cinfo->dest = (struct jpeg_destination_mgr *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_PERMANENT, sizeof(mem_dest));
  else if (cinfo->dest->init_destination != mem_init) ERREXIT(cinfo, JERR_BUFFER_SIZE);
  d = (mem_dest *)cinfo->dest;
  d->pub.init_destination = mem_init;
  d->pub.empty_output_buffer = mem_empty;
  d->pub.term_destination = mem_term;
  d->outbuffer = outbuffer;
  d->outsize = outsize;
  d->newbuffer = NULL;
  if (*outbuffer == NULL || *outsize == 0) {
    d->newbuffer = *outbuffer = (unsigned char *)malloc(OUTPUT_BUF_SIZE);
    if (d->newbuffer == NULL) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 10);
    *outsize = OUTPUT_BUF_SIZE;
  }
  d->pub.next_output_byte = d->buffer = *outbuffer;
  d->pub.free_in_buffer = d->bufsize = *outsize;
}

/* =====================================================================================================================
 * markers written by the application -- jcapimin.c:235-300, jcicc.c
 * ===================================================================================================================== */
static void need_marker_state(j_compress_ptr cinfo)
{ /* only between jpeg_start_compress and the first scanline / jpeg_write_coefficients and finish */
  if (cinfo->next_scanline != 0 || (cinfo->global_state != CSTATE_SCANNING && cinfo->global_state != CSTATE_RAW_OK && cinfo->global_state != CSTATE_WRCOEFS))
    ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state);
}

void jpeg_write_marker(j_compress_ptr cinfo, int marker, const JOCTET *dataptr, unsigned int datalen)
{ /* jcapimin.c:243-262 */
  need_marker_state(cinfo);
  (*cinfo->marker->write_marker_header) (cinfo, marker, datalen);
  while (datalen--) { (*cinfo->marker->write_marker_byte) (cinfo, *dataptr); dataptr++; }
}

void jpeg_write_m_header(j_compress_ptr cinfo, int marker, unsigned int datalen)
{ /* jcapimin.c:266-276 */
  need_marker_state(cinfo);
  (*cinfo->marker->write_marker_header) (cinfo, marker, datalen);
}

void jpeg_write_m_byte(j_compress_ptr cinfo, int val) { (*cinfo->marker->write_marker_byte) (cinfo, val); }   /* jcapimin.c:279-283 */

void jpeg_write_icc_profile(j_compress_ptr cinfo, const JOCTET *icc_data_ptr, unsigned int icc_data_len)
{ /* jcicc.c:49-105: APP2 "ICC_PROFILE\0" segments of at most 65519 profile bytes, numbered from 1 */
  static const char tag[12] = "ICC_PROFILE";
  const unsigned int per = 65533u - 14u;
  unsigned int num, cur = 1;
  if (icc_data_ptr == NULL || icc_data_len == 0) ERREXIT(cinfo, JERR_BUFFER_SIZE);
  need_marker_state(cinfo);
  num = (icc_data_len + per - 1) / per;
  while (icc_data_len > 0) {
    unsigned int n = icc_data_len > per ? per : icc_data_len, i;
    jpeg_write_m_header(cinfo, JPEG_APP0 + 2, n + 14u);
    for (i = 0; i < 12; i++) jpeg_write_m_byte(cinfo, tag[i]);
    jpeg_write_m_byte(cinfo, (int)cur);
    jpeg_write_m_byte(cinfo, (int)num);
    for (i = 0; i < n; i++) jpeg_write_m_byte(cinfo, icc_data_ptr[i]);
    icc_data_ptr += n; icc_data_len -= n; cur++;
  }
}

static void dest_byte(j_compress_ptr cinfo, int v)
{
  struct jpeg_destination_mgr *dest = cinfo->dest;
  *(dest->next_output_byte)++ = (JOCTET)v;
  if (--dest->free_in_buffer == 0) if (!(*dest->empty_output_buffer) (cinfo)) ERREXIT(cinfo, JERR_CANT_SUSPEND);
}

void jpeg_write_tables(j_compress_ptr cinfo)
{ /* jcapimin.c:300-328 + write_tables_only jcmarker.c:805-831: SOI, the unsent DQT / DHT tables, EOI */
  int i, k;
  NEED_START(cinfo);
  (*cinfo->err->reset_error_mgr) ((j_common_ptr)cinfo);
  (*cinfo->dest->init_destination) (cinfo);
  dest_byte(cinfo, 0xFF); dest_byte(cinfo, 0xD8);
  for (i = 0; i < NUM_QUANT_TBLS; i++) {
    JQUANT_TBL *q = cinfo->quant_tbl_ptrs[i];
    int prec = 0;
    if (q == NULL || q->sent_table) continue;
    for (k = 0; k < DCTSIZE2; k++) if (q->quantval[k] > 255) prec = 1;
    dest_byte(cinfo, 0xFF); dest_byte(cinfo, 0xDB);
    k = prec ? DCTSIZE2 * 2 + 1 + 2 : DCTSIZE2 + 1 + 2;
    dest_byte(cinfo, k >> 8); dest_byte(cinfo, k & 0xFF);
    dest_byte(cinfo, i + (prec << 4));
    for (k = 0; k < DCTSIZE2; k++) {
      const unsigned int v = q->quantval[jpeg_natural_order[k]];
      if (prec) dest_byte(cinfo, (int)(v >> 8));
      dest_byte(cinfo, (int)(v & 0xFF));
    }
    q->sent_table = TRUE;
  }
  if (!cinfo->arith_code)
    for (i = 0; i < 2 * NUM_HUFF_TBLS; i++) {   /* DC 0, AC 0, DC 1, AC 1 ... as write_tables_only orders them */
      JHUFF_TBL *h = (i & 1) ? cinfo->ac_huff_tbl_ptrs[i >> 1] : cinfo->dc_huff_tbl_ptrs[i >> 1];
      int n = 0;
      if (h == NULL || h->sent_table) continue;
      for (k = 1; k <= 16; k++) n += h->bits[k];
      dest_byte(cinfo, 0xFF); dest_byte(cinfo, 0xC4);
      dest_byte(cinfo, (n + 2 + 1 + 16) >> 8); dest_byte(cinfo, (n + 2 + 1 + 16) & 0xFF);
      dest_byte(cinfo, (i >> 1) + ((i & 1) ? 0x10 : 0));
      for (k = 1; k <= 16; k++) dest_byte(cinfo, h->bits[k]);
      for (k = 0; k < n; k++) dest_byte(cinfo, h->huffval[k]);
      h->sent_table = TRUE;
    }
  dest_byte(cinfo, 0xFF); dest_byte(cinfo, 0xD9);
  (*cinfo->dest->term_destination) (cinfo);
}

/* =====================================================================================================================
 * transcoding parameters -- jctrans.c:70-171
 * ===================================================================================================================== */
void jpeg_copy_critical_parameters(const j_decompress_ptr srcinfo, j_compress_ptr dstinfo)
{
  int tblno, ci;
  if (srcinfo->master->lossless) ERREXIT(dstinfo, JERR_NOTIMPL);
  NEED_START(dstinfo);
  dstinfo->image_width = srcinfo->image_width;
  dstinfo->image_height = srcinfo->image_height;
  dstinfo->input_components = srcinfo->num_components;
  dstinfo->in_color_space = srcinfo->jpeg_color_space;
#if JPEG_LIB_VERSION >= 70
  dstinfo->jpeg_width = srcinfo->output_width;
  dstinfo->jpeg_height = srcinfo->output_height;
  dstinfo->min_DCT_h_scaled_size = srcinfo->min_DCT_h_scaled_size;
  dstinfo->min_DCT_v_scaled_size = srcinfo->min_DCT_v_scaled_size;
#endif
  jpeg_set_defaults(dstinfo);
  dstinfo->master->trellis_quant = FALSE;
  jpeg_set_colorspace(dstinfo, srcinfo->jpeg_color_space);
  dstinfo->data_precision = srcinfo->data_precision;
  dstinfo->CCIR601_sampling = srcinfo->CCIR601_sampling;
  for (tblno = 0; tblno < NUM_QUANT_TBLS; tblno++)
    if (srcinfo->quant_tbl_ptrs[tblno] != NULL) {
      JQUANT_TBL **slot = &dstinfo->quant_tbl_ptrs[tblno];
      if (*slot == NULL) *slot = jpeg_alloc_quant_table((j_common_ptr)dstinfo);
      memcpy((*slot)->quantval, srcinfo->quant_tbl_ptrs[tblno]->quantval, sizeof((*slot)->quantval));
      (*slot)->sent_table = FALSE;
    }
  dstinfo->num_components = srcinfo->num_components;
  if (dstinfo->num_components < 1 || dstinfo->num_components > MAX_COMPONENTS) ERREXIT2(dstinfo, JERR_COMPONENT_COUNT, dstinfo->num_components, MAX_COMPONENTS);
  for (ci = 0; ci < dstinfo->num_components; ci++) {
    const jpeg_component_info *in = &srcinfo->comp_info[ci];
    jpeg_component_info *out = &dstinfo->comp_info[ci];
    out->component_id = in->component_id;
    out->h_samp_factor = in->h_samp_factor;
    out->v_samp_factor = in->v_samp_factor;
    out->quant_tbl_no = tblno = in->quant_tbl_no;
    if (tblno < 0 || tblno >= NUM_QUANT_TBLS || srcinfo->quant_tbl_ptrs[tblno] == NULL) ERREXIT1(dstinfo, JERR_NO_QUANT_TABLE, tblno);
    if (in->quant_table != NULL && memcmp(in->quant_table->quantval, srcinfo->quant_tbl_ptrs[tblno]->quantval, sizeof(in->quant_table->quantval)) != 0)
      ERREXIT1(dstinfo, JERR_MISMATCHED_QUANT_TABLE, tblno);   /* the file re-used this table slot */
  }
  if (srcinfo->saw_JFIF_marker) {
    if (srcinfo->JFIF_major_version == 1) { dstinfo->JFIF_major_version = srcinfo->JFIF_major_version; dstinfo->JFIF_minor_version = srcinfo->JFIF_minor_version; }
    dstinfo->density_unit = srcinfo->density_unit;
    dstinfo->X_density = srcinfo->X_density;
    dstinfo->Y_density = srcinfo->Y_density;
  }
}

/* =====================================================================================================================
 * what this library does not do.  An unchanged cjpeg is linked with immediate binding, so every symbol it names must
 * exist; the ones of the decompressor (cjpeg can take a JPEG file as input) and of 16-bit lossless input raise
 * JERR_NOT_COMPILED when they are actually called.
 * ===================================================================================================================== */
#ifdef MJH_STANDALONE
static void not_here(j_common_ptr cinfo, const char *what)
{
  fprintf(stderr, "mozjpeg_hip: %s is not part of this library (compress API only)\n", what);
  ERREXIT(cinfo, JERR_NOT_COMPILED);
}
void jpeg_CreateDecompress(j_decompress_ptr cinfo, int version, size_t structsize) { (void)version; (void)structsize; cinfo->mem = NULL; not_here((j_common_ptr)cinfo, "jpeg_CreateDecompress"); }
void jpeg_destroy_decompress(j_decompress_ptr cinfo) { jpeg_destroy((j_common_ptr)cinfo); }
void jpeg_abort_decompress(j_decompress_ptr cinfo) { jpeg_abort((j_common_ptr)cinfo); }
boolean jpeg_finish_decompress(j_decompress_ptr cinfo) { not_here((j_common_ptr)cinfo, "jpeg_finish_decompress"); return FALSE; }
int jpeg_read_header(j_decompress_ptr cinfo, boolean require_image) { (void)require_image; not_here((j_common_ptr)cinfo, "jpeg_read_header"); return 0; }
JDIMENSION jpeg_read_scanlines(j_decompress_ptr cinfo, JSAMPARRAY scanlines, JDIMENSION max_lines) { (void)scanlines; (void)max_lines; not_here((j_common_ptr)cinfo, "jpeg_read_scanlines"); return 0; }
boolean jpeg_start_decompress(j_decompress_ptr cinfo) { not_here((j_common_ptr)cinfo, "jpeg_start_decompress"); return FALSE; }
void jpeg_save_markers(j_decompress_ptr cinfo, int marker_code, unsigned int length_limit) { (void)marker_code; (void)length_limit; not_here((j_common_ptr)cinfo, "jpeg_save_markers"); }
void jpeg_stdio_src(j_decompress_ptr cinfo, FILE *infile) { (void)infile; not_here((j_common_ptr)cinfo, "jpeg_stdio_src"); }
void jpeg_mem_src(j_decompress_ptr cinfo, const unsigned char *inbuffer, unsigned long insize) { (void)inbuffer; (void)insize; not_here((j_common_ptr)cinfo, "jpeg_mem_src"); }
jvirt_barray_ptr *jpeg_read_coefficients(j_decompress_ptr cinfo) { not_here((j_common_ptr)cinfo, "jpeg_read_coefficients"); return NULL; }
JDIMENSION jpeg16_write_scanlines(j_compress_ptr cinfo, J16SAMPARRAY scanlines, JDIMENSION num_lines) { (void)scanlines; (void)num_lines; not_here((j_common_ptr)cinfo, "jpeg16_write_scanlines (lossless mode)"); return 0; }
#endif
