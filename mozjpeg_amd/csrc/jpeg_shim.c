/*
 * jpeg_shim.c -- libjpeg drop-in entry points on top of the MI355X batch encoder.
 * See include/mozjpeg_hip_jpeglib.h for the contract.  Compiled against the libjpeg headers of the
 * tree it drops into; links only to libmozjpeg_hip.so (and libdl).
 *
 * Two builds of this file:
 *   (default)        libmozjpeg_hip_jpeg62.so -- placed IN FRONT of a libjpeg (link order / LD_PRELOAD): exports only
 *                    the entry points that bracket the hot path plus the abort / destroy hooks, everything else of
 *                    the API keeps coming from the library behind it (dlsym(RTLD_NEXT)).
 *   -DMJH_STANDALONE standalone/libjpeg.so.62 -- together with jpeg_api.c a complete replacement of the libjpeg
 *                    COMPRESS API: nothing of the reference is needed at run time.
 *
 * State: one shim_state per active compress object (start .. finish / abort / destroy), found through a small
 * table keyed by the cinfo address.  A state OWNS its encoder from start to finish: encoders come from a process-wide
 * cache keyed by (parameters, device), so two interleaved compress objects of one thread can never meet in one
 * encoder, and repeated compressions with the same parameters reuse device buffers.  Devices: MOZJPEG_HIP_DEVICE=n
 * pins the process to one GPU; otherwise host threads are dealt round-robin over the visible GPUs.
 */
#define _GNU_SOURCE
#define JPEG_INTERNALS
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "jinclude.h"
#include "jpeglib.h"   /* JPEG_INTERNALS: pulls in jpegint.h and jerror.h */

#include "mozjpeg_hip.h"
#include "jpeg_shim.h"

typedef struct shim_state {
  j_compress_ptr cinfo;
  mjh_params p;
  mjh_encoder *enc;          /* owned from start to finish / abort */
  unsigned char *pixels;     /* scanlines go straight into the encoder's pinned staging buffer (not owned) */
  size_t row_bytes;
  int raw;                   /* raw_data_in: component planes arrive through jpeg_write_raw_data */
  jvirt_barray_ptr *coef_arrays;   /* jpeg_write_coefficients: the caller's virtual arrays, read at jpeg_finish_compress */
  unsigned char *planes[MAX_COMPONENTS];   /* staged planes / coefficient blocks (owned) */
  size_t plane_pitch[MAX_COMPONENTS];
  int header_bytes;          /* SOI (+APP0, +APP14) already written by jpeg_start_compress */
  int total_passes;          /* what the reference's master would report to a progress monitor */
  struct batcher *bt;        /* pixel input: the batcher of this parameter set (concurrent clients are coalesced, below) */
  int reading_arena;         /* >= 0: this object's file lies in that result arena of the batch encoder until it has been handed over */
  int end_dc;                /* scan search: bit t = end_dc_bits / vals[t] is the DC table the search's last coded DC scan left in slot t */
  unsigned char end_dc_bits[NUM_HUFF_TBLS][17], end_dc_vals[NUM_HUFF_TBLS][256];
  struct shim_state *next;
} shim_state;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static shim_state *g_states = NULL;

/* ---- encoder cache ------------------------------------------------------------------------------------------------ */
#define CACHE_MAX_IDLE 6
typedef struct cache_ent {
  mjh_params p;
  int device;
  mjh_encoder *enc;
  int busy;
  unsigned long stamp;
  struct cache_ent *next;
} cache_ent;
static cache_ent *g_cache = NULL;
static unsigned long g_stamp = 0;
static int g_next_device = 0;
static __thread int t_device = -1;

static int pick_device(void)
{
  if (t_device < 0) {
    const char *v = getenv("MOZJPEG_HIP_DEVICE");
    int n = mjh_device_count();
    if (n < 1) n = 1;
    if (v && *v >= '0' && *v <= '9') t_device = atoi(v) % n;
    else {   /* host threads are dealt round-robin over the GPUs: N client threads reach N devices */
      pthread_mutex_lock(&g_lock);
      t_device = g_next_device++ % n;
      pthread_mutex_unlock(&g_lock);
    }
    /* MOZJPEG_HIP_BIND=1: this client thread stages its rows into ITS device's pinned buffers from now on, so it may be moved
     * to the CPUs of that device's NUMA node (mjh_numa.cpp).  Opt-in: the reference never touches a caller's scheduling, and an
     * affinity mask set on an application thread is inherited by every thread it starts afterwards. */
    {
      const char *b = getenv("MOZJPEG_HIP_BIND");
      if (b && atoi(b) != 0) (void)mjh_bind_thread_to_device(t_device);
    }
  }
  return t_device;
}

static mjh_encoder *cache_acquire(const mjh_params *p, int device)
{
  cache_ent *c, **pp, *lru = NULL, **lru_pp = NULL;
  mjh_encoder *enc = NULL;
  int idle = 0;
  pthread_mutex_lock(&g_lock);
  for (c = g_cache; c; c = c->next)
    if (!c->busy && c->device == device && memcmp(&c->p, p, sizeof(*p)) == 0) { c->busy = 1; c->stamp = ++g_stamp; enc = c->enc; break; }
  if (!enc) {   /* make room: the least recently used idle encoder goes (each holds device buffers for one image) */
    for (pp = &g_cache; *pp; pp = &(*pp)->next)
      if (!(*pp)->busy) { idle++; if (!lru || (*pp)->stamp < lru->stamp) { lru = *pp; lru_pp = pp; } }
    if (idle >= CACHE_MAX_IDLE && lru) { *lru_pp = lru->next; } else lru = NULL;
  }
  pthread_mutex_unlock(&g_lock);
  if (enc) return enc;
  if (lru) { mjh_encoder_destroy(lru->enc); free(lru); }
  if (mjh_encoder_create(p, 1, device, &enc) != MJH_OK) return NULL;
  c = (cache_ent *)calloc(1, sizeof(*c));
  if (!c) { mjh_encoder_destroy(enc); return NULL; }
  c->p = *p; c->device = device; c->enc = enc; c->busy = 1;
  pthread_mutex_lock(&g_lock);
  c->stamp = ++g_stamp;
  c->next = g_cache; g_cache = c;
  pthread_mutex_unlock(&g_lock);
  return enc;
}

static void cache_release(mjh_encoder *enc)
{
  cache_ent *c;
  if (!enc) return;
  pthread_mutex_lock(&g_lock);
  for (c = g_cache; c; c = c->next) if (c->enc == enc) { c->busy = 0; break; }
  pthread_mutex_unlock(&g_lock);
}

/* ---- coalescing of concurrent clients --------------------------------------------------------------------------------
 * libjpeg hands the library ONE image per compress object, and a single-image schedule leaves most of the device idle
 * (measured, 16 client threads with an encoder each: 979 4K images/s, every image waiting 6.6 ms for its ~25 small
 * launches among the others').  Clients whose jpeg_finish_compress calls meet are therefore encoded as ONE batch: the first
 * to arrive while no batch is running becomes the leader, takes everybody who queued up meanwhile (up to BATCH_MAX),
 * gathers their staged images on the device (mjh_encode_gather) and hands every member its file; arrivals during a batch
 * form the next one -- no timer, the batch size follows the load.  A process with a single client never gets here
 * (`contended` stays 0) and keeps the direct path.  MOZJPEG_HIP_BATCH=0 turns the coalescing off. */
#define BATCH_MAX 8
typedef struct bnode { shim_state *s; const unsigned char *file; size_t n; int rc, done, arena; struct bnode *next; } bnode;
typedef struct batcher {
  mjh_params p;
  int device;
  mjh_encoder *enc;                 /* max_batch = BATCH_MAX, created when the first real batch forms */
  int busy, inflight, contended, arena_next, readers[2];
  bnode *qhead, *qtail;
  int qlen;
  pthread_mutex_t m;
  pthread_cond_t cv;
  struct batcher *next;
} batcher;
static batcher *g_batchers = NULL;
#define BATCH_PRIVATE 1000          /* node result: no batch encoder to be had, encode through the private one */

static int batching_enabled(void)
{
  static int on = -1;
  if (on < 0) { const char *v = getenv("MOZJPEG_HIP_BATCH"); on = !(v && atoi(v) == 0); }
  return on;
}

static batcher *batcher_enter(const mjh_params *p, int device)
{
  batcher *b;
  pthread_mutex_lock(&g_lock);
  for (b = g_batchers; b; b = b->next) if (b->device == device && memcmp(&b->p, p, sizeof(*p)) == 0) break;
  if (!b && (b = (batcher *)calloc(1, sizeof(*b))) != NULL) {
    b->p = *p; b->device = device;
    pthread_mutex_init(&b->m, NULL); pthread_cond_init(&b->cv, NULL);
    b->next = g_batchers; g_batchers = b;
  }
  pthread_mutex_unlock(&g_lock);
  if (b) {
    pthread_mutex_lock(&b->m);
    if (++b->inflight > 1) b->contended = 1;
    pthread_mutex_unlock(&b->m);
  }
  return b;
}

static void batcher_leave(batcher *b)
{
  if (!b) return;
  pthread_mutex_lock(&b->m);
  b->inflight--;
  pthread_mutex_unlock(&b->m);
}

static void batcher_done_reading(batcher *b, int arena)
{
  pthread_mutex_lock(&b->m);
  b->readers[arena]--;
  pthread_cond_broadcast(&b->cv);
  pthread_mutex_unlock(&b->m);
}

/* A scan search codes every candidate scan, and what the object's DC table slots hold afterwards is the table of the LAST CODED
 * scan that used the slot -- chroma: the Cr-alone candidate (jcparam.c:817-822), whether or not the file carries it.  The file
 * tells the other slots; these come from the encoder while the image's tables are still there (image `index` of enc's last batch). */
static void fetch_end_tables(shim_state *s, mjh_encoder *enc, int index)
{
  int si, k, t;
  s->end_dc = 0;
  if (!s->p.optimize_scans || s->p.arith_code) return;
  for (t = 0; t < NUM_HUFF_TBLS; t++) {
    int last = -1;
    for (si = 0; si < s->p.num_scans; si++) {
      const mjh_scan *sc = &s->p.scan_info[si];
      if (sc->Ss != 0 || sc->Ah != 0) continue;
      for (k = 0; k < sc->comps_in_scan; k++) if (s->p.dc_tbl_no[sc->component_index[k]] == t) last = si;
    }
    if (last >= 0 && mjh_get_scan_table(enc, index, last, t, s->end_dc_bits[t], s->end_dc_vals[t]) == MJH_OK) s->end_dc |= 1 << t;
  }
}

/* jpeg_finish_compress of a pixel-input object whose batcher has seen company: returns MJH_OK with *file / *n pointing into the
 * batch encoder's result arena (valid until batcher_done_reading(b, *arena)), BATCH_PRIVATE, or an encoder error */
static int batched_encode(shim_state *s, const unsigned char **file, size_t *n, int *arena)
{
  batcher *b = s->bt;
  bnode me;
  memset(&me, 0, sizeof(me));
  me.s = s;
  pthread_mutex_lock(&b->m);
  if (b->qtail) b->qtail->next = &me; else b->qhead = &me;
  b->qtail = &me; b->qlen++;
  while (!me.done) {
    if (!b->busy && b->qhead == &me) {          /* lead the next batch: everybody queued so far, in arrival order */
      bnode *mem[BATCH_MAX];
      mjh_encoder *encs[BATCH_MAX];
      const void *base = NULL; const mjh_result *res = NULL;
      int k = 0, i, cnt = 0, rc = MJH_OK, ar;
      b->busy = 1;
      while (b->qhead && k < BATCH_MAX) { mem[k] = b->qhead; encs[k] = mem[k]->s->enc; b->qhead = b->qhead->next; k++; }
      if (!b->qhead) b->qtail = NULL;
      b->qlen -= k;
      ar = b->arena_next;
      while (b->readers[ar] > 0) pthread_cond_wait(&b->cv, &b->m);   /* the files of the batch before the last one are still being handed over */
      pthread_mutex_unlock(&b->m);
      if (!b->enc && mjh_encoder_create(&b->p, BATCH_MAX, b->device, &b->enc) != MJH_OK) { b->enc = NULL; rc = BATCH_PRIVATE; }
      if (rc == MJH_OK) rc = mjh_encode_gather(b->enc, encs, k);
      if (rc == MJH_OK) rc = mjh_collect(b->enc, 0, &base, &res, &cnt);
      if (rc == MJH_OK && cnt != k) rc = MJH_EHIP;
      for (i = 0; rc == MJH_OK && i < k; i++) fetch_end_tables(mem[i]->s, b->enc, i);   /* (before the next batch overwrites them) */
      pthread_mutex_lock(&b->m);
      for (i = 0; i < k; i++) {
        mem[i]->rc = rc; mem[i]->arena = ar;
        if (rc == MJH_OK) { mem[i]->file = (const unsigned char *)base + res[i].offset; mem[i]->n = (size_t)res[i].size; }
        mem[i]->done = 1;
      }
      if (rc == MJH_OK) { b->readers[ar] += k; b->arena_next ^= 1; }
      if (rc == BATCH_PRIVATE) b->contended = -1;        /* no batch encoder: everybody keeps to the private path from now on */
      else if (rc != MJH_OK && b->enc) {
        /* a failed batch may have flipped the encoder's result arenas without this bookkeeping knowing (mjh_encode_gather counts
         * the call before the steps that can still fail): start over with a fresh batch encoder, once nobody reads the old
         * one's arenas any more */
        mjh_encoder *old = b->enc;
        while (b->readers[0] > 0 || b->readers[1] > 0) pthread_cond_wait(&b->cv, &b->m);
        b->enc = NULL; b->arena_next = 0;
        pthread_mutex_unlock(&b->m);
        mjh_encoder_destroy(old);
        pthread_mutex_lock(&b->m);
      }
      b->busy = 0;
      pthread_cond_broadcast(&b->cv);
      continue;
    }
    pthread_cond_wait(&b->cv, &b->m);
  }
  pthread_mutex_unlock(&b->m);
  *file = me.file; *n = me.n; *arena = me.arena;
  return me.rc;
}

/* ---- state table ---------------------------------------------------------------------------------------------------- */
static shim_state *find_state(j_compress_ptr cinfo, int remove)
{
  shim_state **pp, *s = NULL;
  pthread_mutex_lock(&g_lock);
  for (pp = &g_states; *pp; pp = &(*pp)->next)
    if ((*pp)->cinfo == cinfo) { s = *pp; if (remove) *pp = s->next; break; }
  pthread_mutex_unlock(&g_lock);
  return s;
}

static void free_state(shim_state *s)
{
  int ci;
  if (!s) return;
  for (ci = 0; ci < MAX_COMPONENTS; ci++) free(s->planes[ci]);
  if (s->bt && s->reading_arena >= 0) batcher_done_reading(s->bt, s->reading_arena);   /* (also on an error exit in the middle of the hand-over) */
  batcher_leave(s->bt);
  cache_release(s->enc);
  free(s);
}

/* jpeg_abort / jpeg_destroy on an object that is in the middle of a compression (the application's error_exit
 * longjmp'ed out, or it simply gave up): forget the staged image, hand the encoder back.  Returns 1 if there was one. */
int mjh_shim_drop(void *cinfo)
{
  shim_state *s = find_state((j_compress_ptr)cinfo, 1);
  if (!s) return 0;
  free_state(s);
  return 1;
}

/* every ERREXIT of this file goes through here once a state exists: the state must not outlive the error */
#define FAIL_WITH_STATE(cinfo, stmt) do { mjh_shim_drop(cinfo); stmt; } while (0)

static void emit_byte(j_compress_ptr cinfo, int v)
{ /* same protocol as jcmarker.c:113-123 */
  struct jpeg_destination_mgr *dest = cinfo->dest;
  *(dest->next_output_byte)++ = (JOCTET)v;
  if (--dest->free_in_buffer == 0)
    if (!(*dest->empty_output_buffer) (cinfo)) FAIL_WITH_STATE(cinfo, ERREXIT(cinfo, JERR_CANT_SUSPEND));
}
static void emit_bytes(j_compress_ptr cinfo, const unsigned char *p, size_t n)
{
  struct jpeg_destination_mgr *dest = cinfo->dest;
  while (n > 0) {
    size_t k = n < dest->free_in_buffer ? n : dest->free_in_buffer;
    memcpy(dest->next_output_byte, p, k);
    dest->next_output_byte += k; dest->free_in_buffer -= k; p += k; n -= k;
    if (dest->free_in_buffer == 0)
      if (!(*dest->empty_output_buffer) (cinfo)) FAIL_WITH_STATE(cinfo, ERREXIT(cinfo, JERR_CANT_SUSPEND));
  }
}

/* ---- DQT / DHT markers under the reference's sent_table protocol ----------------------------------------------------------
 * The device writes a COMPLETE file: every table in front of the first scan that needs it, as a compress object does whose
 * tables are all unsent.  What a libjpeg client may ask for on top of that lives in the object, not in the image: a table whose
 * `sent_table` flag is set is left out (abbreviated datastreams: jpeg_write_tables / jpeg_suppress_tables /
 * jpeg_start_compress(cinfo, FALSE); libjpeg.txt "Abbreviated datastreams and multiple images"), every table written gets the
 * flag, and the two marker layouts (one marker per table, or mozjpeg's one marker for all of them) are chosen per call from
 * the flags.  The shim therefore writes the table markers itself, from the object's table slots, with the reference's own
 * rules; the device's file supplies the table CONTENTS (optimal Huffman tables, re-estimated quantization tables: copied into
 * the slots first, unsent, as jpeg_gen_optimal_table leaves them, jchuff.c:1104-1105) and everything that is not a table.
 * Restated from jcmarker.c: emit_dqt :149-186, emit_multi_dqt :189-254, emit_dht :257-290, emit_multi_dht :293-401 (including
 * the way its sizing loop skips a component's AC table behind a DC table that was sent or seen -- the reference's bytes, whatever
 * one thinks of them), write_frame_header :674-697, write_scan_header :757-775. */
static const unsigned char shim_zz[DCTSIZE2] = {
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

static void emit_2(j_compress_ptr cinfo, int v) { emit_byte(cinfo, (v >> 8) & 0xFF); emit_byte(cinfo, v & 0xFF); }

static int qtbl_wide(const JQUANT_TBL *q)
{
  int i, wide = 0;
  for (i = 0; i < DCTSIZE2; i++) if (q->quantval[i] > 255) wide = 1;
  return wide;
}

static void tw_qtbl_body(j_compress_ptr cinfo, JQUANT_TBL *q, int index, int wide)
{
  int i;
  emit_byte(cinfo, index + (wide << 4));
  for (i = 0; i < DCTSIZE2; i++) {
    const unsigned int v = q->quantval[shim_zz[i]];
    if (wide) emit_byte(cinfo, (int)(v >> 8));
    emit_byte(cinfo, (int)(v & 0xFF));
  }
  q->sent_table = TRUE;
}

static void tw_frame_tables(j_compress_ptr cinfo)
{
  int ci, one_marker = cinfo->master->compress_profile != JCP_FASTEST;
  for (ci = 0; ci < cinfo->num_components && one_marker; ci++) {   /* one marker for all: only while nothing has been sent */
    const JQUANT_TBL *q = cinfo->quant_tbl_ptrs[cinfo->comp_info[ci].quant_tbl_no];
    if (q == NULL || q->sent_table) one_marker = 0;
  }
  if (one_marker) {
    int size = 2, counted[NUM_QUANT_TBLS] = { 0, 0, 0, 0 };
    for (ci = 0; ci < cinfo->num_components; ci++) {
      const int t = cinfo->comp_info[ci].quant_tbl_no;
      if (!counted[t]) { size += DCTSIZE2 * (qtbl_wide(cinfo->quant_tbl_ptrs[t]) + 1) + 1; counted[t] = 1; }
    }
    emit_byte(cinfo, 0xFF); emit_byte(cinfo, 0xDB);
    emit_2(cinfo, size);
    for (ci = 0; ci < cinfo->num_components; ci++) {
      const int t = cinfo->comp_info[ci].quant_tbl_no;
      JQUANT_TBL *q = cinfo->quant_tbl_ptrs[t];
      if (!q->sent_table) tw_qtbl_body(cinfo, q, t, qtbl_wide(q));
    }
    return;
  }
  for (ci = 0; ci < cinfo->num_components; ci++) {
    const int t = cinfo->comp_info[ci].quant_tbl_no;
    JQUANT_TBL *q = cinfo->quant_tbl_ptrs[t];
    if (q == NULL) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_NO_QUANT_TABLE, t));
    if (!q->sent_table) {
      const int wide = qtbl_wide(q);
      emit_byte(cinfo, 0xFF); emit_byte(cinfo, 0xDB);
      emit_2(cinfo, wide ? DCTSIZE2 * 2 + 1 + 2 : DCTSIZE2 + 1 + 2);
      tw_qtbl_body(cinfo, q, t, wide);
    }
  }
}

static int htbl_count(const JHUFF_TBL *h) { int l, n = 0; for (l = 1; l <= 16; l++) n += h->bits[l]; return n; }

static void tw_htbl_body(j_compress_ptr cinfo, JHUFF_TBL *h, int id, int nvals)
{
  int i;
  emit_byte(cinfo, id);
  for (i = 1; i <= 16; i++) emit_byte(cinfo, h->bits[i]);
  for (i = 0; i < nvals; i++) emit_byte(cinfo, h->huffval[i]);
  h->sent_table = TRUE;
}

/* the tables in front of one scan's SOS: comps in scan order */
static void tw_scan_tables(j_compress_ptr cinfo, jpeg_component_info *const *comps, int n, int Ss, int Se, int Ah)
{
  const int want_dc = Ss == 0 && Ah == 0, want_ac = Se != 0;
  int i, j;
  if (cinfo->master->compress_profile != JCP_FASTEST) {
    /* one marker: sized in a first loop over the components, written in a second one */
    int length = 2, nd[MAX_COMPS_IN_SCAN] = { 0, 0, 0, 0 }, na[MAX_COMPS_IN_SCAN] = { 0, 0, 0, 0 };
    const JHUFF_TBL *dseen[NUM_HUFF_TBLS] = { NULL, NULL, NULL, NULL }, *aseen[NUM_HUFF_TBLS] = { NULL, NULL, NULL, NULL };
    for (i = 0; i < n; i++) {
      const int d = comps[i]->dc_tbl_no, a = comps[i]->ac_tbl_no;
      const JHUFF_TBL *dt = cinfo->dc_huff_tbl_ptrs[d], *at = cinfo->ac_huff_tbl_ptrs[a];
      int dup = 0;
      if (want_dc) {
        if (dt == NULL) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_NO_HUFF_TABLE, d));
        if (dt->sent_table) continue;                       /* (past this component's AC table as well) */
        for (j = 0; j < NUM_HUFF_TBLS; j++) dup += dt == dseen[j];
        if (dup) continue;
        dseen[i] = dt;
        length += (nd[i] = htbl_count(dt)) + 16 + 1;
      }
      if (want_ac) {
        if (at == NULL) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_NO_HUFF_TABLE, a + 0x10));
        if (at->sent_table) continue;
        for (dup = 0, j = 0; j < NUM_HUFF_TBLS; j++) dup += at == aseen[j];
        if (dup) continue;
        aseen[i] = at;
        length += (na[i] = htbl_count(at)) + 16 + 1;
      }
    }
    if (length <= 65535) {
      emit_byte(cinfo, 0xFF); emit_byte(cinfo, 0xC4);
      emit_2(cinfo, length);
      for (i = 0; i < n; i++) {
        JHUFF_TBL *dt = cinfo->dc_huff_tbl_ptrs[comps[i]->dc_tbl_no], *at = cinfo->ac_huff_tbl_ptrs[comps[i]->ac_tbl_no];
        if (want_dc && !dt->sent_table) tw_htbl_body(cinfo, dt, comps[i]->dc_tbl_no, nd[i]);
        if (want_ac && !at->sent_table) tw_htbl_body(cinfo, at, comps[i]->ac_tbl_no + 0x10, na[i]);
      }
      return;
    }
  }
  for (i = 0; i < n; i++) {          /* one marker per table not sent yet */
    int which;
    for (which = 0; which < 2; which++) {
      const int t = which ? comps[i]->ac_tbl_no : comps[i]->dc_tbl_no;
      JHUFF_TBL *h = which ? cinfo->ac_huff_tbl_ptrs[t] : cinfo->dc_huff_tbl_ptrs[t];
      if (which ? !want_ac : !want_dc) continue;
      if (h == NULL) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_NO_HUFF_TABLE, t + (which ? 0x10 : 0)));
      if (!h->sent_table) {
        const int nv = htbl_count(h);
        emit_byte(cinfo, 0xFF); emit_byte(cinfo, 0xC4);
        emit_2(cinfo, nv + 2 + 1 + 16);
        tw_htbl_body(cinfo, h, t + (which ? 0x10 : 0), nv);
      }
    }
  }
}

/* first marker at or behind p that is neither a stuffed zero, a fill byte nor RSTn: the end of a scan's entropy-coded data */
static const unsigned char *scan_data_end(const unsigned char *p, const unsigned char *end)
{
  while (p < end && (p = (const unsigned char *)memchr(p, 0xFF, (size_t)(end - p))) != NULL) {
    if (p + 1 >= end) break;
    if (p[1] == 0 || (p[1] >= 0xD0 && p[1] <= 0xD7)) p += 2;
    else if (p[1] == 0xFF) p++;
    else return p;
  }
  return end;
}

/* The device's file from its frame header on (p: the first DQT) goes to the destination manager: tables through the writers
 * above, the rest as it is.  own_tables: this image's Huffman tables were made for it on the device (optimize_coding, forced or
 * asked for) -- they replace what the slots hold, and count as unsent; otherwise the file's tables ARE the slots' (the encoder got
 * them from there) and only the flags decide.  The quantization tables are copied back where the device re-estimated them
 * (trellis_q_opt: jcmaster.c:1016-1030 writes cinfo->quant_tbl_ptrs too, flags untouched).  On return cinfo->Ss / Se / Ah / Al are
 * what the reference's last select_scan_parameters call of the image leaves (jcmaster.c:468-512). */
static void hand_over_file(j_compress_ptr cinfo, const shim_state *s, const unsigned char *p, const unsigned char *end)
{
  const int own_tables = !cinfo->arith_code && (s->p.optimize_coding || s->p.data_precision == 12);
  const int one_scan = s->p.num_scans == 0;
  int search_al = -1, last[4] = { 0, DCTSIZE2 - 1, 0, 0 };
  const unsigned char *held[4];
  int nheld = 0, k;
  /* frame header: DQT ... SOF */
  while (p + 4 <= end && p[0] == 0xFF && p[1] == 0xDB) {
    const unsigned char *q = p + 4, *qe = p + 2 + ((p[2] << 8) | p[3]);
    if (s->p.trellis_quant && s->p.trellis_q_opt)
      while (q < qe) {
        const int wide = q[0] >> 4, t = q[0] & 15;
        JQUANT_TBL *qt = t < NUM_QUANT_TBLS ? cinfo->quant_tbl_ptrs[t] : NULL;
        q++;
        for (k = 0; k < DCTSIZE2; k++, q += 1 + wide) if (qt) qt->quantval[shim_zz[k]] = (UINT16)(wide ? (q[0] << 8) | q[1] : q[0]);
      }
    p = qe;
  }
  tw_frame_tables(cinfo);
  for (;;) {
    /* one scan: DHT / DAC / DRI ... SOS, entropy-coded data */
    jpeg_component_info *comps[MAX_COMPS_IN_SCAN];
    int n, Ss, Se, Ah, Al;
    nheld = 0;
    while (p + 4 <= end && p[0] == 0xFF && p[1] != 0xDA && p[1] != 0xD9) {
      const unsigned char *seg_end = p + 2 + ((p[2] << 8) | p[3]);
      if (seg_end > end) seg_end = end;
      if (p[1] == 0xC4) {
        const unsigned char *q = p + 4;
        while (own_tables && q + 17 <= seg_end) {
          const int ac = q[0] >> 4, t = q[0] & 15;
          JHUFF_TBL **slot = t < NUM_HUFF_TBLS ? (ac ? &cinfo->ac_huff_tbl_ptrs[t] : &cinfo->dc_huff_tbl_ptrs[t]) : NULL;
          int nv = 0;
          for (k = 1; k <= 16; k++) nv += q[k];
          if (slot) {
            if (*slot == NULL) *slot = jpeg_alloc_huff_table((j_common_ptr)cinfo);
            (*slot)->bits[0] = 0;
            memcpy((*slot)->bits + 1, q + 1, 16);
            memcpy((*slot)->huffval, q + 17, (size_t)(nv < 256 ? nv : 256));
            (*slot)->sent_table = FALSE;
          }
          q += 17 + nv;
        }
      } else if (nheld < 4) held[nheld++] = p;      /* SOF in front of the first scan, DAC, DRI: behind the tables, in the device's order */
      p = seg_end;
    }
    if (p + 4 > end || p[1] == 0xD9) {
      for (k = 0; k < nheld; k++) emit_bytes(cinfo, held[k], (size_t)(2 + ((held[k][2] << 8) | held[k][3])));
      if (p + 2 <= end) emit_bytes(cinfo, p, 2);
      break;
    }
    /* p: SOS -- Ls, Ns, (Cs, Td/Ta) x Ns, Ss, Se, Ah/Al */
    n = p[4] <= MAX_COMPS_IN_SCAN ? p[4] : MAX_COMPS_IN_SCAN;
    for (k = 0; k < n; k++) {
      int ci;
      comps[k] = &cinfo->comp_info[0];
      for (ci = 0; ci < cinfo->num_components; ci++) if (cinfo->comp_info[ci].component_id == p[5 + 2 * k]) { comps[k] = &cinfo->comp_info[ci]; break; }
    }
    Ss = p[5 + 2 * n]; Se = p[6 + 2 * n]; Ah = p[7 + 2 * n] >> 4; Al = p[7 + 2 * n] & 15;
    /* the SOF travels with the frame header, in front of the first scan's tables */
    k = 0;
    if (nheld && held[0][1] >= 0xC0 && held[0][1] <= 0xCF && held[0][1] != 0xC4 && held[0][1] != 0xCC) { emit_bytes(cinfo, held[0], (size_t)(2 + ((held[0][2] << 8) | held[0][3]))); k = 1; }
    if (!cinfo->arith_code) tw_scan_tables(cinfo, comps, n, Ss, Se, Ah);
    for (; k < nheld; k++) emit_bytes(cinfo, held[k], (size_t)(2 + ((held[k][2] << 8) | held[k][3])));
    last[0] = Ss; last[1] = Se; last[2] = Ah; last[3] = Al;
    if (Ss > 0 && Ah == 0 && comps[0] == &cinfo->comp_info[cinfo->num_components - 1]) search_al = Al;
    if (one_scan) { emit_bytes(cinfo, p, (size_t)(end - p)); break; }
    {
      const unsigned char *e = scan_data_end(p + 2 + ((p[2] << 8) | p[3]), end);
      emit_bytes(cinfo, p, (size_t)(e - p));
      p = e;
    }
  }
  for (k = 0; k < NUM_HUFF_TBLS; k++)
    if (s->end_dc >> k & 1) {
      JHUFF_TBL **slot = &cinfo->dc_huff_tbl_ptrs[k];
      if (*slot == NULL) *slot = jpeg_alloc_huff_table((j_common_ptr)cinfo);
      memcpy((*slot)->bits, s->end_dc_bits[k], 17);
      memcpy((*slot)->huffval, s->end_dc_vals[k], 256);
      (*slot)->sent_table = TRUE;
    }
  if (s->p.optimize_scans && search_al >= 0) { last[2] = 0; last[3] = search_al; }   /* the search's last coded scan: a frequency split of the last component at its best Al (jcmaster.c:482-495) */
  cinfo->Ss = last[0]; cinfo->Se = last[1]; cinfo->Ah = last[2]; cinfo->Al = last[3];
}

/* minimal marker writer so that jpeg_write_marker / jpeg_write_m_header keep working
 * (jcmarker.c:590-615) between jpeg_start_compress and the first scanline */
static void mw_header(j_compress_ptr cinfo, int marker, unsigned int datalen)
{
  if (datalen > 65533u) FAIL_WITH_STATE(cinfo, ERREXIT(cinfo, JERR_BAD_LENGTH));
  emit_byte(cinfo, 0xFF); emit_byte(cinfo, marker);
  emit_byte(cinfo, (int)((datalen + 2) >> 8) & 0xFF); emit_byte(cinfo, (int)(datalen + 2) & 0xFF);
}
static void mw_byte(j_compress_ptr cinfo, int val) { emit_byte(cinfo, val); }
static void mw_nop(j_compress_ptr cinfo) { (void)cinfo; }

typedef void (*start_fn)(j_compress_ptr, boolean);
typedef JDIMENSION (*write_fn)(j_compress_ptr, JSAMPARRAY, JDIMENSION);
typedef void (*finish_fn)(j_compress_ptr);
typedef void (*wrcoef_fn)(j_compress_ptr, jvirt_barray_ptr *);
typedef JDIMENSION (*raw_fn)(j_compress_ptr, JSAMPIMAGE, JDIMENSION);

#ifdef MJH_STANDALONE
#define NEXT_SYMBOL(name) NULL
#else
#define NEXT_SYMBOL(name) dlsym(RTLD_NEXT, name)
#endif

static const char *capture_params(j_compress_ptr cinfo, mjh_params *p, int no_pixels)
{
  int ci, i;
  memset(p, 0, sizeof(*p));
  if (cinfo->data_precision != 8 && cinfo->data_precision != 12) return "data_precision other than 8 or 12";
  p->data_precision = cinfo->data_precision;
  if (cinfo->arith_code) {
    /* cjpeg -arithmetic; the conditioning of tables 0 / 1 as the application set it (jpeg_set_defaults: 0 / 1 / 5, jcparam.c:417-419) */
    for (i = 0; i < 2; i++) { p->arith_dc_L[i] = cinfo->arith_dc_L[i]; p->arith_dc_U[i] = cinfo->arith_dc_U[i]; p->arith_ac_K[i] = cinfo->arith_ac_K[i]; }
    p->arith_code = 1;
    /* (mjh_params reads a table with L = U = K = 0 as "not set: the defaults 0 / 1 / 5"; an application that really conditions a
     * table it uses that way would get other bytes than the reference writes) */
    for (ci = 0; ci < cinfo->num_components; ci++) {
      const int td = cinfo->comp_info[ci].dc_tbl_no, ta = cinfo->comp_info[ci].ac_tbl_no;
      if ((td >= 0 && td < 2 && ta >= 0 && ta < 2) && ((!cinfo->arith_dc_L[td] && !cinfo->arith_dc_U[td] && !cinfo->arith_ac_K[td]) || (!cinfo->arith_dc_L[ta] && !cinfo->arith_dc_U[ta] && !cinfo->arith_ac_K[ta])))
        return "arithmetic conditioning L = U = K = 0 for a table in use";
    }
  }
  if (cinfo->master->lossless) return "lossless mode";
  p->smoothing_factor = cinfo->smoothing_factor;   /* cjpeg -smooth N; ignored for raw data / coefficients like in the reference */
  if (cinfo->dct_method == JDCT_IFAST) {
    p->dct_method = 1;      /* jfdctfst.c: what TurboJPEG's legacy calls select below quality 96, `cjpeg -dct fast` */
  } else if (cinfo->dct_method != JDCT_ISLOW) return "dct_method JDCT_FLOAT (floating point: not a bit-exact path, SURVEY F5)";
  {
    /* rgb_red/green/blue/pixelsize of jccolor.c / jmorecfg.h for the extended colour spaces */
    int ps = 0, ro = 0, go = 1, bo = 2;
    switch (cinfo->in_color_space) {
    case JCS_RGB: case JCS_EXT_RGB: ps = 3; break;
    case JCS_EXT_BGR: ps = 3; ro = 2; bo = 0; break;
    case JCS_EXT_RGBX: case JCS_EXT_RGBA: ps = 4; break;
    case JCS_EXT_BGRX: case JCS_EXT_BGRA: ps = 4; ro = 2; bo = 0; break;
    case JCS_EXT_XBGR: case JCS_EXT_ABGR: ps = 4; ro = 3; go = 2; bo = 1; break;
    case JCS_EXT_XRGB: case JCS_EXT_ARGB: ps = 4; ro = 1; go = 2; bo = 3; break;
    case JCS_GRAYSCALE: ps = 1; break;
    case JCS_YCbCr: ps = 3; break;     /* samples that are YCbCr already: only into a YCbCr file (below) */
    default:
      if (!no_pixels) return "input colour space (RGB family / grayscale only)";
      ps = 3;
      break;
    }
    if (no_pixels) { ps = cinfo->num_components == 1 ? 1 : 3; ro = 0; go = 1; bo = 2; }   /* planes / coefficients in: the input pixel format is never looked at */
    if (!no_pixels && cinfo->input_components != ps) return "input_components does not match in_color_space";
    if (ps == 1) p->input_components = 1;
    else { p->input_components = 3; p->input_pixel_size = ps; p->rgb_offset[0] = ro; p->rgb_offset[1] = go; p->rgb_offset[2] = bo; }
  }
  if (!no_pixels && cinfo->in_color_space == JCS_YCbCr) {
    /* jinit_color_converter jccolor.c:687-692: YCbCr in -> YCbCr out is null_convert (:479); -> grayscale takes the Y samples
     * (grayscale_convert :448-466): the same null conversion with one component kept */
    if (cinfo->jpeg_color_space == JCS_YCbCr && cinfo->num_components == 3) p->num_components = 3;
    else if (cinfo->jpeg_color_space == JCS_GRAYSCALE && cinfo->num_components == 1) p->num_components = 1;
    else return "YCbCr input into anything but a YCbCr or a grayscale file";
    p->color_transform = MJH_COLOR_YCC_IN;
  } else if (cinfo->jpeg_color_space == JCS_YCbCr && cinfo->num_components == 3) p->num_components = 3;
  else if (cinfo->jpeg_color_space == JCS_GRAYSCALE && cinfo->num_components == 1) p->num_components = 1;
  else if (cinfo->jpeg_color_space == JCS_RGB && cinfo->num_components == 3 && (no_pixels || p->input_components == 3)) {
    p->num_components = 3;           /* cjpeg -rgb: null_convert (jccolor.c:479), the samples go through unconverted */
    p->color_transform = MJH_COLOR_NONE;
  } else return "JPEG colour space (only YCbCr / grayscale / RGB)";
  /* an Adobe APP14 marker the application asks for is written by this shim (shim_begin) with the transform code of the file's
   * colour space; the device writes its own only into RGB files, where it is dropped like the APP0 */
  /* JFIF version / density: the APP0 segment is written by this shim from the cinfo fields (emit_jfif_app0
   * jcmarker.c:422-449), the device's fixed APP0 is dropped, so any values are fine */
  p->image_width = (int)cinfo->image_width;
  p->image_height = (int)cinfo->image_height;
  for (ci = 0; ci < cinfo->num_components; ci++) {
    jpeg_component_info *c = &cinfo->comp_info[ci];
    p->h_samp_factor[ci] = c->h_samp_factor; p->v_samp_factor[ci] = c->v_samp_factor;
    p->quant_tbl_no[ci] = c->quant_tbl_no; p->dc_tbl_no[ci] = c->dc_tbl_no; p->ac_tbl_no[ci] = c->ac_tbl_no;
    p->component_id[ci] = c->component_id;
    if (c->quant_tbl_no < 0 || c->quant_tbl_no >= NUM_QUANT_TBLS || cinfo->quant_tbl_ptrs[c->quant_tbl_no] == NULL)
      return "component without a quantization table";
    for (i = 0; i < 64; i++) p->quantval[c->quant_tbl_no][i] = cinfo->quant_tbl_ptrs[c->quant_tbl_no]->quantval[i];
  }
  p->compress_profile = jpeg_c_get_int_param(cinfo, JINT_COMPRESS_PROFILE) == JCP_FASTEST ? MJH_PROFILE_FASTEST
                                                                                            : MJH_PROFILE_MAX_COMPRESSION;
  p->optimize_coding = cinfo->optimize_coding;
  p->trellis_quant = jpeg_c_get_bool_param(cinfo, JBOOLEAN_TRELLIS_QUANT);
  p->trellis_quant_dc = jpeg_c_get_bool_param(cinfo, JBOOLEAN_TRELLIS_QUANT_DC);
  p->overshoot_deringing = jpeg_c_get_bool_param(cinfo, JBOOLEAN_OVERSHOOT_DERINGING);
  p->lambda_log_scale1 = jpeg_c_get_float_param(cinfo, JFLOAT_LAMBDA_LOG_SCALE1);
  p->lambda_log_scale2 = jpeg_c_get_float_param(cinfo, JFLOAT_LAMBDA_LOG_SCALE2);
  if (no_pixels == 2) { p->trellis_quant = p->trellis_quant_dc = p->overshoot_deringing = 0; p->dct_method = 0; }   /* no trellis passes when transcoding (jcmaster.c transcode_only) */
  if (cinfo->data_precision == 12 && p->trellis_quant) return "12-bit trellis (the reference itself aborts: jccoefct.c:132-138)";
  /* the remaining extension parameters travel as they are; mjh_encoder_create refuses what the device path lacks */
  p->trellis_eob_opt = jpeg_c_get_bool_param(cinfo, JBOOLEAN_TRELLIS_EOB_OPT);
  p->use_scans_in_trellis = jpeg_c_get_bool_param(cinfo, JBOOLEAN_USE_SCANS_IN_TRELLIS);
  p->trellis_freq_split = jpeg_c_get_int_param(cinfo, JINT_TRELLIS_FREQ_SPLIT);
  p->trellis_q_opt = jpeg_c_get_bool_param(cinfo, JBOOLEAN_TRELLIS_Q_OPT);
  p->trellis_delta_dc_weight = jpeg_c_get_float_param(cinfo, JFLOAT_TRELLIS_DELTA_DC_WEIGHT);
  p->dc_scan_opt_mode = jpeg_c_get_int_param(cinfo, JINT_DC_SCAN_OPT_MODE);
  p->trellis_num_loops = jpeg_c_get_int_param(cinfo, JINT_TRELLIS_NUM_LOOPS);
  if (p->trellis_num_loops < 1 || p->trellis_num_loops > 16) return "trellis_num_loops outside 1..16";
  if (p->arith_code) p->optimize_coding = 0;   /* jinit_c_master_control, jcmaster.c:1088-1089 */
  p->restart_interval = cinfo->restart_interval;
  p->restart_in_rows = cinfo->restart_in_rows;
  if (cinfo->scan_info != NULL && cinfo->num_scans > 0) {
    /* progressive script: jpeg_scan_info (jpeglib.h:196-201) -> mjh_scan; jpeg_start_compress drops
     * optimize_scans when no search script was set up (jcapistd.c:53-56) */
    int si;
    if (cinfo->num_scans > MJH_MAX_SCANS) return "more than 64 scans";
    p->num_scans = cinfo->num_scans;
    for (si = 0; si < cinfo->num_scans; si++) {
      const jpeg_scan_info *js = &cinfo->scan_info[si];
      mjh_scan *ms = &p->scan_info[si];
      int k;
      ms->comps_in_scan = js->comps_in_scan;
      for (k = 0; k < js->comps_in_scan && k < MJH_MAX_COMPS; k++) ms->component_index[k] = js->component_index[k];
      ms->Ss = js->Ss; ms->Se = js->Se; ms->Ah = js->Ah; ms->Al = js->Al;
    }
    p->optimize_scans = jpeg_c_get_bool_param(cinfo, JBOOLEAN_OPTIMIZE_SCANS) && cinfo->master->num_scans_luma != 0;
    /* a progressive script forces optimal tables (jcmaster.c:1088-1094); a script of whole-block scans is a sequential
     * multi-scan file (validate_script :309-330) and keeps the application's choice */
    if (p->optimize_scans || !(p->scan_info[0].Ss == 0 && p->scan_info[0].Se == 63)) p->optimize_coding = p->arith_code ? 0 : 1;
  }
  {
    /* what the reference reads from the object's Huffman table slots (mozjpeg_hip.h: huff_tables_given): the coding tables when the
     * application does not ask for optimal ones (start_pass_huff: jpeg_make_c_derived_tbl of dc / ac_huff_tbl_ptrs, jchuff.c:190-196)
     * -- for 12-bit samples only tables of its own, the Annex K ones are replaced by optimal tables (jcmaster.c:1096-1105) -- and the DC
     * tables a progressive image's DC trellis takes its rates from (jccoefct.c:388-389).  A slot that holds the Annex K
     * table travels as "not given". */
    const int progressive = p->num_scans > 0 && (p->optimize_scans || !(p->scan_info[0].Ss == 0 && p->scan_info[0].Se == 63));
    int coding, own = 0;
    /* the trellis with optimize_coding switched off by hand, ONE component: the reference's passes are then the schedule of
     * optimize_coding itself (statistics, trellis pass(es), output with the tables the last pass left in the slots) and its file the
     * same bytes (mjh_encoder.cpp: check_supported); cinfo->optimize_coding stays what the application set */
    if (p->trellis_quant && !p->optimize_coding && !p->arith_code && p->num_components == 1 && p->num_scans == 0 && !p->use_scans_in_trellis &&
        !(p->trellis_q_opt && p->trellis_num_loops > 1))
      p->optimize_coding = 1;
    coding = !p->optimize_coding && !p->arith_code;
    const int dc_rates = progressive && !p->arith_code && p->trellis_quant && p->trellis_quant_dc;
    if (coding || dc_rates)
      for (ci = 0; ci < cinfo->num_components; ci++) {
        int which;
        for (which = 0; which < (coding ? 2 : 1); which++) {
          const int t = which ? cinfo->comp_info[ci].ac_tbl_no : cinfo->comp_info[ci].dc_tbl_no;
          const JHUFF_TBL *h = t >= 0 && t < NUM_HUFF_TBLS ? (which ? cinfo->ac_huff_tbl_ptrs[t] : cinfo->dc_huff_tbl_ptrs[t]) : NULL;
          const uint8_t *bits, *vals;
          int nv = 0, l;
          if (!h) return "a component's Huffman table slot is empty (JERR_NO_HUFF_TABLE)";
          for (l = 1; l <= 16; l++) nv += h->bits[l];
          if (nv > 256) return "Huffman table with more than 256 symbols (JERR_BAD_HUFF_TABLE)";
          if (t <= 1 && mjh_std_huffman_table(which, t, &bits, &vals, &l) == MJH_OK && nv == l && memcmp(h->bits + 1, bits + 1, 16) == 0 &&
              memcmp(h->huffval, vals, (size_t)nv) == 0)
            continue;
          own = 1;
          p->huff_tables_given |= 1 << (2 * t + which);
          p->huff_bits[2 * t + which][0] = 0;
          memcpy(p->huff_bits[2 * t + which] + 1, h->bits + 1, 16);
          memset(p->huff_vals[2 * t + which], 0, 256);
          memcpy(p->huff_vals[2 * t + which], h->huffval, (size_t)nv);
        }
      }
    if (coding && cinfo->data_precision == 12 && !own) p->optimize_coding = 1, coding = 0;   /* (and nothing of the slots is read) */
  }
  if (p->trellis_quant && !p->optimize_coding && !p->arith_code)      /* (one component: coded, see above) */
    return "trellis without optimize_coding (the reference codes such an image with tables no pass made for it: its own djpeg rejects the colour files)";
  if (p->num_scans > 0 && p->optimize_coding && p->trellis_quant && no_pixels != 2) { p->trellis_stats_Ah = cinfo->Ah; p->trellis_stats_Al = cinfo->Al; }   /* SURVEY T15 */
  p->write_JFIF_header = cinfo->write_JFIF_header;
  return NULL;
}

/* what the reference's master would report as total_passes (jcmaster.c:1121-1139) */
static int reference_total_passes(j_compress_ptr cinfo, const mjh_params *p)
{
  const int nscans = p->num_scans > 0 ? p->num_scans : 1;
  int total = p->optimize_coding ? nscans * 2 : nscans;
  if (p->trellis_quant) {
    const int loops = p->trellis_num_loops > 0 ? p->trellis_num_loops : 1;
    const int bands = p->use_scans_in_trellis ? 2 : 1;   /* jcmaster.c:1129-1138 */
    total += p->optimize_coding ? 2 * bands * cinfo->num_components * loops : bands * cinfo->num_components * loops + 1;
  }
  return total;
}

/* common start of jpeg_start_compress (mode 0 pixels / 1 raw data) and jpeg_write_coefficients (mode 2) */
static void shim_begin(j_compress_ptr cinfo, boolean write_all_tables, int mode, jvirt_barray_ptr *coef_arrays)
{
  const char *why;
  shim_state *s;
  struct jpeg_marker_writer *mw;
  static __thread char whybuf[320];

  if (cinfo->global_state != CSTATE_START) ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state);
  mjh_shim_drop(cinfo);   /* a stale entry for this address (object destroyed behind our back) must never be found again */
  s = (shim_state *)calloc(1, sizeof(*s));
  if (!s) ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 0);
  s->reading_arena = -1;
  if (mode == 2) cinfo->input_components = 1;   /* transencode_master_selection jctrans.c:186 */
  why = capture_params(cinfo, &s->p, mode);
  if (!why && mjh_params_size() != sizeof(mjh_params))   /* (mjh_params grows at its end between versions: a stale library next to a new shim) */
    why = "libmozjpeg_hip.so was built with another mjh_params layout than this shim (rebuild both)";
  if (!why) {
    /* the encoder validates too (geometry limits, sampling factors ...) and is this object's until finish / abort */
    s->enc = cache_acquire(&s->p, pick_device());
    if (!s->enc) { snprintf(whybuf, sizeof(whybuf), "%s", mjh_last_error()); why = whybuf; }
  }
  if (why) {
    free_state(s);
    if (getenv("MOZJPEG_HIP_PASSTHROUGH")) {
      if (mode == 2) {
        wrcoef_fn next = (wrcoef_fn)NEXT_SYMBOL("jpeg_write_coefficients");
        if (next) { fprintf(stderr, "mozjpeg_hip: %s is outside the GPU path; MOZJPEG_HIP_PASSTHROUGH set, handing over to the host libjpeg\n", why); next(cinfo, coef_arrays); return; }
      } else {
        start_fn next = (start_fn)NEXT_SYMBOL("jpeg_start_compress");
        if (next) { fprintf(stderr, "mozjpeg_hip: %s is outside the GPU path; MOZJPEG_HIP_PASSTHROUGH set, handing over to the host libjpeg\n", why); next(cinfo, write_all_tables); return; }
      }
    }
    fprintf(stderr, "mozjpeg_hip: unsupported configuration (%s); no CPU fallback\n", why);
    ERREXIT(cinfo, JERR_NOT_COMPILED);
  }
  s->cinfo = cinfo;
  s->raw = mode == 1;
  s->coef_arrays = mode == 2 ? coef_arrays : NULL;
  s->total_passes = reference_total_passes(cinfo, &s->p);
  pthread_mutex_lock(&g_lock);
  s->next = g_states; g_states = s;
  pthread_mutex_unlock(&g_lock);
  /* from here on an error exit has to drop the state first (FAIL_WITH_STATE) */
  if (write_all_tables) jpeg_suppress_tables(cinfo, FALSE);
  (*cinfo->err->reset_error_mgr) ((j_common_ptr)cinfo);
  (*cinfo->dest->init_destination) (cinfo);
  /* marker writer object for jpeg_write_marker (lives in the image pool like the reference's) */
  mw = (struct jpeg_marker_writer *)(*cinfo->mem->alloc_small) ((j_common_ptr)cinfo, JPOOL_IMAGE, sizeof(*mw));
  mw->write_file_header = mw_nop; mw->write_frame_header = mw_nop; mw->write_scan_header = mw_nop;
  mw->write_file_trailer = mw_nop; mw->write_tables_only = mw_nop;
  mw->write_marker_header = mw_header; mw->write_marker_byte = mw_byte;
  cinfo->marker = mw;
  /* write_file_header jcmarker.c:649: SOI + JFIF APP0 (+ Adobe APP14) */
  emit_byte(cinfo, 0xFF); emit_byte(cinfo, 0xD8);
  s->header_bytes = 2;
  if (cinfo->write_JFIF_header) {   /* emit_jfif_app0 jcmarker.c:422-449 */
    unsigned char app0[18] = { 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 };
    app0[9] = cinfo->JFIF_major_version; app0[10] = cinfo->JFIF_minor_version; app0[11] = cinfo->density_unit;
    app0[12] = (unsigned char)(cinfo->X_density >> 8); app0[13] = (unsigned char)cinfo->X_density;
    app0[14] = (unsigned char)(cinfo->Y_density >> 8); app0[15] = (unsigned char)cinfo->Y_density;
    emit_bytes(cinfo, app0, sizeof(app0));
    s->header_bytes += 18;   /* the device file carries the default APP0 at the same place: skipped on output */
  }
  if (cinfo->write_Adobe_marker) {  /* emit_adobe_app14 jcmarker.c:452-486: version 100, flags 0, transform 1 = YCbCr, 2 = YCCK, 0 = anything else */
    unsigned char app14[16] = { 0xFF, 0xEE, 0, 14, 'A', 'd', 'o', 'b', 'e', 0, 100, 0, 0, 0, 0, 0 };
    app14[15] = cinfo->jpeg_color_space == JCS_YCbCr ? 1 : cinfo->jpeg_color_space == JCS_YCCK ? 2 : 0;
    emit_bytes(cinfo, app14, sizeof(app14));   /* (an RGB file from the device carries its own 16 bytes: skipped on output) */
  }
  {
    /* the geometry fields callers read back after jpeg_start_compress (initial_setup jcmaster.c:237-259);
     * tj3CompressFromYUVPlanes8 sizes its row buffers from width_in_blocks / max_*_samp_factor */
    int ci;
    jpeg_component_info *c;
    cinfo->max_h_samp_factor = cinfo->max_v_samp_factor = 1;
    for (ci = 0, c = cinfo->comp_info; ci < cinfo->num_components; ci++, c++) {
      if (c->h_samp_factor > cinfo->max_h_samp_factor) cinfo->max_h_samp_factor = c->h_samp_factor;
      if (c->v_samp_factor > cinfo->max_v_samp_factor) cinfo->max_v_samp_factor = c->v_samp_factor;
    }
    for (ci = 0, c = cinfo->comp_info; ci < cinfo->num_components; ci++, c++) {
      const long hd = (long)cinfo->max_h_samp_factor * DCTSIZE, vd = (long)cinfo->max_v_samp_factor * DCTSIZE;
      c->component_index = ci;
      c->DCT_scaled_size = DCTSIZE;
      c->width_in_blocks = (JDIMENSION)(((long)cinfo->image_width * c->h_samp_factor + hd - 1) / hd);
      c->height_in_blocks = (JDIMENSION)(((long)cinfo->image_height * c->v_samp_factor + vd - 1) / vd);
      c->downsampled_width = (JDIMENSION)(((long)cinfo->image_width * c->h_samp_factor + cinfo->max_h_samp_factor - 1) / cinfo->max_h_samp_factor);
      c->downsampled_height = (JDIMENSION)(((long)cinfo->image_height * c->v_samp_factor + cinfo->max_v_samp_factor - 1) / cinfo->max_v_samp_factor);
      c->component_needed = TRUE;
    }
    cinfo->total_iMCU_rows = (JDIMENSION)(((long)cinfo->image_height + cinfo->max_v_samp_factor * DCTSIZE - 1) /
                                          ((long)cinfo->max_v_samp_factor * DCTSIZE));
  }
  s->row_bytes = (size_t)cinfo->image_width * cinfo->input_components * (cinfo->data_precision == 12 ? 2 : 1);
  /* arrays requested from this object's memory manager get realised here, as in the reference (jinit_compress_master
   * jcinit.c:143 / transencode_master_selection jctrans.c:214): an application may have asked for its own before starting --
   * cjpeg's BMP and bottom-up Targa readers keep the whole picture in one (rdbmp.c:605-609, rdtarga.c) */
  (*cinfo->mem->realize_virt_arrays) ((j_common_ptr)cinfo);
  if (mode == 2) {
    /* (coefficients in: nothing to stage) */
  } else if (s->raw) {
    int ci, bad = 0;
    for (ci = 0; ci < cinfo->num_components; ci++) {
      jpeg_component_info *c = &cinfo->comp_info[ci];
      s->plane_pitch[ci] = (size_t)c->width_in_blocks * DCTSIZE * (cinfo->data_precision == 12 ? 2 : 1);
      s->planes[ci] = (unsigned char *)malloc(s->plane_pitch[ci] * c->height_in_blocks * DCTSIZE);
      if (!s->planes[ci]) bad = 1;
    }
    if (bad) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 0));
  } else {
    /* scanlines are written straight into the encoder's pinned staging buffer: no second host copy at finish */
    void *buf = NULL;
    size_t cap = 0;
    if (mjh_host_staging(s->enc, &buf, &cap) != MJH_OK || cap < s->row_bytes * cinfo->image_height) {
      fprintf(stderr, "mozjpeg_hip: %s\n", mjh_last_error());
      FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 0));
    }
    s->pixels = (unsigned char *)buf;
    if (batching_enabled()) s->bt = batcher_enter(&s->p, pick_device());
  }
  if (cinfo->progress != NULL) { cinfo->progress->completed_passes = 0; cinfo->progress->total_passes = s->total_passes; }
  cinfo->next_scanline = 0;
  cinfo->global_state = mode == 2 ? CSTATE_WRCOEFS : (s->raw ? CSTATE_RAW_OK : CSTATE_SCANNING);   /* jcapistd.c:62, jctrans.c:67 */
}

void jpeg_start_compress(j_compress_ptr cinfo, boolean write_all_tables)
{
  shim_begin(cinfo, write_all_tables, cinfo->raw_data_in ? 1 : 0, NULL);
}

/* jpeg_write_coefficients jctrans.c:44-68: start of a lossless re-encode (jpegtran).  All tables are written; the
 * virtual arrays may still be filled by the caller (jtransform_execute_transformation) until jpeg_finish_compress. */
void jpeg_write_coefficients(j_compress_ptr cinfo, jvirt_barray_ptr *coef_arrays)
{
  if (cinfo->master->num_scans_luma == 0) cinfo->master->optimize_scans = FALSE;
  shim_begin(cinfo, TRUE, 2, coef_arrays);
}

#define STAGE_ROWS 256
/* MOZJPEG_HIP_TIMING=1: seconds spent per phase, summed over threads, printed when the library is unloaded */
static int shim_timing = 0;
static double shim_t[4];        /* rows -> staging, submit, wait for the files, hand-over to the destination manager */
static unsigned long shim_images;
static pthread_mutex_t shim_tlock = PTHREAD_MUTEX_INITIALIZER;
static double shim_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void shim_acc(int what, double dt, int image) { pthread_mutex_lock(&shim_tlock); shim_t[what] += dt; shim_images += image; pthread_mutex_unlock(&shim_tlock); }
static void __attribute__((constructor)) shim_timing_init(void) { const char *v = getenv("MOZJPEG_HIP_TIMING"); shim_timing = v && atoi(v) > 0; }
static void __attribute__((destructor)) shim_report(void)
{
  if (shim_timing > 0 && shim_images)
    fprintf(stderr, "mozjpeg_hip timing: %lu images; per image: rows->staging %.3f ms, submit %.3f ms, wait %.3f ms, hand-over %.3f ms\n", shim_images,
            1e3 * shim_t[0] / shim_images, 1e3 * shim_t[1] / shim_images, 1e3 * shim_t[2] / shim_images, 1e3 * shim_t[3] / shim_images);
}

/* a row into the pinned staging buffer.  The buffer is only ever read by the device's copy engine, so the row is written
 * with non-temporal stores where the CPU has them (x86-64 AVX2): no read-for-ownership of the destination lines -- half
 * the DRAM traffic of a plain memcpy, which is what bounds many client threads feeding one GPU -- and the client's own
 * working set stays in its caches.  MOZJPEG_HIP_NT=0 keeps memcpy.  stage_fence() before the rows are handed on. */
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
static int stage_nt = -1;
__attribute__((target("avx2"))) static void stage_copy_nt(unsigned char *dst, const unsigned char *src, size_t n)
{
  size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
  if (head > n) head = n;
  if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
  for (; n >= 128; n -= 128, dst += 128, src += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i *)src), b = _mm256_loadu_si256((const __m256i *)(src + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i *)(src + 64)), d = _mm256_loadu_si256((const __m256i *)(src + 96));
    _mm256_stream_si256((__m256i *)dst, a); _mm256_stream_si256((__m256i *)(dst + 32), b);
    _mm256_stream_si256((__m256i *)(dst + 64), c); _mm256_stream_si256((__m256i *)(dst + 96), d);
  }
  if (n) memcpy(dst, src, n);
}
static void stage_copy(void *dst, const void *src, size_t n)
{
  if (stage_nt < 0) {
    const char *e = getenv("MOZJPEG_HIP_NT");
    stage_nt = (e ? atoi(e) != 0 : 1) && __builtin_cpu_supports("avx2");
  }
  if (stage_nt && n >= 1024) stage_copy_nt((unsigned char *)dst, (const unsigned char *)src, n);
  else memcpy(dst, src, n);
}
static void stage_fence(void) { if (stage_nt > 0) _mm_sfence(); }
#else
static void stage_copy(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }
static void stage_fence(void) { }
#endif

static JDIMENSION write_rows(j_compress_ptr cinfo, void **scanlines, JDIMENSION num_lines, int precision, const char *name)
{
  shim_state *s = find_state(cinfo, 0);
  JDIMENSION rows_left, i;
  if (!s) {
    write_fn next = (write_fn)NEXT_SYMBOL(name);
    if (next) return next(cinfo, (JSAMPARRAY)scanlines, num_lines);
    ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state);
  }
  if (cinfo->data_precision != precision) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_BAD_PRECISION, cinfo->data_precision));   /* jcapistd.c:96-97 */
  if (cinfo->global_state != CSTATE_SCANNING) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state));
  if (cinfo->next_scanline >= cinfo->image_height) WARNMS(cinfo, JWRN_TOO_MUCH_DATA);
  if (cinfo->progress != NULL) {
    cinfo->progress->pass_counter = (long)cinfo->next_scanline;
    cinfo->progress->pass_limit = (long)cinfo->image_height;
    (*cinfo->progress->progress_monitor) ((j_common_ptr)cinfo);
  }
  rows_left = cinfo->image_height - cinfo->next_scanline;
  if (num_lines > rows_left) num_lines = rows_left;
  {
    /* the rows go straight into the encoder's pinned staging buffer; every STAGE_ROWS rows the finished part is sent on its
     * way to the device, so that the PCIe copy runs under the client's production of the rest of the image */
    const double t0 = shim_timing ? shim_now() : 0.0;
    for (i = 0; i < num_lines; i++) {
      stage_copy(s->pixels + (size_t)(cinfo->next_scanline + i) * s->row_bytes, scanlines[i], s->row_bytes);
      if (((cinfo->next_scanline + i + 1) % STAGE_ROWS) == 0 && cinfo->next_scanline + i + 1 < cinfo->image_height) {
        stage_fence();
        (void)mjh_stage_commit(s->enc, (size_t)(cinfo->next_scanline + i + 1) * s->row_bytes);
      }
    }
    stage_fence();
    if (shim_timing) shim_acc(0, shim_now() - t0, 0);
  }
  cinfo->next_scanline += num_lines;
  return num_lines;
}

JDIMENSION jpeg_write_scanlines(j_compress_ptr cinfo, JSAMPARRAY scanlines, JDIMENSION num_lines)
{
  return write_rows(cinfo, (void **)scanlines, num_lines, 8, "jpeg_write_scanlines");
}

/* 12-bit twin (jpeglib.h:1070, J12SAMPLE = short): rows of 16-bit samples */
JDIMENSION jpeg12_write_scanlines(j_compress_ptr cinfo, J12SAMPARRAY scanlines, JDIMENSION num_lines)
{
  return write_rows(cinfo, (void **)scanlines, num_lines, 12, "jpeg12_write_scanlines");
}

/* jpeg_write_raw_data jcapistd.c:145-199: exactly one iMCU row of caller-made component planes per call
 * (data[ci] = v_samp_factor*8 row pointers of width_in_blocks*8 samples); the rows are staged and the whole
 * image goes to the GPU at jpeg_finish_compress through mjh_encode_planes_host. */
static JDIMENSION write_raw(j_compress_ptr cinfo, void ***data, JDIMENSION num_lines, int precision, const char *name)
{
  shim_state *s = find_state(cinfo, 0);
  JDIMENSION lines_per_iMCU_row, imcu;
  int ci;
  if (!s) {
    raw_fn next = (raw_fn)NEXT_SYMBOL(name);
    if (next) return next(cinfo, (JSAMPIMAGE)data, num_lines);
    ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state);
  }
  if (cinfo->data_precision != precision) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_BAD_PRECISION, cinfo->data_precision));
  if (cinfo->global_state != CSTATE_RAW_OK) FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state));
  if (cinfo->next_scanline >= cinfo->image_height) { WARNMS(cinfo, JWRN_TOO_MUCH_DATA); return 0; }
  if (cinfo->progress != NULL) {
    cinfo->progress->pass_counter = (long)cinfo->next_scanline;
    cinfo->progress->pass_limit = (long)cinfo->image_height;
    (*cinfo->progress->progress_monitor) ((j_common_ptr)cinfo);
  }
  lines_per_iMCU_row = (JDIMENSION)cinfo->max_v_samp_factor * DCTSIZE;
  if (num_lines < lines_per_iMCU_row) FAIL_WITH_STATE(cinfo, ERREXIT(cinfo, JERR_BUFFER_SIZE));
  imcu = cinfo->next_scanline / lines_per_iMCU_row;
  for (ci = 0; ci < cinfo->num_components; ci++) {
    jpeg_component_info *c = &cinfo->comp_info[ci];
    const JDIMENSION rows = (JDIMENSION)c->v_samp_factor * DCTSIZE, plane_rows = c->height_in_blocks * DCTSIZE;
    JDIMENSION r;
    for (r = 0; r < rows; r++) {
      const JDIMENSION dr = imcu * rows + r;
      if (dr >= plane_rows) break;      /* rows of dummy blocks below the image are never read (jccoefct.c:327-345) */
      memcpy(s->planes[ci] + (size_t)dr * s->plane_pitch[ci], data[ci][r], s->plane_pitch[ci]);
    }
  }
  cinfo->next_scanline += lines_per_iMCU_row;
  return lines_per_iMCU_row;
}

JDIMENSION jpeg_write_raw_data(j_compress_ptr cinfo, JSAMPIMAGE data, JDIMENSION num_lines)
{
  return write_raw(cinfo, (void ***)data, num_lines, 8, "jpeg_write_raw_data");
}

JDIMENSION jpeg12_write_raw_data(j_compress_ptr cinfo, J12SAMPIMAGE data, JDIMENSION num_lines)
{
  return write_raw(cinfo, (void ***)data, num_lines, 12, "jpeg12_write_raw_data");
}

static int encode_staged(j_compress_ptr cinfo, shim_state *s)
{
  if (s->coef_arrays) {
    const void *cf[MJH_MAX_COMPS] = { 0, 0, 0, 0 };
    size_t bpr[MJH_MAX_COMPS] = { 0, 0, 0, 0 };
    int ci;
    for (ci = 0; ci < cinfo->num_components && ci < MJH_MAX_COMPS; ci++) {
      jpeg_component_info *c = &cinfo->comp_info[ci];
      JDIMENSION r;
      s->planes[ci] = (unsigned char *)malloc((size_t)c->width_in_blocks * c->height_in_blocks * sizeof(JBLOCK));
      if (!s->planes[ci]) return MJH_ENOMEM;
      for (r = 0; r < c->height_in_blocks; r++) {   /* one block row at a time: always within the array's maxaccess */
        JBLOCKARRAY ba = (*cinfo->mem->access_virt_barray) ((j_common_ptr)cinfo, s->coef_arrays[ci], r, 1, FALSE);
        memcpy(s->planes[ci] + (size_t)r * c->width_in_blocks * sizeof(JBLOCK), ba[0], (size_t)c->width_in_blocks * sizeof(JBLOCK));
      }
      cf[ci] = s->planes[ci]; bpr[ci] = c->width_in_blocks;
    }
    return mjh_encode_coefficients_host(s->enc, cf, bpr, NULL, 1);
  }
  if (s->raw) {
    const void *pl[MJH_MAX_COMPS] = { 0, 0, 0, 0 };
    size_t pitch[MJH_MAX_COMPS] = { 0, 0, 0, 0 };
    int pw[MJH_MAX_COMPS] = { 0, 0, 0, 0 }, ph[MJH_MAX_COMPS] = { 0, 0, 0, 0 }, ci;
    for (ci = 0; ci < cinfo->num_components && ci < MJH_MAX_COMPS; ci++) {
      pl[ci] = s->planes[ci]; pitch[ci] = s->plane_pitch[ci];
      pw[ci] = (int)cinfo->comp_info[ci].width_in_blocks * DCTSIZE;
      ph[ci] = (int)cinfo->comp_info[ci].height_in_blocks * DCTSIZE;
    }
    return mjh_encode_planes_host(s->enc, pl, pitch, NULL, pw, ph, 1);
  }
  return mjh_encode_host(s->enc, s->pixels, s->row_bytes, s->row_bytes * cinfo->image_height, 1);
}

/* an encoder error ends the compression: message on stderr, state dropped, the libjpeg error code that fits */
static void encoder_failed(j_compress_ptr cinfo)
{
  const char *msg = mjh_last_error();
  const int bad_coef = strstr(msg, "JERR_BAD_DCT_COEF") != NULL;
  fprintf(stderr, "mozjpeg_hip: %s\n", msg);
  mjh_shim_drop(cinfo);
  if (bad_coef) ERREXIT(cinfo, JERR_BAD_DCT_COEF);
  ERREXIT1(cinfo, JERR_OUT_OF_MEMORY, 0);
}

void jpeg_finish_compress(j_compress_ptr cinfo)
{
  shim_state *s = find_state(cinfo, 0);
  size_t n = 0;
  const void *base = NULL;
  const mjh_result *res = NULL;
  unsigned char *copy = NULL;
  const unsigned char *file;
  int cnt = 0, pass, batched = 0, arena = 0;
  double t_wait0 = 0.0;
  if (!s) {
    finish_fn next = (finish_fn)NEXT_SYMBOL("jpeg_finish_compress");
    if (next) { next(cinfo); return; }
    ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state);
  }
  if (cinfo->global_state == CSTATE_SCANNING || cinfo->global_state == CSTATE_RAW_OK) {
    if (cinfo->next_scanline < cinfo->image_height) FAIL_WITH_STATE(cinfo, ERREXIT(cinfo, JERR_TOO_LITTLE_DATA));
  } else if (cinfo->global_state != CSTATE_WRCOEFS)
    FAIL_WITH_STATE(cinfo, ERREXIT1(cinfo, JERR_BAD_STATE, cinfo->global_state));   /* jcapimin.c:180-189 */
  if (s->bt && !s->raw && !s->coef_arrays) {       /* pixel input with company: one batch with whoever else is finishing now */
    int contended;
    pthread_mutex_lock(&s->bt->m);
    contended = s->bt->contended;
    pthread_mutex_unlock(&s->bt->m);
    if (contended > 0) {
      const double t0 = shim_timing ? shim_now() : 0.0;
      const int rc = batched_encode(s, &file, &n, &arena);
      if (shim_timing) shim_acc(2, shim_now() - t0, 1);
      if (rc == MJH_OK) { batched = 1; s->reading_arena = arena; }
      else if (rc != BATCH_PRIVATE) encoder_failed(cinfo);
    }
  }
  if (!batched) {
    const double t0 = shim_timing ? shim_now() : 0.0;
    if (encode_staged(cinfo, s) != MJH_OK) encoder_failed(cinfo);
    if (shim_timing) shim_acc(1, shim_now() - t0, 1);
  }
  t_wait0 = shim_timing ? shim_now() : 0.0;
  /* the remaining passes run on the device; a progress monitor sees them go by (jcmaster.c:708-713) */
  if (cinfo->progress != NULL) {
    for (pass = 1; pass < s->total_passes; pass++) {
      cinfo->progress->completed_passes = pass;
      cinfo->progress->total_passes = s->total_passes;
      cinfo->progress->pass_counter = 0;
      cinfo->progress->pass_limit = (long)cinfo->total_iMCU_rows;
      (*cinfo->progress->progress_monitor) ((j_common_ptr)cinfo);
    }
  }
  if (batched) {
    /* (file / n point into the batch encoder's result arena) */
  } else if (mjh_collect(s->enc, 0, &base, &res, &cnt) == MJH_OK && cnt == 1) {   /* zero-copy: the file lies in pinned host memory */
    file = (const unsigned char *)base + res[0].offset;
    n = (size_t)res[0].size;
  } else if (mjh_get_jpeg_size(s->enc, 0, &n) == MJH_OK && (copy = (unsigned char *)malloc(n)) != NULL &&
             mjh_get_jpeg(s->enc, 0, copy, n, &n) == MJH_OK) {
    file = copy;
  } else {
    free(copy);
    encoder_failed(cinfo);
    return;
  }
  if (!batched) fetch_end_tables(s, s->enc, 0);
  if (shim_timing) { const double t = shim_now(); shim_acc(2, t - t_wait0, 0); t_wait0 = t; }
  /* the device wrote a complete file; SOI(+APP0) went out in jpeg_start_compress already */
  {
    /* what the DEVICE wrote in front of DQT: SOI, APP0 when asked for, and always an Adobe APP14 for RGB output (build_prefix);
     * the markers the application asked for went out from the cinfo flags in jpeg_start_compress (an application may clear
     * write_Adobe_marker for JCS_RGB, which the reference honours: the device's APP14 is dropped all the same) */
    const size_t skip = (size_t)(2 + (cinfo->write_JFIF_header ? 18 : 0) + (cinfo->jpeg_color_space == JCS_RGB ? 16 : 0));
    hand_over_file(cinfo, s, file + skip, file + n);     /* (table markers under the object's sent_table flags) */
  }
  free(copy);
  if (shim_timing) shim_acc(3, shim_now() - t_wait0, 0);
  mjh_shim_drop(cinfo);
  (*cinfo->dest->term_destination) (cinfo);
  jpeg_abort((j_common_ptr)cinfo);   /* releases JPOOL_IMAGE, global_state = CSTATE_START (jcapimin.c:228) */
}

#ifndef MJH_STANDALONE
/* ---- abort / destroy hooks (preload build): drop our state, then let the library behind us do its part -------------- */
typedef void (*common_fn)(j_common_ptr);
typedef void (*compress_fn)(j_compress_ptr);

void jpeg_abort(j_common_ptr cinfo)
{
  common_fn next = (common_fn)dlsym(RTLD_NEXT, "jpeg_abort");
  mjh_shim_drop(cinfo);
  if (next) next(cinfo);
}

void jpeg_destroy(j_common_ptr cinfo)
{
  common_fn next = (common_fn)dlsym(RTLD_NEXT, "jpeg_destroy");
  mjh_shim_drop(cinfo);
  if (next) next(cinfo);
}

void jpeg_abort_compress(j_compress_ptr cinfo)
{
  compress_fn next = (compress_fn)dlsym(RTLD_NEXT, "jpeg_abort_compress");
  mjh_shim_drop(cinfo);
  if (next) next(cinfo);
}

void jpeg_destroy_compress(j_compress_ptr cinfo)
{
  compress_fn next = (compress_fn)dlsym(RTLD_NEXT, "jpeg_destroy_compress");
  mjh_shim_drop(cinfo);
  if (next) next(cinfo);
}
#endif
