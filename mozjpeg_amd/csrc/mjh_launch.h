// mjh_launch.h -- host-callable launch wrappers implemented in mjh_kernels.hip
#ifndef MJH_LAUNCH_H
#define MJH_LAUNCH_H
#include <hip/hip_runtime.h>
#include "mjh_internal.h"

void mjh_launch_import_coefs(const MjhConst &C, const MjhCoefSrc &S, void *coef_q, void *meta, int n, hipStream_t s);
void mjh_launch_import_planes(const MjhConst &C, const MjhPlaneSrc &S, void *planes, int n, hipStream_t s);
void mjh_launch_color(const MjhConst &C, const void *pix, size_t row_pitch, size_t img_stride, void *planes, int n, hipStream_t s);
void mjh_launch_dct(const MjhConst &C, const MjhQuant *Q, const void *planes, void *uq, void *q, float *lambda,
                    MjhHuffTable *stat_tabs, int spi, const int stat_slot[4], uint8_t *nq8, int n, hipStream_t s, int fastdiv = 0, int ifast = 0);   // ifast: JDCT_IFAST (its own kernel); fastdiv: every table in use has q <= 255 (MjhQuant.mdiv)
// nzmask != nullptr (here and in mjh_launch_encode / mjh_launch_trellis_ac): the AC planes hold COMPACT records (plane i+1 = the block's i-th
// non-zero value in position order, nzmask = its non-zero positions) instead of one plane per position
void mjh_launch_stats_ac(const MjhConst &C, const void *q, const unsigned long long *nzmask, MjhHuffTable *tabs, int spi, const int slot[4], int count_dummies, int n, hipStream_t s);
void mjh_launch_stats_dc(const MjhConst &C, const void *q, MjhHuffTable *tabs, int spi, const int slot[4], int mcu_order, const int comp_restart[4], int n, hipStream_t s);
void mjh_launch_gen_tables(MjhHuffTable *tabs, int spi, const int *slots, int nslots, int n, hipStream_t s);
void mjh_launch_trellis_ac(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, MjhHuffTable *tabs, int spi, const int ac_slot[4], const float *lambda,
                           unsigned *worklist, unsigned *worklist2, void *dense, unsigned dense_cap, int variant,
                           int Ss, int Se, void *eob_cost, int *eob_has, unsigned long long *nzmask, int qstride, int n, hipStream_t s,
                           uint8_t *nq8 = nullptr, int v3_passes = 0, int fastdiv = 0, hipEvent_t after_first_tier = nullptr, hipEvent_t after_first_tier2 = nullptr,
                           int chunks = 1, hipStream_t side = nullptr, hipEvent_t *ev_chunk = nullptr, int count_mask = 0);   // count_mask: the heavy-block counts come from the tiles with (tile & count_mask) == 0   // chunks > 1: the tile-sorted tier over that many image ranges, the general tiers of all but the last on `side` (ev_chunk[chunks] events)   // after_first_tier: recorded behind the tile-sorted kernel, in front of the general tiers   // v3_passes > 0 (plain compact pass): the tile-sorted kernel with that many passes per tile
// trellis_eob_opt: the block-row pass behind a (band-limited) AC trellis; eob_cost / eob_has as written by mjh_launch_trellis_ac
void mjh_launch_trellis_eob_chain(const MjhConst &C, void *q, const MjhHuffTable *tabs, int spi, const int ac_slot[4], const void *eob_cost, const int *eob_has,
                                  int Ss, int Se, int n, hipStream_t s);
// trellis_q_opt: sums[image][4 tables][64][2] += over all blocks; new entries patched into the DQT bytes of the finished files
void mjh_launch_qopt_accumulate(const MjhConst &C, const void *uq, const void *q, void *sums, int n, hipStream_t s);
void mjh_launch_qopt_update(void *sums, MjhQuant *Q, int n, hipStream_t s);   // Q: one MjhQuant per image
// the DQT segment(s) of the finished files rebuilt from every image's final tables ([dqt_start, sof_off) as first written; tabs: the
// tables in marker order); shrinks the file and fixes SOF0/SOF1 when a 16-bit table has become an 8-bit one
void mjh_launch_qopt_fix(const MjhQuant *Q, void *out, size_t out_stride, unsigned *sizes, int dqt_start, int sof_off, const int *tabs, int ntab,
                         int multi, int baseline_capable, int n, hipStream_t s);
// window_ok: every component's DC quantizer step 8q is >= 40 (candidate values are then never clamped: the sliding-window kernel applies)
void mjh_launch_trellis_dc(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, const MjhHuffTable *tabs, int spi, const int dc_slot[4], const float *lambda, void *back, int n, hipStream_t s,
                           int window_ok = 0, int chain0 = 0, int chain1 = -1);   // chains [chain0, chain1) of every image (component-major, one per iMCU row); -1: all
void mjh_launch_encode(const MjhConst &C, const void *q, const unsigned long long *nzmask, const MjhHuffTable *tabs, int spi, const int dc_slot[4], const int ac_slot[4],
                       void *len16, void *off32, unsigned *sums, int chunks_per_image, unsigned *totals,
                       unsigned *stream, size_t stream_words_per_image, void *meta,
                       unsigned *seg_x, unsigned *seg_E, unsigned *seg_sums, unsigned *seg_totals, unsigned *mpos, int nseg,
                       int n, hipStream_t s);
void mjh_launch_header(const void *prefix, int prefix_len, const void *sos, int sos_len, const MjhHuffTable *tabs, int spi,
                       const int dht_slots[8], const int dht_ids[8], int ndht, int multi_dht, void *out, size_t out_stride, void *meta, int n, hipStream_t s, const unsigned *append_sizes = nullptr);   // append_sizes: write over the EOI of the files so far (later scans of a sequential script)
void mjh_launch_stuff(const unsigned *stream, size_t stream_words_per_image, const unsigned *totals, unsigned *ffsums, int ff_chunks_per_image,
                      unsigned *ff_totals, void *out, size_t out_stride, void *meta, unsigned *sizes, const unsigned *mpos, int nseg, int n, hipStream_t s);
void mjh_launch_pack_results(const void *out, size_t out_stride, const unsigned *sizes, const void *meta, const void *prog_ctl, int n,
                             void *dst, size_t cap, void *table, hipStream_t s);
void mjh_launch_spin(unsigned long long ticks_100mhz, hipStream_t s);   // one wave busy for that long (hardware-queue probe)
void mjh_launch_gen_tables_list(MjhHuffTable *tabs, int spi, const int *d_slots, int nslots, int n, hipStream_t s);
// progressive mode (mjh_prog.hip)
void mjh_launch_prog_reset(void *ctl, int nscans, int n, hipStream_t s);
void mjh_launch_prog_stats(const MjhConst &C, const void *scans, const int *list, int nlist, void *ctl, const void *q,
                           MjhHuffTable *tabs, int spi, unsigned *mpos, int mpos_per_image, int n, hipStream_t s);
void mjh_launch_prog_stats_par(const MjhConst &C, const void *scans, const int *list, int nlist, void *ctl, const void *q,
                               MjhHuffTable *tabs, int spi, const MjhProgPE &pe, bool any_refine, const unsigned long long *nzmask, int n, hipStream_t s,
                               int nacf);   // the list starts with its nacf first-pass AC scans
void mjh_launch_prog_encode(const MjhConst &C, const void *scans, const int *list, int nlist, const int *seq_list, int nseq,
                            const int *par_list, int npar, const MjhProgPE &pe, void *ctl, const void *q,
                            MjhHuffTable *tabs, int spi, unsigned *pool, size_t pool_words, const void *frame_hdr, int frame_hdr_len,
                            int multi_dht, void *outpool, size_t out_bytes, unsigned *mpos, int mpos_per_image, unsigned *ffsums, const unsigned long long *nzmask, int n, hipStream_t s,
                            hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, int nacf);
void mjh_launch_scan16(const void *len16, int n_per, unsigned *sums, int chunks, unsigned *totals, unsigned *off32, int npairs, hipStream_t s);
void mjh_launch_prog_select(void *ctl, int ncomp, int phase, int dc_scan_opt_mode, int n, hipStream_t s);
void mjh_launch_prog_concat(const void *ctl, const void *file_hdr, int file_hdr_len, const void *outpool, size_t out_bytes,
                            void *out, size_t out_stride, unsigned *sizes, int n, hipStream_t s);
bool mjh_trellis_dc_speculative_ok(const MjhConst &C, int window_ok);
void mjh_launch_trellis_dc_speculative(const MjhConst &C, const MjhQuant *Q, const void *uq, void *q, const MjhHuffTable *tabs, int spi, const int dc_slot[4], const float *lambda,
                                       void *back9, int *jfin, void *qspec, int n, hipStream_t s);
// arithmetic entropy coding (mjh_arith.hip)
void mjh_launch_arith_scans(const MjhConst &C, const void *scans, const int *list, int nlist, void *ctl, const void *q,
                            const uint8_t *frame_hdr, int frame_hdr_len, const uint8_t *file_hdr, int file_hdr_len,
                            uint8_t *out, size_t out_stride, unsigned *sizes, int whole_blocks, int mode, int n, hipStream_t s);
void mjh_launch_arith_layout(void *ctl, const uint8_t *file_hdr, int file_hdr_len, uint8_t *out, size_t out_stride, unsigned *sizes, int n, hipStream_t s);
void mjh_launch_trellis_arith(const MjhConst &C, const MjhQuant *Q, int qstride, const void *uq, void *q, const float *lambda, const void *rate_tab, void *back,
                              int Ss, int Se, int quant_dc, float delta_dc_weight, int restart_blocks, int prog_file, int n, hipStream_t s);
#endif
