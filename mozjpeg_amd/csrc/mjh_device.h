// mjh_device.h -- device helpers shared by the gfx950 kernel files (mjh_kernels.hip, mjh_prog.hip)
#ifndef MJH_DEVICE_H
#define MJH_DEVICE_H
// MJH_DIVERGENT_SCOPE marks a divergent region (`if (act) { ... }`) that holds a cross-lane operation.  Nothing on the device;
// the host emulator (tools/simt, test infrastructure) needs it to know that the lanes inside go before the lanes that
// skipped the region and already wait where the wave reconverges.
// MJH_WAVE_GROUPS(16) at the top of a kernel: its wave holds independent groups of 16 lanes (one chain per DPP row, cross-lane
// operations never leave the row) that follow their own control flow.  Nothing on the device either.
#ifndef MJH_SIMT_HOST
#define MJH_DIVERGENT_SCOPE
#define MJH_WAVE_GROUPS(n) ((void)0)
#define MJH_WAVE_SYNC() ((void)0)
#define MJH_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)      // the instruction scheduler moves nothing across (the emulator: nothing)
#endif
// MJH_WAVE_SYNC(): the lanes of a wave execute in lock step, so "every lane reads an LDS word, then lane 0 overwrites it" needs
// no barrier on the device; the emulator runs the lanes one after the other between cross-lane operations and needs the point
// between the reads and the write marked.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mjh_internal.h"

__device__ __forceinline__ int bitlen(unsigned v) { return 32 - __clz((int)v); }  // JPEG_NBITS; clz(0)=32

// 24-bit multiply (full rate on CDNA; the 32-bit v_mul_lo_u32 is quarter rate): exact low 32 bits of the product for
// operands in [-2^23, 2^23) -- raw coefficients (|x| <= 2^15), quantizer steps 8q (< 2^20), candidates (<= 1023) and the
// differences cand*8q - x (|.| < 2^21) all are
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
// (float)(v * v) for |v| < 2^15 without the integer multiply: (float)v is exact, the float product rounds the exact square once,
// as the conversion of the integer square does -- v_cvt + v_mul_f32 (4 + 2 issue cycles) instead of v_mul_i32_i24 + v_cvt (4 + 4;
// profiles/r04a_valu_rate_summary.md).  Used by the opt-in kernels only (mjh_sorted.hip: record mode and tile-sorted planes): the default kernels keep
// the machine code they were validated with.
__device__ __forceinline__ float squaref(int v) { const float f = (float)v; return f * f; }

// exact floor(n/d) for 0 <= n < 2^23, 1 <= d < 2^23, rcp = RN(1/d)
__device__ __forceinline__ int udiv_exact(int n, int d, float rcp)
{
  int q = (int)((float)n * rcp);
  int r = n - mul24(q, d);
  if (r < 0) q--; else if (r >= d) q++;
  return q;
}


// exact floor(n / d) for 0 <= n < 2^16 through the per-entry constants of MjhQuant (mdiv, sdiv): one shift + one 24-bit multiply-high
__device__ __forceinline__ int udiv_mh(int n, int sh, unsigned m)
{
  return (int)__umulhi(((unsigned)n << sh) & 0xFFFFFFu, m & 0xFFFFFFu);
}

__device__ __forceinline__ int dc_source_block(const MjhComp &cc, int r, int c)
{
  if (r >= cc.hib) { c = (c / cc.h) * cc.h + cc.h - 1; r = cc.hib - 1; }
  if (c > cc.wib - 1) c = cc.wib - 1;
  return r * cc.wib + c;
}

// previous block of the same component in interleaved MCU order; returns false if there is
// none (first MCU of the scan or of a restart interval).  (pr,pc) in padded coordinates.
__device__ __forceinline__ bool mcu_prev_block(const MjhConst &C, const MjhComp &cc, int r, int c, int &pr, int &pc)
{
  const int xi = c % cc.h, yi = r % cc.v;
  if (xi > 0) { pr = r; pc = c - 1; return true; }
  if (yi > 0) { pr = r - 1; pc = c + cc.h - 1; return true; }
  const int m = (r / cc.v) * C.mcus_per_row + c / cc.h;
  if (m == 0) return false;
  if (C.restart_interval && (m % C.restart_interval) == 0) return false;
  const int pm = m - 1;
  const int pmy = pm / C.mcus_per_row, pmx = pm - pmy * C.mcus_per_row;
  pr = pmy * cc.v + cc.v - 1;
  pc = pmx * cc.h + cc.h - 1;
  return true;
}


__device__ __forceinline__ unsigned block_reduce_256(unsigned v, unsigned *sh)
{
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  const unsigned tot = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return tot;
}

// exclusive scan of one value per thread within a 256-thread block; returns exclusive prefix
__device__ __forceinline__ unsigned block_excl_scan_256(unsigned v, unsigned *sh, unsigned *total)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  unsigned base = 0;
  for (int i = 0; i < w; i++) base += sh[i];
  const unsigned tot = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  if (total) *total = tot;
  return base + inc - v;
}

// inclusive prefix sum over the 64 lanes of a wave: four DPP row shifts (lanes in front of the row read 0), then the totals
// of the rows in front (lanes 15, 31, 47) through scalar registers
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v)
{
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
  const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47);
  const int row = (int)(threadIdx.x & 63) >> 4;
  x += row == 0 ? 0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : t0 + t1 + t2));
  return (unsigned)x;
}

// exclusive prefix within a 256-thread workgroup with ONE barrier: sh holds 8 words, `parity` alternates between
// consecutive calls (a wave cannot be two calls ahead of another: every call has its barrier)
__device__ __forceinline__ unsigned block_excl_scan_256_1b(unsigned v, unsigned *sh, int parity, unsigned *total)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned inc = wave_incl_scan(v);
  if (lane == 63) sh[parity * 4 + w] = inc;
  __syncthreads();
  const unsigned a = sh[parity * 4], b = sh[parity * 4 + 1], c = sh[parity * 4 + 2], d = sh[parity * 4 + 3];
  *total = a + b + c + d;
  return (w == 0 ? 0u : (w == 1 ? a : (w == 2 ? a + b : a + b + c))) + inc - v;
}


// bit writer: ORs big-endian bit strings into a zero-initialised word array
struct BitWriter {
  unsigned *words;       // stream base (32-bit words, bytes are big-endian inside the stream)
  unsigned long long acc;
  int nacc;              // valid bits in acc (low end)
  unsigned widx;
  __device__ __forceinline__ void init(unsigned *w, unsigned bitoff) { words = w; widx = bitoff >> 5; nacc = (int)(bitoff & 31); acc = 0; }
  __device__ __forceinline__ void put(unsigned code, int n)
  {
    acc = (acc << n) | (unsigned long long)(code & ((1u << n) - 1u));
    nacc += n;
    if (nacc >= 32) {
      const unsigned w = (unsigned)(acc >> (nacc - 32));
      atomicOr(&words[widx], __builtin_bswap32(w));
      widx++;
      nacc -= 32;
    }
  }
  // a symbol's code (e = size << 16 | code) and its nbits value bits in ONE accumulator step (size + nbits <= 16 + 16): the same bits in
  // the same order as put(code, size); put(val, nbits) -- half the shifts / compares, and half the dependent chain through acc
  __device__ __forceinline__ void put_sym(unsigned e, unsigned val, int nbits)
  {
    const int n = (int)(e >> 16) + nbits;
    acc = (acc << n) | (unsigned long long)(((e & 0xFFFFu) << nbits) | (val & ((1u << nbits) - 1u)));
    nacc += n;
    if (nacc >= 32) {
      const unsigned w = (unsigned)(acc >> (nacc - 32));
      atomicOr(&words[widx], __builtin_bswap32(w));
      widx++;
      nacc -= 32;
    }
  }
  __device__ __forceinline__ void flush()
  {
    if (nacc > 0) {
      const unsigned w = (unsigned)(acc << (32 - nacc));
      atomicOr(&words[widx], __builtin_bswap32(w));
    }
  }
};


// One round of the byte-stuffing passes (256 threads x 8 stream words -> up to 16 KB of output): the stuffed bytes are laid
// out in LDS exactly as they will lie in memory (same alignment mod 4) and leave as whole, coalesced 32-bit stores -- a
// thread's own bytes start at an arbitrary byte offset, so writing them directly costs one scattered byte store each.
// dst0 / total: first output byte of the round and the round's output length (uniform); my_dst: where this thread's first
// byte goes; nvalid: how many of its 32 input bytes exist; keep: bit (4 * i + b) set = byte b of word i is the 0xFF of a
// marker (no zero byte behind it).  lds: STUFF_LDS_WORDS words.  Two barriers.
#define STUFF_LDS_WORDS (4096 + 2)
__device__ __forceinline__ void stuff_store_round(uint8_t *__restrict__ o, unsigned dst0, unsigned total, unsigned my_dst, const unsigned (&w)[8],
                                                  int nvalid, unsigned keep, unsigned *lds)
{
  uint8_t *l8 = reinterpret_cast<uint8_t *>(lds);
  const unsigned shift = (unsigned)((uintptr_t)(o + dst0) & 3u);
  unsigned at = my_dst - dst0 + shift;
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const unsigned byte = (w[i] >> (8 * b)) & 0xFFu;    // little-endian word = stream byte order
      if (4 * i + b < nvalid) {
        l8[at++] = (uint8_t)byte;
        if (byte == 0xFFu && !((keep >> (4 * i + b)) & 1u)) l8[at++] = 0;
      }
    }
  }
  __syncthreads();
  uint8_t *g = o + dst0 - shift;                           // 4-byte aligned
  const unsigned end = shift + total, nw = (end + 3u) >> 2;
  for (unsigned k = threadIdx.x; k < nw; k += 256) {
    if (4 * k >= shift && 4 * k + 4 <= end) reinterpret_cast<unsigned *>(g)[k] = lds[k];
    else
      for (unsigned b = 4 * k; b < 4 * k + 4; b++) if (b >= shift && b < end) g[b] = l8[b];
  }
  __syncthreads();
}

// the same writer for a window of the stream kept in LDS (the words become ds_or) or, LDSW = false, for the stream itself
template <bool LDSW>   // (two instantiations so that the window's words become ds_or and the direct path's global atomics)
struct BitSink {
  unsigned *words;
  unsigned long long acc;
  int nacc;
  unsigned widx;
  __device__ __forceinline__ void init(unsigned *w, unsigned bitoff) { words = w; widx = bitoff >> 5; nacc = (int)(bitoff & 31); acc = 0; }
  __device__ __forceinline__ void put(unsigned code, int n)
  {
    acc = (acc << n) | (unsigned long long)(code & ((1u << n) - 1u));
    nacc += n;
    if (nacc >= 32) {
      const unsigned w = (unsigned)(acc >> (nacc - 32));
      atomicOr(&words[widx], __builtin_bswap32(w));
      widx++;
      nacc -= 32;
    }
  }
  // a symbol's code (e = size << 16 | code) and its nbits value bits in ONE accumulator step (size + nbits <= 16 + 16): the same bits in
  // the same order as put(code, size); put(val, nbits) -- half the shifts / compares, and half the dependent chain through acc
  __device__ __forceinline__ void put_sym(unsigned e, unsigned val, int nbits)
  {
    const int n = (int)(e >> 16) + nbits;
    acc = (acc << n) | (unsigned long long)(((e & 0xFFFFu) << nbits) | (val & ((1u << nbits) - 1u)));
    nacc += n;
    if (nacc >= 32) {
      const unsigned w = (unsigned)(acc >> (nacc - 32));
      atomicOr(&words[widx], __builtin_bswap32(w));
      widx++;
      nacc -= 32;
    }
  }
  __device__ __forceinline__ void flush()
  {
    if (nacc > 0) atomicOr(&words[widx], __builtin_bswap32((unsigned)(acc << (32 - nacc))));
  }
  __device__ __forceinline__ unsigned bitpos() const { return widx * 32u + (unsigned)nacc; }
};

#endif
