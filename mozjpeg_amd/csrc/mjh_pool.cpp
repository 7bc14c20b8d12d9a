// mjh_pool.cpp -- one process driving several GPUs ("a batch of independent images is partitioned across
// the GPUs of one node, no exchange step").  A pool owns one encoder per device and one host thread per device; a batch
// handed to mjh_pool_encode_host is dealt round-robin (image i -> device i mod N, the same rule as the one-process-per-GPU
// launch of bench.py / mozjpeg_amd.shard), every device pipelines its share through the double-buffered host path
// (mjh_encode_host / mjh_collect), and the finished files come back in the caller's image order.  Built on the public
// ABI only; nothing is exchanged between devices.  A single libjpeg / TurboJPEG client that holds many images -- the
// reference's tjbench-style callers (turbojpeg-mp.c:69-136 loops tj3Compress8 over its tiles) -- scales this way
// without becoming N processes.
#include <stdio.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/mozjpeg_hip.h"

int mjh_internal_fail(int code, const char *msg);   // mjh_encoder.cpp: sets mjh_last_error()

struct mjh_pool {
  std::vector<mjh_encoder *> enc;
  std::vector<int> dev;
  int max_batch = 0;
  std::vector<std::vector<uint8_t>> store;   // per device: the files of its share, back to back
  std::vector<const uint8_t *> ptrs;         // per image of the last call
  std::vector<size_t> sizes;
  std::string error;
};

extern "C" int mjh_pool_create(const mjh_params *p, int max_batch_per_device, const int *devices, int ndevices, mjh_pool **out)
{
  if (!p || !out || max_batch_per_device < 1) return mjh_internal_fail(MJH_EINVAL, "mjh_pool_create: bad arguments");
  std::vector<int> devs;
  if (devices && ndevices > 0) devs.assign(devices, devices + ndevices);
  else { const int n = mjh_device_count(); for (int i = 0; i < n; i++) devs.push_back(i); }
  if (devs.empty()) return mjh_internal_fail(MJH_EHIP, "mjh_pool_create: no HIP device available (libmozjpeg_hip has no CPU fallback)");
  mjh_pool *pl = new mjh_pool;
  pl->max_batch = max_batch_per_device;
  for (int d : devs) {
    mjh_encoder *e = nullptr;
    const int rc = mjh_encoder_create(p, max_batch_per_device, d, &e);
    if (rc != MJH_OK) { for (mjh_encoder *x : pl->enc) mjh_encoder_destroy(x); delete pl; return rc; }   // mjh_last_error() holds the reason
    pl->enc.push_back(e); pl->dev.push_back(d);
  }
  pl->store.resize(pl->enc.size());
  *out = pl;
  return MJH_OK;
}

extern "C" void mjh_pool_destroy(mjh_pool *pl)
{
  if (!pl) return;
  for (mjh_encoder *e : pl->enc) mjh_encoder_destroy(e);
  delete pl;
}

extern "C" int mjh_pool_device_count(const mjh_pool *pl) { return pl ? (int)pl->enc.size() : 0; }
extern "C" const char *mjh_pool_last_error(const mjh_pool *pl) { return pl ? pl->error.c_str() : "null pool"; }

namespace {
struct Share { std::vector<int> images; std::vector<size_t> off, len; int rc = MJH_OK; std::string err; };

// device d's share: gather its (strided) images into the encoder's pinned staging buffer, queue them, and pick the
// previous batch up while this one runs
void run_share(mjh_pool *pl, int d, const uint8_t *pixels, size_t row_pitch, size_t image_stride, size_t row_bytes, int rows, Share *sh)
{
  mjh_encoder *e = pl->enc[d];
  // this thread gathers the device's images into its pinned staging buffer: it runs on the CPUs of the device's NUMA node
  // (the staging buffer was pinned there), so the copy's stores stay on the socket the DMA engine reads from
  (void)mjh_bind_thread_to_device(pl->dev[d]);
  std::vector<uint8_t> &st = pl->store[d];
  st.clear();
  const int total = (int)sh->images.size();
  sh->off.assign(total, 0); sh->len.assign(total, 0);
  auto pick = [&](int age, int first, int count) -> int {
    const void *base; const mjh_result *res; int cnt = 0;
    const int rc = mjh_collect(e, age, &base, &res, &cnt);
    if (rc != MJH_OK) return rc;
    if (cnt != count) return mjh_internal_fail(MJH_EINVAL, "pool: a batch came back with a different number of files than was queued");
    for (int i = 0; i < cnt; i++) {
      sh->off[first + i] = st.size(); sh->len[first + i] = (size_t)res[i].size;
      st.insert(st.end(), (const uint8_t *)base + res[i].offset, (const uint8_t *)base + res[i].offset + res[i].size);
    }
    return MJH_OK;
  };
  int prev_first = -1, prev_count = 0;
  for (int first = 0; first < total && sh->rc == MJH_OK; first += pl->max_batch) {
    const int count = total - first < pl->max_batch ? total - first : pl->max_batch;
    void *stage; size_t stage_bytes;
    int rc = mjh_host_staging(e, &stage, &stage_bytes);
    if (rc == MJH_OK) {
      for (int i = 0; i < count; i++) {
        const uint8_t *src = pixels + (size_t)sh->images[first + i] * image_stride;
        uint8_t *dst = (uint8_t *)stage + (size_t)i * row_bytes * rows;
        if (row_pitch == row_bytes) memcpy(dst, src, row_bytes * rows);
        else for (int r = 0; r < rows; r++) memcpy(dst + (size_t)r * row_bytes, src + (size_t)r * row_pitch, row_bytes);
      }
      rc = mjh_encode_host(e, stage, row_bytes, row_bytes * rows, count);
    }
    if (rc == MJH_OK && prev_first >= 0) rc = pick(1, prev_first, prev_count);
    if (rc != MJH_OK) { sh->rc = rc; sh->err = mjh_last_error(); break; }
    prev_first = first; prev_count = count;
  }
  if (sh->rc == MJH_OK && prev_first >= 0) {
    const int rc = pick(0, prev_first, prev_count);
    if (rc != MJH_OK) { sh->rc = rc; sh->err = mjh_last_error(); }
  }
}
}  // namespace

extern "C" int mjh_pool_encode_host(mjh_pool *pl, const void *pixels, size_t row_pitch, size_t image_stride, int n,
                                    const uint8_t *const **jpegs, const size_t **sizes)
{
  if (!pl || !pixels || n < 1 || !jpegs || !sizes) return mjh_internal_fail(MJH_EINVAL, "mjh_pool_encode_host: bad arguments");
  const mjh_params *p = mjh_encoder_params(pl->enc[0]);
  const size_t row_bytes = (size_t)p->image_width * (p->input_components == 1 ? 1 : (p->input_pixel_size ? p->input_pixel_size : 3)) * (p->data_precision == 12 ? 2 : 1);
  if (row_pitch < row_bytes) { pl->error = "row_pitch smaller than one row"; return mjh_internal_fail(MJH_EINVAL, pl->error.c_str()); }
  if (n > 1 && image_stride < row_pitch * (size_t)(p->image_height - 1) + row_bytes) {
    pl->error = "image_stride smaller than one image: images would overlap";
    return mjh_internal_fail(MJH_EINVAL, pl->error.c_str());
  }
  const int nd = (int)pl->enc.size();
  std::vector<Share> shares(nd);
  for (int i = 0; i < n; i++) shares[i % nd].images.push_back(i);
  std::vector<std::thread> th;
  for (int d = 0; d < nd; d++)
    if (!shares[d].images.empty())
      th.emplace_back(run_share, pl, d, (const uint8_t *)pixels, row_pitch, image_stride, row_bytes, p->image_height, &shares[d]);
  for (std::thread &t : th) t.join();
  pl->ptrs.assign(n, nullptr); pl->sizes.assign(n, 0);
  for (int d = 0; d < nd; d++) {
    if (shares[d].rc != MJH_OK) { pl->error = "device " + std::to_string(pl->dev[d]) + ": " + shares[d].err; return shares[d].rc; }
    for (size_t k = 0; k < shares[d].images.size(); k++) {
      pl->ptrs[shares[d].images[k]] = pl->store[d].data() + shares[d].off[k];
      pl->sizes[shares[d].images[k]] = shares[d].len[k];
    }
  }
  *jpegs = pl->ptrs.data(); *sizes = pl->sizes.data();
  return MJH_OK;
}
