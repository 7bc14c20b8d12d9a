#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the MI355X JPEG encode hot path (BASELINE.json metric).

One "step" = one pass of the whole hot path (colour/downsample -> FDCT/quantize -> statistics ->
Huffman tables -> AC+DC trellis -> final tables -> bit packing/stuffing/headers) over one batch
of synthetic frames.  Default workload (--config metric): 3840x2160 RGB, quality 75, 4:2:0,
baseline (sequential) mode with trellis quantization, overshoot deringing and optimal Huffman
tables = `cjpeg -quality 75 -baseline` = the configuration the metric is quoted on.  --config
c2|c3|c4|c5|c5t selects the other BASELINE.json configurations (same JSON line, their own workload).
N>1: one process per GPU, every rank encodes its own share of the batch (images are independent:
weak scaling, no collective on the data path; c4 shards its 1024 frames: strong scaling).
`python bench.py --gpus N` WITHOUT a launcher starts the N ranks itself (re-executes under
torch.distributed.run on 127.0.0.1) and fails loudly when fewer than N devices are visible; under a
launcher (WORLD_SIZE set) --gpus must agree with it.  At N>1 the line also carries `host_inclusive`
measured on EVERY rank at the same time (sum, per-rank min/max, total input GB/s: the PCIe side is
what can fail to scale) and `pool_host` (ONE process driving all N devices, mjh_pool_encode_host).

Prints ONE JSON line (rank 0).  `value` is the DEVICE-RESIDENT rate (frames already in HBM when the
timed region starts, complete JPEG files left in HBM), as the contract defines it.  Extra objects:
  host_inclusive -- the wall the reference's own definition asks for: pixels in pinned host
                  memory -> JPEG files in host memory, H2D and D2H included, >= 3 s of back-to-back batches through the
                  asynchronous double-buffered mjh_encode_host / mjh_collect path.  Never `value`.
  bit_exact     -- EVERY frame of the batch compared byte for byte with the real reference (oracle/_ref/refenc, the
                  reference's libjpeg driven like cjpeg) where that binary exists, else with the C port (oracle/).
  roofline      -- dominant kernel by HIP-event time inside the timed region (events are recorded on the encoder's own
                  stream), algorithmic bytes = input samples + JPEG bytes of one batch, peak = 8 TB/s HBM.
  cpu_baseline  -- the reference mozjpeg (oracle/_ref, C build) or, if that binary is absent, the C port in oracle/,
                  timed on this box's host cores on a bounded sample.
"""
import argparse
import concurrent.futures as cf
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0
PCIE_PEAK_GBS = 63.0   # PCIe Gen5 x16 per GPU (MI355X_MICROARCH.md)

# BASELINE.json configs (SURVEY 8d "Config -> concrete runs").  batch = frames per step per GPU.
CONFIGS = {
    # (96 frames per encode call since the round's third session: 64 / 96 / 128 frames = 140.1 / 144.1 / 144.3 Gpixels/s -- every
    # launch of the schedule drains once per call, gpurun_out/r06x)
    "metric": dict(w=3840, h=2160, kw=dict(quality=75, baseline=True), batch=96, steps=400,
                   name="3840x2160 synthetic RGB, q75 4:2:0 baseline, trellis+deringing+optimal Huffman (cjpeg -quality 75 -baseline)"),
    # (frames per encode call, third session of round 6: c2 64 / 128 / 256 / 512 = 108.8 / 124.7 / 134.3 / 135.1 Gpixels/s; c3 32 / 48 / 64 =
    # 25.5 / 26.1 / 26.3; c5t 1 / 2 / 4 / 8 images = 23.7 / 26.3 / 25.8 / 27.1 and c5 1 / 4 / 8 = 40.4 / 41.8 / 42.1: one image stays the case)
    "c2": dict(w=1920, h=1080, kw=dict(quality=75, baseline=True), batch=256, steps=600,
               name="C2: 1920x1080 synthetic RGB, q75 4:2:0 baseline, trellis on (cjpeg -quality 75 -baseline)"),
    "c3": dict(w=3840, h=2160, kw=dict(quality=85, sample=(2, 2)), batch=64, steps=100,
               name="C3: 3840x2160 synthetic RGB, q85 4:2:0 progressive + scan search (cjpeg -quality 85 -sample 2x2)"),
    "c4": dict(w=1920, h=1080, kw=dict(quality=75, baseline=True), batch=256, total=1024, steps=150,
               name="C4: batch of 1024 x 1920x1080 frames, q75 trellis baseline, sharded over the GPUs (256 per encode call)"),
    "c5": dict(w=8192, h=8192, kw=dict(precision=12, baseline=True, notrellis=True, quality=90, sample=(1, 1), restart=1), batch=1, steps=600,
               name="C5: 8192x8192 12-bit, q90 4:4:4, restart interval = MCU row, -notrellis (the reference aborts on 12-bit + trellis, SURVEY F1)"),
    # SURVEY 8f row 4 (completeness path, not a throughput path: an adaptive coder is one dependent chain per scan)
    # (arithmetic coding is one dependent chain per scan and image: a step takes the same ~4 s for 16 frames or 256, so the
    # batch is the throughput knob; `distinct` synthetic frames are generated and repeated to fill it)
    "arith": dict(w=3840, h=2160, kw=dict(quality=75, baseline=True, arithmetic=True), batch=256, distinct=32, steps=2,
                  name="4K q75 4:2:0 sequential, arithmetic coding + the coder's trellis (cjpeg -quality 75 -baseline -arithmetic)"),
    "arith_prog": dict(w=3840, h=2160, kw=dict(quality=75, arithmetic=True), batch=256, distinct=32, steps=2,
                       name="4K q75 4:2:0 progressive + scan search, arithmetic coding (cjpeg -quality 75 -arithmetic)"),
    "c5t": dict(w=8192, h=8192, kw=dict(baseline=True, quality=90, sample=(1, 1), restart=1), batch=1, steps=400,
                name="C5 8-bit twin: 8192x8192, q90 4:4:4 trellis, restart interval = MCU row"),
}


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _refenc_cmd(w, h, kw):
    import oracle_lib as O
    return [os.path.join(O.REF_DIR, "refenc")] + O.ref_switches(**kw) + ["-raw", str(w), str(h)]


def _ref_one(args):
    cmd, raw, out = args
    subprocess.check_output(cmd + [raw, out])
    with open(out, "rb") as f:
        return f.read()


def verify_frames(frames, jpegs, w, h, kw, limit=None):
    """EVERY frame (or the first `limit`) against the real reference where it exists (oracle/_ref/refenc), else the
    C port.  Checker only: nothing here is timed or shipped."""
    import oracle_lib as O
    n = len(jpegs) if limit is None else min(limit, len(jpegs))
    if O.have_ref():
        cmd = _refenc_cmd(w, h, kw)
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            jobs = []
            for i in range(n):
                raw = os.path.join(td, "f%d.raw" % i)
                frames[i].tofile(raw)
                jobs.append((cmd, raw, os.path.join(td, "o%d.jpg" % i)))
            with cf.ThreadPoolExecutor(max_workers=usable_cores()) as ex:
                refs = list(ex.map(_ref_one, jobs))
        kind = "reference (oracle/_ref/refenc)"
    else:
        po = O.make_params(w, h, **kw)
        refs = [O.encode(po, frames[i]) for i in range(n)]
        kind = "port (oracle/libmjoracle.so)"
    same = sum(1 for i in range(n) if refs[i] == jpegs[i])
    return {"checked": n, "identical": same, "against": kind, "ok": same == n}


def cpu_baseline(frame, kw, budget_s=20.0):
    """Reference encoder on the host cores, bounded sample.  Checker/baseline only."""
    import oracle_lib as O
    cores = usable_cores()
    h, w = frame.shape[:2]
    if O.have_ref():
        # one process per core, each encodes the same frame `reps` times in memory
        with tempfile.TemporaryDirectory() as td:
            raw = os.path.join(td, "f.rgb")
            frame.tofile(raw)
            base = _refenc_cmd(w, h, kw)
            t0 = time.time()
            subprocess.check_output(base + ["-reps", "1", raw, os.path.join(td, "o.jpg")])
            one = time.time() - t0
            reps = max(1, min(8, int(budget_s / max(one, 1e-3) / 1.5)))
            t0 = time.time()
            procs = [subprocess.Popen(base + ["-reps", str(reps), raw, os.path.join(td, "o%d.jpg" % i)],
                                      stdout=subprocess.PIPE) for i in range(cores)]
            outs = [p.communicate()[0] for p in procs]
            wall = time.time() - t0
            best1 = max(json.loads(o.decode())["mpix_per_s_best"] for o in outs)
        return {"value": round(cores * reps * w * h / wall / 1e6, 2), "unit": "Mpixels/s", "cores": cores,
                "kind": "reference",
                "sample": "%d procs x %d reps of one %dx%d frame, refenc (mozjpeg C build, no SIMD: no NASM), "
                          "in-memory libjpeg API; best single-core %.1f Mpixels/s" % (cores, reps, w, h, best1)}
    p = O.make_params(w, h, **kw)
    t0 = time.time()
    O.encode(p, frame)
    dt = time.time() - t0
    return {"value": round(w * h / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "1 encode of one %dx%d frame with oracle/libmjoracle.so (scalar C port)" % (w, h)}


def other_config_line(M, torch, key, device, budget_s, t_end):
    """Compact driver-visible evidence for a BASELINE configuration other than the metric's (never `value`): a few device-
    resident steps of the configuration with its own frames, the first frame(s) compared with the real reference (outside
    the timed steps), the dominant interval's roofline fraction.  Bounded: frames and steps shrink to fit `budget_s`."""
    cfg = CONFIGS[key]
    w, h, kw = cfg["w"], cfg["h"], cfg["kw"]
    twelve = kw.get("precision", 8) == 12
    B = cfg["batch"]
    calls_per_step = 1
    if cfg.get("total"):                       # C4: the whole job of `total` frames = total / batch encode calls of `batch` frames
        B = cfg["batch"]
        calls_per_step = cfg["total"] // cfg["batch"]
    t_start = time.perf_counter()
    distinct = min(B, 4)                       # synthetic frames are the expensive part on the host: 4 distinct, repeated
    frames = make_frames(w, h, [1234 + i for i in range(distinct)], twelve, 1)
    reps = B // distinct
    batch = np.concatenate([frames] * reps) if reps > 1 else frames
    d = torch.from_numpy(batch.view(np.int16) if twelve else batch).to(device)
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=len(batch), device=device.index or 0)
    try:
        for _ in range(2):
            enc.encode_tensor(d, stream="own")
            enc.sync()
        jp = [enc.get_jpeg(i) for i in range(len(batch))]
        nver = 1 if w * h > 30000000 else 2
        be = verify_frames(frames, jp, w, h, kw, nver)
        be["repeats_identical_to_first_copy"] = all(jp[i] == jp[i % distinct] for i in range(len(jp)))
        be["ok"] = bool(be["ok"] and be["repeats_identical_to_first_copy"])
        # the dominant interval ONE BATCH AT A TIME (its kernels alone on the chip: what the roofline fraction is about) ...
        enc.set_inflight(1)
        enc.set_profiling(1)
        enc.encode_tensor(d, stream="own"); enc.sync()          # creates the events
        enc.set_profiling(1)
        enc.encode_tensor(d, stream="own"); enc.sync()
        kt = {k: v for k, v in dict(enc.kernel_times()).items() if "side stream" not in k and not k.startswith("join(")}
        focus = max(kt, key=kt.get) if kt else None
        enc.set_profiling(2, focus=focus)
        for _ in range(6):
            enc.encode_tensor(d, stream="own")
        enc.sync()
        dom = dict(enc.kernel_times())
        dom_ms = dom.get(focus) if focus else None
        enc.set_profiling(0)
        # ... and the throughput as the library runs consecutive device-resident batches: two in flight (mjh_set_inflight)
        enc.set_inflight(2)
        for _ in range(2):
            enc.encode_tensor(d, stream="own")
        enc.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 0
        left = min(budget_s - (t0 - t_start), t_end - t0)
        while steps < 3 or (time.perf_counter() - t0 < min(1.5, left) and steps < 400):
            for _ in range(calls_per_step):
                enc.encode_tensor(d, stream="own")
            steps += 1
        enc.sync()
        dt = (time.perf_counter() - t0) / steps
        algo = float(batch.nbytes) + float(sum(len(j) for j in jp))
        out = {"workload": cfg["name"], "frames_per_step": int(len(batch)) * calls_per_step, "steps": steps, "ms_per_step": round(dt * 1e3, 3),
               "value": round(float(w) * h * len(batch) * calls_per_step / dt / 1e6, 1), "unit": "Mpixels/s", "bit_exact": be}
        if calls_per_step > 1:
            out["encode_calls_per_step"] = calls_per_step
        if dom_ms:
            out["roofline"] = {"kernel": focus, "kernel_ms": round(dom_ms, 4), "achieved": round(algo / (dom_ms * 1e-3) / 1e9, 1),
                               "unit": "GB/s", "frac": round(algo / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                               "kernel_ms_is": "the interval with one batch at a time on the chip (mjh_set_inflight 1); `value` is with two batches in flight"}
        return out
    finally:
        enc.close()


# kernels behind each interval name of the schedule (mjh_get_kernel_times), for the PMC traffic lookup
INTERVAL_KERNELS = {
    "trellis_ac": ("k_trellis_ac",), "dct_quant": ("k_dct_quant",), "color": ("k_color",),
    "huff_encode": ("k_enc_len", "k_enc_write", "k_chunk_sums", "k_scan_sums", "k_offsets", "k_zero_stream", "k_seg_extra"),
    "byte_stuff": ("k_ff_chunk_sums", "k_stuff_write", "k_finish_bits"),
    "stats_ac(final)": ("k_stats_ac",), "stats_dc(final)": ("k_stats_dc",),
    "prog_encode": ("k_pp_len", "k_pp_write", "k_pp_emit", "k_pp_chunk_bits", "k_pp_finish", "k_prog_", "k_chunk_sums", "k_scan_sums", "k_offsets"),
    "prog_stats": ("k_pp_init", "k_pp_stats", "k_pp_carry", "k_pp_cuts", "k_pp_runs", "k_pp_resolve", "k_prog_scan"),
}


def kernel_source_stamp():
    """sha256 over the sources the kernels of every PROFILED configuration are built from (mozjpeg_amd/csrc/*.hip, *.h):
    tools/pmc_traffic.py writes it into the traffic summaries it produces.  mjh_arith.hip (+ its table) is left out: a
    translation unit of its own that holds only the arithmetic-coding kernels, which no configuration with PMC passes
    launches (the arith lines carry traffic = null).  A summary whose stamp equals the tree's is quoted as it is; one whose
    stamp differs is quoted only for kernels whose MACHINE CODE is still the same (kernel_fingerprints below)."""
    import hashlib
    hsh = hashlib.sha256()
    src = os.path.join(ROOT, "mozjpeg_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")) and not name.startswith("mjh_arith"):
            hsh.update(name.encode())
            hsh.update(open(os.path.join(src, name), "rb").read())
    return hsh.hexdigest()[:16]


def kernel_fingerprints():
    """{kernel: fingerprint of its gfx950 machine code} of the library that runs, from mozjpeg_amd/kernel_isa.json
    (tools/kernel_isa.py, run by build(): the compiler's assembly per kernel with labels and comments normalised).  Trusted
    only while the file was computed from the sources in the tree; {} otherwise."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_isa
        j = json.load(open(os.path.join(ROOT, "mozjpeg_amd", "kernel_isa.json")))
        if j.get("source_stamp") != kernel_isa.source_stamp():
            return {}
        return {k: v["sha"] for k, v in j["kernels"].items()}
    except Exception:
        return {}


def dominant_traffic(config, dom, frames_per_call):
    """HBM bytes per launch of the dominant interval's kernels from the newest committed PMC passes of THIS configuration
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, fetch corrected by the factor calibrated on k_color;
    tools/pmc_traffic.py), scaled to this batch; (None, reason) when no passes of the configuration are committed or the
    interval's kernels are no longer the ones the passes were taken on."""
    pat = "r*_pmc_hbm_traffic_batch*.json" if config == "metric" else "r*_%s_pmc_hbm_traffic_batch*.json" % config
    paths = [p for p in glob.glob(os.path.join(ROOT, "profiles", pat))
             if config != "metric" or not any(("_%s_" % c) in os.path.basename(p) for c in CONFIGS if c != "metric")]
    prefixes = INTERVAL_KERNELS.get(dom.split("(")[0] if dom.startswith("prog_") else dom, ())
    stamp = kernel_source_stamp()
    now = None
    stale = None
    for path in sorted(paths, reverse=True):
        try:
            pmc = json.load(open(path))
            mine = {k: v for k, v in pmc["kernels"].items() if any(k.startswith(pf) for pf in prefixes)}
            per_frame = sum(v["hbm_bytes_per_frame"] for v in mine.values())
            if per_frame <= 0:
                continue
            how = "kernel sources %s" % stamp
            if pmc.get("kernel_source_stamp") != stamp:     # other sources than the ones in the tree: same machine code, kernel by kernel?
                if now is None:
                    now = kernel_fingerprints()
                then = pmc.get("kernel_isa") or {}
                if not (mine and all(k in then and now.get(k) == then[k] for k in mine)):
                    stale = stale or os.path.relpath(path, ROOT)
                    continue
                how = "machine code of the interval's %d kernels identical to the profiled tree's (tools/kernel_isa.py)" % len(mine)
            return int(per_frame * frames_per_call), "%s (bytes per launch, scaled to this batch; passes taken at git %s, %s)" % (
                os.path.relpath(path, ROOT), pmc.get("profile_head", "?"), how)
        except Exception:
            continue
    if stale:
        return None, "stale: the kernels of this interval changed after the newest PMC passes of this configuration (%s) -- re-run tools/profile_round.sh" % stale
    return None, "no PMC passes committed for this configuration / interval"


def dominant_issue(config, dom, frames_per_call, dom_ms):
    """What bounds the dominant interval when it is not the memory system: VALU issue.  From the newest committed SQ counter
    passes of this configuration (profiles/r*_pmc_sq_batch*.json, tools/pmc_sq.py): wave-instructions per launch of the interval's
    kernels, scaled to this batch; a SIMD issues one wave64 VALU instruction per 4 cycles (1024 SIMDs x 2.4 GHz,
    MI355X_MICROARCH.md), so issue_frac = instructions x 4 / (1024 x 2.4e9 x the interval's measured time).  lanes_per_instruction =
    SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU.  None when the interval's kernels are not the profiled machine code."""
    if config != "metric":
        return None
    prefixes = INTERVAL_KERNELS.get(dom, ())
    now = kernel_fingerprints()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_batch*.json")), reverse=True):
        try:
            j = json.load(open(path))
            meta = j.get("_meta") or {}
            batch = float(meta.get("frames_per_launch") or os.path.basename(path).split("batch")[1].split(".")[0])
            mine = {k: v for k, v in j.items() if k != "_meta" and any(k.startswith(pf) for pf in prefixes)}
            then = meta.get("kernel_isa") or {}
            if not mine or not all(k in then and now.get(k) == then[k] for k in mine):
                continue
            insts = sum(v["SQ_INSTS_VALU"] * v.get("launches_per_step", 1) for v in mine.values()) * frames_per_call / batch
            thr = sum(v.get("SQ_THREAD_CYCLES_VALU", 0.0) for v in mine.values())
            act = sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in mine.values())
            return {"valu_wave_instructions_per_launch": int(insts), "issue_frac": round(insts * 4.0 / (1024 * 2.4e9 * dom_ms * 1e-3), 4),
                    "lanes_per_instruction": round(thr / act, 1) if act else None,
                    "source": "%s (SQ_INSTS_VALU per launch, scaled to this batch; machine code of the interval's kernels identical to the profiled tree's)" % os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


def baseline_metric():
    """the metric exactly as BASELINE.json names it"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mpixels/s encode (4K RGB q75 trellis) at 1/2/4/8 GPUs; bit-exact vs cjpeg"


def _make_frame(w, h, seed, twelve):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return (O.synthetic_frame12 if twelve else O.synthetic_frame)(w, h, seed)


def make_frames(w, h, seeds, twelve, world):
    """synthetic frames, one seed per frame, generated by a few worker processes (numpy takes seconds per
    4K frame) -- input preparation, outside every timed region"""
    workers = max(1, min(len(seeds), usable_cores() // max(1, world)))
    if workers > 1:
        try:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(workers) as pool:
                return np.stack(pool.starmap(_make_frame, [(w, h, sd, twelve) for sd in seeds]))
        except Exception as exc:   # no worker processes available: same frames, just slower
            print("bench: frame workers unavailable (%s), generating serially" % exc, file=sys.stderr)
    return np.stack([_make_frame(w, h, sd, twelve) for sd in seeds])


def host_inclusive(M, params, frames, hb, device, min_s, ref_jpegs):
    """pixels in PINNED host memory -> JPEG files in host memory, >= min_s seconds of back-to-back batches of hb frames:
    batch k+1 is queued (H2D on the copy stream) before batch k is picked up, so copies, kernels and the hand-over of
    the files overlap.  Every file of the first pass is compared with the device-resident run's (already verified)."""
    n = frames.shape[0]
    pinned = M.pinned_empty(frames.shape, frames.dtype)
    pinned[...] = frames
    enc = M.Encoder(params, max_batch=hb, device=device)
    starts = list(range(0, n - hb + 1, hb)) or [0]
    hb = min(hb, n)

    def take(age):
        return enc.collect(age=age, copy=False)

    # warm-up + check of every file of one pass over the frames
    same = checked = 0
    enc.submit_host(pinned[starts[0]:starts[0] + hb])
    for k in range(1, len(starts)):
        enc.submit_host(pinned[starts[k]:starts[k] + hb])
        for j, mv in enumerate(take(1)):
            checked += 1
            same += bytes(mv) == ref_jpegs[starts[k - 1] + j]
    for j, mv in enumerate(take(0)):
        checked += 1
        same += bytes(mv) == ref_jpegs[starts[-1] + j]
    # timed: the host touches every output byte once (sums the sizes; the files ARE in host memory)
    done = 0
    out_bytes = 0
    t0 = time.perf_counter()
    enc.submit_host(pinned[starts[0]:starts[0] + hb])
    k = 1
    while True:
        enc.submit_host(pinned[starts[k % len(starts)]:starts[k % len(starts)] + hb])
        res = take(1)
        done += len(res)
        out_bytes += sum(len(r) for r in res)
        k += 1
        if time.perf_counter() - t0 >= min_s and k >= 3:
            break
    res = take(0)
    done += len(res)
    out_bytes += sum(len(r) for r in res)
    dt = time.perf_counter() - t0
    h, w = frames.shape[1:3]
    enc.close()
    in_bytes = done * frames[0].nbytes
    return {"value": round(done * w * h / dt / 1e6, 2), "unit": "Mpixels/s", "seconds": round(dt, 3), "frames": done,
            "frames_per_call": hb, "input_GBps": round(in_bytes / dt / 1e9, 2), "output_GBps": round(out_bytes / dt / 1e9, 3),
            "files_identical_to_device_run": "%d/%d" % (same, checked),
            "path": "pinned host pixels -> hipMemcpyAsync on the copy stream -> kernels -> files packed into pinned host memory by "
                    "the device (mjh_encode_host / mjh_collect, double-buffered); one host thread"}


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_plan(gpus, env, argv, visible_devices):
    """What `bench.py --gpus N` does about its ranks.  Returns ("run", world) when this process IS a rank (or N == 1),
    ("spawn", command) when it has to start the N ranks itself; raises SystemExit with the reason when the request cannot
    be honoured.  (A function of its arguments only, so that the CPU suite can check it.)"""
    world_env = env.get("WORLD_SIZE")
    share = env.get("MJH_BENCH_DIST_BACKEND") == "gloo"     # test switch: ranks may share devices (numbers mean nothing)
    if world_env is not None:
        world = int(world_env)
        if gpus != world and not (gpus == 1 and "--gpus" not in " ".join(argv)):
            raise SystemExit("bench: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
        if world > visible_devices and not share and visible_devices >= 0:
            raise SystemExit("bench: %d ranks but only %d GPU(s) visible" % (world, visible_devices))
        return "run", world
    if gpus <= 1:
        return "run", 1
    if gpus > visible_devices and not share and visible_devices >= 0:
        raise SystemExit("bench: --gpus %d but only %d GPU(s) visible (no CPU fallback, no device sharing)" % (gpus, visible_devices))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def gather_objects(obj, dist, world):
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def pool_host_leg(M, params, frames, n_devices, per_device, min_s, ref_jpegs):
    """ONE process, N devices: mjh_pool_encode_host deals the images of a call round-robin over its devices (one host
    thread + one double-buffered encoder per device) and returns the files in image order."""
    import torch
    ndev = max(1, torch.cuda.device_count())
    pool = M.Pool(params, max_batch_per_device=per_device, devices=[i % ndev for i in range(n_devices)])
    n = frames.shape[0]
    out = pool.encode_host(frames)                      # warm-up + check
    same = sum(1 for i in range(n) if out[i] == ref_jpegs[i])
    done = 0
    t0 = time.perf_counter()
    while True:
        pool.encode_host(frames)
        done += n
        if time.perf_counter() - t0 >= min_s:
            break
    dt = time.perf_counter() - t0
    pool.close()
    h, w = frames.shape[1:3]
    return {"value": round(done * w * h / dt / 1e6, 2), "unit": "Mpixels/s", "devices": n_devices, "frames": done,
            "seconds": round(dt, 3), "input_GBps": round(done * frames[0].nbytes / dt / 1e9, 2),
            "files_identical_to_device_run": "%d/%d" % (same, n),
            "path": "one process, pageable host pixels -> mjh_pool_encode_host (one host thread + encoder per device)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = the configuration's default: a timed region of about 3-5 s, BASELINE.md section 3)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps before (-1 = a tenth of the steps)")
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="frames per encode call per GPU (0 = the config's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of host CPU work for the cpu_baseline leg")
    ap.add_argument("--no-host-leg", action="store_true")
    ap.add_argument("--no-inflight-leg", action="store_true", help="(kept for the scripts of earlier rounds: the extra legs it switched off are gone or folded into the line)")
    ap.add_argument("--host-seconds", type=float, default=3.0)
    ap.add_argument("--host-batch", type=int, default=16)
    ap.add_argument("--verify", default="all", help="frames per batch to compare with the reference: all | N")
    ap.add_argument("--other-configs", default="auto", help="auto: the default N=1 run of the metric configuration appends a compact line per other BASELINE configuration (c2, c3, c5, c5t) under --other-budget seconds; none: skip")
    ap.add_argument("--other-budget", type=float, default=60.0)
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, form the process group (gloo) and print n_gpus: no GPU is touched (CPU test of the launch path)")
    args = ap.parse_args()

    import torch
    if args.launch_check:
        os.environ["MJH_BENCH_DIST_BACKEND"] = "gloo"
    ndev = -1 if args.launch_check else (torch.cuda.device_count() if torch.cuda.is_available() else 0)
    action, what = launch_plan(args.gpus, os.environ, sys.argv[1:], ndev)
    if action == "spawn":     # --gpus N without a launcher: this process becomes the launcher of N ranks
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
        sys.exit(subprocess.call(what, env=env))
    if args.launch_check:
        import torch.distributed as dist
        rank, world = int(os.environ.get("RANK", "0")), what
        seen = [rank]
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            seen = gather_objects(rank, dist, world)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": sorted(seen)}), flush=True)
        return
    import mozjpeg_amd as M

    cfg = CONFIGS[args.config]
    if args.steps <= 0:
        args.steps = cfg["steps"]
    if args.warmup < 0:
        args.warmup = max(3, args.steps // 10)
    w, h, kw = cfg["w"], cfg["h"], cfg["kw"]
    twelve = kw.get("precision", 8) == 12
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = what
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") on a multi-GPU node.  MJH_BENCH_DIST_BACKEND=gloo exists only to exercise this code path on a
        # box with fewer GPUs than ranks (ranks then share devices; numbers from such a run mean nothing)
        dist.init_process_group(os.environ.get("MJH_BENCH_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.batch or cfg["batch"]
    # frames this rank encodes per step: its own batch (weak scaling), or its shard of a fixed total (c4: strong)
    total = cfg.get("total")
    from mozjpeg_amd import shard
    if total:
        mine = shard.shard_indices(total, rank, world)
        nframes = len(mine)
        B = min(B, nframes)
        seeds = [1234 + i for i in mine]
    else:
        nframes = B
        seeds = [1234 + rank * B + i for i in range(B)]
    distinct = cfg.get("distinct", 0)
    repeats = B // distinct if (distinct and not total and B > distinct and B % distinct == 0) else 1
    if repeats > 1:
        seeds = seeds[:distinct]
    frames = make_frames(w, h, seeds, twelve, world)
    if repeats > 1:
        frames = np.concatenate([frames] * repeats)      # frame i of the batch = distinct frame i % distinct
    d_frames = torch.from_numpy(frames.view(np.int16) if twelve else frames).to(dev)
    params = M.make_params(w, h, **kw)
    enc = M.Encoder(params, max_batch=B, device=local_rank)
    calls = [(s, min(B, nframes - s)) for s in range(0, nframes, B)]   # encode calls per step

    def step(keep=None):
        for s0, cnt in calls:
            enc.encode_tensor(d_frames[s0:s0 + cnt], stream="own")
            if keep is not None:
                keep.extend(enc.get_jpeg(i) for i in range(cnt))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        enc.sync()

    # Warm-up: W untimed steps, ordered so that the chip is in the timed region's state when the clock starts (round 6, third
    # session: the files used to be fetched and compared with the reference BETWEEN the warm-up and the timed region -- seconds of
    # host work with an idle GPU right in front of a region that lasts 0.1 s at the driver's --steps 20; and no warm-up step ran
    # with two batches in flight).
    #   step 1      fetches every file; the comparison with the reference follows at once (outside every timed region)
    #   nprobe      steps with every kernel bracketed (profiling level 1): the DOMINANT interval of the schedule is measured on
    #               this workload, not assumed; the timed region then brackets exactly that one (level 2)
    #   the rest    exactly like the timed steps: back to back, two batches in flight, the dominant interval bracketed
    W = args.warmup
    nprobe = 2 if W >= 4 else 0            # (the first bracketed step creates the events: two are the least)
    focus, probe = None, {}
    jpegs = []
    late_probe = nprobe == 0
    if late_probe:
        # a warm-up too short for probe steps (configurations whose step takes seconds, or --warmup 0: the one step below is then
        # an extra untimed one): the file-fetching step is the bracketed one -- its intervals include the creation of the events,
        # which does not change which one is largest
        enc.set_profiling(1)
    step(keep=jpegs)
    barrier()
    if late_probe:
        probe = dict(enc.kernel_times())
        enc.set_profiling(0)
    jpeg_bytes = sum(len(j) for j in jpegs)
    # bit-exactness of EVERY frame against the real reference (outside the timed region)
    bitexact = None
    if rank == 0:
        lim = None if args.verify == "all" else int(args.verify)
        if repeats > 1:     # the distinct frames against the reference, every repeat against the file of its first copy
            bitexact = verify_frames(frames[:distinct], jpegs[:distinct], w, h, kw, lim)
            same = all(jpegs[i] == jpegs[i % distinct] for i in range(len(jpegs)))
            bitexact["repeats_identical_to_first_copy"] = same
            bitexact["ok"] = bool(bitexact["ok"] and same)
        else:
            bitexact = verify_frames(frames, jpegs, w, h, kw, lim)
    barrier()              # (every rank waits for rank 0's comparison, so that the remaining warm-up steps run right before the clock starts on all of them)
    if nprobe:
        enc.set_profiling(1)
        for i in range(nprobe):
            if i == 1:
                enc.set_profiling(1)   # the first bracketed step creates the events: its intervals include that; start over
            step()
            enc.sync()     # (the encoder tunes itself from what the previous batch reported: let every probe batch finish)
        probe = dict(enc.kernel_times())
        enc.set_profiling(0)
    main_stream = {k: v for k, v in probe.items() if "side stream" not in k and not k.startswith("join(")}
    if main_stream:
        focus = max(main_stream, key=main_stream.get)
    enc.set_profiling(2, focus=focus)
    for _ in range(max(0, W - 1 - nprobe)):
        step()

    # Timed region: K steps back to back.  HIP events bracket only the dominant kernel here (profiling
    # level 2: two events per step on the encoder's stream, read once after the loop), so the event
    # barriers of a full per-kernel breakdown do not slow the measured steps.
    enc.set_profiling(2, focus=focus)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    dom_times = dict(enc.kernel_times())          # average ms per encode call over the timed region
    elapsed = shard.max_over_ranks(elapsed, dist, dev)

    # Untimed extra pass with every kernel bracketed (level 1) for the per-kernel breakdown, one batch at a time (with two in
    # flight an interval between two events on one stream also holds the other batch's kernels)
    enc.set_inflight(1)
    enc.set_profiling(1)
    for _ in range(min(args.steps, 5)):
        step()
    ktimes = dict(enc.kernel_times())
    # ... and the reference figures of the same steps one batch at a time (never `value`)
    enc.set_profiling(2, focus=focus)
    ksteps = max(3, min(args.steps, 40))
    barrier()
    t0s = time.perf_counter()
    for _ in range(ksteps):
        step()
    barrier()
    serial_ms = (time.perf_counter() - t0s) / ksteps * 1e3
    serial_dom = dict(enc.kernel_times())
    enc.set_profiling(0)
    enc.set_inflight(2)

    # (Rounds 2-5 measured here what three encoders taken round-robin gain over one: since round 6 the encoder itself keeps two
    # batches in flight -- mjh_set_inflight -- and `value` is that; the one-batch-at-a-time figures are in `one_batch_at_a_time`.)
    pipelined = None
    enc.close()
    del d_frames

    # Driver-visible evidence for the other BASELINE configurations (extra information, never `value`): only in the
    # default run of the metric configuration on one GPU, under a wall-clock budget
    others = None
    if args.other_configs == "auto" and args.config == "metric" and world == 1 and not args.no_inflight_leg:
        others = {}
        t_end = time.perf_counter() + args.other_budget
        keys = ["c2", "c3", "c4", "c5t", "c5"]
        for i, key in enumerate(keys):
            left = t_end - time.perf_counter()
            if left < 4.0:
                others[key] = {"skipped": "budget of %.0f s used up" % args.other_budget}
                continue
            try:
                others[key] = other_config_line(M, torch, key, dev, left / (len(keys) - i), t_end)
            except Exception as exc:
                others[key] = {"error": str(exc)}
        torch.cuda.empty_cache()

    # PCIe-inclusive legs (never `value`).  N > 1: EVERY rank runs its host path at the same time -- the host side (N x
    # pinned reads through one root complex, N result streams back) is the part of the job that can fail to scale.
    host_all = None
    if not args.no_host_leg:
        try:
            if dist is not None:
                dist.barrier()
            host_res = host_inclusive(M, params, frames, min(args.host_batch, nframes), local_rank, args.host_seconds, jpegs)
        except Exception as exc:   # the leg is extra information: the contract line must still come out
            host_res = {"error": str(exc)}
        host_all = gather_objects(host_res, dist, world)
    pool_res = None
    if not args.no_host_leg and world > 1:
        if dist is not None:
            dist.barrier()          # the other ranks idle while rank 0 drives every device from one process
        if rank == 0:
            try:
                ndev = max(1, torch.cuda.device_count())
                sample = frames[:min(nframes, 8)]
                big = np.concatenate([sample] * world)           # `world` x the sample, dealt over `world` devices
                pool_res = pool_host_leg(M, params, big, world, min(args.host_batch, len(sample)), args.host_seconds,
                                         [jpegs[i % len(sample)] for i in range(len(big))])
                pool_res["device_ids"] = [i % ndev for i in range(world)]
            except Exception as exc:
                pool_res = {"error": str(exc)}
        if dist is not None:
            dist.barrier()

    if rank == 0:
        frames_per_step_all = total if total else B * world
        total_px = float(w) * h * frames_per_step_all * args.steps
        dom = max(dom_times, key=dom_times.get) if dom_times else max(ktimes, key=ktimes.get)
        dom_ms = dom_times.get(dom, ktimes.get(dom))
        per_call = float(calls[0][1])
        algo_bytes = frames[0].nbytes * per_call + jpeg_bytes * per_call / nframes
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate runs, fetch corrected by the factor calibrated on k_color; tools/pmc_traffic.py) -- per launch
        traffic, traffic_src = dominant_traffic(args.config, dom, per_call)
        out = {
            "metric": baseline_metric(),
            "value": round(total_px / elapsed / 1e6, 2), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if total else "weak", "vs_baseline": None,
            "dtype": "u16/int16" if twelve else "u8/int16 (+f32 trellis costs)",
            "data": "synthetic",
            "value_is": "device-resident rate: frames in HBM when the timed region starts, complete JPEG files left in HBM "
                        "(the PCIe-inclusive wall is host_inclusive.value)",
            "config": {"workload": cfg["name"], "config_key": args.config,
                       "frames_per_step_per_gpu": nframes, "frames_per_encode_call": int(per_call),
                       "distinct_frames": int(nframes // repeats),
                       "value_definition": "device-resident (this line's `value`); SURVEY 8d / BASELINE.md section 3 define the metric from pinned host "
                                           "pixels to JPEG bytes in host memory: that number is `value_host_inclusive`",
                       "batches_in_flight": 2,
                       "input": "resident in HBM", "output": "complete JPEG files in HBM",
                       "parallelism": "images sharded, 1 process per GPU, no collective"},
            "bit_exact": bitexact,
            "bit_exact_vs_reference": bool(bitexact and bitexact["ok"]),
            "jpeg_bytes_per_frame": int(jpeg_bytes / nframes),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "kernel_ms": round(dom_ms, 4),
                         "focus_probe_ms": {k: round(v, 4) for k, v in sorted(probe.items(), key=lambda kv: -kv[1])[:4]},
                         "kernel_ms_source": "HIP events around the interval in every encode call of the timed region; the interval was chosen as "
                                             "the largest of a per-kernel pass over the warm-up steps",
                         "algorithmic_bytes": "input samples + JPEG bytes of one encode call (SURVEY 8d)",
                         "whole_step_GBps": round(algo_bytes * len(calls) / (elapsed / args.steps) / 1e9, 1),
                         "kernel_ms_per_call(untimed pass, every kernel bracketed)":
                             {k: round(v, 4) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1])}},
        }
        out["one_batch_at_a_time"] = {"ms_per_step": round(serial_ms, 3), "steps": ksteps, "kernel": dom,
                                      "kernel_ms": round(serial_dom.get(dom, 0.0), 4),
                                      "frac": round(algo_bytes / (serial_dom[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if serial_dom.get(dom) else None,
                                      "what": "the same steps with mjh_set_inflight(1): every kernel alone on the chip (rounds 1-5 measured this way)"}
        issue = dominant_issue(args.config, dom, per_call, dom_ms)
        if issue is not None:
            out["roofline"]["valu_issue"] = issue      # (the interval is VALU-issue bound, not HBM bound: this is the ceiling it is near)
        if elapsed < 1.0:
            out["timed_region_warning"] = ("the timed region is %.2f s (%d steps): one noisy box moves it by percent -- the default run "
                                           "(no --steps) times 3-5 s" % (elapsed, args.steps))
        if others is not None:
            out["other_configs"] = others
        if pipelined is not None:
            out["pipelined"] = pipelined
        if host_all is not None:
            good = [h for h in host_all if h and "error" not in h]
            if world == 1 or not good:
                out["host_inclusive"] = host_all[0]
            else:
                vals = [h["value"] for h in good]
                out["host_inclusive"] = {
                    "value": round(sum(vals), 2), "unit": "Mpixels/s", "ranks_measured": len(good), "ranks": world,
                    "per_rank_min": min(vals), "per_rank_max": max(vals),
                    "input_GBps": round(sum(h["input_GBps"] for h in good), 2),
                    "per_rank_input_GBps": [h["input_GBps"] for h in good],
                    "host_dram_read_GBps": round(sum(h["input_GBps"] for h in good), 2),   # pinned frames are read in place by the DMA engines: one pass over host DRAM
                    "output_GBps": round(sum(h["output_GBps"] for h in good), 3),
                    "frames_per_call": good[0]["frames_per_call"],
                    "files_identical_to_device_run": [h["files_identical_to_device_run"] for h in good],
                    "errors": [h["error"] for h in host_all if h and "error" in h],
                    "path": "every rank at the same time: " + good[0]["path"]}
        hi = out.get("host_inclusive")
        if isinstance(hi, dict) and "value" in hi:
            # SURVEY 8d / BASELINE.md section 3 define the metric host -> host: this is that number, lifted to the top level;
            # at one GPU it is bound by the host link, so its roofline is the link's, not HBM's
            out["value_host_inclusive"] = hi["value"]
            out["host_roofline"] = {"bound": "pcie", "achieved": hi.get("input_GBps"), "peak": PCIE_PEAK_GBS * world, "unit": "GB/s",
                                    "frac": round(hi.get("input_GBps", 0.0) / (PCIE_PEAK_GBS * world), 4),
                                    "what": "pinned host pixels read over PCIe Gen5 x16 (63 GB/s per GPU, MI355X_MICROARCH.md); the files going back add output_GBps"}
        if pool_res is not None:
            out["pool_host"] = pool_res
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(frames[0], kw, args.cpu_budget)
            out["cpu_baseline"]["simd_note"] = ("C-only build of the reference (no NASM in the image); SURVEY section 6 estimates its SIMD build at ~17 instead of "
                                                "~14 Mpixels/s per core on the trellis encode (the trellis and the entropy coder are scalar either way, "
                                                "simd/CMakeLists.txt:44-52), i.e. the ratio would move by ~20 %")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
