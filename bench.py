#!/usr/bin/env python
"""bench.py -- end-to-end throughput of the MIMAMO-Net per-video inference hot path on MI355X.

Metric (BASELINE.json): face-frames/sec end-to-end (phase-diff + ResNet50 + 2-stream GRU), 64-frame clips.
Workload at N=1 (BASELINE configs[3]): a batch of independent 64-frame clips, full two-stream path -- PIL-exact
preprocessing of the uint8 112x112x3 aligned faces (the raw boundary, resident in HBM before the timed region),
steerable pyramid + phase difference, ResNet50 pool5, PhaseNet/MLP/GRU head -- fp32, random-init weights of the
reference architecture.  One step = one pass of the hot path over `--clips` clips (default 32 -> 2048 frames) per GPU.

N>1 (BASELINE configs[4]): one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).  `--gpus N`
without a torchrun environment spawns the N ranks itself; under torchrun (WORLD_SIZE set) it must equal WORLD_SIZE.
Rank 0 builds the work queue (`--total-clips` clip ids + lengths, default 10 000 at N>1), broadcasts it (RCCL),
every rank takes its shard (`dist.shard`: clips c mod N == rank) and walks it `--clips` clips per step; per-step
results ([frames,2], 8 B/frame) are all-gathered asynchronously.  No collective on the data path; weak scaling
(per-GPU work per step is fixed); the time is the max over ranks.  `--whole-job` walks the entire queue once.

Prints ONE JSON line (rank 0) with `roofline` (conv/GEMM engine on the fp32 matrix cores, timed live with
hipEvents on the launch stream by the library's measurement hook), `cpu_baseline` (the oracle's PyTorch-CPU
restatement of the same path on a bounded sample, rank 0 / N=1 only) and `extra` (N=1: the same step in direct
form without Winograd, and a step of multi-snippet 309-frame videos -- the reference's run_example shape: 5 snippets,
GRU seq_len 5).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

T_PROCESS_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per HIP stream this process uses: the three lane streams, torch's default stream, the copy stream of a
# streamed input and RCCL's own stream.  The runtime's default of 4 queues makes two of them share one as soon as the RCCL
# process group exists, which serialises two lanes: measured 111.6 vs 107.5 ms per step with / without a (one-rank) RCCL group,
# 108.2 vs 107.9 with 8 queues (tools/ab_force_dist.sh).  Read by the HIP runtime when it initialises: set before torch loads it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0
FRAMES_PER_CLIP = 64
EXAMPLE_VIDEO_FRAMES = 309      # api/readme.md:100 (utterance_1.mp4): 5 snippets of 64 (snippet_sampler.py:112-126)


LIVE_CHILD_TIMEOUT_S = 40      # one profiler child (a healthy one takes ~5-8 s after the parent has paged the image in)
LIVE_BUDGET_S = 80             # all profiler children together; a hung profiler costs ONE child timeout, then the leg gives up
N_SIMDS = 256 * 4              # MI355X: 256 CUs x 4 SIMDs
NOMINAL_CLOCK_GHZ = 2.4        # the clock PEAK_FP32_MFMA_TFLOPS is quoted at (MI355X_MICROARCH.md)


def _profiler_child(cmd, env, timeout):
    """One child under the profiler, in its OWN process group: on timeout the whole group is killed (the profiler forks the python child;
    killing only the wrapper would leave a grandchild sharing the GPU with the legs that follow).  Returns (returncode | None, output)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True,
                         universal_newlines=True)
    try:
        out, _ = p.communicate(timeout=timeout)
        return p.returncode, out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        try:
            out, _ = p.communicate(timeout=10)
        except Exception:      # noqa: BLE001
            out = ""
        return None, out


def schedule_flags(args):
    """The flags of this run that change WHICH kernels a step launches: a profiler child must run the same schedule as its parent, or its
    PMC bytes would be reported as 'measured live by this run' for another configuration (round-5 ADVICE)."""
    f = ["--lanes", str(args.lanes), "--winograd", str(args.winograd)]
    if args.no_winograd:
        f.append("--no-winograd")
    if args.from_f32:
        f.append("--from-f32")
    if args.stream_input:
        f.append("--stream-input")
    return f


def measure_live_traffic(args, per_step):
    """PMC counters of one step, measured now: this script run as a child (1 timed + 1 warm-up step, no extras) under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace` -- the collection MI355X_MICROARCH.md prescribes for HBM
    bytes (separate passes; FETCH_SIZE doubled on gfx950), the recipe of tools/pmc_traffic.sh -- and a third pass with SQ / GRBM counters
    (`SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`) for the sustained clock of the conv launches and the VALU-issue
    floor of the phase kernels.  Returns {"conv": bytes, "phase": bytes, "winograd_transforms": bytes, "seconds": s, "clock": {...} | None,
    "valu": {...} | None} or {"error": "..."}; never raises, never takes longer than LIVE_BUDGET_S (+ one kill grace)."""
    import csv
    import glob
    import shutil
    import tempfile
    t0 = time.time()
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {"error": "rocprofv3 not found"}
    steps, warm = 1, 1
    groups = {"conv": ("conv_mfma_kernel", "wino_fused_kernel"), "winograd_transforms": ("wino_in", "wino_out"),
              "phase": ("pyramid_wave_kernel", "pyramid_frame_kernel", "pyramid_kernel", "phase_window2_kernel")}
    # (the per-frame stage is pyramid_wave_kernel for whole rounds of 2 048 frames, pyramid_frame_kernel for a small remainder: one tag)
    phase_kernels = {"pyramid_frame": ("pyramid_wave_kernel", "pyramid_frame_kernel"), "phase_window2<48>": "phase_window2_kernel<48",
                     "phase_window2<24>": "phase_window2_kernel<24"}
    tot = {g: {} for g in groups}
    tmp = None
    try:
        tmp = tempfile.mkdtemp(prefix="mm_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")

        def child(tag, counters):
            """-> (counter_collection.csv, kernel_trace.csv, steps the child ran) or raises RuntimeError"""
            left = LIVE_BUDGET_S - (time.time() - t0)
            if left < 10:
                raise RuntimeError("live PMC budget of %d s used up before the %s pass" % (LIVE_BUDGET_S, tag))
            d = os.path.join(tmp, tag)
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "out", "--", sys.executable,
                                               os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warm), "--clips", str(per_step),
                                               "--no-cpu-baseline", "--no-extra", "--no-live-traffic"] + schedule_flags(args)
            rc, out = _profiler_child(cmd, env, min(LIVE_CHILD_TIMEOUT_S, left))
            if rc is None:
                raise RuntimeError("rocprofv3 --pmc %s: no result after %d s (process group killed)" % (tag, min(LIVE_CHILD_TIMEOUT_S, left)))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
            if rc != 0 or not files:
                raise RuntimeError("rocprofv3 --pmc %s: rc %d, %d counter files" % (tag, rc, len(files)))
            # the child says how many steps of the hot path it ran (its line's `hot_path_steps_executed`): the divisor of the per-step bytes
            n = None
            for line in out.splitlines():
                if line.startswith('{"metric"'):
                    try:
                        n = json.loads(line).get("hot_path_steps_executed")
                    except ValueError:
                        pass
            if not n:
                raise RuntimeError("the %s child printed no bench line with hot_path_steps_executed" % tag)
            return files[0], (traces[0] if traces else None), int(n)

        nsteps = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            path, _, nsteps[c] = child(c, [c])
            with open(path) as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") != c:
                        continue
                    for g, pats in groups.items():
                        if any(p_ in row["Kernel_Name"] for p_ in pats):
                            tot[g][c] = tot[g].get(c, 0.0) + float(row["Counter_Value"])
        out = {}
        for g in groups:
            if "FETCH_SIZE" not in tot[g] or "WRITE_SIZE" not in tot[g]:
                return {"error": "no %s kernels in the counter files" % g}
            out[g] = tot[g]["FETCH_SIZE"] * 1024 * 2 / nsteps["FETCH_SIZE"] + tot[g]["WRITE_SIZE"] * 1024 / nsteps["WRITE_SIZE"]
        out["steps_profiled"] = nsteps["FETCH_SIZE"]
        # third pass (optional: a failure here keeps the traffic): sustained clock + matrix-pipe busy of the conv launches, VALU instructions
        # of the phase kernels
        out["clock"], out["valu"] = None, None
        try:
            path, trace, n3 = child("SQ", ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
            out["clock"], out["valu"] = sq_pass_summary(path, trace, n3, groups["conv"], phase_kernels)
        except Exception as e:      # noqa: BLE001
            out["sq_pass_error"] = "%s: %s" % (type(e).__name__, e)
        out["seconds"] = time.time() - t0
        return out
    except Exception as e:      # noqa: BLE001 -- a profiler problem must never take the bench line down
        return {"error": "%s: %s" % (type(e).__name__, e), "seconds": time.time() - t0}
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)


def sq_pass_summary(counter_csv, trace_csv, nsteps, conv_pats, phase_kernels):
    """From one `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` run:
    clock = {"GHz": sum(GRBM_GUI_ACTIVE / 8 XCDs) / sum(kernel durations) over the conv launches (MI355X_MICROARCH.md: effective clock =
             GRBM_GUI_ACTIVE / kernel wall time), "mfma_busy": SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)};
    valu = {tag: non-MFMA VALU wave instructions per step} for the phase kernels.  Pure function of the two csv files (CPU-tested)."""
    import csv
    gui = busy = dur_ns = 0.0
    valu = {t: 0.0 for t in phase_kernels}
    with open(counter_csv) as f:
        for row in csv.DictReader(f):
            name, c, v = row["Kernel_Name"], row.get("Counter_Name"), float(row["Counter_Value"])
            if any(p_ in name for p_ in conv_pats):
                if c == "GRBM_GUI_ACTIVE":
                    gui += v
                elif c == "SQ_VALU_MFMA_BUSY_CYCLES":
                    busy += v
            for t, pat in phase_kernels.items():
                if any(p_ in name for p_ in ((pat,) if isinstance(pat, str) else pat)):
                    if c == "SQ_INSTS_VALU":
                        valu[t] += v
                    elif c == "SQ_INSTS_MFMA":
                        valu[t] -= v
    clock = None
    if trace_csv and gui > 0:
        with open(trace_csv) as f:
            for row in csv.DictReader(f):
                if any(p_ in row["Kernel_Name"] for p_ in conv_pats):
                    dur_ns += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        if dur_ns > 0:
            clock = {"GHz": gui / 8.0 / dur_ns, "mfma_busy": busy / (N_SIMDS * gui / 8.0),
                     "how": "sum over the conv launches of GRBM_GUI_ACTIVE / 8 XCDs, divided by the sum of their kernel-trace durations "
                            "(under the profiler: launches serialised); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x those cycles)"}
    return clock, ({t: v / nsteps for t, v in valu.items()} if any(v > 0 for v in valu.values()) else None)


def plausible_clock(clk):
    """The live SQ pass derives the sustained clock as GRBM_GUI_ACTIVE / 8 XCDs / kernel-trace time.  The chip cannot run above its nominal
    2.4 GHz and does not drop below ~1.5 GHz under this load: a value outside [1.2, 2.4 x 1.03] means the counter file and the kernel trace
    of that profiler run disagree (seen once: 2.98 GHz, with the matrix-pipe busy fraction low by the same factor) -- the line then carries
    no sustained-clock figures rather than wrong ones.  -> (clock dict or None, note or None)"""
    if not clk:
        return None, None
    g = clk.get("GHz")
    if g is None or not (1.2 <= g <= NOMINAL_CLOCK_GHZ * 1.03):
        return None, ("live SQ / GRBM pass rejected: it implies %s GHz (nominal %.1f) -- GRBM_GUI_ACTIVE and the kernel trace of that profiler "
                      "run disagree" % ("%.2f" % g if g is not None else "no", NOMINAL_CLOCK_GHZ))
    return clk, None


def phase_floors(live_rows, n_frames, valu_insts=None):
    """Per-kernel floors of the phase stage (round-5 verdict: the stage's floor is NOT the pyramid's MFMA floor alone -- the window kernels
    execute no MFMA and are VALU / LDS-issue bound).  live_rows: (cat, work, ms, tag) of one single-stream step from the library's
    measurement hook; valu_insts: {tag: non-MFMA VALU wave instructions per step} from the live SQ pass, or None.
      floor_hbm_ms  = the kernel's algorithmic bytes / 8 TB/s (pyramid: frame in + the four planes per (frame, band, level) it hands to
                      the window kernels are INTERNAL traffic, not counted; windows: the difference planes written)
      floor_mfma_ms = pyramid only: the 4 344 v_mfma_f32_16x16x4_f32 = 8.9 MFLOP per frame of the pyramid products / 157.3 TFLOP/s
                      (pyramid_wave_kernel EXECUTES 4 840: 144 exact-zero spectrum blocks skipped, 640 added by the blurs it runs on the
                      matrix pipe instead of the vector pipe -- the floor stays the algorithm's)
      floor_valu_ms = VALU wave instructions x 4 issue cycles / 1 024 SIMDs / 2.4 GHz (a wave64 VALU instruction occupies its SIMD 4 cycles)
    A kernel's floor is the largest of its floors; the stage's limiting floor is the SUM of the kernels' floors (they run back to back)."""
    kern = {}
    for c_, w_, t_, tag in live_rows:
        if c_ not in (1, 2):
            continue
        k = kern.setdefault(tag, {"ms": 0.0, "bytes": 0.0, "launches": 0})
        k["ms"] += t_
        k["bytes"] += w_
        k["launches"] += 1
    total = 0.0
    for tag, k in kern.items():
        fl = {"hbm": k["bytes"] / (PEAK_HBM_GBS * 1e9) * 1e3}
        if tag.startswith("pyramid"):
            fl["mfma"] = n_frames * 4344 * 2048.0 / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
        if valu_insts and valu_insts.get(tag):
            fl["valu"] = valu_insts[tag] * 4.0 / N_SIMDS / (NOMINAL_CLOCK_GHZ * 1e9) * 1e3
        for n_, v in fl.items():
            k["floor_%s_ms" % n_] = v
        k["limiter"] = max(fl, key=fl.get)
        k["floor_ms"] = fl[k["limiter"]]
        k["frac_of_floor"] = k["floor_ms"] / k["ms"] if k["ms"] > 0 else None
        total += k["floor_ms"]
    return kern, total


def kernel_source_hash():
    """sha256 over the HIP/C++ sources + headers: PMC traffic summaries under profiles/ record the hash of the kernels
    they were measured on, and are only quoted when it still matches (they cannot be re-measured live: PMC counters need
    rocprofv3)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mimamo-net_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = test infrastructure; used here only as the thing timed beside the HIP path)
# ---------------------------------------------------------------------------------------------------------
def _cpu_worker(first_clip, n_clips, threads, out_path, dedup):
    """One host process of the CPU baseline: the oracle on `n_clips` 64-frame clips with `threads` PyTorch threads.
    dedup=0: reference semantics incl. the 13x redundant pyramid (tester.py:122-139 on windowed input);
    dedup=1: one pyramid per unique frame (variant ii);
    dedup=2: "as tuned" -- variant ii with the ResNet50 trunk on torch.channels_last tensors (same ops and arithmetic; oneDNN's NCHW
    1x1 convolution, the reference's layout, is pathologically slow on AMD hosts).  Prints one JSON line with its timings."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mm_oracle
    import mimamo_net_amd  # noqa: F401
    from mimamo_net_amd import synthetic, sampler, weights
    torch.set_num_threads(threads)
    head_sd = weights.make_two_stream_state_dict(seed=0)
    resnet_sd = weights.make_resnet50_state_dict(seed=0)
    ids = sampler.window_ids(0, FRAMES_PER_CLIP, FRAMES_PER_CLIP)
    gray, rgb = synthetic.preprocess_host(synthetic.make_clip_u8(first_clip, FRAMES_PER_CLIP))
    mm_oracle.resnet50_pool5(resnet_sd, rgb[:4])                        # warm-up (thread pool, oneDNN primitives)
    tp = tr = th = 0.0
    t_start = time.time()
    for c in range(n_clips):
        if c:
            gray, rgb = synthetic.preprocess_host(synthetic.make_clip_u8(first_clip + c, FRAMES_PER_CLIP))
        t0 = time.time()
        if dedup:
            p0, p1 = mm_oracle.phase_diff_from_frames(gray, ids)
            p0, p1 = p0[None], p1[None]
        else:
            p0, p1 = mm_oracle.phase_diff_output(gray[ids][None])       # tester.py:122-139 (windowed, 13x redundant)
        t1 = time.time()
        feats = mm_oracle.resnet50_pool5(resnet_sd, rgb, channels_last=dedup == 2)   # resnet50_extractor.py:74-83
        t2 = time.time()
        out = mm_oracle.two_stream_forward(head_sd, p0, p1, feats[None])  # mimamo_net.py:129-143
        t3 = time.time()
        tp, tr, th = tp + t1 - t0, tr + t2 - t1, th + t3 - t2
        if c == 0 and out_path:
            np.save(out_path, out)
    print(json.dumps({"t_start": t_start, "t_end": time.time(), "phase": tp, "resnet": tr, "head": th,
                      "frames": n_clips * FRAMES_PER_CLIP}))


def _cpu_run(n_clips, procs, threads, dedup, out0):
    import subprocess
    per = max(1, n_clips // procs)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i * per), str(per), str(threads),
                            out0 if i == 0 else "", str(int(dedup))], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           env=env, universal_newlines=True) for i in range(procs)]
    recs = []
    for pr in ps:
        so, _ = pr.communicate()
        recs.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    wall = max(r["t_end"] for r in recs) - min(r["t_start"] for r in recs)
    frames = sum(r["frames"] for r in recs)
    return frames / wall, wall, procs * per, recs


def cpu_baseline(n_clips, resnet_sd):
    """The oracle on the host's cores: PyTorch's intra-op pool stops scaling at ~16 threads on this path (one thread per
    logical core collapses to 0.4 frames/s on the 256-thread bench host), so the host is filled with several
    processes of a calibrated thread count each, every process working on its own clips -- the same sharding the GPU
    path uses.  value = all frames / wall time from the first process's start of compute to the last one's end.
    Both variants of BASELINE.md section 3: (i) reference-faithful (`value`), (ii) deduplicated pyramid."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mm_oracle
    from mimamo_net_amd import synthetic
    ncpu = os.cpu_count() or 1
    gray, rgb = synthetic.preprocess_host(synthetic.make_clip_u8(0, FRAMES_PER_CLIP))
    best, best_t = 1, float("inf")
    for nt in sorted({min(ncpu, t) for t in (8, 16, 32)}):
        torch.set_num_threads(nt)
        mm_oracle.resnet50_pool5(resnet_sd, rgb[:4])
        t0 = time.time()
        mm_oracle.resnet50_pool5(resnet_sd, rgb[:16])
        dt = (time.time() - t0) * nt          # core-seconds: prefer the thread count that uses cores best
        if dt < best_t:
            best, best_t = nt, dt
    procs = max(1, min(n_clips, (ncpu // 2) // best))      # physical cores (SMT pairs) / threads per process
    tmp = tempfile.mkdtemp(prefix="mm_cpu_")
    out0 = os.path.join(tmp, "clip0.npy")
    v, wall, clips, recs = _cpu_run(n_clips, procs, best, 0, out0)
    v2, wall2, clips2, recs2 = _cpu_run(n_clips, procs, best, 1, "")
    v3, wall3, clips3, recs3 = _cpu_run(n_clips, procs, best, 2, "")
    cpu_out = np.load(out0)

    def stages(rs):
        return "phase %.1f s, resnet50 %.1f s, head %.1f s" % (np.mean([r["phase"] for r in rs]),
                                                                np.mean([r["resnet"] for r in rs]), np.mean([r["head"] for r in rs]))

    def stage_rates(rs):
        """Per process (its `threads` cores): GFLOP/s of the ResNet50 trunk and of the head on the direct-form counts of SURVEY 8(d)
        (7.712 / 0.396 GFLOP per frame), frames/s of the phase stage."""
        fr = float(np.mean([r["frames"] for r in rs]))
        return {"resnet50_GFLOP_per_s_per_process": fr * 7.712 / max(np.mean([r["resnet"] for r in rs]), 1e-9),
                "head_GFLOP_per_s_per_process": fr * 0.396 / max(np.mean([r["head"] for r in rs]), 1e-9),
                "phase_frames_per_s_per_process": fr / max(np.mean([r["phase"] for r in rs]), 1e-9)}
    cfg = torch.__config__.show()
    libs = "; ".join(l.strip(" -") for l in cfg.splitlines() if any(k in l for k in ("Math Kernel Library", "MKL-DNN", "OpenMP", "CPU capability")))
    return {"value": v, "unit": "frames/s", "cores": procs * best, "kind": "port", "cpu_model": cpu_model(),
            "logical_cpus": ncpu, "torch": torch.__version__, "torch_cpu_libraries": libs,
            "sample": "%d clips x 64 frames in %d processes x %d threads, oracle/mm_oracle.py on PyTorch-CPU fp32 with the "
                      "reference's semantics (13x redundant pyramid, NCHW tensors), %.1f s wall; per process: %s"
                      % (clips, procs, best, wall, stages(recs)),
            "stage_rates": stage_rates(recs),
            "deduplicated": {"value": v2, "unit": "frames/s",
                             "sample": "same clips/processes/threads, one pyramid per unique frame, %.1f s wall; per process: %s"
                                       % (wall2, stages(recs2))},
            "as_tuned": {"value": v3, "unit": "frames/s", "stage_rates": stage_rates(recs3),
                         "sample": "same clips/processes/threads, one pyramid per unique frame AND the ResNet50 trunk on "
                                   "torch.channels_last tensors (same ops, same fp32 arithmetic: oneDNN's NCHW 1x1 convolution -- the "
                                   "reference's layout -- is what the reference-faithful figure mostly measures on this host), "
                                   "%.1f s wall; per process: %s" % (wall3, stages(recs3))}}, cpu_out


# ---------------------------------------------------------------------------------------------------------
# compute back ends
# ---------------------------------------------------------------------------------------------------------
class HipCompute(object):
    """The product path: HotPath on libmimamo_hip.so (fails loudly without the library or a GPU)."""

    def __init__(self, args, device):
        import mimamo_net_amd  # noqa: F401
        from mimamo_net_amd import weights
        from mimamo_net_amd.pipeline import HotPath
        self.args, self.device = args, device
        self.head_sd = weights.make_two_stream_state_dict(seed=0)
        self.resnet_sd = weights.make_resnet50_state_dict(seed=0)
        self.hot = HotPath(self.head_sd, self.resnet_sd, device)
        self.hot.resnet.set_winograd(0 if args.no_winograd else args.winograd)
        self.pool = {}          # clip content id -> row in the device-resident pool
        self.frames_u8 = None   # [P*64,112,112,3] uint8
        self.pre = None         # (gray, rgb) of the pool for --from-f32
        self.n_forward = 0      # passes of the hot path this process has run (a profiler child reports it: the divisor of per-step counters)

    def load(self, content_ids, share=None):
        """Synthetic clips (seed 1000 + id) -> HBM, outside the timed region.
        share = (rank, world, distinct, collective device) at N > 1: the pool of `distinct` clip contents is generated ONCE across
        the job -- rank r makes contents [r * per, (r + 1) * per) -- and all-gathered (RCCL, device to device), so every rank holds the
        whole pool (any clip of the queue can run anywhere) and no content is generated twice."""
        from mimamo_net_amd import synthetic
        from mimamo_net_amd import stream as mstream
        if share is not None and share[1] > 1 and not self.args.from_f32:
            from mimamo_net_amd import dist as mdist
            rank, world, distinct, coll_dev = share
            per = (distinct + world - 1) // world
            mine = [c for c in range(rank * per, (rank + 1) * per)]
            part = np.zeros((per * FRAMES_PER_CLIP, 112, 112, 3), dtype=np.uint8)
            for i, c in enumerate(mine):
                if c < distinct:
                    part[i * FRAMES_PER_CLIP:(i + 1) * FRAMES_PER_CLIP] = synthetic.make_clip_u8(c, FRAMES_PER_CLIP)
            allf = mdist.all_gather_bytes(torch.from_numpy(part), world, coll_dev)[:distinct * FRAMES_PER_CLIP]
            self.pool = {c: c for c in range(distinct)}
            self.pool_generated_here = len([c for c in mine if c < distinct])
            self.frames_u8 = allf.to(self.device)
            self.frames_host = mstream.pin(allf.cpu()) if self.args.stream_input else None
            self._fs = None
            self._sel = {}
            return
        ids = sorted(set(int(c) for c in content_ids))
        self.pool = {c: i for i, c in enumerate(ids)}
        self.pool_generated_here = len(ids)
        clips = [synthetic.make_clip_u8(c, FRAMES_PER_CLIP) for c in ids]
        self.frames_host = mstream.pin(np.concatenate(clips))           # the raw boundary in page-locked host memory
        self.frames_u8 = self.frames_host.to(self.device)
        self._fs = None
        if self.args.from_f32:
            g, r = zip(*[synthetic.preprocess_host(c) for c in clips])
            self.pre = (torch.from_numpy(np.concatenate(g)).to(self.device), torch.from_numpy(np.concatenate(r)).to(self.device))
        self._sel = {}

    def _select(self, content_ids):
        """Frames of the step's clips: the pool itself when the step is the pool in order, else a device gather."""
        key = tuple(content_ids)
        sel = self._sel.get(key)
        if sel is None:
            slots = [self.pool[c] for c in content_ids]
            if slots == list(range(len(self.pool))):
                sel = False
            else:
                sel = (torch.tensor(slots, device=self.device, dtype=torch.int64)[:, None] * FRAMES_PER_CLIP
                       + torch.arange(FRAMES_PER_CLIP, device=self.device)[None, :]).reshape(-1)
            if len(self._sel) > 64:
                self._sel.clear()
            self._sel[key] = sel
        srcs = self.pre if self.args.from_f32 else (self.frames_u8,)
        return srcs if sel is False else tuple(t.index_select(0, sel) for t in srcs)

    def _host_pieces(self, content_ids):
        """Pinned host views of the step's clips, adjacent pool slots merged into one copy."""
        runs, F = [], FRAMES_PER_CLIP
        for c in content_ids:
            slot = self.pool[c]
            if runs and runs[-1][1] == slot:
                runs[-1][1] = slot + 1
            else:
                runs.append([slot, slot + 1])
        return [self.frames_host[a * F:b * F] for a, b in runs]

    def step_streamed(self, content_ids, next_ids=None, lanes=None):
        """The same step with its uint8 frames coming from pinned HOST memory: a dedicated copy stream uploads the NEXT step's
        frames into the other half of a double buffer while this step computes (stream.FrameStream); the compute streams only
        wait for the upload event of their own buffer.  Same rows as step(), bit for bit."""
        from mimamo_net_amd.stream import FrameStream
        rows = len(content_ids) * FRAMES_PER_CLIP
        if self._fs is None or self._fs.slots[0].shape[0] < rows:
            if self._fs is not None:
                self._fs.close()
            self._fs = FrameStream(self.device, max(rows, self.args.clips * FRAMES_PER_CLIP))
            self._fs_next = 0
        fs = self._fs
        key = tuple(content_ids)
        slot = fs.find(("step", key))
        if slot is None:                                  # cold start: nothing was prefetched for this step
            slot = self._fs_next
            fs.upload(slot, self._host_pieces(content_ids), ("step", key))
        if next_ids:
            fs.upload(1 - slot, self._host_pieces(next_ids), ("step", tuple(next_ids)))
        self._fs_next = 1 - slot
        frames = fs.acquire(slot)
        out = self._forward((frames,), len(content_ids), self.args.lanes if lanes is None else lanes, True)
        fs.release(slot)
        return out

    def step(self, content_ids, lanes=None):
        ins = self._select(content_ids)
        lanes = self.args.lanes if lanes is None else lanes
        return self._forward(ins, len(content_ids), lanes, not self.args.from_f32)

    def _forward(self, ins, n_clips, lanes, u8):
        self.n_forward += 1
        lengths = [FRAMES_PER_CLIP] * n_clips
        if lanes > 1:
            return self.hot.forward_lanes(ins, lengths, lanes, independent_clips=True, from_u8=u8)
        plan = self.hot.plan(lengths) if getattr(self, "_plan_n", None) != len(lengths) else self._plan
        self._plan, self._plan_n = plan, len(lengths)
        return self.hot.forward_u8(ins[0], plan, True) if u8 else self.hot.forward(ins[0], ins[1], plan, True)

    def sync(self):
        torch.cuda.synchronize()

    def sync_compute(self):
        torch.cuda.current_stream().synchronize()      # lanes have been joined into the current stream by forward_lanes


class StubCompute(object):
    """Launcher / work-queue plumbing test on hosts without a GPU (`--stub-compute`): rows depend only on
    (clip id, frame).  Never a measurement: the JSON line says data = "stub"."""

    def __init__(self, args, device):
        self.device = device

    pool_generated_here = 0

    def load(self, content_ids, share=None):
        pass

    def step(self, content_ids, lanes=None):
        c = torch.tensor(list(content_ids), dtype=torch.float32)[:, None].expand(len(content_ids), FRAMES_PER_CLIP)
        f = torch.arange(FRAMES_PER_CLIP, dtype=torch.float32)[None, :].expand_as(c)
        return torch.stack([c, f], -1).reshape(-1, 2).contiguous()

    def sync(self):
        pass

    def sync_compute(self):
        pass


# ---------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=32, help="64-frame clips per GPU per step")
    ap.add_argument("--total-clips", type=int, default=0,
                    help="size of the work queue (clips of the whole job); default: 10000 for N>1 (BASELINE configs[4]), "
                         "--clips for N=1")
    ap.add_argument("--whole-job", action="store_true", help="ignore --steps: walk the whole work queue once")
    ap.add_argument("--distinct-clips", type=int, default=0,
                    help="distinct synthetic clip contents (clip id mod this); default min(total, 2 x clips)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the direct-form / multi-snippet legs (extra)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic live (two child runs under rocprofv3 --pmc, ~1 min); quote the committed PMC summary")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--backend", default=None, help="torch.distributed backend for N>1 (default nccl = RCCL on a GPU host; "
                    "gloo only for plumbing tests, e.g. two ranks on one GPU with --same-device)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing only: create the process group (RCCL) even with one rank, to exercise the broadcast / "
                         "all-gather path on a single-GPU box")
    ap.add_argument("--no-cpu-bind", action="store_true", help="N > 1: do not pin the rank's host threads to its GPU's NUMA node")
    ap.add_argument("--no-step-gather", action="store_true",
                    help="diagnosis only: keep the process group but skip the per-step all-gather of the result rows")
    ap.add_argument("--stub-compute", action="store_true", help="testing only: no GPU work (launcher / work-queue plumbing)")
    ap.add_argument("--dump-out", default="", help="testing only: rank 0 saves the gathered rows of the last step (.npy)")
    ap.add_argument("--no-winograd", action="store_true", help="run every 3x3 layer in the direct implicit-GEMM form")
    ap.add_argument("--winograd", type=int, default=1, help="1 = default (F(4x4,3x3); conv2_x..conv4_x with the output transform fused into the position GEMMs), 2 = F(2x2,3x3), 4 = F(4x4,3x3) as three kernels everywhere, 5 = fused everywhere")
    ap.add_argument("--lanes", type=int, default=3, help="HIP streams the clips of a step are spread over (1 = single stream)")
    ap.add_argument("--from-f32", action="store_true",
                    help="start every step from host-preprocessed fp32 tensors (gray 48x48, RGB 224x224) instead of the "
                         "raw uint8 boundary")
    ap.add_argument("--from-u8", action="store_true", help="(default) start every step from uint8 112x112x3 frames in HBM")
    ap.add_argument("--stream-input", action="store_true",
                    help="every timed step takes its uint8 frames from pinned host memory through a copy stream, double-buffered "
                         "against compute (the headline keeps inputs resident in HBM; the default line reports this as extra.streamed)")
    ap.add_argument("--cpu-clips", type=int, default=16, help="64-frame clips timed on the host for cpu_baseline (all processes together)")
    return ap.parse_args(argv)


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":
        _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 0)
        return 0
    args = parse_args()
    if args.stream_input and args.from_f32:
        raise SystemExit("bench.py: --stream-input streams the uint8 boundary; it cannot be combined with --from-f32")
    import mimamo_net_amd  # noqa: F401
    from mimamo_net_amd import dist as mdist
    _, env_w, _ = mdist.env_world()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_w is None and args.gpus > 1:
        # self-spawn: one process per GPU with the torchrun environment; rank 0's stdout is the JSON line
        code, out0 = mdist.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.stdout.write(out0)
        sys.stdout.flush()
        return code
    if env_w is not None and env_w != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher environment" % (args.gpus, env_w))
    return run_rank(args)


def run_rank(args):
    from mimamo_net_amd import dist as mdist
    gpu = not args.stub_compute
    if gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    backend = args.backend or ("nccl" if gpu else "gloo")
    _, env_w, env_local = mdist.env_world()
    dev_index = 0 if args.same_device else env_local
    t_init = time.perf_counter()
    rank, world, local_rank = mdist.init(backend, device_index=dev_index if gpu else None, force=args.force_dist)
    init_s = time.perf_counter() - t_init
    # host side of a rank: its threads stay on the CPUs next to its GPU (NUMA node from sysfs; an even slice of the allowed CPUs
    # when the platform does not say).  N = 1 stays unbound: rank 0 alone runs the cpu_baseline on the whole host.
    # the ranks of THIS node (LOCAL_WORLD_SIZE from the launcher; one node = the whole job): a multi-node launch must not split the
    # node's CPUs by the global world size or probe device indices that live on other nodes
    local_world = max(1, min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    cpu_bind = mdist.bind_rank_cpus(local_rank, local_world,
                                    ([0] * local_world if args.same_device else list(range(local_world))) if gpu else None) \
        if world > 1 and not args.no_cpu_bind else {"numa_node": None, "cpus": len(os.sched_getaffinity(0)), "first_cpu": -1, "bound": False}
    if gpu:
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    else:
        device = torch.device("cpu")
    coll_dev = device if (gpu and backend == "nccl") else torch.device("cpu")

    # ---- the work queue: rank 0 builds it, everyone receives it (RCCL broadcast), everyone shards it the same way
    total = args.total_clips or (10000 if world > 1 else args.clips)
    work0 = np.stack([np.arange(total), np.full(total, FRAMES_PER_CLIP)], 1) if rank == 0 else None
    t_b = time.perf_counter()
    work = mdist.broadcast_work(work0, rank, world, coll_dev)
    bcast_ms = (time.perf_counter() - t_b) * 1e3     # first collective of the job: includes RCCL's communicator set-up
    assert work.shape == (total, 2) and (work[:, 1] == FRAMES_PER_CLIP).all()
    mine = [int(work[i, 0]) for i in mdist.shard(total, rank, world)]          # clip ids of this rank, queue order
    if not mine:
        raise SystemExit("rank %d has no clips: --total-clips %d < %d ranks" % (rank, total, world))
    distinct = args.distinct_clips or min(total, 2 * args.clips)
    per_step = min(args.clips, len(mine))
    if args.whole_job:
        steps_all = [mine[i:i + args.clips] for i in range(0, len(mine), args.clips)]
        n_steps = int(mdist.max_over_ranks(len(steps_all), coll_dev))
        warm = [steps_all[0]] * args.warmup
        timed = steps_all + [[]] * (n_steps - len(steps_all))
    else:
        n_steps = args.steps
        seq = [[mine[(s * per_step + k) % len(mine)] for k in range(per_step)] for s in range(args.warmup + n_steps)]
        warm, timed = seq[:args.warmup], seq[args.warmup:]
    content = lambda ids: [c % distinct for c in ids]                         # noqa: E731

    t_w = time.perf_counter()
    comp = (StubCompute if args.stub_compute else HipCompute)(args, device)
    weights_s = time.perf_counter() - t_w            # weights generated / folded / uploaded (each rank: replicated, 105 MB)
    t_l = time.perf_counter()
    comp.load({c % distinct for s in warm + timed for c in s}, share=(rank, world, distinct, coll_dev))
    load_s = time.perf_counter() - t_l
    startup_s = time.time() - T_PROCESS_START        # interpreter start -> ready to run the first warm-up step

    # The [frames,2] results of a step are all-gathered (8 B/frame, the only collective besides the queue broadcast).  It
    # is issued asynchronously on RCCL's stream and only waited for one step later, so a rank never stalls on a slower
    # peer inside a step -- ranks are independent shards and the job time is the slowest rank's, not a sum of maxima.
    pending = []
    last = {}

    def drain():
        while pending:
            h, bufs, keep = pending.pop()
            if h is not None:
                h.wait()
            last["gathered"] = bufs

    def step(ids, nxt=None):
        if not ids:
            out = torch.zeros((0, 2), device=device)
        elif args.stream_input and gpu:
            out = comp.step_streamed(content(ids), content(nxt) if nxt else None)
        else:
            out = comp.step(content(ids))
        if (world > 1 or args.force_dist) and not args.no_step_gather:
            drain()
            rows = args.clips * FRAMES_PER_CLIP
            if out.shape[0] != rows:      # ragged tail of the queue (--whole-job): pad to the step shape, NaN = no clip
                pad = torch.full((rows, 2), float("nan"), dtype=out.dtype, device=out.device)
                pad[: out.shape[0]] = out
                out_g = pad
            else:
                out_g = out
            h, bufs = mdist.all_gather_rows_async(out_g, world)
            pending.append((h, bufs, out_g))
        else:
            last["gathered"] = [out]
        return out

    def fence():
        drain()
        mdist.barrier()
        comp.sync()

    with torch.no_grad():
        for i, ids in enumerate(warm):
            out = step(ids, (warm + timed)[i + 1] if i + 1 < len(warm) + len(timed) else None)
        fence()
        t0 = time.perf_counter()
        for i, ids in enumerate(timed):
            out = step(ids, timed[i + 1] if i + 1 < len(timed) else None)
        # this rank's own finish time (its compute stream only: the result all-gather runs on RCCL's stream and the barrier
        # below waits for the slowest peer) -- reported per rank so a poor scaling number can be attributed
        comp.sync_compute()
        dt_local = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
    dt_own = dt
    dt = mdist.max_over_ranks(dt, coll_dev)
    frames_rank = sum(len(s) for s in timed) * FRAMES_PER_CLIP
    frames_all = total * FRAMES_PER_CLIP if args.whole_job else world * frames_rank
    if gpu:
        assert torch.isfinite(out).all()
    if args.dump_out and rank == 0:
        np.save(args.dump_out, torch.cat([t.cpu() for t in last["gathered"]], 0).numpy())

    n_frames = per_step * FRAMES_PER_CLIP
    result = {
        "metric": "face-frames/sec end-to-end (phase-diff + ResNet50 + 2-stream GRU), 64-frame clips",
        "value": frames_all / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": n_steps,
        "warmup": args.warmup,
        "ms_per_step": dt / max(n_steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if args.whole_job else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "stub (no GPU work: launcher plumbing test)" if args.stub_compute else "synthetic",
        "config": {"workload": ("full two-stream hot path (BASELINE configs[%d]): %d clips x 64 frames per GPU per step from "
                                "uint8 112x112x3 aligned-face frames resident in HBM (PIL-exact on-GPU preprocessing, pyramid + "
                                "phase difference, ResNet50 pool5, two-stream GRU head inside the timed region); random-init "
                                "weights of the reference architecture" % (4 if world > 1 else 3, per_step))
                               if not args.from_f32 else
                               ("full two-stream hot path (BASELINE configs[3]) from host-preprocessed fp32 tensors (gray 48x48 + "
                                "RGB 224x224): %d clips x 64 frames per GPU per step" % per_step),
                   "input": "preprocessed fp32 tensors" if args.from_f32 else
                            ("uint8 112x112x3 frames streamed from pinned host memory (copy stream, double-buffered)" if args.stream_input
                             else "uint8 112x112x3 frames"),
                   "lanes": args.lanes, "clips_per_gpu": per_step, "frames_per_step_per_gpu": n_frames,
                   "work_queue": {"total_clips": total, "clips_this_rank": len(mine), "distinct_clip_contents": distinct,
                                  "broadcast": backend if (world > 1 or args.force_dist) else None,
                                  "whole_job": bool(args.whole_job)},
                   "parallelism": "videos sharded, dp%d" % world},
    }
    if world > 1 or args.force_dist:
        # one row per rank, gathered with the collective the results use: enough to tell a slow rank (ms_per_step_local), a
        # rank held up by its peers (ms_per_step >> local) or a slow communicator set-up (queue_broadcast_ms) apart
        rows = mdist.all_gather_floats([rank, dt_local / max(n_steps, 1) * 1e3, dt_own / max(n_steps, 1) * 1e3, frames_rank,
                                        bcast_ms, load_s, dev_index if gpu else -1, startup_s, init_s, weights_s,
                                        cpu_bind["cpus"], cpu_bind["first_cpu"],
                                        -1 if cpu_bind["numa_node"] is None else cpu_bind["numa_node"], comp.pool_generated_here],
                                       coll_dev)
        result["per_rank"] = [{"rank": int(r[0]), "ms_per_step_local": r[1], "ms_per_step": r[2], "frames": int(r[3]),
                               "queue_broadcast_ms": r[4], "input_load_s": r[5], "device": int(r[6]),
                               "startup_s": r[7], "process_group_init_s": r[8], "weights_s": r[9],
                               "cpus": int(r[10]), "first_cpu": int(r[11]), "numa_node": None if r[12] < 0 else int(r[12]),
                               "clip_contents_generated": int(r[13])} for r in rows]
        result["config"]["work_queue"]["broadcast_ms_rank0"] = bcast_ms if rank == 0 else None
    if args.stub_compute:
        if rank == 0:
            print(json.dumps(result))
        mdist.shutdown()
        return 0

    # ---- roofline leg: hipEvent-timed launches of one more step (same stream), by kernel category
    from mimamo_net_amd import _lib
    L = _lib.lib()
    ms = (ctypes.c_double * 5)()
    work_ = (ctypes.c_double * 5)()
    launches = (ctypes.c_int64 * 5)()
    ids0 = content(timed[0] if timed[0] else warm[0] if warm else mine[:per_step])
    import tempfile
    dump_path = os.path.join(tempfile.mkdtemp(prefix="mm_prof_"), "launches.csv")
    os.environ["MM_PROF_DUMP"] = dump_path          # per-launch rows (category, work, ms, tag) for the mixed roofline below
    with torch.no_grad():
        comp.step(ids0, lanes=1)
        comp.sync()
        L.mm_profile_begin()
        # single stream for this leg: with several lanes in flight a kernel's event bracket also counts the time it
        # shares the GPU with another lane's kernel, which would under-state the per-kernel rate
        comp.step(ids0, lanes=1)
        rc = L.mm_profile_end(ms, work_, launches)
    assert rc == 0
    os.environ.pop("MM_PROF_DUMP", None)
    live = []
    try:
        with open(dump_path) as f:
            for line in f:
                c_, w_, t_, tag_ = line.rstrip("\n").split(",", 3)
                live.append((int(c_), float(w_), float(t_), tag_))
    except OSError:
        pass
    import shutil
    shutil.rmtree(os.path.dirname(dump_path), ignore_errors=True)
    conv_tflops = work_[0] / (ms[0] * 1e-3) / 1e12
    result["hot_path_steps_executed"] = comp.n_forward     # warm-up + timed + the two single-stream steps of this leg (read by a profiler parent)
    phase_ms = ms[1] + ms[2]
    phase_gbs = (work_[1] + work_[2]) / (phase_ms * 1e-3) / 1e9

    def committed_traffic(kind):
        """PMC HBM traffic from profiles/*<kind>_traffic*.json -- whichever summary was measured on THESE kernel sources (hash)
        at this step size; file names (rounds) do not matter.  None when the kernels changed since the last PMC run."""
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s_traffic*.json" % kind)), reverse=True):
            try:
                with open(path) as f:
                    tj = json.load(f)
            except (OSError, ValueError):
                continue
            if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("clips_per_gpu") == per_step:
                return tj.get("bytes_per_step"), os.path.basename(path)
        return None, None

    traffic, traffic_file = committed_traffic("conv")
    ptraffic, ptraffic_file = committed_traffic("phase")
    # (round 5) ... and measured LIVE when this is the default single-GPU run and rocprofv3 is on the box: two child runs of this script
    # under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes), so
    # that the driver's line shows a traffic regression by itself; any failure leaves the committed figures in place
    live_traffic = None
    if rank == 0 and world == 1 and not args.no_extra and not args.no_live_traffic:
        live_traffic = measure_live_traffic(args, per_step)
    live_note = None
    if live_traffic and live_traffic.get("conv"):
        live_note = ("HBM bytes per step over all conv launches, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE)*1024, measured LIVE by this run "
                     "(child runs of bench.py under rocprofv3: FETCH_SIZE, WRITE_SIZE and an SQ / GRBM pass, %.0f s in all); committed summary for the same sources: %s"
                     % (live_traffic["seconds"], ("%.4g bytes (profiles/%s)" % (traffic, traffic_file)) if traffic else "none"))
        traffic, traffic_file = live_traffic["conv"], "live"
        if live_traffic.get("phase"):
            ptraffic, ptraffic_file = live_traffic["phase"], "live"

    def mixed_roofline():
        """Sum over the launches of this step of max(FLOPs / MFMA peak, HBM bytes / HBM peak) against the sum of their measured
        (live, hipEvent) times: the fraction of the LIMITING roofline per launch.  FLOPs and times are live; the
        HBM bytes per launch are rocprofv3 PMC counters (tools/layer_roofline.sh) committed under profiles/ for these kernel
        sources (hash) -- they cannot be read without the profiler."""
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*layer_bytes*.json")), reverse=True):
            try:
                with open(path) as f:
                    lb = json.load(f)
            except (OSError, ValueError):
                continue
            if lb.get("kernel_source_hash") != kernel_source_hash() or lb.get("clips_per_gpu") != per_step:
                continue
            rows = lb["launches"]
            if len(rows) != len(live) or any(r["tag"] != l[3] or r["cat"] != l[0] for r, l in zip(rows, live)):
                return None, "profiles/%s lists other launches than this step ran (%d vs %d)" % (os.path.basename(path), len(rows), len(live))
            floor = meas = 0.0
            by_bound = {"mfma": [0.0, 0.0], "hbm": [0.0, 0.0]}
            for r, (c_, w_, t_, _) in zip(rows, live):
                t_m = (w_ if c_ == 0 else 0.0) / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
                t_h = r["pmc_bytes"] / (PEAK_HBM_GBS * 1e9) * 1e3
                floor += max(t_m, t_h)
                meas += t_
                b = by_bound["mfma" if t_m >= t_h else "hbm"]
                b[0] += max(t_m, t_h)
                b[1] += t_
            return {"step_floor_ms": floor, "measured_ms": meas, "mixed_frac": floor / meas, "launches": len(live),
                    "mfma_bound_launches": {"floor_ms": by_bound["mfma"][0], "measured_ms": by_bound["mfma"][1]},
                    "hbm_bound_launches": {"floor_ms": by_bound["hbm"][0], "measured_ms": by_bound["hbm"][1]},
                    "bytes_source": "committed PMC bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per launch, not live); times and FLOPs live",
                    "bytes_file": os.path.basename(path), "bytes_file_kernel_source_hash": lb.get("kernel_source_hash"),
                    "bytes_file_mtime": time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))}, None
        return None, "no per-launch PMC byte list under profiles/ for these kernel sources (hash %s)" % kernel_source_hash()

    mixed, mixed_note = mixed_roofline() if live else (None, "no per-launch dump")
    wino = 0 if args.no_winograd else {1: "F(4x4,3x3); output transform fused into the GEMMs for Cin <= 256", 2: "F(2x2,3x3)", 4: "F(4x4,3x3), three kernels", 5: "F(4x4,3x3), fused everywhere"}.get(args.winograd, args.winograd)
    result["roofline"] = {
        "bound": "mfma", "kernel": "conv_mfma_kernel (fp32 implicit-GEMM conv/GEMM engine, all %d launches of one step)" % launches[0],
        "achieved": conv_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": conv_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
        "traffic_note": live_note if live_note else
                        ("HBM bytes per step over all conv launches, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE)*1024, from "
                         "profiles/%s (same kernel sources, not live%s)" % (traffic_file, "; live measurement failed: " + live_traffic["error"]
                                                                         if live_traffic and live_traffic.get("error") else "")) if traffic else
                        "no PMC summary under profiles/ for these kernel sources (hash %s)" % kernel_source_hash(),
        "flops_per_step": work_[0], "ms_per_step": ms[0], "launches_per_step": int(launches[0]),
        "note": "flops = executed on the matrix cores; with Winograd F(4x4,3x3) (or F(2x2,3x3)) on the stride-1 3x3 layers of conv2_x..conv5_x "
                "that is less than the direct-form count (algorithmic_direct_flops_per_step = 8.108 GFLOP/frame)",
        "algorithmic_direct_flops_per_step": 8.108e9 * n_frames,
        # SURVEY 8(d) figure (direct-form 8.108 GFLOP/frame) over the conv launches AND the Winograd transforms
        "algorithmic_equiv_tflops": 8.108e9 * n_frames / ((ms[0] + ms[3]) * 1e-3) / 1e12,
        "algorithmic_equiv_frac": 8.108e9 * n_frames / ((ms[0] + ms[3]) * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "winograd": wino,
        # per-launch limiting roofline over EVERY kernel of the step (conv / GEMM, Winograd transforms, phase stage, pools,
        # preprocessing): sum of max(t_mfma, t_hbm) and its ratio to the sum of the measured launch times
        "step_floor_ms": mixed["step_floor_ms"] if mixed else None,
        "mixed_frac": mixed["mixed_frac"] if mixed else None,
        "mixed": mixed if mixed else {"note": mixed_note},
        "winograd_transforms": {"ms_per_step": ms[3], "bytes_per_step": work_[3], "launches_per_step": int(launches[3]),
                                "GB_per_s": (work_[3] / (ms[3] * 1e-3) / 1e9) if ms[3] > 0 else None,
                                "pmc_bytes_per_step_live": live_traffic.get("winograd_transforms") if live_traffic else None},
        "traffic_source": "live" if live_note else ("committed" if traffic else None)}
    # (round 6) the 157.3 TFLOP/s denominator is the peak at the NOMINAL 2.4 GHz; under fp32-MFMA load the chip sustains 2.0-2.4 GHz.  `frac`
    # stays on the nominal peak; beside it the clock the conv launches actually ran at (GRBM_GUI_ACTIVE / kernel time, the live SQ pass) and
    # the fraction of the peak AT THAT CLOCK, so that "0.89 at 2.36 GHz, 91 % busy" reads as done and "0.60 at 2.17 GHz, 61 % busy" as open
    clk, clk_rejected = plausible_clock(live_traffic.get("clock") if live_traffic else None)
    result["roofline"]["sustained_clock_GHz"] = clk["GHz"] if clk else None
    result["roofline"]["mfma_busy_frac"] = clk["mfma_busy"] if clk else None
    result["roofline"]["peak_at_sustained_clock"] = PEAK_FP32_MFMA_TFLOPS * clk["GHz"] / NOMINAL_CLOCK_GHZ if clk else None
    result["roofline"]["frac_at_sustained_clock"] = conv_tflops / (PEAK_FP32_MFMA_TFLOPS * clk["GHz"] / NOMINAL_CLOCK_GHZ) if clk else None
    result["roofline"]["clock_note"] = (clk["how"] if clk else clk_rejected if clk_rejected else
                                        "no live SQ / GRBM pass (%s)" % ((live_traffic or {}).get("sq_pass_error") or (live_traffic or {}).get("error")
                                                                         or "not requested: --no-extra / --no-live-traffic / N > 1"))
    # both floors of the phase stage: HBM (algorithmic bytes at 8 TB/s) and the matrix pipes (the pyramid products: 4 344
    # v_mfma_f32_16x16x4_f32 = 8.9 MFLOP per frame) -- the MFMA floor is the higher one
    ph_floor_hbm = (work_[1] + work_[2]) / (PEAK_HBM_GBS * 1e9) * 1e3
    ph_floor_mfma = n_frames * 4344 * 2048.0 / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
    ph_valu = live_traffic.get("valu") if live_traffic else None
    ph_kernels, ph_floor_sum = phase_floors(live, n_frames, ph_valu)
    ph_floor_valu = sum(k.get("floor_valu_ms", 0.0) for k in ph_kernels.values()) if ph_valu else None
    result["roofline_phase"] = {"bound": "hbm", "kernel": "pyramid_wave_kernel + phase_window2_kernel<48|24> (all phase-stage launches of one step)",
                                "achieved": phase_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": phase_gbs / PEAK_HBM_GBS,
                                "floor_hbm_ms": ph_floor_hbm, "floor_mfma_ms": ph_floor_mfma,
                                "floor_valu_ms": ph_floor_valu,
                                # per kernel the largest of its floors (HBM on its algorithmic bytes, MFMA for the pyramid, VALU issue from the
                                # live SQ_INSTS_VALU count), summed over the kernels: they run back to back on one stream
                                "kernels": ph_kernels,
                                "limiting_floor_ms": ph_floor_sum,
                                "limiting_floor": "+".join("%s:%s" % (t, k["limiter"]) for t, k in sorted(ph_kernels.items())),
                                "frac_of_limiting_floor": ph_floor_sum / phase_ms if phase_ms > 0 else None,
                                "valu_source": "live (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA, this run)" if ph_valu else
                                               "none: no live SQ pass -- floors are HBM / MFMA only",
                                "traffic": ptraffic, "traffic_file": ptraffic_file, "bytes_per_step": work_[1] + work_[2], "ms_per_step": phase_ms,
                                "ms_pyramid": ms[1], "ms_window": ms[2]}

    if rank == 0 and world == 1 and not args.no_extra:
        result["extra"] = extra_legs(args, comp, ids0, n_frames)

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_out = cpu_baseline(args.cpu_clips, comp.resnet_sd)
            result["cpu_baseline"] = cb
            # clip 0 of the CPU sample is content id 0: report the parity of the two paths next to the numbers
            with torch.no_grad():
                gpu0 = comp.step([0], lanes=1).cpu().numpy() if 0 in comp.pool else None
            if gpu0 is not None:
                result["parity_vs_cpu_sample"] = {"max_abs_err_valence_arousal": float(np.abs(gpu0 - cpu_out[0]).max()),
                                                  "tolerance": 1e-4}
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    mdist.shutdown()
    return 0


def extra_legs(args, comp, ids0, n_frames):
    """N=1 only, after the headline: the same step (a) in direct form (no Winograd), (b) on multi-snippet videos of the reference's run_example shape (309 frames -> 5 snippets, GRU seq_len 5,
    tail snippet overwriting; api/run_example.py:7-13, snippet_sampler.py:112-126)."""
    from mimamo_net_amd import synthetic
    hot, K = comp.hot, max(1, args.extra_steps)

    def timeit(fn):
        with torch.no_grad():
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K

    ex = {"steps": K}
    wino = 0 if args.no_winograd else args.winograd
    hot.resnet.set_winograd(0)
    dt = timeit(lambda: comp.step(ids0))
    hot.resnet.set_winograd(wino)
    ex["direct_form"] = {"value": n_frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                         "what": "same step with every 3x3 layer as a direct implicit GEMM (no Winograd)"}
    # (a2) the headline step with its input streamed over PCIe instead of resident in HBM
    if not args.from_f32:
        with torch.no_grad():
            comp.step_streamed(ids0, ids0)             # primes the pipeline: the first timed step's frames are in flight
            torch.cuda.synchronize()
            b0 = comp._fs.bytes_uploaded
            t0 = time.perf_counter()
            for i in range(K):                         # steady state: every step uploads the next step's frames while it computes
                comp.step_streamed(ids0, ids0)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / K
            up = (comp._fs.bytes_uploaded - b0) / K
        dtr = timeit(lambda: comp.step(ids0))
        ex["streamed"] = {"value": n_frames / dts, "unit": "frames/s", "ms_per_step": dts * 1e3,
                          "resident_value": n_frames / dtr, "resident_ms_per_step": dtr * 1e3, "streamed_over_resident": dtr / dts,
                          "pcie_bytes_per_step": up, "pcie_GB_per_s": up / dts / 1e9,
                          "what": "same step, uint8 frames uploaded from pinned host memory on a copy stream while the previous step "
                                  "computes (double buffer); resident_* = the headline form measured back to back in this leg"}
    # (b) multi-snippet videos
    nv = max(1, round(n_frames / EXAMPLE_VIDEO_FRAMES))
    vids = torch.from_numpy(np.concatenate([synthetic.make_clip_u8(5000 + v, EXAMPLE_VIDEO_FRAMES) for v in range(nv)])).to(comp.device)
    lengths = [EXAMPLE_VIDEO_FRAMES] * nv
    dt = timeit(lambda: hot.forward_lanes((vids,), lengths, min(args.lanes, nv), from_u8=True))
    plan = hot.plan(lengths)
    ex["multi_snippet"] = {"value": nv * EXAMPLE_VIDEO_FRAMES / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                           "videos_per_step": nv, "frames_per_video": EXAMPLE_VIDEO_FRAMES,
                           "snippets_per_video": len(plan["videos"][0]["ranges"]), "gru_seq_len": plan["groups"][0]["bs"],
                           "head_calls_per_step": len(plan["groups"]) if args.lanes <= 1 else None,
                           "what": "uint8 videos of the reference's run_example length; the GRU runs over each video's "
                                   "snippets; unique frames counted once (snippet rows: %d per video)"
                                   % (len(plan["videos"][0]["ranges"]) * plan["videos"][0]["T"])}
    del vids
    ex["bf16x3"] = bf16x3_leg(comp, ids0, n_frames, timeit)
    # (c) per-stage rates for the other single-GPU BASELINE configurations (SURVEY 8(d) "also per-stage frames/s")
    ex["phase_only"] = phase_only_leg(comp)
    ex["resnet50_only"] = resnet50_only_leg(comp)
    return ex


PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), measured 2 495


def bf16x3_leg(comp, ids0, n_frames, timeit):
    """OPTIONAL, never the headline (the reference computes in fp32; `dtype` of the line stays f32): the same step with the 1x1 layers of
    K >= 512 -- conv3_x's reduce convs, conv4_x / conv5_x's 1x1 layers and projection contractions, conv5_x's Winograd position GEMMs -- on the
    bf16 matrix pipes through a three-way bf16 split of both fp32 operands (6 of the 9 partial products, fp32 accumulate;
    mm_resnet50_set_precision, csrc/conv_mfma.hip X3).  Reported against the bf16 roof next to the native-fp32 time of the same launches."""
    from mimamo_net_amd import _lib
    import tempfile
    import shutil
    hot, L = comp.hot, _lib.lib()

    def x3_launches():
        """(ms, algorithmic FLOPs, launches) of the layers the mode touches, from one single-stream step under the measurement hook"""
        d = tempfile.mkdtemp(prefix="mm_prof_x3_")
        os.environ["MM_PROF_DUMP"] = os.path.join(d, "l.csv")
        ms, wk, ln = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
        with torch.no_grad():
            comp.step(ids0, lanes=1)
            comp.sync()
            L.mm_profile_begin()
            comp.step(ids0, lanes=1)
            L.mm_profile_end(ms, wk, ln)
        os.environ.pop("MM_PROF_DUMP", None)
        rows = []
        with open(os.path.join(d, "l.csv")) as f:
            for line in f:
                c_, w_, t_, tag_ = line.rstrip("\n").split(",", 3)
                rows.append((int(c_), float(w_), float(t_), tag_))
        shutil.rmtree(d, ignore_errors=True)
        return rows, ms[0]

    with torch.no_grad():
        ref = comp.step(ids0, lanes=1).clone()
    rows32, conv32 = x3_launches()
    hot.resnet.set_precision("bf16x3")
    try:
        dt = timeit(lambda: comp.step(ids0))
        with torch.no_grad():
            out = comp.step(ids0, lanes=1)
        diff = float((out - ref).abs().max())
        rows, conv_x3 = x3_launches()
    finally:
        hot.resnet.set_precision("fp32")
    sel = [r for r in rows if r[3].endswith(" x3")]

    def x3_layer(tag):           # the same layers in the fp32 schedule: 1x1 GEMM launches with K >= 512 (bulk + tail-split remainder rows)
        f = dict(kv.split("=") for kv in tag.split() if "=" in kv)
        return tag.startswith("M=") and " k1 " in tag + " " and int(f.get("K", 0)) >= 512
    sel32 = [r for r in rows32 if r[0] == 0 and x3_layer(r[3])]
    fl, ms = sum(r[1] for r in sel), sum(r[2] for r in sel)
    ms32 = sum(r[2] for r in sel32)
    return {"value": n_frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "dtype": "bf16x3 (fp32 in / out / accumulate)",
            "max_abs_diff_vs_fp32_outputs": diff, "conv_launches_ms": conv_x3, "conv_launches_ms_fp32": conv32,
            "layers": {"launches": len(sel), "algorithmic_flops": fl, "ms": ms, "ms_fp32_mfma": ms32,
                       "fp32_equivalent_TFLOP_per_s": fl / (ms * 1e-3) / 1e12 if ms else None,
                       "bf16_executed_TFLOP_per_s": 6 * fl / (ms * 1e-3) / 1e12 if ms else None,
                       "frac_of_bf16_mfma_peak": 6 * fl / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS if ms else None},
            "what": "NOT the headline: same step, 1x1 layers with K >= 512 as six bf16 MFMA products of three-way split fp32 operands "
                    "(fp32 accumulate); everything else fp32 as in the headline"}


def _time_stream(fn, budget_s=1.0, min_iters=3):
    """Mean seconds per call on the current stream: warm-up call, then as many calls as fit `budget_s` (at least min_iters),
    bracketed by device synchronisation."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    iters = max(min_iters, min(200, int(budget_s / one)))
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, iters


def phase_only_leg(comp):
    """BASELINE configs[1]: steerable pyramid + phase difference only, 64-frame clips, one stream.  Input = the gray 48x48 fp32
    frames (the pyramid's input, api/phase_difference_extractor.py:38; from the bench's textured clips -- the wrap rate matters:
    the window kernel skips blurs until a window's first wrap), output = phase_0 [N,48,48,24] + phase_1 into PhaseNet's concat buffer,
    i.e. exactly the phase stage of the full step."""
    from mimamo_net_amd import sampler
    hot, dev = comp.hot, comp.device
    if hot._pre is None:
        from mimamo_net_amd.preprocess import FramePreprocessor
        hot._pre = FramePreprocessor(phase_size=hot.phase_size, mean=hot.resnet.meta['mean'], device=dev)
    out = {"unit": "frames/s", "algorithmic_bytes_per_frame": 285696,
           "what": "pyramid_wave_kernel + phase_window2_kernel<48|24> on gray 48x48 fp32 frames resident in HBM, 13-frame clamped windows "
                   "of 64-frame clips, single stream (BASELINE configs[1]); GB_per_s on SURVEY 8(d)'s 285 696 B/frame"}
    pool_clips = comp.frames_u8.shape[0] // FRAMES_PER_CLIP
    with torch.no_grad():
        gray_pool = hot._pre(comp.frames_u8, want_rgb=False)[0]
        for clips in (32, 256):
            n = clips * FRAMES_PER_CLIP
            reps = (clips + pool_clips - 1) // pool_clips
            gray = gray_pool.repeat(reps, 1, 1)[:n].contiguous()
            ids = torch.from_numpy(np.concatenate([sampler.window_ids(0, FRAMES_PER_CLIP, FRAMES_PER_CLIP) + FRAMES_PER_CLIP * c
                                                   for c in range(clips)]).astype(np.int32)).to(dev)
            dt, iters = _time_stream(lambda: hot.pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64,
                                                                       ids_checked=True), 0.6)
            out["clips_%d" % clips] = {"value": n / dt, "ms_per_pass": dt * 1e3, "frames": n, "iters": iters,
                                       "GB_per_s": n * 285696 / dt / 1e9, "frac_of_hbm_peak": n * 285696 / dt / 1e9 / PEAK_HBM_GBS}
            del gray, ids
    return out


def resnet50_only_leg(comp):
    """BASELINE configs[2]: ResNet50 pool5 extractor on synthetic 224x224 face batches, fp32 [B,3,224,224] in the reference's NCHW
    layout resident in HBM (the layout conversion is part of the call), batch 64 (the reference's `run()` batch,
    api/resnet50_extractor.py:42) and 256, single stream."""
    hot, dev = comp.hot, comp.device
    out = {"unit": "frames/s", "algorithmic_flops_per_frame": 7.712e9,
           "what": "Resnet50_Extractor.get_vec on fp32 [B,3,224,224] (255*x - mean) resident in HBM, single stream (BASELINE configs[2]); "
                   "TFLOP_per_s on SURVEY 8(d)'s direct-form 7.712 GFLOP/frame (Winograd executes fewer)"}
    with torch.no_grad():
        for bs in (64, 256):
            x = (torch.rand((bs, 3, 224, 224), device=dev) * 255.0 - 110.0).contiguous()
            dt, iters = _time_stream(lambda: hot.resnet.get_vec(x), 1.0)
            out["batch_%d" % bs] = {"value": bs / dt, "ms_per_batch": dt * 1e3, "iters": iters,
                                    "TFLOP_per_s_algorithmic": bs * 7.712e9 / dt / 1e12,
                                    "frac_of_fp32_mfma_peak_algorithmic": bs * 7.712e9 / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS}
            del x
    return out


if __name__ == "__main__":
    sys.exit(main())
