#!/usr/bin/env python
"""Per-launch LIMITING roofline of one single-stream step (SURVEY 8(d) "fraction of the limiting roofline").

Joins, launch by launch and in launch order,
  * the library's measurement-hook dump of a plain run (MM_PROF_DUMP of tools/layer_table.py: category, executed FLOPs or
    algorithmic bytes, hipEvent ms, shape tag), and
  * rocprofv3 PMC rows of the same program under --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace;
    HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md)
and prints per shape: launches, measured ms, executed GFLOP, PMC GB, t_mfma = FLOPs / 157.3 TFLOP/s, t_hbm = bytes / 8 TB/s, the
floor max(t_mfma, t_hbm) and floor / measured; the totals give roofline.step_floor_ms and roofline.mixed_frac of the bench line.
Writes the per-launch byte list (stamped with the kernel-source hash) that bench.py joins with ITS live hipEvent times.

usage: layer_roofline.py <dump.csv> <fetch_dir> <write_dir> <clips> <out.txt> <out.json>   (driven by tools/layer_roofline.sh)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TFLOPS, PEAK_GBS = 157.3, 8000.0
NAMES = {0: "conv", 1: "pyramid", 2: "window", 3: "wino-xf", 4: "other"}


def kernel_of(tag):
    """substring a dispatch's Kernel_Name must contain for a dump row with this tag"""
    if tag.startswith("wino-fused"):
        return "wino_fused_kernel"
    if tag.startswith("M="):
        return "conv_mfma_kernel"
    return {"phase_window2<48>": "phase_window2_kernel", "phase_window2<24>": "phase_window2_kernel", "pyramid_frame": "pyramid_wave_kernel",
            "maxpool3x3s2": "maxpool_kernel", "maxpool+reduce64": "maxpool_reduce64_kernel", "vpool+reduce64": "maxpool_reduce64_kernel", "avgpool": "avgpool", "gru_gates": "gru_gates_kernel"}.get(tag, tag)


def pmc_rows(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and "mm::" in r["Kernel_Name"]]
    rows.sort()
    return rows


def main():
    dump, fdir, wdir, clips, out_txt, out_json = sys.argv[1:7]
    clips = int(clips)
    launches = []
    for line in open(dump):
        cat, work, t, tag = line.rstrip("\n").split(",", 3)
        launches.append({"cat": int(cat), "work": float(work), "ms": float(t), "tag": tag})
    n = len(launches)
    per = {}
    for counter, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
        rows = pmc_rows(d, counter)
        assert len(rows) % n == 0 and len(rows) >= n, "%s: %d mm:: dispatches do not tile %d hooked launches" % (counter, len(rows), n)
        rows = rows[-n:]                      # the last pass of the program = the profiled step (earlier ones are its warm-ups)
        for l, (_, name, v) in zip(launches, rows):
            assert kernel_of(l["tag"]) in name, (l["tag"], name)
        per[counter] = [v for _, _, v in rows]
    for l, f, w in zip(launches, per["FETCH_SIZE"], per["WRITE_SIZE"]):
        l["pmc_bytes"] = (2.0 * f + w) * 1024.0
    agg = collections.OrderedDict()
    for l in launches:
        a = agg.setdefault((l["cat"], l["tag"]), {"n": 0, "ms": 0.0, "flops": 0.0, "alg": 0.0, "pmc": 0.0, "floor": 0.0})
        fl = l["work"] if l["cat"] == 0 else 0.0
        t_m, t_h = fl / (PEAK_TFLOPS * 1e12) * 1e3, l["pmc_bytes"] / (PEAK_GBS * 1e9) * 1e3
        l["floor_ms"] = max(t_m, t_h)
        a["n"] += 1; a["ms"] += l["ms"]; a["flops"] += fl; a["alg"] += 0.0 if l["cat"] == 0 else l["work"]; a["pmc"] += l["pmc_bytes"]
        a["floor"] += l["floor_ms"]
        a["t_m"] = a.get("t_m", 0.0) + t_m
        a["t_h"] = a.get("t_h", 0.0) + t_h
    lines = ["%-8s %-46s %4s %9s %9s %9s %8s %8s %8s %6s  %s" % ("kind", "shape", "n", "ms", "GFLOP", "PMC GB", "t_mfma", "t_hbm", "floor", "f/ms", "bound  rate")]
    tot = collections.Counter()
    for (cat, tag), a in agg.items():
        bound = "mfma" if a["t_m"] >= a["t_h"] else "hbm"
        rate = "%7.1f TFLOP/s" % (a["flops"] / a["ms"] / 1e9) if cat == 0 else "%7.0f GB/s (PMC)" % (a["pmc"] / a["ms"] / 1e6)
        lines.append("%-8s %-46s x%-3d %9.3f %9.1f %9.3f %8.3f %8.3f %8.3f %6.3f  %-5s %s" % (
            NAMES[cat], tag, a["n"], a["ms"], a["flops"] / 1e9, a["pmc"] / 1e9, a["t_m"], a["t_h"], a["floor"], a["floor"] / a["ms"], bound, rate))
        tot["ms"] += a["ms"]; tot["floor"] += a["floor"]; tot["pmc"] += a["pmc"]; tot["flops"] += a["flops"]
        tot["ms_%d" % cat] += a["ms"]; tot["floor_%d" % cat] += a["floor"]
    lines.append("totals: %d launches, measured %.2f ms, step_floor_ms (sum of max(t_mfma, t_hbm)) %.2f ms, mixed_frac %.4f; PMC %.1f GB, executed %.2f TFLOP, frames %d"
                 % (n, tot["ms"], tot["floor"], tot["floor"] / tot["ms"], tot["pmc"] / 1e9, tot["flops"] / 1e12, clips * 64))
    lines.append("by kind: " + ", ".join("%s %.2f ms (floor %.2f)" % (NAMES[c], tot["ms_%d" % c], tot["floor_%d" % c]) for c in sorted(NAMES) if tot["ms_%d" % c]))
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    sys.path.insert(0, ROOT)
    import bench
    json.dump({"kernel_source_hash": bench.kernel_source_hash(), "clips_per_gpu": clips, "peak_tflops": PEAK_TFLOPS, "peak_GBs": PEAK_GBS,
               "how": "tools/layer_roofline.sh: tools/layer_table.py (one single-stream step, winograd default) plain for times and under "
                      "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (separate passes); per launch (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B",
               "step_floor_ms": tot["floor"], "measured_ms": tot["ms"], "mixed_frac": tot["floor"] / tot["ms"],
               "launches": [{"cat": l["cat"], "tag": l["tag"], "pmc_bytes": l["pmc_bytes"]} for l in launches]}, open(out_json, "w"), indent=0)


if __name__ == "__main__":
    main()
