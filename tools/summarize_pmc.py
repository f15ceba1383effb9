#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, csv output) into the per-execute HBM traffic
figure used by bench.py's roofline.traffic.

    tools/summarize_pmc.py <src> <dst.json> --config c3 --kernels stockham_strided --alg-bytes 4294967296 \
        [--cal-dir <dir> --cal-kernel 'copy_cols<128' --cal-kib 2097152]

<src>/FETCH_SIZE/runc/*_counter_collection.csv and <src>/WRITE_SIZE/runc/*_counter_collection.csv come from
    rocprofv3 --pmc <counter> --kernel-trace --output-format csv -d <src>/<counter>/runc -- python3 bench.py ...
(tools/run_pmc.sh).  Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters
are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, WRITE_SIZE is exact.  The
guide calibrates that for 16 B per lane only, so the factor is re-measured on a copy kernel with a known byte count and
the access shape of the kernel in question (--cal-*: tools/probes/copy_sweep2.hip for 8 B per lane rows,
tools/probes/copy_strided.hip for 128-byte column segments at 16 B per lane) and the measured factor is what is applied.
An execute may consist of several launches (two-pass plans): the per-launch means of every matching kernel are summed.
"""
import argparse
import collections
import csv
import glob
import json


def per_kernel(directory, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/**/*_counter_collection.csv" % directory, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def short(name):
    return name if len(name) <= 160 else name[:157] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--kernels", default="stockham_", help="comma-separated substrings of the kernels of one execute")
    ap.add_argument("--alg-bytes", type=float, required=True)
    ap.add_argument("--label", default="")
    ap.add_argument("--cal-dir", default=None, help="directory with cal FETCH_SIZE / WRITE_SIZE passes of a copy kernel")
    ap.add_argument("--cal-kernel", default="rows_copy")
    ap.add_argument("--cal-kib", type=float, default=2 * 1024 * 1024, help="KiB the calibration kernel reads (= writes)")
    ap.add_argument("--launches-per-execute", type=int, default=1,
                    help="launches per execute of the LEAST-launched matching kernel (plans that run chunk by chunk)")
    a = ap.parse_args()
    subs = [s for s in a.kernels.split(",") if s]
    out = {"config": a.config, "label": a.label,
           "source": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python3 bench.py --config %s "
                     "--steps 5 --warmup 2 --no-cpu-baseline (one pass per counter, tools/run_pmc.sh)" % a.config,
           "unit_of_counters": "KiB"}
    fetch_corr, write_corr = 2.0, 1.0
    if a.cal_dir:
        kf = [v for k, vs in per_kernel(a.cal_dir + "/FETCH_SIZE", "FETCH_SIZE").items() if a.cal_kernel in k for v in vs]
        kw = [v for k, vs in per_kernel(a.cal_dir + "/WRITE_SIZE", "WRITE_SIZE").items() if a.cal_kernel in k for v in vs]
        if kf and kw:
            fetch_corr = a.cal_kib / (sum(kf) / len(kf))
            write_corr = a.cal_kib / (sum(kw) / len(kw))
            out["calibration"] = {"kernel": a.cal_kernel, "known_KiB_per_launch": a.cal_kib,
                                  "FETCH_SIZE_mean": sum(kf) / len(kf), "WRITE_SIZE_mean": sum(kw) / len(kw),
                                  "fetch_correction": round(fetch_corr, 4), "write_correction": round(write_corr, 4),
                                  "launches": [len(kf), len(kw)]}
    if "calibration" not in out:
        out["calibration"] = {"note": "guide's gfx950 factors (FETCH_SIZE x2, WRITE_SIZE x1); "
                                      "confirmed for 8 B per lane rows in profiles/r1_pmc_traffic.json"}
    fetch = per_kernel(a.src + "/FETCH_SIZE", "FETCH_SIZE")
    write = per_kernel(a.src + "/WRITE_SIZE", "WRITE_SIZE")
    kernels = []
    read_b = write_b = 0.0
    matching = [n for n in set(fetch) & set(write) if any(sub in n for sub in subs) and fetch[n] and write[n]]
    least = min((len(fetch[n]) for n in matching), default=0)
    for name in sorted(set(fetch) | set(write)):
        if not any(s in name for s in subs):
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        if not f or not w:
            continue
        # kernels of one plan may run a different number of chunks (three-stage plan: 4 + 4 + 1 launches): a kernel
        # sampled k times as often as the least-sampled one runs k times as many launches per execute
        ratio = len(f) / float(least) if least else 1.0
        rb = sum(f) / len(f) * 1024 * fetch_corr * a.launches_per_execute * ratio
        wb = sum(w) / len(w) * 1024 * write_corr * a.launches_per_execute * ratio
        kernels.append({"kernel": short(name), "launches_sampled": [len(f), len(w)],
                        "launches_per_execute": a.launches_per_execute * ratio,
                        "FETCH_SIZE_mean_KiB": sum(f) / len(f), "WRITE_SIZE_mean_KiB": sum(w) / len(w),
                        "hbm_read_bytes_per_execute": rb, "hbm_write_bytes_per_execute": wb})
        read_b += rb
        write_b += wb
    out["kernels"] = kernels
    out["hbm_read_bytes_per_launch"] = read_b
    out["hbm_write_bytes_per_launch"] = write_b
    out["traffic_bytes_per_launch"] = read_b + write_b
    out["algorithmic_bytes_per_launch"] = a.alg_bytes
    out["hbm_passes"] = len(kernels)
    out["launches_of_each_kernel_per_execute"] = a.launches_per_execute
    out["traffic_over_algorithmic"] = round((read_b + write_b) / a.alg_bytes, 4) if a.alg_bytes else None
    with open(a.dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
    for k in kernels:
        print("  %-100s read %.4g B write %.4g B" % (k["kernel"][:100], k["hbm_read_bytes_per_execute"],
                                                    k["hbm_write_bytes_per_execute"]))


if __name__ == "__main__":
    main()
