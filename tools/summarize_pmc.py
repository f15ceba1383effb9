#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, csv output) into the per-launch HBM traffic
figure used by bench.py's roofline.traffic.

    tools/summarize_pmc.py gpurun_out/pmc profiles/r1_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read, WRITE_SIZE is exact.  The
halving is re-checked here on a calibration kernel with a known byte count and the FFT kernel's own access shape
(tools/probes/copy_sweep2.hip: 32 KiB rows, 8 B per lane): cal_* directories.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(directory, counter):
    files = glob.glob("%s/runc/*_counter_collection.csv" % directory)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(src, dst):
    out = {"source": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python3 bench.py --steps 5 "
                     "--warmup 2 --no-cpu-baseline (one pass per counter)", "unit_of_counters": "KiB"}
    cal_f = per_kernel(src + "/cal_FETCH_SIZE", "FETCH_SIZE")
    cal_w = per_kernel(src + "/cal_WRITE_SIZE", "WRITE_SIZE")
    known_kib = 2 * 1024 * 1024  # the calibration copy reads 2 GiB and writes 2 GiB per launch
    kf = [v for k, vs in cal_f.items() if "rows_copy" in k for v in vs]
    kw = [v for k, vs in cal_w.items() if "rows_copy" in k for v in vs]
    fetch_corr = known_kib / (sum(kf) / len(kf))
    write_corr = known_kib / (sum(kw) / len(kw))
    out["calibration"] = {"known_KiB_per_launch": known_kib, "FETCH_SIZE_mean": sum(kf) / len(kf),
                          "WRITE_SIZE_mean": sum(kw) / len(kw), "fetch_correction": round(fetch_corr, 4),
                          "write_correction": round(write_corr, 4)}
    f = [v for k, vs in per_kernel(src + "/FETCH_SIZE", "FETCH_SIZE").items() if "stockham_wg" in k for v in vs]
    w = [v for k, vs in per_kernel(src + "/WRITE_SIZE", "WRITE_SIZE").items() if "stockham_wg" in k for v in vs]
    fetch_b = sum(f) / len(f) * 1024 * 2.0  # guide's gfx950 correction (confirmed by the calibration above)
    write_b = sum(w) / len(w) * 1024
    out["kernel"] = "stockham_wg_prefetch_kernel<f32, 16x16x16, wg256> (bench.py workload: N=4096 batch=65536)"
    out["launches_sampled"] = {"FETCH_SIZE": len(f), "WRITE_SIZE": len(w)}
    out["FETCH_SIZE_mean_KiB"] = sum(f) / len(f)
    out["WRITE_SIZE_mean_KiB"] = sum(w) / len(w)
    out["hbm_read_bytes_per_launch"] = fetch_b
    out["hbm_write_bytes_per_launch"] = write_b
    out["traffic_bytes_per_launch"] = fetch_b + write_b
    out["algorithmic_bytes_per_launch"] = 2.0 * 4096 * 65536 * 8
    out["traffic_over_algorithmic"] = round((fetch_b + write_b) / out["algorithmic_bytes_per_launch"], 4)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
