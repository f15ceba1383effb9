# FETCH_SIZE / WRITE_SIZE of the gang-synchronous prototype (tools/probes/xcd_gang.hpp) next to the ticket-queue kernel and the
# two-launch pair of the same stage bodies: does its 1-3 MiB ring stay in the L2?
export TMPDIR=/tmp
out=gpurun_out/r5_gang; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  TUNE_REPS=2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- ./build/tune_gang1_16 > $out/pmc_$c.log 2>&1
done
python3 - $out <<'PY' | tee $out/gang_pmc.txt
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(root + "/pmc_%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[(r["Kernel_Name"][:110], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))][c].append(float(r["Counter_Value"]))
alg = 4096 * 65536 * 8 / 1024.0  # KiB read (= written) per launch over the whole batch
for k, v in sorted(agg.items()):
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
    w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print("%-112s grid %-8s lds %-6s launches %d  FETCH x2 / alg %.2f   WRITE / alg %.2f" % (k[0], k[1], k[2], len(v["FETCH_SIZE"]), 2 * f / alg, w / alg))
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
