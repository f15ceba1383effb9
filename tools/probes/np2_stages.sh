# per-stage kernel times of runtime-specialised four-step plans (rocprofv3 kernel stats)
export TMPDIR=/tmp
for spec in "40000 3355" "120000 1118" "68640 1955" "1000000 134" "131072 1024"; do
  set -- $spec
  d=gpurun_out/np2_stages/$1; mkdir -p $d
  if [ $1 = 131072 ]; then export PFFT_NO_XCD_LOCAL=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --manual d=cpx,n=$1,b=$2 --precision float --no-cpu-baseline --steps 20 --warmup 3 > $d/log.txt 2>&1
  f=$(ls $d/*/*kernel_stats.csv | head -1)
  echo "== n=$1 batch $2 (1 GiB): $(grep -o 'radices/factors [0-9x]*' $d/log.txt | head -1)"
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"]
    if "stockham" not in name and "pfa" not in name: continue
    print("   %-90s calls %5s avg %9.1f us  total %8.2f ms" % (name[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
