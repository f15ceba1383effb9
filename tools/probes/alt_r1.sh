# alternating runs of a few layouts under the round-1 tree and the current tree (3 rounds): ms per execute
run() { (cd $1; f=tools/probes/one_layout.py; [ -f $f ] || f=tools/one_layout.py; python3 $f $2 $3 $4 $5 $6 2>&1 | grep TB/s | tail -1 | awk '{for (i=1;i<=NF;i++) if ($i=="ms") print $(i-1)}'); }
for spec in "f32 30000 4096 P P" "f32 2985984 40 P P" "f32 1024 262144 P BI" "f64 1024 131072 BI BI" "f32 1000 262144 BI BI" "f32 4096 65536 BI BI"; do
  a=""; b=""
  for i in 1 2 3; do a="$a $(run build/r1src $spec)"; b="$b $(run . $spec)"; done
  echo "$spec | r1:$a | now:$b"
done
