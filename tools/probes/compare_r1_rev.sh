# order check of compare_r1.sh: current tree first, round-1 tree second, then current again (perf_generic + perf_strided)
for t in perf_generic perf_strided; do
  python tools/$t.py 2>&1 | grep "TB/s" | awk '{for (i=1;i<=NF;i++) if ($i=="ms") ms=$(i-1); print ms}' > /tmp/a.txt
  (cd build/r1src && python tools/$t.py 2>&1 | grep "TB/s" | awk '{for (i=1;i<=NF;i++) if ($i=="ms") ms=$(i-1); print ms}') > /tmp/b.txt
  python tools/$t.py 2>&1 | grep "TB/s" | awk '{n=""; for (i=1;i<=NF;i++) if ($i ~ /^tiers/) break; else n=n" "$i; for (i=1;i<=NF;i++) if ($i=="ms") ms=$(i-1); print n "|" ms}' > /tmp/c.txt
  paste -d'|' /tmp/c.txt /tmp/a.txt /tmp/b.txt | awk -F'|' '{printf "%-44s now1 %s  r1 %s  now2 %s   r1/now1 %.3f r1/now2 %.3f\n", $1, $3, $4, $2, $4/$3, $4/$2}'
done
