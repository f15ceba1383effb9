# whole-round regression check (set-up: git worktree add -f build/r1src <round-1 commit>; make -C build/r1src/portfft_amd/csrc):
# the same survey scripts under the round-1 tree (build/r1src, its own library) and
# under the current tree, side by side:  tools/probes/compare_r1.sh > out.txt   (run through gpurun)
for t in perf_survey perf_f64s perf_strided perf_generic perf_global_f32 perf_global_np2 perf_split perf_split_global perf_unpacked; do
  (cd build/r1src && python tools/$t.py 2>&1 | grep "TB/s" | awk '{n=""; for (i=1;i<=NF;i++) if ($i ~ /^tiers/) break; else n=n" "$i; for (i=1;i<=NF;i++) if ($i=="ms") ms=$(i-1); print n "|" ms}') > /tmp/r1_$t.txt
  python tools/$t.py 2>&1 | grep "TB/s" | awk '{n=""; for (i=1;i<=NF;i++) if ($i ~ /^tiers/) break; else n=n" "$i; for (i=1;i<=NF;i++) if ($i=="ms") ms=$(i-1); print n "|" ms}' > /tmp/now_$t.txt
  python3 - $t <<'PY'
import sys
t = sys.argv[1]
def rd(p):
    d = {}
    for l in open(p):
        k, v = l.rsplit("|", 1)
        d[k.strip()] = float(v)
    return d
a, b = rd("/tmp/r1_%s.txt" % t), rd("/tmp/now_%s.txt" % t)
for k in b:
    if k in a:
        r = a[k] / b[k]
        print("%-18s %-44s r1 %8.4f ms  now %8.4f ms  speed-up %.3f%s" % (t, k, a[k], b[k], r, "   <-- slower" if r < 0.97 else ""))
    else:
        print("%-18s %-44s (new row) now %8.4f ms" % (t, k, b[k]))
PY
done
