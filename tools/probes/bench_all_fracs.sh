#!/bin/bash
# event-timed roofline fraction of every bench.py config (run through gpurun): a regression sweep against profiles/r6_bench_*.json
cd "$GRAFT_REPO_ROOT" || exit 1
export PFFT_JIT_CACHE_DIR=$GRAFT_REPO_ROOT/build/jit_cache
for c in c2 c3 c5 ref16 ref256 ref4096 ref65536 ref9800 ref15360 ref68640 g32_14 g32_15 g32_17 g32_18 g32_19 g32_20 g32_21 g32_22 g32_24 g64_13 g64_14 g64_16 g64_17 g64_18 bi32_2048 bi64_2048; do
  out=$(python bench.py --config $c --steps 20 --no-cpu-baseline 2>/dev/null | tail -1)
  python3 - "$c" "$out" <<'PY'
import json, sys, glob, os
c, line = sys.argv[1], sys.argv[2]
d = json.loads(line); r = d["roofline"]
ref = None
p = os.path.join("profiles", "r6_bench_%s.json" % c)
if os.path.exists(p):
    ref = json.loads(open(p).read().strip().splitlines()[-1])["roofline"]["frac"]
print("%-10s frac %.4f  wall %.4f  kernel_ms %.4f  committed %s  %s" % (c, r["frac"], r.get("frac_wall") or 0, r["kernel_ms"], ("%.4f" % ref) if ref else "-", ("%+.1f %%" % (100 * (r["frac"] / ref - 1))) if ref else ""))
PY
done
