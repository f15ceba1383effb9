# the lanes rule for prime-factor lengths (jit.cpp choose_spec_params) against the planner's earlier choice (PFFT_NO_PRIME_LANES=1)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], d['config']['parity_rel_l2_vs_numpy'], r['kernel'][20:95])"; }
man() { python bench.py --manual d=cpx,n=$1,b=$2 --precision $3 --no-cpu-baseline --steps 30 2>/dev/null | pr; }
for prec in float double; do
  es=8; [ $prec = double ] && es=16
  for n in 976 1696 2021 3481 2368 2624 592 1376 1952 3392 944 1369 1763 3721 1480 2928 1184; do
    b=$(( (1<<30) / (n*es) ))
    echo "$prec n=$n"
    echo -n "   rule:   "; man $n $b $prec
    echo -n "   before: "; PFFT_NO_PRIME_LANES=1 man $n $b $prec
  done
done
