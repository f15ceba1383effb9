# would the lanes rule pay for the primes 23 ... 31 too?  PFFT_PRIME_LANES_MIN=23 against the default (37)
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel'][20:95])"; }
man() { python bench.py --manual d=cpx,n=$1,b=$2 --precision $3 --no-cpu-baseline --steps 30 2>/dev/null | pr; }
for prec in float double; do
  es=8; [ $prec = double ] && es=16
  for n in 368 736 1104 928 1856 841 496 1984 1488 529 667 899 992 961 464 1472 608 1216 912; do
    b=$(( (1<<29) / (n*es) ))
    echo "$prec n=$n"
    echo -n "   rule from 19: "; PFFT_PRIME_LANES_MIN=19 man $n $b $prec
    echo -n "   default:      "; man $n $b $prec
  done
done
