# medium primes (17 ... 31): lanes per transform and radix order, fp32, 0.5 GiB
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel'][20:95])"; }
man() { python bench.py --manual d=cpx,n=$1,b=$2 --precision ${3:-float} --no-cpu-baseline --steps 30 2>/dev/null | pr; }
for spec in "992 31x32 32x31" "464 29x16 16x29" "1472 23x8x8 8x8x23 23x64 64x23" "1088 17x8x8 8x8x17 17x64" "961 31x31" "2976 31x12x8 12x8x31" "5704 31x23x8 8x23x31"; do
  set -- $spec; n=$1; b=$(( (1<<26) / n )); shift
  echo "n=$n"; echo -n "  default: "; man $n $b
  for rad in "$@"; do
    for tpf in 16 32 64 128; do
      echo -n "  $rad tpf $tpf: "; PFFT_JIT_SPEC_RADICES=$n:$rad PFFT_JIT_FORCE_TPF=$tpf man $n $b
    done
  done
done
