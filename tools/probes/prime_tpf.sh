# prime-factor lengths (radix 37 ... 61 butterflies): lanes per transform and radix order, fp32, 1 GiB
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel'][-70:])"; }
man() { python bench.py --manual d=cpx,n=$1,b=$2 --precision ${3:-float} --no-cpu-baseline --steps 30 2>/dev/null | pr; }
for spec in "976 137518 61x16 16x61" "2021 66411 47x43 43x47" "2368 56680 37x8x8 8x8x37 37x64" "3481 38557 59x59" "1696 79137 53x32 32x53 53x8x4"; do
  set -- $spec; n=$1; b=$2; shift 2
  echo "n=$n"; echo -n "  default: "; man $n $b
  for rad in "$@"; do
    for tpf in 8 16 32 64 128; do
      echo -n "  $rad tpf $tpf: "; PFFT_JIT_SPEC_RADICES=$n:$rad PFFT_JIT_FORCE_TPF=$tpf man $n $b
    done
  done
done
