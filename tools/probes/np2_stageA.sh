# stage A of 500 points (fp32 N = 40000 = 500 x 80): forced configurations n:fpw:lanes_per_fft:radices[:twl]
pr() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel_ms'], r['kernel'][-48:])"; }
man() { python bench.py --manual d=cpx,n=$1,b=$2 --precision float --no-cpu-baseline --steps 40 2>/dev/null | pr; }
echo -n "default: "; man 40000 3355
for f in 500:32:26:10x10x5 500:32:25:20x25 500:32:20:25x20 500:32:25:25x20 500:16:50:10x10x5 500:16:25:20x25 500:16:26:10x10x5 500:32:32:10x10x5 500:32:16:25x20 500:16:32:20x25 500:32:26:5x10x10 500:32:20:5x10x10 500:32:26:10x10x5:1 500:32:26:10x10x5:2; do
  echo -n "$f: "; PFFT_JIT_STRIDED_FORCE=$f man 40000 3355
done
