for sl in "2 1" "3 1" "3 2" "4 2"; do set -- $sl; TUNE_SLOTS=$1 TUNE_LAG=$2 timeout 120 build/tune_xcd_g_120_0 2>&1 | grep -E "two launches|XCD-local" | tail -2; done
echo "--- fp64 2^19"; for sl in "2 1" "3 1" "3 2" "4 2"; do set -- $sl; TUNE_SLOTS=$1 TUNE_LAG=$2 timeout 120 build/tune_xcd_g_119_0 2>&1 | grep -E "XCD-local" | tail -1; done
echo "--- fp32 2^20"; for sl in "2 1" "3 1" "3 2" "4 2"; do set -- $sl; TUNE_SLOTS=$1 TUNE_LAG=$2 timeout 120 build/tune_xcd_g_20_0 2>&1 | grep -E "XCD-local" | tail -1; done
echo "--- split rule"; python bench.py --config ref68640 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(r['frac'], r['kernel'][-60:])"
