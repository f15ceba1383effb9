# A/B of the control-op placement: in front of the pass-0 loads (o4_fr0) / behind them (ci) / one task ahead in duties() (p3)
for rep in 1 2; do
for c in 18 16; do for pt in "12 8" "8 5"; do set -- $pt; for b in ci p3; do echo -n "$b case $c: "; TUNE_SLOTS=$1 TUNE_LAG=$2 TUNE_REPS=7 timeout 40 ./build/tune_xcd_${b}_$c | grep "XCD-local" | cut -c1-125; done; done; done
done
TUNE_SLOTS=16 TUNE_LAG=8 TUNE_REPS=7 timeout 40 ./build/tune_xcd_p3_116 | grep "XCD-local\|two l" | cut -c1-125
TUNE_SLOTS=12 TUNE_LAG=8 TUNE_REPS=7 timeout 40 ./build/tune_xcd_p3_116 | grep "XCD-local\|two l" | cut -c1-125
