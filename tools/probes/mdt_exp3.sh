export PFFT_XCD_CHECK=1 PFFT_XCD_DUMP=1
run() { echo "--- $*"; env "$@" timeout 180 build/multi_device_test 2>&1 | grep -E "gave up|OK|FAILED|thread 0|terminate|xcd ctl|queue" | head -14; }
for i in 1 2 3 4 5; do run MDT_THREADS=8 MDT_XCD_BATCH=256; done
run MDT_THREADS=12 MDT_XCD_BATCH=256
run MDT_THREADS=16 MDT_XCD_BATCH=128 PFFT_XCD_MIN_BATCH=0
for i in 1 2 3; do d=$(mktemp -d); run MDT_THREADS=4 PFFT_JIT_CACHE_DIR=$d; done
