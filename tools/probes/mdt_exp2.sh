export PFFT_XCD_CHECK=1 PFFT_XCD_DUMP=1
run() { echo "--- $*"; env "$@" timeout 180 build/multi_device_test 2>&1 | grep -E "gave up|OK|FAILED|thread 0|terminate|xcd ctl|queue" | head -14; }
run MDT_THREADS=8 MDT_XCD_BATCH=256
run MDT_THREADS=8 MDT_XCD_BATCH=256
run MDT_THREADS=6 MDT_XCD_BATCH=256
run MDT_THREADS=5 MDT_XCD_BATCH=256
run MDT_THREADS=8 MDT_XCD_BATCH=256 GPU_MAX_HW_QUEUES=8
