mkdir -p gpurun_out/r5_gang
for g in 2 1; do timeout 300 ./build/tune_gang${g}_16 2>&1 | grep -E "^N =|two launches|gang|XCD-local|!!" ; done | tee gpurun_out/r5_gang/gang16.txt
