#!/bin/bash
# BATCH_INTERLEAVED N=1024 / 256 at different batch sizes (the row pitch is batch * 8 B): tools/probes/bi_batch.sh
for n in 1024 256; do
for b in 512 2048 8192 32768 131072; do
  reps=$(( 4194304 / b )); [ $reps -gt 200 ] && reps=200; [ $reps -lt 10 ] && reps=10
  python3 tools/probes/one_desc.py float "domain=complex,lengths=$n,batch=$b,fwd_strides=$b,bwd_strides=$b,fwd_dist=1,bwd_dist=1" $reps 2>&1 | grep " ms" | awk -v n=$n -v b=$b '{ms=$NF=="ms"?$(NF-1):$NF; printf "BI N=%d batch=%d: %s ms  %.2f TB/s\n", n, b, $(NF-1), 2*n*b*8/($(NF-1)*1e-3)/1e12}'
done; done
