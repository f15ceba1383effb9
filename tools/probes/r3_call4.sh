#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3_c4; mkdir -p $O
tools/survey.sh $O/survey > $O/survey.log 2>&1
tools/prof_1d.sh $O/prof f32:1048576:128 f32:262144:512 f32:4194304:32 f32:65536:2048 > $O/prof_default.txt 2>&1
PFFT_CACHE_CHUNK_MIB=0 tools/prof_1d.sh $O/prof_nochunk f32:1048576:128 > $O/prof_nochunk.txt 2>&1
cat $O/prof_default.txt $O/prof_nochunk.txt
