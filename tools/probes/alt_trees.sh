# alternating runs of one layout under several source trees (3 rounds): tools/probes/alt_trees.sh "<one_layout args>" tree1 tree2 ...
spec=$1; shift
declare -A acc
for i in 1 2 3; do
  for d in "$@"; do
    v=$(cd $d; f=tools/probes/one_layout.py; [ -f $f ] || f=tools/one_layout.py; python3 $f $spec 2>&1 | grep TB/s | tail -1 | awk '{for (i=1;i<=NF;i++) if ($i=="ms") print $(i-1)}')
    acc["$d"]="${acc["$d"]} $v"
  done
done
for d in "$@"; do echo "$spec | $d |${acc["$d"]}"; done
