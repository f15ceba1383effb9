# alternating runs of one layout under several environment settings (3 rounds): tools/probes/alt_env.sh "<one_layout args>" "ENV=a" "ENV=b" ...
spec=$1; shift
declare -A acc
for i in 1 2 3; do
  for e in "A=1" "$@"; do
    v=$(env $e python3 tools/probes/one_layout.py $spec 2>&1 | grep TB/s | tail -1 | awk '{for (i=1;i<=NF;i++) if ($i=="ms") print $(i-1)}')
    acc["$e"]="${acc["$e"]} $v"
  done
done
for e in "A=1" "$@"; do echo "$spec | $e |${acc["$e"]}"; done
