# how often the 76 -> 8 shape of tools/conv_shapes_ab.py (24th of 30 in the process) takes ~2 ms instead of ~0.16: one process per sample
for v in "$@"; do
  line=""
  for i in 1 2 3 4 5 6; do
    t=$(env $v python tools/conv_shapes_ab.py x 2>/dev/null | grep -E "76->  8" | awk '{print $(NF-5)}')
    line="$line $t"
  done
  echo "$v:$line"
done
