#!/bin/bash
# tools/shard2_trace.sh [WORLD] — WORLD (default 2) ranks of tools/shard2_dry.py on one GPU, rank 0 under rocprofv3 --kernel-trace
# --stats; prints rank 0's kernel statistics (calls per kernel: the launches of a sharded iteration) and leaves them in
# gpurun_out/shard_trace/.
W=${1:-2}
PORT=$((20000 + RANDOM % 20000))
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/shard_trace
rm -rf "$OUT"; mkdir -p "$OUT"
for r in $(seq 1 $((W - 1))); do
  python3 tools/shard2_dry.py $r $W $PORT > "$OUT/rank$r.log" 2>&1 &
done
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o r0 --output-format csv -- python3 tools/shard2_dry.py 0 $W $PORT > "$OUT/rank0.log" 2>&1
rc=$?
wait
cat "$OUT"/rank*.log | grep -E "rank [0-9]|same basis"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/rank0_kernel_stats.csv" && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>7s}  avg {float(r["AverageNs"]) / 1000:8.2f} us  {r["Percentage"]:>6s} %')
PY
rm -rf "$OUT/prof"
exit $rc
