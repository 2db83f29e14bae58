#!/bin/bash
# tools/rocprof_bench.sh TAG [bench args...] — run bench.py under rocprofv3 --kernel-trace --stats
# on the GPU box and leave <TAG>_kernel_stats.csv + <TAG>_bench.json under gpurun_out/.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT/prof_$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAG" -o "$TAG" --output-format csv -- python3 "$ROOT/bench.py" "$@" > "$OUT/${TAG}_bench.log" 2>&1
grep "^{\"metric\"" "$OUT/${TAG}_bench.log" > "$OUT/${TAG}_bench.json"
find "$OUT/prof_$TAG" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from ellp_amd.build import engine_source_hash as h; import json; print(json.dumps({'engine_source_hash': h(), 'command': 'bench.py ' + ' '.join(sys.argv[1:])}))" "$@" > "$OUT/${TAG}_kernel_stats.meta.json"
rm -rf "$OUT/prof_$TAG"  # the trace itself is tens of MB, and gpurun_out/ is merged back only up to 64 MiB
head -12 "$OUT/${TAG}_kernel_stats.csv"
