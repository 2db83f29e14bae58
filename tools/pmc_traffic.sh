#!/bin/bash
# tools/pmc_traffic.sh TAG [bench args...] — HBM traffic per kernel launch from two SEPARATE
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench.py run; writes
# gpurun_out/<TAG>_pmc_traffic.json (copy it to profiles/).  MI355X_MICROARCH.md §HBM:
# counter unit KiB... FETCH_SIZE reads half of a wide coalesced read stream on gfx950 (x2).
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o pmc --output-format csv -- python3 "$ROOT/bench.py" --steps 400 --warmup 50 --profile-steps 0 --cpu-pivots 0 --config5 0 --config4 0 --full-solve 0 "$@" > "$OUT/${TAG}_pmc_$C.log" 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" "$TAG" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "$@" > "$OUT/${TAG}_pmc_traffic.json"
head -c 1500 "$OUT/${TAG}_pmc_traffic.json"
