#!/bin/bash
# A/B of two builds of libcreg.so inside ONE gpurun call (boxes differ by 2-3 %: numbers from different calls do not compare).
#   tools/ab_bench.sh <old-git-ref> "<bench.py arguments>" [repetitions]
# Run HERE (the build container): builds the library at <old-git-ref> in a scratch worktree and at the working tree, ships both as
# ab_old.bin / ab_new.bin, alternates them on the GPU box with identical arguments and prints value / ms_per_step / pose_checksum.
set -eu
REF=$1; ARGS=$2; REPS=${3:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
TREE=$ROOT/gpurun_out/_ab_tree
rm -rf "$TREE"; git worktree prune; git worktree add -f "$TREE" "$REF" -q
(cd "$TREE" && python autourdf_amd/build.py | tail -1)
cp "$TREE/autourdf_amd/libcreg.so" ab_old.bin
git worktree remove --force "$TREE"
python autourdf_amd/build.py | tail -1
cp autourdf_amd/libcreg.so ab_new.bin
trap 'rm -f ab_old.bin ab_new.bin' EXIT
/usr/local/graft/bin/gpurun --timeout 2400 -- "for r in \$(seq $REPS); do for v in old new; do cp ab_\$v.bin autourdf_amd/libcreg.so; echo -n \"\$v \"; timeout 600 python bench.py $ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(\"{\")][0]); print(d[\"value\"], d[\"ms_per_step\"], d.get(\"pose_checksum\"))'; done; done" 2>&1 | tail -$((2 * REPS + 1))
