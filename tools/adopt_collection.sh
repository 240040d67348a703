#!/bin/bash
# After `gpurun -- bash tools/collect_round.sh rN`: copy the summaries of gpurun_out/rN_final/ into profiles/rN/ (logs and raw
# profiler output stay behind), keeping the hand-written header lines of the two probe files, and check that the PMC index was
# taken on the sources in the tree (bench.py drops a record taken on other sources).   usage: bash tools/adopt_collection.sh r5
set -eu
R=${1:-r5}
SRC=gpurun_out/${R}_final
DST=profiles/$R
for f in $SRC/*.json $SRC/*.csv $SRC/*.txt; do
  b=$(basename $f)
  case $b in
    placed_probe.txt) (head -1 $DST/$b; cat $f) > /tmp/adopt.$$ && mv /tmp/adopt.$$ $DST/$b ;;
    launch_head.txt) (head -2 $DST/$b; cat $f) > /tmp/adopt.$$ && mv /tmp/adopt.$$ $DST/$b ;;
    two_leg_probe_final.txt) (grep -B100 "^# the final library:" $DST/two_leg_probe.txt; cat $f) > /tmp/adopt.$$ && mv /tmp/adopt.$$ $DST/two_leg_probe.txt ;;
    *) cp $f $DST/$b ;;
  esac
done
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
idx = json.load(open("$DST/pmc_index.json"))
hashes = set(r.get("source_hash") for r in idx["records"])
print("PMC index taken on", hashes, "- the tree is", bench.source_hash())
sys.exit(0 if hashes == {bench.source_hash()} else 1)
PY
