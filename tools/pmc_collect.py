#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into one JSON (per-launch means for one kernel).

Run on the GPU box, one counter group per pass (gpurun refuses --pmc together with tracing other than
--kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"):

  cd /tmp && export TMPDIR=/tmp
  for g in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
    rocprofv3 --kernel-trace --pmc $g -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o "pass_${g%% *}" -- \\
        python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 25 --warmup 5 --no-cpu-baseline
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_collect.py $GRAFT_REPO_ROOT/gpurun_out/pmc balance_coop_kernel out.json
"""
import glob
import json
import os
import sqlite3
import sys


def per_launch(dbfile, kernel):
    db = sqlite3.connect(dbfile)
    # counters_collection: one row per (dispatch, counter, dimension instance); sum the instances of a dispatch
    acc = {}
    for cname, _, v in db.execute("select counter_name, dispatch_id, sum(value) from counters_collection "
                                  "where kernel_name like ? group by counter_name, dispatch_id order by dispatch_id",
                                  ("%" + kernel + "%",)):
        acc.setdefault(cname, []).append(v)
    out = {}
    for cname, vals in acc.items():
        vals = vals[5:] if len(vals) > 10 else vals          # drop the warm-up launches
        out[cname] = {"mean": sum(vals) / len(vals), "launches": len(vals)}
    return out


def main():
    d, kernel, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    res = {}
    for f in sorted(glob.glob(os.path.join(d, "*_results.db"))):
        res.update(per_launch(f, kernel))
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
