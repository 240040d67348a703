cd "${GRAFT_REPO_ROOT:-.}"
P=tools/experiments/packed_record_probe.py
for cfg in "--gait static --batch 4096 --steps 200" "--gait trot --batch 8192 --steps 200" "--gait trot --batch 65536 --steps 64"; do
  for rep in 1 2; do
    python $P quadruped_locomotion_amd/libqlamd.so 0 $cfg 2>/dev/null
    python $P variants/libqlamd_rec48.so 48 $cfg 2>/dev/null
    python $P variants/libqlamd_rec40.so 40 $cfg 2>/dev/null
  done
done
# bytes per launch: rocprofv3 counter passes, eager launches
cd /tmp; export TMPDIR=/tmp
for what in "quadruped_locomotion_amd/libqlamd.so 0" "variants/libqlamd_rec48.so 48" "variants/libqlamd_rec40.so 40"; do
  set -- $what
  for cfg in "--gait static --batch 4096" "--gait trot --batch 65536"; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pr; rocprofv3 --kernel-trace --pmc $c -d /tmp/pr -o p -- python3 $GRAFT_REPO_ROOT/$P $GRAFT_REPO_ROOT/$1 $2 $cfg --steps 30 --no-graph > /dev/null 2>&1
      python3 - "$1" "$2" "$cfg" $c <<'PY'
import glob, sqlite3, sys
db = glob.glob("/tmp/pr/**/*_results.db", recursive=True)
if not db:
    print(sys.argv[1:], "no db"); sys.exit()
con = sqlite3.connect(db[0])
rows = con.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like '%balance_coop_kernel%' group by dispatch_id order by dispatch_id").fetchall()
v = [r[1] for r in rows][10:]
print("%s record %s %s %s: %.3f MB per launch" % (sys.argv[1].split("/")[-1], sys.argv[2], sys.argv[3], sys.argv[4], sum(v) / len(v) * 1024 / 1e6))
PY
    done
  done
done
