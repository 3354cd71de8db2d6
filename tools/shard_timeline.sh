cd /root/repo
export TMPDIR=/tmp
rm -rf /tmp/prof_sh
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sh -o s -- python3 /root/repo/tools/run_shard_one.py 3 8 > /tmp/sh.log 2>&1)
DB=$(find /tmp/prof_sh -name "*_results.db" | head -1)
python3 tools/step_timeline.py $DB 2
