#!/bin/bash
# Round-end evidence run (on the GPU box): bench line, rocprofv3 kernel stats of the same command, HBM traffic counters.
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 bench.py --steps 2 --warmup 1 --no-extras > $OUT/stats_bench.json 2> $OUT/stats.err || { echo stats failed; tail -5 $OUT/stats.err; exit 1; }
cat $OUT/stats_bench.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-extras > $OUT/fetch_bench.json 2> $OUT/fetch.err || { echo fetch failed; tail -5 $OUT/fetch.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 bench.py --steps 1 --warmup 0 --no-extras > $OUT/write_bench.json 2> $OUT/write.err || { echo write failed; tail -5 $OUT/write.err; exit 1; }
find $OUT -name "*kernel_stats.csv" -exec cat {} \;
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(set(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True))):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k, dict(d))
PY
