#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/pmc_icache
mkdir -p $OUT
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python3 bench.py --steps 1 --warmup 0 --spp 100 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(set(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True))):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg["render"][r["Counter_Name"]] += float(r["Counter_Value"])
for c, v in sorted(agg["render"].items()):
    print(f"   {c:32s} {v:.6g}")
PY
