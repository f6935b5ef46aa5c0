#!/bin/bash
# usage: tools/prof_pmc.sh <tag> [bench args...]   (run on the GPU box from the repo root)
# One rocprofv3 run per counter group (rocprofv3 does not split counters over passes).
set -u
TAG=$1; shift
ARGS="$@"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  [ "$i" -ge "${PMC_PASSES:-99}" ] && break   # PMC_PASSES=1: only the VALU / wave-cycle group
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done <<'GRP'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
TA_BUSY_avr TA_TA_BUSY_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
GRP
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write(f"   {c:32s} {v:.6g}\n")
print(open(out + "/summary.txt").read())
PY
