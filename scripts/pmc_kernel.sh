#!/usr/bin/env bash
# SQ / GRBM counters of ONE kernel (name substring) over separate rocprofv3 --pmc passes (8 SQ slots per pass; no trace
# domains in the same run).  usage: scripts/pmc_kernel.sh <tag> <kernel substring> <python script> [args...]
# The profiled program is python itself, directly after `--`.
set -u
tag=$1; export KSUB=$2; shift 2
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python "$@" > "$out/p$i.log" 2>&1
  rc=$?; echo "pass $i rc=$rc [$set]" | tee -a "$out/summary.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  f=$(find "$out/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a "$out/summary.txt"
import csv, sys, os, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in rows:
    if os.environ["KSUB"].replace(" ","") not in r["Kernel_Name"].replace(" ",""): continue
    agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for c,val in agg.items():
    print(f"   {c} = {val/cnt[c]:.5g} (per dispatch, {cnt[c]} dispatches)")
PY
  find "$out/p$i" -name "*.csv" -size +5M -delete 2>/dev/null
done
