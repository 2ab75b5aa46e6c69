#!/usr/bin/env bash
set -u
tag=${1:-pmcf}; out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python scripts/flash_pmc.py 4096 8 40 > "$out/p$i.log" 2>&1
  rc=$?; echo "pass $i rc=$rc [$set]"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  f=$(find "$out/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in rows:
    if "flash" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for c,val in agg.items():
    print(f"   {c} = {val/cnt[c]:.4g} (per dispatch, {cnt[c]} dispatches)")
PY
done
