#!/usr/bin/env bash
# PMC passes over the dominant conv shape (64x64x320->320, K=2880.. and 5760): one counter set per pass.
set -u
tag=${1:-pmc}; out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || exit 1
shape="64 640 320 9 1 0 5"     # M=16384, K=5760, N=320, DMA kernel
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_WAVE32_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python scripts/igemm_pmc.py $shape > "$out/p$i.log" 2>&1
  rc=$?; echo "pass $i rc=$rc [$set]"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  f=$(find "$out/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"]
    if "igemm" not in k and "halo" not in k: continue
    agg[k.split("(")[0][-40:]][r["Counter_Name"]]+=float(r["Counter_Value"])
    cnt[(k.split("(")[0][-40:], r["Counter_Name"])]+=1
for k,v in agg.items():
    for c,val in v.items():
        print(f"   {k} {c} = {val/cnt[(k,c)]:.4g} (per dispatch, {cnt[(k,c)]} dispatches)")
PY
done
