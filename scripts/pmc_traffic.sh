#!/usr/bin/env bash
# HBM traffic of the dominant kernel, per MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (no trace domains), KB units, FETCH_SIZE doubled on gfx950.
set -u
tag=${1:-traffic}; export DOM="${2:-igemm_dma_kernel<128, 160, false, false, 0>}"; out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- python scripts/step_pmc.py > "$out/$c.log" 2>&1
  rc=$?; echo "pass $c rc=$rc"; tail -2 "$out/$c.log"
  if [ $rc -ne 0 ]; then exit $rc; fi
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections, re
out = sys.argv[1]
res = {}
def kname(s):
    s = s.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", s)
    return m.group(1) if m else s[:60]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = kname(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    res[c] = {k: {"sum_kb": v[0], "dispatches": v[1], "avg_kb": v[0] / v[1]} for k, v in agg.items()}
dom = [k for k in res["FETCH_SIZE"] if os.environ["DOM"].replace(" ", "") in k.replace(" ", "")]
summary = {"counters": res}
if dom:
    k = dom[0]
    f, w = res["FETCH_SIZE"][k], res["WRITE_SIZE"].get(k, {"avg_kb": 0.0, "dispatches": 0})
    summary["dominant"] = {"kernel": k, "dispatches": f["dispatches"], "fetch_size_avg_kb": f["avg_kb"],
                           "write_size_avg_kb": w["avg_kb"],
                           "traffic_bytes_per_launch": (2.0 * f["avg_kb"] + w["avg_kb"]) * 1024.0,
                           "correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B), WRITE_SIZE as read; KB units"}
json.dump(summary, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(summary.get("dominant"), indent=1))
with open(f"{out}/traffic.txt", "w") as fh:
    fh.write(f"{'kernel':64s} {'launches':>8s} {'fetch MB/launch (x2)':>22s} {'write MB/launch':>16s}\n")
    for k, v in sorted(res["FETCH_SIZE"].items(), key=lambda kv: -kv[1]["sum_kb"])[:24]:
        w = res["WRITE_SIZE"].get(k, {"avg_kb": 0.0})
        fh.write(f"{k[:64]:64s} {v['dispatches']:8d} {v['avg_kb'] / 1024 * 2:22.2f} {w['avg_kb'] / 1024:16.2f}\n")
print(open(f"{out}/traffic.txt").read())
tot_f = sum(v["sum_kb"] for v in res["FETCH_SIZE"].values()); tot_w = sum(v["sum_kb"] for v in res["WRITE_SIZE"].values())
print(f"whole 2 steps: fetch(x2) {2*tot_f/1024/1024:.2f} GiB, write {tot_w/1024/1024:.2f} GiB")
PY
