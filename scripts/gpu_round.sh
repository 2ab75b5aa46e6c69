#!/usr/bin/env bash
# One gpurun call = one box acquisition (minutes of budget): run the whole GPU checklist in it.
# A step that TIMES OUT or is KILLED ends the call (no further GPU work on a possibly wedged card);
# ordinary test failures are logged and the later steps still run.
# usage: scripts/gpu_round.sh <tag> [steps...]   steps: full kernels parity smoke stepprof stepprofvae breakdown bench prof pmc
set -u
tag=${1:-r01}; shift || true
steps=${*:-"kernels parity smoke bench prof"}
out=gpurun_out/$tag; mkdir -p "$out"
run() {  # run <name> <timeout_s> <cmd...>
  local name=$1 to=$2; shift 2
  echo "=== [$name] $(date +%T) $*" | tee -a "$out/summary.txt"
  timeout -k 10 "$to" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "=== [$name] rc=$rc" | tee -a "$out/summary.txt"
  tail -n "${TAILN:-15}" "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out/killed: stopping" | tee -a "$out/summary.txt"; exit $rc; fi
  return 0
}
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || { tail -20 "$out/build.log"; exit 1; }
for s in $steps; do
  case $s in
    full)    # the driver's exact round-end commands, in its order: run this ONCE as the last GPU action after the last code change
             run full 1100 python -m pytest tests/ -x -q -m gpu
             run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    kernels) run kernels 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 ;;
    parity)  run parity 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s --timeout=600 ;;
    smoke)   run smoke 300 python __graft_entry__.py smoke ;;
    stepprof) TAILN=60 run stepprof 600 python scripts/step_profile.py --list --out "$out/step_profile.txt" ;;
    stepprofab) # same-box A/B of one UNet step: SET_A / SET_B = engine policy overrides (NAME=VALUE), e.g. SET_A=HALO_DUO=False
                TAILN=5 run stepprof_a 600 python scripts/step_profile.py --list --out "$out/step_profile_a.txt" ${SET_A:+--set $SET_A}
                TAILN=5 run stepprof_b 600 python scripts/step_profile.py --list --out "$out/step_profile_b.txt" ${SET_B:+--set $SET_B} ;;
    benchab) # same-box A/B of the whole pass: BENCH_A / BENCH_B = extra bench.py arguments (e.g. "--set LN_STATS_FROM_PRODUCER=False")
             TAILN=2 run bench_a 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_A:-}
             TAILN=2 run bench_b 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_B:-}
             TAILN=2 run bench_a2 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_A:-}
             TAILN=2 run bench_b2 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_B:-}
             grep -h -o '"ms_per_step": [0-9.]*' "$out"/bench_a.log "$out"/bench_b.log "$out"/bench_a2.log "$out"/bench_b2.log ;;
    stepprofvae) TAILN=40 run stepprofvae 600 python scripts/step_profile.py --vae --list --out "$out/step_profile_vae.txt" ;;
    opbench) TAILN=80 run opbench 600 python scripts/op_bench.py "$tag" ;;
    opbenchvae) TAILN=60 run opbenchvae 600 python scripts/op_bench.py "$tag" --vae ;;
    breakdown) TAILN=20 run breakdown 600 python scripts/pass_breakdown.py ;;
    bench)   run bench 900 python bench.py --steps 3 --warmup 1 ;;
    prof)    export TMPDIR=/tmp
             run prof 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
             find "$out/prof" -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} sh -c 'head -40 "{}" > '"$out"'/kernel_stats_top.csv'
             python scripts/trace_gaps.py "$out/prof" > "$out/trace_gaps.txt" 2>&1; tail -n 45 "$out/trace_gaps.txt"
             find "$out/prof" -name "*kernel_trace*.csv" -size +20M -delete 2>/dev/null ;;
    pmc)     export TMPDIR=/tmp
             run pmc 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ;;
  esac
done
echo "=== done $(date +%T)" | tee -a "$out/summary.txt"
