#!/bin/bash
# The one parameterised gpurun script (round 5; the 59 one-off scripts of rounds 1-4 are archived under tools/history/ next to the profiles
# they produced).  Usage on the GPU box:   bash tools/gpu.sh <out-tag> <step> [<step> ...]      - every step writes under gpurun_out/<out-tag>/
#   tests[:<pytest -k expression>]      python -m pytest tests -m gpu -x (whole suite, or the selection)
#   testsall:<pytest -k expression>     the selection without -x and with -s (printed diagnostics kept)
#   smoke                               __graft_entry__.smoke()
#   bench[:<bench.py flags>]            bench.py (default: --sweep off --no-cpu-baseline --steps 10 --warmup 3); prints value / fractions
#   benchfull                           the driver's line: python bench.py (full sweep, scores, CPU baseline)
#   ab:<lib-a>,<lib-b>[,...]            bench (sweep off) alternating over VISREP_LIB variant libraries (`default` = the product library), 2 rounds
#   trace:<python file + args>          rocprofv3 --kernel-trace --stats of the command, summary via tools/summarize_pmc.py
#   pmc:<counters,comma>:<python file>  one rocprofv3 --pmc pass (counters in their own run: never combined with a trace domain)
#   py:<python file + args>             plain run, output to <step index>.log
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=law_of_vision_representation_in_mllms_amd
short() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    k = {n.split()[0]: (v.get("ms_in_layer_mix"), v.get("ms")) for n, v in (r.get("kernels") or {}).items()}
    print("  value", d["value"], "ms/step", d["ms_per_step"], "fc1 frac", r.get("frac"), "b2b", r.get("frac_back_to_back"), "practical", (r.get("practical_roof") or {}).get("tflops"), k)
    s = d.get("sweep")
    if s and "wall_s" in s:
        print("  sweep wall_s", s["wall_s"], "all_bf16", s.get("wall_s_all_bf16"), "setup", s.get("setup_s"), {n: (v.get("c_s"), v.get("a_s")) for n, v in s["per_setting"].items()})
except Exception as e:
    print("  (no bench line:", e, ")")
PY
}
i=0
for step in "$@"; do
  i=$((i+1)); kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -k "$arg" > $O/$i.pytest.log 2>&1; else timeout 1800 python -m pytest tests -m gpu -q -x --tb=short > $O/$i.pytest.log 2>&1; fi
           echo "[$i tests] rc=$?"; tail -4 $O/$i.pytest.log | cut -c1-300 ;;
    testsall) timeout 1800 python -m pytest tests -m gpu -q --tb=line -s -k "$arg" > $O/$i.pytest.log 2>&1      # no -x, prints kept: a first look at a new test file
           echo "[$i testsall] rc=$?"; grep -E "^hostile|passed|failed" $O/$i.pytest.log | cut -c1-420 | tail -80 ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/$i.smoke.log 2>&1; echo "[$i smoke] rc=$?"; tail -1 $O/$i.smoke.log ;;
    bench) timeout 900 python bench.py ${arg:---sweep off --no-cpu-baseline --steps 10 --warmup 3} > $O/$i.bench.json 2> $O/$i.bench.err; echo "[$i bench] rc=$?"; short $O/$i.bench.json ;;
    benchfull) timeout 2400 python bench.py > $O/$i.bench.json 2> $O/$i.bench.err; echo "[$i benchfull] rc=$?"; short $O/$i.bench.json ;;
    ab) for r in 1 2; do for v in ${arg//,/ }; do
          if [ $v = default ]; then unset VISREP_LIB; else export VISREP_LIB=$PWD/$P/libvisrep_hip_$v.so; fi
          timeout 300 python bench.py --sweep off --no-cpu-baseline --no-scores --steps 10 --warmup 3 > $O/$i.ab.$v.$r.json 2> $O/$i.ab.$v.$r.err; echo "[$i ab $v $r] rc=$?"; short $O/$i.ab.$v.$r.json
        done; done; unset VISREP_LIB ;;
    trace) rm -rf $O/$i.trace; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$i.trace -- python $arg > $O/$i.trace.log 2>&1; echo "[$i trace] rc=$?"
           python tools/summarize_pmc.py $O/$i.trace > $O/$i.trace.md 2>/dev/null; head -30 $O/$i.trace.md ;;
    pmc) ctr=${arg%%:*}; cmd=${arg#*:}; rm -rf $O/$i.pmc; timeout 400 rocprofv3 --pmc ${ctr//,/ } --output-format csv -d $O/$i.pmc -- python $cmd > $O/$i.pmc.log 2>&1; echo "[$i pmc $ctr] rc=$?"
         python tools/summarize_pmc.py $O/$i.pmc > $O/$i.pmc.md 2>/dev/null; head -20 $O/$i.pmc.md ;;
    py) timeout 1200 python $arg > $O/$i.log 2>&1; echo "[$i py $arg] rc=$?"; tail -25 $O/$i.log | cut -c1-400 ;;
    *) echo "unknown step $step" ;;
  esac
done
