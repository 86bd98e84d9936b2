#!/bin/bash
# One lease: the driver's exact bench command, then the same step loop with and without a monitoring tool sampling the card next to it
# (the driver's round-end run keeps smi.<unix time>.json samples every 5 s beside the bench: tools/lease_check.sh finds out what such a
# sampler costs a 35-ms timed region).   usage: tools/lease_check.sh <tag> [driver|smi|hog|all]
set -o pipefail
tag=${1:-lease}; what=${2:-all}
out=gpurun_out/r5_${tag}; mkdir -p $out
if [ "$what" = driver ] || [ "$what" = all ]; then
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_args.json 2> $out/driver_args.err || exit 1
fi
if [ "$what" = smi ] || [ "$what" = all ]; then
  short="--gpus 1 --steps 20 --warmup 5 --repeats 12 --no-cpu-baseline --no-parity"
  python bench.py $short > $out/quiet.json 2> $out/quiet.err || exit 1
  for tool in "rocm-smi -a --json" "amd-smi metric --json" "rocm-smi --showuse --showpower --showclocks --json"; do
    name=$(echo "$tool" | tr -c 'a-z\n' '_' | cut -c1-24)
    ( while true; do $tool > $out/smi_$name.last 2>&1; date +%s.%N >> $out/smi_$name.times; sleep 0.5; done ) &
    sp=$!
    python bench.py $short > $out/with_$name.json 2> $out/with_$name.err; rc=$?
    kill $sp; wait $sp 2>/dev/null
    [ $rc = 0 ] || exit 1
  done
fi
if [ "$what" = hog ] || [ "$what" = all ]; then
  # a busy host: 3 spinning processes per CPU this box gives us, next to the bench's one enqueueing thread
  short="--gpus 1 --steps 20 --warmup 5 --repeats 6 --no-cpu-baseline --no-parity"
  n=${HOGS:-48}; pids=""      # (nproc shows the whole host: 256; the box gives one GPU's lease a share of 16)
  for i in $(seq $n); do ( exec timeout 120 sh -c 'while :; do :; done' ) & pids="$pids $!"; done
  python bench.py $short > $out/with_host_hogs_x$n.json 2> $out/with_host_hogs.err; rc=$?
  kill $pids 2>/dev/null; wait 2>/dev/null
  [ $rc = 0 ] || exit 1
fi
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex); continue
    print("%-40s value %9.1f  ms %.3f  median step %.3f  repeats %s\n    %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["step_ms"]["median"], d["repeats_ms"], d["diagnosis"]))
    if d.get("slow_steps"): print("    slow:", d["slow_steps"])
PY
