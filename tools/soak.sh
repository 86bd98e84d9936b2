NP="--no-cpu-baseline --no-parity"
run() { echo -n "$*   "; "${ENVV[@]}" python bench.py $NP "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'triplets/s', d['ms_per_step'], 'ms/step  elbo/sentence', d.get('elbo_per_sentence'), ' seq_fallbacks', d.get('seq_fallbacks'), 'steps_skipped', d.get('steps_skipped'))"; }
ENVV=(env X=1)
run --steps 10000 --warmup 10
run --steps 4000 --warmup 10 --config script
run --steps 6000 --warmup 10 --config script --batch 40
run --steps 3000 --warmup 10 --conditional
run --steps 300 --warmup 5 --config 5
run --steps 4000 --warmup 10 --lengths ragged --config script
