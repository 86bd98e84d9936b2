# same-box A/B of environment switches: bash tools/ab_env.sh "<bench args>" NAME=VAL[,NAME=VAL...] ...   (first arm: no switches; schedule attributes of the
# engine go through the one switch left for them: VMMT_ENGINE_ATTRS=bg_adam_blocks=224 -- a comma inside it separates ARMS here, so one attribute per arm)
ARGS="$1"; shift
run() { env "$@" python bench.py --no-cpu-baseline --no-parity $ARGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
  line="base $(run X=1)"
  for arm in "$@"; do line="$line | $arm $(run $(echo $arm | tr ',' ' '))"; done
  echo "$line"
done
