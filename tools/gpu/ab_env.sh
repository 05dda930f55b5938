# A/B of an environment switch on ONE box: bash tools/gpu/ab_env.sh VAR  (runs tools/ab_step.py with VAR=0 then VAR=1, twice)
V=$1
for r in 1 2; do
  env $V=0 python tools/ab_step.py 0 | head -1 | sed "s/^/$V=0  /"
  env $V=1 python tools/ab_step.py 0 | head -1 | sed "s/^/$V=1  /"
done
