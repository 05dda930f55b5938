mkdir -p gpurun_out/r02l
for W in MPI_Multi_SurS1 HM36_Multi_SurS2; do
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --workload $W > gpurun_out/r02l/b_$W.json 2> gpurun_out/r02l/b_$W.err; tail -1 gpurun_out/r02l/b_$W.err
done
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --workload HM36_Multi_SynthS2 --batch 64 > gpurun_out/r02l/b_synth64.json 2> gpurun_out/r02l/b_synth64.err; tail -2 gpurun_out/r02l/b_synth64.err
python - <<'PY'
import json
for f in ('b_MPI_Multi_SurS1','b_HM36_Multi_SurS2','b_synth64'):
    try:
        d=json.loads(open('gpurun_out/r02l/%s.json'%f).read()); print(f, round(d['ms_per_step'],1),'ms', round(d['value'],1),'img/s', round(d['roofline']['achieved'],1))
    except Exception as e: print(f,'ERR',e)
PY
