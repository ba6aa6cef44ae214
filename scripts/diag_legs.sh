# leg_train / leg_train_b8 inside the default bench process (streams of the earlier legs alive) against the stand-alone workload
for v in "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "HIMO_TRAIN_SIDE_STREAM=0"; do
  env $v python bench.py --no-cpu-baseline --leg-fastnsf-fits 0 --leg-fit-steps 0 --no-hostfed-leg --no-extra-precisions --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'value', round(d['value'],1), {k: (round(d[k]['frames_per_s'],1), round(d[k]['ms_per_step'],2)) for k in ('leg_train','leg_train_b8','leg_train_bf16x3','leg_train_rings') if k in d and 'frames_per_s' in d[k]})
"
done
python bench.py --workload train --train-batch 8 --steps 10 --warmup 2 --no-extra-workloads 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('standalone b8', round(d['value'],1), round(d['ms_per_step'],2))"
