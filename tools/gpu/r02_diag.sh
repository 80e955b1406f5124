mkdir -p gpurun_out/r02
for i in 1 2 3 4; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > gpurun_out/r02/driver_cmd_$i.json
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r02/driver_cmd_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms_median'], d['step_ms_max'], d['host_enqueue_ms'], d['cpu_baseline']['value'])"
done
