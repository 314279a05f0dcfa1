mkdir -p gpurun_out
set -x
for w in n4096 n1024; do
 python bench.py --workload $w --graph 0 --no-cpu-baseline > gpurun_out/e_$w.json 2>gpurun_out/e_$w.err
 EKF_DEBUG_SKIP_FLUSH=1 python bench.py --workload $w --graph 0 --no-cpu-baseline --no-flush-profile > gpurun_out/e_${w}_nf.json 2>gpurun_out/e_${w}_nf.err
done
for f in n4096 n4096_nf n1024 n1024_nf; do python3 -c "
import json
try:
  d=json.load(open('gpurun_out/e_$f.json')); print('$f', round(d['value'],1), 'steps/s dev ms/step', round(d['device_ms_per_step'],4), d['roofline']['avg_launch_us'])
except Exception as e: print('$f', 'ERR', open('gpurun_out/e_$f.err').read()[-300:])
"; done
