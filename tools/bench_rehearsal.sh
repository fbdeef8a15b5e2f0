set -e
B="timeout -k 10 300 python bench.py --steps 3 --warmup 1 --spp-per-step 16 --no-cpu-baseline --no-parity"
$B --gpus 1 > gpurun_out/r5_g1.json 2> gpurun_out/r5_g1.err
$B --gpus 2 --backend gloo > gpurun_out/r5_g2s.json 2> gpurun_out/r5_g2s.err
$B --gpus 2 --backend gloo --sharding tiles > gpurun_out/r5_g2t.json 2> gpurun_out/r5_g2t.err
$B --gpus 2 --backend gloo --split-each-step > gpurun_out/r5_g2e.json 2> gpurun_out/r5_g2e.err
$B --gpus 4 --backend gloo > gpurun_out/r5_g4s.json 2> gpurun_out/r5_g4s.err
$B --gpus 2 --backend gloo --scaling weak > gpurun_out/r5_g2w.json 2> gpurun_out/r5_g2w.err
for f in g1 g2s g2t g2e g4s g2w; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r5_$f.json').read().strip().splitlines()[-1]); c=d['config']
print('$f', d['n_gpus'], round(d['value'],1), d['scaling'], c['sharding'], c['calls_per_rank'], c['spp_per_call'], c['batch_samples'], d['frame_checksum'], c['job_split'])"; done
