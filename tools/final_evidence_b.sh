set -e
mkdir -p gpurun_out/final
bash tools/profile_round.sh gpurun_out/prof_10m r03 synth10m --spp-per-step 8 --steps 2 > gpurun_out/prof_10m.log 2>&1
cp gpurun_out/prof_10m/counters.json profiles/counters.json
cp profiles/counters.json gpurun_out/final/counters.json
timeout -k 10 400 python bench.py --workload synth10m --spp-per-step 8 --steps 2 --no-cpu-baseline > gpurun_out/final/bench_synth10m.json 2> gpurun_out/final/bench_synth10m.err
: > gpurun_out/final/other.jsonl
for w in spheres cornell hitchcock cornell_area hitchcock_mesh; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline >> gpurun_out/final/other.jsonl 2>> gpurun_out/final/other.err
done
timeout -k 10 300 python bench.py --workload synth1m_glossy_terrain --max-depth 8 --no-cpu-baseline >> gpurun_out/final/other.jsonl 2>> gpurun_out/final/other.err
echo done
