set -e
mkdir -p gpurun_out/final
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/final/tests.log 2>&1
bash tools/profile_round.sh gpurun_out/prof_1m r03 synth1m > gpurun_out/prof_1m.log 2>&1
cp gpurun_out/prof_1m/counters.json profiles/counters.json
cp profiles/counters.json gpurun_out/final/counters_a.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_synth1m.json 2> gpurun_out/final/bench_synth1m.err
timeout -k 10 200 python tools/workspace_size.py > gpurun_out/final/workspace.txt 2>&1
echo done
