# Copies what tools/final_evidence.sh left under gpurun_out/ into profiles/ (run on the build machine after both halves).
set -e
cp gpurun_out/final/counters.json profiles/counters.json
cp gpurun_out/prof_1m/kernel_stats_r03_synth1m.csv profiles/r03_kernel_stats_synth1m.csv
cp gpurun_out/prof_10m/kernel_stats_r03_synth10m.csv profiles/r03_kernel_stats_synth10m.csv
cp gpurun_out/prof_1m/stats_bench.json profiles/r03_kernel_stats_synth1m_bench.json
cp gpurun_out/prof_10m/stats_bench.json profiles/r03_kernel_stats_synth10m_bench.json
cp gpurun_out/final/bench_synth1m.json profiles/r03_bench_synth1m.json
cp gpurun_out/final/bench_synth10m.json profiles/r03_bench_synth10m.json
cp gpurun_out/final/other.jsonl profiles/r03_bench_other_workloads.jsonl
