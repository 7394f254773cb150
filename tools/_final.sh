set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_e.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gputests_e.txt; tail -3 gpurun_out/r04_gputests_e.txt
bash tools/r04_profiles.sh > gpurun_out/r04_profiles.log 2>&1; tail -3 gpurun_out/r04_profiles.log
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || tail -5 gpurun_out/r04_bench_final.err
wc -c gpurun_out/r04_bench_final.json
python bench.py --config 3 --steps 20 > gpurun_out/r04_bench_config3_1gpu.json 2> gpurun_out/r04_bench_config3_1gpu.err
