set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_g.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gputests_g.txt; tail -3 gpurun_out/r04_gputests_g.txt
bash tools/r04_profiles.sh > gpurun_out/r04_profiles.log 2>&1; tail -2 gpurun_out/r04_profiles.log
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || tail -5 gpurun_out/r04_bench_final.err
wc -c gpurun_out/r04_bench_final.json
