set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_g.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gputests_g.txt; tail -3 gpurun_out/r04_gputests_g.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.txt 2>&1; tail -1 gpurun_out/r04_smoke.txt
bash tools/r04_profiles.sh > gpurun_out/r04_profiles.log 2>&1; tail -2 gpurun_out/r04_profiles.log
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || tail -5 gpurun_out/r04_bench_final.err
wc -c gpurun_out/r04_bench_final.json
