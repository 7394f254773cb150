set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_a.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gputests_a.txt; tail -5 gpurun_out/r04_gputests_a.txt
python tools/populate_probe.py > gpurun_out/r04_populate_probe.txt 2>&1; cat gpurun_out/r04_populate_probe.txt
python tools/pairwise_two_pass_ab.py > gpurun_out/r04_pairwise_two_pass.txt 2>&1; cat gpurun_out/r04_pairwise_two_pass.txt
