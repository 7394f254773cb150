set -o pipefail
for t in 0 4 2 8; do
  GRAPHPOPE_PREFAULT_THREADS=$t python bench.py --no-sage --steps 20 > gpurun_out/r04_bench_prefault_$t.json 2> gpurun_out/r04_bench_prefault_$t.err || { tail -5 gpurun_out/r04_bench_prefault_$t.err; exit 1; }
  python - <<PY
import json
r = json.loads(open("gpurun_out/r04_bench_prefault_$t.json").read().strip().split("\n")[-1])
print("prefault threads $t: first call %.2f ms, repeated %.2f ms, bit exact %s" % (r["host_to_host_first_call_ms"], r["host_to_host_repeat_call_ms"], r["host_to_host"]["bit_exact_vs_cpu"]))
PY
done
for t in 0 4; do
  echo "== fresh process, GRAPHPOPE_PREFAULT_THREADS=$t =="
  GRAPHPOPE_PREFAULT_THREADS=$t python tools/first_call_breakdown.py whole 2>&1 | tail -8
done
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_b.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04_gputests_b.txt; tail -4 gpurun_out/r04_gputests_b.txt
