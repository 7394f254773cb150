#!/bin/bash
# CPU-only race / memory check of graphpope_amd/csrc/host.cc (build container, no GPU): the real file compiled with a sanitizer
# against tools/host_race_harness/fake_hip.cc instead of libamdhip64.   bash tools/host_race_harness/run.sh [thread|address|none]...
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/tools/_diag/host_harness
mkdir -p "$OUT"
for san in "${@:-thread address}"; do
  for s in $san; do
    flags="-fsanitize=$s"; [ "$s" = none ] && flags=""
    g++ -O1 -g -std=c++17 -pthread $flags -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" \
        "$ROOT/graphpope_amd/csrc/host.cc" "$ROOT/tools/host_race_harness/fake_hip.cc" "$ROOT/tools/host_race_harness/harness.cc" -o "$OUT/harness_$s"
    echo "== $s =="
    TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" ASAN_OPTIONS="detect_leaks=0" "$OUT/harness_$s"
  done
done
