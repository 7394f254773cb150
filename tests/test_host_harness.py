"""csrc/host.cc (the host half of the host -> host boundary, utils.py:129-147) without a GPU: the real file compiled against
tools/host_race_harness/fake_hip.cc -- copy-engine threads that execute enqueued copies and event records asynchronously -- and
driven through every transport (pinned ring with floats and with hop codes, registered pages, the bounce buffer, refusals),
aborts with the workers asleep, and two assemblies at once; every result is compared with a plain reference inside the harness.
The same binary is what `bash tools/host_race_harness/run.sh thread address` runs under ThreadSanitizer / AddressSanitizer
(profiles/r04_host_harness_sanitizers.txt); here it runs unsanitised so that the CPU suite stays fast."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="needs g++ and the HIP headers")
def test_host_boundary_against_a_fake_copy_engine():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_race_harness", "run.sh"), "none"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all cases passed" in r.stdout
