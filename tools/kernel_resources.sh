#!/bin/bash
# Registers / occupancy / LDS of the kernels of one source file whose name matches a pattern (build container: hipcc's own remarks).
#   bash tools/kernel_resources.sh geodesic.hip k_bfs_level [-DPOPE_WT8=1 ...]
R=$(cd "$(dirname "$0")/.." && pwd)
src=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include "$@" -Rpass-analysis=kernel-resource-usage -c $R/graphpope_amd/csrc/$src -o /dev/null 2>&1 |
  python3 -c "
import re,sys,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().split('(')[0]; vals={}
        continue
    m=re.search(r'remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',l)
    if m and cur is not None:
        vals[m.group(1).split(' ')[0]]=m.group(2)
        if m.group(1).startswith('LDS') and '$pat' in cur:
            print(f'{cur:60s} ' + ' '.join(f'{k} {v}' for k,v in vals.items()))
"
