#!/bin/bash
# Register / spill / LDS summary of the matrix-core kernels (hipcc -Rpass-analysis=kernel-resource-usage):
#   bash tools/kernel_resources.sh [file.hip]
f=${1:-cs_corr_mfma.hip}
cd "$(dirname "$0")/../chromosight_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -DCS_HAVE_FAST $CS_EXTRA -c $f -o /tmp/kr_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name'):
        if cur: print(cur)
        cur=t.split(': ')[1][:70]
    elif any(t.startswith(k) for k in ('VGPRs:','SGPRs Spill','VGPRs Spill','ScratchSize','Occupancy')):
        cur+=' | '+t.replace(' [bytes/lane]','').replace(' [bytes/block]','').replace(' [waves/SIMD]','')
if cur: print(cur)
"
rm -f /tmp/kr_$$.o
