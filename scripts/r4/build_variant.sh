#!/bin/bash
# builds cxrmate_amd/lib/libcxrmate_hip_$1.so from the CURRENT csrc/ (for same-call A/B of kernel variants: select with CXR_LIB=...); extra hipcc flags after the name
set -e
N=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
for f in $R/cxrmate_amd/csrc/*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c $f -o $T/$(basename $f .hip).o 2>/dev/null ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cxrmate_amd/lib/libcxrmate_hip_$N.so $T/*.o
rm -rf $T
echo built $R/cxrmate_amd/lib/libcxrmate_hip_$N.so
