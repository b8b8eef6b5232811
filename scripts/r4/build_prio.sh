#!/bin/bash
# builds cxrmate_amd/lib/libcxrmate_hip_prio$1.so = the library with -DCXR_MAIN_PRIO=$1 (main-stream kernels raise their wave priority); select it with CXR_LIB=...
set -e
P=${1:-2}
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
for f in $R/cxrmate_amd/csrc/*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -DCXR_MAIN_PRIO=$P -c $f -o $T/$(basename $f .hip).o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cxrmate_amd/lib/libcxrmate_hip_prio$P.so $T/*.o
rm -rf $T
echo built $R/cxrmate_amd/lib/libcxrmate_hip_prio$P.so
