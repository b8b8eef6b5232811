#!/bin/bash
# builds the standalone laboratories (gfx950) next to their sources; the binaries travel to the GPU box with the gpurun snapshot
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-result -o decode_lab decode_lab.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o cu_mask_probe cu_mask_probe.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o libpaced_copy.so paced_copy.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o handoff_lab handoff_lab.hip
