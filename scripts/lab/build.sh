#!/bin/bash
# builds the standalone decode laboratory (gfx950) next to its source; the binary travels to the GPU box with the gpurun snapshot
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-result -o decode_lab decode_lab.hip

