#!/bin/bash
# usage: tools/build_variant.sh <name> <-D flags...>   -> cheetah_amd/lib/variants/lib<name>.so (A/B timing of kernel variants)
name=$1; shift
cd "$(dirname "$0")/../cheetah_amd/csrc" && mkdir -p ../lib/variants && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o ../lib/variants/lib$name.so capi.hip
