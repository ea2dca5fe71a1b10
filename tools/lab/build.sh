#!/bin/bash
# usage: tools/lab/build.sh <name> <-D flags...>   ->  tools/lab/lib<name>.so   (one kernel family, seconds to compile)
name=$1; shift
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o lib$name.so lab.hip
