#!/bin/bash
# tuning aid: build a variant of the library with extra -D flags into exp/<name>.so (git-ignored); used with GI_LIB_PATH (tools/exp_bench.sh)
# usage: tools/build_exp.sh name -DGI_EXP_X=1 ...
set -e
cd "$(dirname "$0")/../gi_raytracer_amd/csrc"
name=$1; shift
mkdir -p ../../exp build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c gi_kernels.hip -o build/exp_$name.o
[ -f build/gi_host.o ] || make build/gi_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared build/exp_$name.o build/gi_host.o -lz -o ../../exp/$name.so
echo exp/$name.so
