#!/bin/bash
# A/B builds of the library: scripts/build_variant.sh NAME SOURCE.hip REPLACES [extra compiler flags] -> exp/libtad_NAME.so
# (SOURCE.hip is compiled in place of the object REPLACES, e.g. mm16)
# (the other objects are the ones of the regular build; select with TAD_HIP_LIB=exp/libtad_NAME.so)
set -e
cd "$(dirname "$0")/../timeautodiff_amd/csrc"
name=$1; src=$2; repl=$3; shift 3
mkdir -p ../../exp
obj=../../exp/${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -I. "$@" -c $src -o $obj
others=$(for f in gemm gemm16 mm16 pack16 rnn rnn_cluster elementwise vae tvmcg draws api; do [ "$f" = "$repl" ] || echo $f.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp/libtad_${name}.so $obj $others
echo built exp/libtad_${name}.so
