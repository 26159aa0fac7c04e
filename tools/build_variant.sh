#!/bin/bash
# usage: tools/build_variant.sh <file.hip> <out.so> [extra hipcc flags...]  -- rebuilds ONE source with extra flags and links it with
# the other objects of the last regular build into an alternative library (load it with MART_HIP_LIB=<out.so>)
set -e
cd "$(dirname "$0")/.."
src=$1; out=$2; shift 2
extra=""; { [ "$src" = attention.hip ] || [ "$src" = precise.hip ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imkg_analogy_amd/csrc -Wno-unused-result -Wno-pass-failed $extra "$@" -c mkg_analogy_amd/csrc/$src -o /tmp/variant_${src%.hip}.o
objs=""; for o in util gemm_nt gemm_tn norm_embed attention fusion head_optim precise; do
  if [ "$o.hip" = "$src" ]; then objs="$objs /tmp/variant_$o.o"; else objs="$objs mkg_analogy_amd/build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
echo built $out
