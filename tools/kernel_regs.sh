#!/bin/bash
# Register / LDS / spill figures of the kernels of one HIP source:  tools/kernel_regs.sh marl_amd/csrc/edge_env.hip [name-filter] [extra flags]
set -e
src=$1; filt=${2:-.}; shift; shift || true
tmp=$(mktemp -d)
extra=""
case "$src" in *edge_env*|*edge_marl_env*) extra="-ffp-contract=off -fno-fast-math";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$(dirname $src)/../../include -I$(dirname $src) $extra "$@" --cuda-device-only -c $src -o $tmp/bundle.o 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/bundle.o --output=$tmp/dev.o --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.o | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.agpr_count:/{a=$2} /\.vgpr_spill_count:/{vs=$2} /\.sgpr_spill_count:/{ss=$2} /\.private_segment_fixed_size:/{p=$2} /\.group_segment_fixed_size:/{l=$2} /\.wavefront_size:/{print n, "vgpr", v, "agpr", a, "sgpr", s, "vspill", vs, "sspill", ss, "scratch", p, "lds", l}' | grep -E "$filt" | while read line; do n=$(echo $line | cut -d' ' -f1 | c++filt | cut -c1-100); echo "$n | $(echo $line | cut -d' ' -f2-)"; done
rm -rf $tmp
