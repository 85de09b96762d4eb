#!/bin/bash
# Registers, scratch and LDS of every kernel of the shipped library (the code object's metadata notes), and the packed-fp32 guard:
#   tools/kernel_resources.sh [pattern]        lines "kernel vgpr sgpr scratch vgpr_spill sgpr_spill lds" (sgpr_spill: scalar registers kept in VGPR lanes) (pattern: grep on the kernel name)
#   tools/kernel_resources.sh --check-march-no-sgpr-spill  exit 1 if k_march_count_wide<16|64, single cascade> keeps scalar registers in VGPR lanes
#   tools/kernel_resources.sh --check-no-pk-f32  exit 1 if the gfx950 code contains a v_pk_{mul,add,fma}_f32 instruction (rnb-neus2_amd/build.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SO=$ROOT/rnb-neus2_amd/librnb_neus2_hip.so
# the ROCm installation: ROCM_PATH, else where hipcc lives, else /opt/rocm (as rnb-neus2_amd/build.py resolves it)
ROCM=${ROCM_PATH:-}
if [ -z "$ROCM" ] && command -v hipcc >/dev/null 2>&1; then ROCM=$(dirname "$(dirname "$(readlink -f "$(command -v hipcc)")")"); fi
[ -d "$ROCM/lib/llvm/bin" ] || ROCM=/opt/rocm
LLVM=$ROCM/lib/llvm/bin
ARCH=${RNB_OFFLOAD_ARCH:-gfx950}
if [ ! -x "$LLVM/clang-offload-bundler" ] || [ ! -x "$LLVM/llvm-objdump" ]; then
  echo "kernel_resources.sh: no clang-offload-bundler / llvm-objdump under $LLVM -- the code-object guards are SKIPPED (the library itself was built by hipcc)" >&2
  exit 0
fi
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--$ARCH --input=<($LLVM/llvm-objcopy --dump-section .hip_fatbin=/dev/stdout "$SO") --output="$TMP/dev.co" --unbundle 2>/dev/null || {
  $LLVM/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "$SO"
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--$ARCH --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
}
if [ "$1" = "--check-no-pk-f32" ]; then
  n=$($LLVM/llvm-objdump -d --mcpu=$ARCH "$TMP/dev.co" | grep -c -E 'v_pk_(mul|add|fma)_f32' || true)
  echo "v_pk_*_f32 instructions in librnb_neus2_hip.so: $n"
  [ "$n" = "0" ]
  exit $?
fi
if [ "$1" = "--check-march-no-sgpr-spill" ]; then
  # the single-cascade march kernels that run on the side stream beside the backward pass keep no scalar register in VGPR lanes (the other half of round 1's hazard)
  bad=$($LLVM/llvm-readelf --notes "$TMP/dev.co" | awk '/\.name:/ {name=$2} /\.sgpr_spill_count:/ {ssp=$2} /\.wavefront_size:/ { if (name ~ /(k_march_count_wideILi(16|64)ELb1E|k_march_count_skip)/ && ssp + 0 > 0) print name, ssp }')
  echo "single-cascade k_march_count_wide instances with SGPR spills: ${bad:-none}"
  [ -z "$bad" ]
  exit $?
fi
$LLVM/llvm-readelf --notes "$TMP/dev.co" | awk -v pat="${1:-.}" '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  /\.vgpr_spill_count:/ {sp=$2} /\.sgpr_spill_count:/ {ssp=$2}
  /\.wavefront_size:/ { if (name ~ pat) printf "%-70s vgpr %3s sgpr %3s scratch %5s vgpr_spill %4s sgpr_spill %4s lds %6s\n", name, v, s, p, sp, ssp, g }'
