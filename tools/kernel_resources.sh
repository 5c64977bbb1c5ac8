#!/bin/bash
# tools/kernel_resources.sh <object.o | lib.so> [name filter] -- VGPRs / spills / scratch / LDS of every kernel in a hipcc object: the
# .hip_fatbin section is dumped, the gfx950 code object unbundled (clang-offload-bundler) and its notes read (llvm-readelf).
# Used to check that a kernel does not spill.
OBJ=$1; FILTER=${2:-.}
BIN=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$BIN/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $OBJ $T/ignored.o || exit 1
$BIN/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o || exit 1
$BIN/llvm-readelf --notes $T/dev.o | awk '
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {scr=$2}
  /\.sgpr_count:/ {sg=$2}
  /\.vgpr_count:/ {vg=$2}
  /\.vgpr_spill_count:/ {sp=$2}
  /\.group_segment_fixed_size:/ {lds=$2}
  /\.wavefront_size:/ {printf "%-120s vgpr %4s spill %4s scratch %6s lds %6s sgpr %4s\n", name, vg, sp, scr, lds, sg}
' | grep -E "$FILTER" | sort -u
[ -n "$KEEP" ] && cp $T/dev.o $KEEP
rm -rf $T
