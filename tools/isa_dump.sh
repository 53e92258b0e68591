#!/bin/bash
# tools/isa_dump.sh <object> <out.s>: disassembles the gfx950 code object of one built object file
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $1
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
$LLVM/llvm-objdump -d $T/dev.co > $2
$LLVM/llvm-readelf --notes $T/dev.co | grep -E "vgpr_count|sgpr_count|spill|private_segment|group_segment" 
rm -rf $T
