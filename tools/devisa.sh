#!/bin/bash
# devisa.sh <host .o> <out.s>: extract the gfx950 code object from a hipcc object and disassemble it
set -e
o=$1; out=$2
objcopy -O binary --only-section=.hip_fatbin $o /tmp/fb_$$.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/fb_$$.bin --output=/tmp/co_$$.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/co_$$.co > $out
rm -f /tmp/fb_$$.bin /tmp/co_$$.co
