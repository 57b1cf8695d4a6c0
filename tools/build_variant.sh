#!/bin/bash
# Developer tool: builds the library from a copy of synthesis_amd/csrc with one file replaced, into synthesis_amd/ab/lib<name>.so
# (picked up through SYNTHESIS_AMD_LIB) — same-box A/B runs of kernel variants in one gpurun call.
#   usage: tools/build_variant.sh <name> [<file in csrc> <replacement file>]...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/variant_$NAME
rm -rf $D; mkdir -p $D/synthesis_amd $D/include
cp -r $R/synthesis_amd/csrc $D/synthesis_amd/csrc
cp $R/include/* $D/include/
rm -f $D/synthesis_amd/csrc/*.o
while [ $# -ge 2 ]; do cp "$2" $D/synthesis_amd/csrc/$1; shift 2; done
mkdir -p $R/synthesis_amd/ab
make -s -j7 -C $D/synthesis_amd/csrc OUT=$R/synthesis_amd/ab/lib$NAME.so 2>&1 | grep -E "error" -A3 || true
ls -la $R/synthesis_amd/ab/lib$NAME.so
