#!/bin/bash
# Diagnostic: build a variant library for tools/cnn_ab.py.
# usage: tools/mkvariant.sh NAME path/to/source.hip replaced_object_basename ["-DSS_VAR=1"]
# -> silent_speech_amd/_ab/libNAME.so = the in-tree objects with csrc/<replaced>.o swapped for the given source.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; src=$(realpath "$2"); repl=$3; defs=$4
cd "$root/silent_speech_amd"
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -Icsrc -I../include $defs -c "$src" -o _ab/${repl}_$name.o
objs=$(ls csrc/*.o | grep -v "csrc/${repl}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib$name.so $objs _ab/${repl}_$name.o
ls -la _ab/lib$name.so
