#!/bin/bash
# Build a variant of libgmf_hip.so from a patched COPY of gmf_amd/csrc (timing experiments; never the library):
#   tools/variant.sh NAME   -> copies csrc (sources AND objects, timestamps kept) to tools/_ab/src_NAME/csrc the first time,
#                              then builds tools/_ab/libgmf_hip_NAME.so from that copy.  Edit the copy and run again.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"
DIR="$ROOT/tools/_ab/src_$NAME/csrc"
mkdir -p "$ROOT/tools/_ab/include"
cp -p "$ROOT/include/gmf_hip.h" "$ROOT/tools/_ab/include/"        # (the sources include ../../include/gmf_hip.h)
if [ ! -d "$DIR" ]; then
  mkdir -p "$DIR"
  cp -p "$ROOT"/gmf_amd/csrc/* "$DIR/"
fi
make -C "$DIR" -j8 OUT="$ROOT/tools/_ab/libgmf_hip_$NAME.so" 2>&1 | grep -E "error|Error" | head -20 || true
ls -la "$ROOT/tools/_ab/libgmf_hip_$NAME.so"
