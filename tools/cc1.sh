#!/bin/bash
# compile one csrc file for gfx950 (same flags as speechclip_plus_amd/build.py); usage: tools/cc1.sh rowops.hip [extra flags]
set -e
cd "$(dirname "$0")/../speechclip_plus_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -x hip -I ../../include -I . -c "$f" -o "${f%.*}.o" "$@"
echo "compiled $f"
