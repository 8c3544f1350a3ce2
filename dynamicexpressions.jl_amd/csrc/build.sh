#!/usr/bin/env bash
# Builds libde_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -ffp-contract=off: Julia never contracts a*b+c, so neither may the kernels.
set -euo pipefail
fail=0
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -Wall -Wno-unused-function"
mkdir -p _obj
build_obj() { # src obj extra...
  local src=$1 obj=$2; shift 2
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find . ../../include -maxdepth 1 \( -name '*.h' \) -newer "$obj" 2>/dev/null | head -1)" ]; then
    echo "  hipcc $src"
    rm -f "$obj"
    $HIPCC $FLAGS "$@" -c "$src" -o "$obj"
  fi
}
build_obj de_lower.cpp _obj/de_lower.o &
build_obj de_api.cpp _obj/de_api.o &
build_obj de_bind.cpp _obj/de_bind.o &
build_obj de_kernels.hip _obj/de_kernels.o ${DE_KERNEL_FLAGS:-} &
build_obj de_grad_kernels.hip _obj/de_grad_kernels.o &
wait
for o in _obj/de_lower.o _obj/de_bind.o _obj/de_api.o _obj/de_kernels.o _obj/de_grad_kernels.o; do [ -f $o ] || { echo "missing $o"; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libde_hip.so _obj/de_lower.o _obj/de_bind.o _obj/de_api.o _obj/de_kernels.o _obj/de_grad_kernels.o
echo "built $(pwd)/libde_hip.so"
