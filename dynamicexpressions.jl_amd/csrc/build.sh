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
# de_kernels.hip goes through the same steps hipcc runs internally, with one extra pass over the optimised
# device IR (irpatch.py: the interpreter's indirect handler calls need none of the implicit kernel inputs).
# DE_NO_IRPATCH=1 builds it the plain way.
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
build_kernels() {
  local src=de_kernels.hip obj=_obj/de_kernels.o tmp=_obj/irp
  if [ -f "$obj" ] && [ ! "$src" -nt "$obj" ] && [ ! irpatch.py -nt "$obj" ] && [ -z "$(find . ../../include -maxdepth 1 -name '*.h' -newer "$obj" 2>/dev/null | head -1)" ]; then return 0; fi
  echo "  hipcc $src (device IR -> irpatch -> gfx950 code object -> host object)"
  rm -f "$obj"; mkdir -p $tmp
  if [ "${DE_NO_IRPATCH:-0}" = 1 ]; then $HIPCC $FLAGS ${DE_KERNEL_FLAGS:-} -c $src -o $obj; return; fi
  $HIPCC $FLAGS ${DE_KERNEL_FLAGS:-} --cuda-device-only -emit-llvm -S $src -o $tmp/k.ll
  python3 irpatch.py $tmp/k.ll $tmp/k2.ll
  $LLVM/clang -x ir $tmp/k2.ll -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -fPIC -ffp-contract=off -Wno-override-module -c -o $tmp/k.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $tmp/k.out $tmp/k.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=$tmp/k.out -output=$tmp/k.hipfb
  $HIPCC $FLAGS ${DE_KERNEL_FLAGS:-} --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $tmp/k.hipfb -c $src -o $obj
}
build_kernels &
build_obj de_grad_kernels.hip _obj/de_grad_kernels.o &
wait
for o in _obj/de_lower.o _obj/de_bind.o _obj/de_api.o _obj/de_kernels.o _obj/de_grad_kernels.o; do [ -f $o ] || { echo "missing $o"; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libde_hip.so _obj/de_lower.o _obj/de_bind.o _obj/de_api.o _obj/de_kernels.o _obj/de_grad_kernels.o
echo "built $(pwd)/libde_hip.so"
