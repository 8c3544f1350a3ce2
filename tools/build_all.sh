#!/usr/bin/env bash
# libde_hip.so and the plain (escape-hatch) variant libde_hip_plain.so — what __graft_entry__.build() builds; use this after every change under csrc/
# (tests/test_gpu_round6.py runs the parity tests against the plain library: a stale one fails there).
set -e
cd "$(dirname "$0")/../dynamicexpressions.jl_amd/csrc"
bash build.sh 2>&1 | grep -v "^asm\|^irpatch\|unused" | tail -2
DE_PLAIN_BUILD=1 DE_OBJ_DIR=_obj_plain DE_OUT_LIB=libde_hip_plain.so bash build.sh 2>&1 | grep -v "^asm\|^irpatch\|unused" | tail -1
