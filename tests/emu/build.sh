#!/bin/sh
# TEST INFRASTRUCTURE: build the CPU execution harness of the HIP kernels (see hip/hip_runtime.h here).
set -e
cd "$(dirname "$0")"
g++ -O2 -g -std=c++17 -fPIC -ffp-contract=off -pthread -I. -shared -o libcc_emu.so emu_main.cpp
