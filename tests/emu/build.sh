#!/bin/sh
# TEST INFRASTRUCTURE: build the CPU execution harness of the HIP translation unit.
set -e
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -pthread -I. -Wno-unused-value -shared -o libcc_emu.so emu_main.cpp -ldl
