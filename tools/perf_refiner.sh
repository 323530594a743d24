#!/bin/bash
# builds and runs tools/cpp/perf_refiner.cpp against the GPU library (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
g++ -O2 -std=c++17 -Iinclude -Imanta_amd/host tools/cpp/perf_refiner.cpp -o tools/cpp/perf_refiner -Lmanta_amd -lmanta_amd -Wl,-rpath,$PWD/manta_amd -lpthread
./tools/cpp/perf_refiner ${1:-10000} 0
./tools/cpp/perf_refiner ${2:-1000} 1
