#!/bin/bash
# Host-only build of libfilm_hip with a sanitizer: the four .cpp translation units (planner, lane analysis, weight packer, tune cache,
# bundle reader, executor bookkeeping) by g++ + abort()ing stubs for the kernel launchers.  For plan-only handles (device = -1), no GPU.
#   tools/sanitize/build_host.sh address|thread|undefined [outdir=/tmp/film_san_<kind>]    -> <outdir>/libfilm_hip_<kind>.so
# Run the CPU suites on it:
#   FILM_NO_TORCH=1 FILM_HIP_LIB=<so> LD_PRELOAD="$(g++ -print-file-name=lib{a,t,ub}san.so) $(g++ -print-file-name=libstdc++.so.6)" \
#     [ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 | TSAN_OPTIONS=halt_on_error=1] python -m pytest tests/test_host_threads_cpu.py ...
# (libstdc++ in LD_PRELOAD: the sanitizer's __cxa_throw interceptor must find the real one at process start; python does not link it)
set -e
KIND=${1:?address|thread|undefined}
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${2:-/tmp/film_san_$KIND}
mkdir -p "$OUT"
FL="-std=c++17 -O1 -g -fsanitize=$KIND -fno-omit-frame-pointer -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
for f in film_bundle film_engine film_layers film_planner; do
  g++ $FL -DFILM_SRC_ID="\"$KIND\"" -c "$R/frame-interpolation_amd/csrc/$f.cpp" -o "$OUT/$f.o" &
done
g++ $FL -c "$R/tools/sanitize/launch_stubs.cpp" -o "$OUT/launch_stubs.o" &
wait
g++ -shared -fPIC -fsanitize=$KIND -o "$OUT/libfilm_hip_$KIND.so" "$OUT"/*.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
echo "$OUT/libfilm_hip_$KIND.so"
