#!/bin/bash
# libfilm_hip with AddressSanitizer on the HOST side and the real kernels: the four .cpp translation units (executor, plan cache, event
# table, graph lifetime, planner, packer, bundle reader) by g++ -fsanitize=address, the .hip units (kernels + their launch stubs) as
# film_hip/build.py built them.  For the GPU box: round-5 verdict item 3, "run the graph executor under ASan where it crashed".
#   tools/sanitize/build_asan_gpu.sh          -> tools/bin/libfilm_hip_asan.so   (tools/bin travels with gpurun, stays out of git)
#   FILM_HIP_LIB=tools/bin/libfilm_hip_asan.so ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0 \
#     LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" FILM_TEST_EXECUTOR=1 python -m pytest ...
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/frame-interpolation_amd/csrc
O=$R/tools/bin/asan_obj
mkdir -p "$O"
(cd "$R/frame-interpolation_amd" && python -m film_hip.build > /dev/null)
FL="-std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
for f in film_bundle film_engine film_layers film_planner; do g++ $FL -DFILM_SRC_ID='"asan-host"' -c "$C/$f.cpp" -o "$O/$f.o" & done
wait
g++ -shared -fPIC -fsanitize=address -o "$R/tools/bin/libfilm_hip_asan.so" "$O"/*.o "$C/build/conv_igemm.o" "$C/build/misc_kernels.o" \
    -Wl,--version-script,"$C/film_hip.map" -L"$C/build/stub" -Wl,--no-as-needed -lamdhip64 -Wl,--as-needed -Wl,-rpath,/opt/rocm/lib
echo "$R/tools/bin/libfilm_hip_asan.so"
