#!/bin/bash
# Host-side AddressSanitizer run of the C ABI (the GPU side is not instrumented).
#   bash tests/cpp/run_host_asan.sh build     here (no GPU needed): tests/cpp/asan/{liborbx_asan.so, api_walk}
#   bash tests/cpp/run_host_asan.sh run       on the GPU box
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/tests/cpp/asan
SRC=$ROOT/my_orb_slam2_amd/csrc
if [ "$1" = build ]; then
    mkdir -p "$OUT"
    objs=()
    for f in orbx_api.hip pyramid.hip fast_cells.hip octree.hip orient_desc.hip match.hip frame_grid.hip search_init.hip search_bow.hip tables.cpp synth.cpp; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address \
            -Xarch_host -fno-omit-frame-pointer -I"$ROOT/include" -I"$SRC" -c "$SRC/$f" -o "$OUT/$f.o" &
        objs+=("$OUT/$f.o")
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -fno-gpu-sanitize -o "$OUT/liborbx_asan.so" "${objs[@]}"
    /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" \
        "$ROOT/tests/cpp/api_walk.cpp" -o "$OUT/api_walk" -L"$OUT" -lorbx_asan -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
    rm -f "$OUT"/*.o
    ls -la "$OUT"
else
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 timeout -k 10 120 "$OUT/api_walk"
fi
