#!/bin/bash
# AddressSanitizer + UBSan over the host-side formats code (text loaders incl. the multi-threaded reader, binary
# files): CPU build only (GPU sanitizers are not available on this pool).  Run from the repository root.
set -e
mkdir -p /tmp/tahoe_asan
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Itahoe_amd/csrc \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/asan/formats_driver.cpp tahoe_amd/csrc/formats.cpp \
    -o /tmp/tahoe_asan/drv -lpthread
/tmp/tahoe_asan/drv tests/golden
