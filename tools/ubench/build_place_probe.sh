#!/bin/bash
# builds tools/ubench/place_probe against the library's own row-kernel object
cd "$(dirname "$0")" || exit 1
make -C ../../sporco_amd/csrc csc_rows.o ck_misc.o >/dev/null || exit 1
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -I../../sporco_amd/csrc -c place_probe.hip -o place_probe.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 place_probe.o ../../sporco_amd/csrc/csc_rows.o ../../sporco_amd/csrc/ck_misc.o -o place_probe && rm -f place_probe.o
