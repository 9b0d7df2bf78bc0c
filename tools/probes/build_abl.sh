#!/bin/bash
# builds ablation copies of the library for the bf16x3 conv engine: tools/probes/libttc_abl<N>.so (TTC_LIB selects one)
cd "$(dirname "$0")/../../sentinel-tree-cover_amd/csrc" || exit 1
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -DTTC_B3_ABL=$n -c conv3x3_bf16x3.hip -o /tmp/b3_abl$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v conv3x3_bf16x3.o) /tmp/b3_abl$n.o -o ../../tools/probes/libttc_abl$n.so
done
