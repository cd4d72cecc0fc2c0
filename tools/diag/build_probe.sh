# Diagnostic builds of the library (never the product): conv.hip compiled with one of
#   -DDIGA_PROBE_STAMP       s_memrealtime stamps of the tile phases of conv_fwd_dma_kernel (tools/diag/f32_tile_stamps.py)
#   -DDIGA_PROBE_NOSTORE     the epilogue's output stores removed   -DDIGA_PROBE_PLAINSTORE  default-policy instead of streaming stores
# linked with the product objects into diga_amd/libdiga_probe_<V>.so; load with DIGA_LIB=<path>.  Run after `python -m diga_amd.build`.
set -e
cd "$(dirname "$0")/../../diga_amd"
mkdir -p build_probe
for v in STAMP NOSTORE PLAINSTORE; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -DDIGA_PROBE_$v -I../include -Icsrc -c csrc/conv.hip -o build_probe/conv_$v.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o libdiga_probe_$v.so build_probe/conv_$v.o $(ls build/*.o | grep -v "/conv.o")
done
