# diagnostic .so with phase stamps in the 3x3 kernel (never shipped): tools/micro/bin/libconv3_stamps.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/micro/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DHD_CONV3_STAMPS -Iinclude -Iheltondetection_amd/csrc \
  -o tools/micro/bin/libconv3_stamps.so heltondetection_amd/csrc/conv3x3.hip heltondetection_amd/csrc/conv3x3_wgrad.hip tools/micro/conv3_stamps_stub.hip
