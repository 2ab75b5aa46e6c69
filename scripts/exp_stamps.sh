#!/usr/bin/env bash
set -u
tag=${1:-stamps}; out=gpurun_out/$tag; mkdir -p "$out" progressive-stable-diffusion_amd/exp
src=progressive-stable-diffusion_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -DDADD_IGEMM_EXP=3 \
  $src/igemm.hip $src/igemm_dma.hip $src/conv_halo.hip $src/norm.hip $src/attention.hip $src/elementwise.hip $src/api.hip \
  -o progressive-stable-diffusion_amd/exp/libdadd_exp3.so > "$out/build.log" 2>&1 || { tail "$out/build.log"; exit 1; }
timeout -k 10 200 python scripts/exp_stamps.py > "$out/stamps.log" 2>&1; echo "rc=$?"; grep -v amdgpu.ids "$out/stamps.log"
