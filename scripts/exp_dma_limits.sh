#!/usr/bin/env bash
# Bound the LDS-DMA GEMM main loop from both sides: diagnostic libraries with (1) the DMA stream but no
# MFMAs and (2) the MFMAs + LDS fragment reads but no DMA stream (both: igemm_dma.hip, i.e. the 1x1 / linear shapes of
# the list), (6) conv3x3_halo_kernel with idle loader waves (the 3x3 shapes), timed on the dominant layer shapes.
set -u
tag=${1:-exp}; out=gpurun_out/$tag; mkdir -p "$out" progressive-stable-diffusion_amd/exp
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || exit 1
src=progressive-stable-diffusion_amd/csrc
for e in 1 2 6; do
  so=progressive-stable-diffusion_amd/exp/libdadd_exp$e.so
  [ -f "$so" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 \
    -DDADD_IGEMM_EXP=$e $src/igemm.hip $src/igemm_dma.hip $src/conv_halo.hip $src/norm.hip $src/attention.hip $src/elementwise.hip $src/api.hip -o "$so" || exit 1
done
for e in 0 1 2 6; do
  timeout -k 10 200 python scripts/exp_dma_limits.py $e > "$out/exp$e.log" 2>&1; rc=$?
  echo "exp $e rc=$rc"; cat "$out/exp$e.log" | grep -v amdgpu.ids
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
