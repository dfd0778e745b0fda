#!/bin/bash
# The instruction-form matrix of profiles/r03_concurrency_hazard.md section 4e (not run in round 3: the GPU budget ended with FORM 0).
#   bash scripts/hazard_form_matrix.sh build          here (hipcc cross-compiles; the binaries travel with the gpurun snapshot)
#   gpurun --timeout 120 -- 'bash scripts/hazard_form_matrix.sh run'     ~20 s on the box; output in gpurun_out/hazard_forms.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
case "${1:-run}" in
  build)
    for v in slp noslp; do
      fl=""; [ $v = noslp ] && fl="-fno-slp-vectorize"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $fl -I include -I geotransformer_amd/csrc -I scripts \
        scripts/packed_fp32_mfma_hazard.hip -o scripts/packed_fp32_mfma_hazard_$v.bin || exit 1
    done
    ls -la scripts/packed_fp32_mfma_hazard_*.bin ;;
  run)
    mkdir -p gpurun_out
    {
      for form in 0 1 2 3 4 5 6; do  # aggressors: loads + bf16 MFMA | LDS DMA + fp32 MFMA (control) | nothing but the LDS reservation (control)
        VICTIM=instruction FORM=$form timeout 20 scripts/packed_fp32_mfma_hazard_slp.bin 200 11 6 14
      done
      VICTIM=instruction FORM=0 timeout 20 scripts/packed_fp32_mfma_hazard_slp.bin 200 13 10 1   # f16 MFMA only | bf16 MFMA only (LDS reserved) | bf16 MFMA only
    } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/hazard_forms.txt ;;
esac
