#!/bin/bash
# Stand-alone GPU programs (no Python start-up: a session costs seconds).  Round 4 ran them all (profiles/r04_hazard_form_matrix.md,
# profiles/r04_ab_runs.md); the Gram-statistics tail prototype lost its A/B (0.27-0.72x) and was deleted with its branch.
#   bash scripts/prepared_gpu_runs.sh build      here (hipcc cross-compiles; the binaries travel with the gpurun snapshot, scripts/*.bin)
#   gpurun --timeout 240 -- 'bash scripts/prepared_gpu_runs.sh run'       outputs in gpurun_out/prepared/
#     hazard_forms.txt   instruction-form matrix of profiles/r03_concurrency_hazard.md 4e (in place or not, low / high lane crossed, add/mul/fma)
#     abi_bench.txt      packed GEMM shapes of the bench workload alone; the pyramid of a 3DMatch and of a KITTI stack; the embedding path
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
HIPCC=/opt/rocm/bin/hipcc
case "${1:-run}" in
  build)
    make -C geotransformer_amd/csrc -j8 > /dev/null || exit 1
    for v in slp noslp; do
      fl=""; [ $v = noslp ] && fl="-fno-slp-vectorize"
      $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $fl -I include -I geotransformer_amd/csrc -I scripts \
        scripts/packed_fp32_mfma_hazard.hip -o scripts/packed_fp32_mfma_hazard_$v.bin || exit 1
    done
    $HIPCC -O2 -std=c++17 -I include scripts/abi_bench.cpp -L geotransformer_amd -lgeotr_hip -Wl,-rpath,'$ORIGIN/../geotransformer_amd' \
      -o scripts/abi_bench.bin || exit 1
    ls -la scripts/*.bin ;;
  run)
    O=gpurun_out/prepared; mkdir -p $O
    {
      for form in 0 1 2 3 4 5 6; do  # aggressors: loads + bf16 MFMA | LDS DMA + fp32 MFMA (control) | nothing but the LDS reservation (control)
        VICTIM=instruction FORM=$form timeout 20 scripts/packed_fp32_mfma_hazard_slp.bin 200 11 6 14
      done
      VICTIM=instruction FORM=0 timeout 20 scripts/packed_fp32_mfma_hazard_slp.bin 200 13 10 1   # f16 MFMA only | bf16 MFMA only (LDS reserved) | bf16 MFMA only
    } 2>&1 | grep -v amdgpu.ids | tee $O/hazard_forms.txt
    { timeout 40 scripts/abi_bench.bin shapes; timeout 30 scripts/abi_bench.bin pyramid 3dmatch 16 5; timeout 40 scripts/abi_bench.bin pyramid kitti 4 5;
      timeout 30 scripts/abi_bench.bin embedding 32 300 5; } 2>&1 | grep -v amdgpu.ids | tee $O/abi_bench.txt ;;
esac
