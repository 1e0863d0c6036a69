set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_draft_forms_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "k_slices or w4a4_gemm_bit_exact or batch32 or two_token or qkv_rope or gate_up" > gpurun_out/t_r04f.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_r04f.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "k5_bs32 or teacher_forced or fused_forward" > gpurun_out/t_r04f2.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/t_r04f2.log
for v in "QSPEC_DOWN_K_SLICES=0" "QSPEC_DOWN_K_SLICES=1" "QSPEC_DOWN_K_SLICES=0" "QSPEC_DOWN_K_SLICES=1"; do env $v python scripts/profile_cycle.py --steps 20 --model llama-3-8b --batch 32 --k 5 2>/dev/null | cut -c1-20 | sed "s/^/8b bs32 k5 $v /"; done
scripts/profile_shapes.sh r04f "llama-3-8b 32 5"
