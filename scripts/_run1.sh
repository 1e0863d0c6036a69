python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear_f16 or lm_head" 2>&1 | tail -4
python scripts/profile_cycle.py --steps 10 --model llama-3-70b --batch 8 --k 3 2>/dev/null | cut -c1-20 | sed "s/^/70b /"
python scripts/profile_cycle.py --steps 10 --model llama-3-70b --batch 4 --k 3 2>/dev/null | cut -c1-20 | sed "s/^/70b bs4 /"
