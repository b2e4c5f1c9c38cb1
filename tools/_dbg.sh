timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "bf16_vs_fp32_loss" 2>&1 | grep -v "^\[W" | tail -12
