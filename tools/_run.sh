timeout 900 python -m pytest tests/test_oracle_vs_reference.py tests/test_kernels.py -m gpu -q -s -k "product_model_golden or torchvision or tuning" 2>&1 | grep -E "^\[|passed|failed|Error" | head
