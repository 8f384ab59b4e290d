for wb in 1 0; do echo "WT_BATCH=$wb"; SGX_WT_BATCH=$wb python tools/_diag.py 2>&1 | tail -15; done
