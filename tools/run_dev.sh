python tools/dev_cbca_check.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
