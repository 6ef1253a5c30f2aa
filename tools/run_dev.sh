timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "
import json; d=json.load(open('gpurun_out/parity_r02.json')); print(d.get('cfg4_feature_stage_peak_bytes'))"
