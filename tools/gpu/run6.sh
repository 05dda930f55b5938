cd _old
python - <<'PY'
import sys
sys.path[:0]=["tests","tests/golden",".","x-as-supervision_amd"]
import torch
import test_gpu_model as t
import pytest
# re-run the test body but print the it=1 numbers
src = open('tests/test_gpu_model.py').read()
try:
    t.test_full_train_step_vs_oracle()
    print('OLD: passed')
except AssertionError as e:
    print('OLD: failed', e)
PY
