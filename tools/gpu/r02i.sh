python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python -m pytest tests/test_fullsize.py -m gpu -q -s -k "simple_test or checkpoint" 2>&1 | grep "440x640\|passed\|failed"
