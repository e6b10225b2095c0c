python -m pytest tests/test_dacs.py -m gpu -x -q -s -k graph 2>&1 | grep -v "^$" | grep "iteration\|passed\|failed\|Error" | head
