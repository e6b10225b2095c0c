python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | grep "injected\|512x512\|passed\|failed\|FAILED\|Error" | cut -c1-400
