timeout 2000 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v "^  File\|^Extension" | tail -6
