#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q -x -k "graphed_model" 2>&1 | grep -B2 -A6 "^E \|passed\|failed" | head -30
