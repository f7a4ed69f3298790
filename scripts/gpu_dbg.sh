#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q -x -k "graphed_model" 2>&1 | grep -B30 "Error\|assert" | head -80
