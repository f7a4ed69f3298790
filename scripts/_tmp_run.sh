python scripts/host_issue.py 100000 480 270 300
python scripts/host_issue.py 1000000 1920 1080 100
FG_MODEL_AB=1 python scripts/model_step_bench.py 100000 200 480 270 2>/dev/null | tr -d '\n' | cut -c1-560; echo
FG_MODEL_AB=1 python scripts/model_step_bench.py 300000 100 960 540 2>/dev/null | tr -d '\n' | cut -c1-560; echo
python -m pytest tests -m gpu -x -q -k "one_call or model or densify or step or rasteriz" 2>&1 | tail -3
