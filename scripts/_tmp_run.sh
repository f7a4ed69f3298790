timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "test_view_dp_training or factored_view_dp or world_size_8" 2>&1 | grep -v "Gloo\|amdgpu.ids" | tail -15
