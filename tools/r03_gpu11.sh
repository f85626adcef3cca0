python -m pytest tests/test_gpu_steps.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head
