set -x
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -40
