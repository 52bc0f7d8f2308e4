#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -12 gpurun_out/t.log
