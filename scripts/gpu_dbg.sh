#!/bin/bash
timeout 300 python scripts/hbm_bandwidth.py 2>&1 | tail -1
