#!/bin/bash
timeout 300 python scripts/prebwd_stamps.py 2>&1 | tail -5
