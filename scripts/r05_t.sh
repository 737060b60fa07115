#!/bin/bash
timeout 600 python -m pytest tests/test_rollout.py -q -m gpu -k "async" 2>&1 | tail -15
