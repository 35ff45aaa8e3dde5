#!/bin/bash
timeout 300 python -m pytest tests/test_cpp_mirror.py -m gpu -x -q 2>&1 | tail -3
