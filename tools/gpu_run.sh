#!/bin/bash
timeout 300 python -m pytest tests -m gpu -x -q -k "labels_only or upper_bound or tokenize_batch_in_chunks" 2>&1 | tail -2
