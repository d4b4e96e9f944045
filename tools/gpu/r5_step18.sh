#!/bin/bash
python tools/microbench_pinning.py 2>&1 | tail -12
