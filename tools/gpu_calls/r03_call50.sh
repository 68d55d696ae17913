#!/bin/bash
# third fuzz campaign (800 cases, other seeds), all failures listed
FLX_FUZZ_CASES=800 FLX_FUZZ_BASE=campaign-3 timeout 420 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k random 2>&1 | tail -40
