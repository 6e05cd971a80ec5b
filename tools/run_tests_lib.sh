#!/bin/bash
# developer tool (GPU box): set-op parity tests against an experimental library.  usage: run_tests_lib.sh TAG [-k expr]
cd $GRAFT_REPO_ROOT
tag=$1; shift
UKM_LIB_PATH=$GRAFT_REPO_ROOT/unikmer_amd/libukm_exp_$tag.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q -m gpu -k "${1:-setop or union or inter or diff}" 2>&1 | grep -E "passed|failed|error" | tail -3
