"""Run the split / opt-in training GPU tests against an alternative build of the library (make -C dm_nerf_amd/csrc variant ...):
    python scripts/run_tests_with_lib.py build_exp/lib_<name>.so
"""
import os, sys
sys.path.insert(0, os.getcwd())
from dm_nerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(["tests/test_gpu_train.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", "split_wgrad or opt_in"]))
