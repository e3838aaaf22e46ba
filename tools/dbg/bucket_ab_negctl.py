"""negative control of tests/test_gpu_bucket_update_ab.py: with the update stream NOT ordered behind the producing streams the A/B
bound must break (run by hand; a race is a matter of timing, so this is not a test)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import test_gpu_bucket_update_ab as T
from erd_amd.engine import BucketedGradSync
ref = T._run(False, 1, 2); twin = T._run(False, 1, 2)
BucketedGradSync._producers = lambda self: []
bad = T._run(True, 1, 2)
try:
    T._compare(ref, twin, bad, "update stream unordered")
    print("NEGATIVE CONTROL NOT DETECTED")
except AssertionError as e:
    print("detected:", str(e)[:200])
