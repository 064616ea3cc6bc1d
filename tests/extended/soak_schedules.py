#!/usr/bin/env python3
"""Soak run of tests/test_gpu_edge_cases.py::test_randomised_schedule_against_oracle over many seeds (GPU box).
usage: python tests/extended/soak_schedules.py <number of seeds>   (from the repo root; test infrastructure: it uses the oracle)   (6 random schedules per seed)"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import conftest, oracle
import test_gpu_edge_cases as t
models = {k: oracle.load_model_yaml(conftest.model_path(k)) for k in conftest.MODEL_FILES}
bad = 0
for seed in range(1000, 1000 + int(sys.argv[1])):
    try:
        t.test_randomised_schedule_against_oracle(models, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300], flush=True)
print("soak done, failures:", bad)
