#!/usr/bin/env python3
"""test_random_one_target_call_sequences_against_oracle over many seeds (GPU box, from the repo root).
usage: python tests/extended/soak_one_target.py <number of seeds>"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import conftest, oracle
import test_gpu_edge_cases as t


class _NoCapture:
    def readouterr(self):
        return None


models = {k: oracle.load_model_yaml(conftest.model_path(k)) for k in conftest.MODEL_FILES}
bad = 0
for seed in range(2000, 2000 + int(sys.argv[1])):
    try:
        t.test_random_one_target_call_sequences_against_oracle(models, seed, _NoCapture())
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300], flush=True)
print("one-target soak done, seeds:", sys.argv[1], "failures:", bad)
sys.exit(1 if bad else 0)
