"""CPU oracle for the per-target Kalman path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Nothing under target_estimation_amd/ does.  See te_oracle.h for the parity status.
"""
from .oracle import (  # noqa: F401
    MODELS, MODEL_DIMS, OracleBatch, OracleGate, OracleTarget, build, load, load_model_yaml,
    ref_test_stream, lowest_real_root, poly_roots, harness_run, stream_fill, stream_sample,
)
