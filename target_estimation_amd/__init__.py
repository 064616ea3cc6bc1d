"""target_estimation_amd -- MI355X-native batched Kalman predict/update path behind the
TargetManager C ABI of graiola/target_estimation.  See DESIGN.md / INTEGRATION.md."""
from .manager import (  # noqa: F401
    ANGULAR_RATES, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY, MODEL_DIMS, MODEL_TYPES, Batch,
    MeasurementIngest, TargetManager,
)
from ._build import build  # noqa: F401
