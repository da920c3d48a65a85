"""MI355X-native batched physics stepper for the HSR pick/place env (hot path of ethanabrooks/hsr-env)."""
from .compiler import CONFIGS, Model, load_config  # noqa: F401
from .env import GoalSpec, HSREnv, VecHSREnv  # noqa: F401
from .spaces import Box  # noqa: F401
