"""Weight-normalised Adam: host-side handle of the optimizer the HIP engine runs.

Mirrors the names of the reference's utils/weightnorm.py.  `AdamWithWeightnorm` (:75-143) is
the default optimizer ('adam-wn', utils/model_utils.py:52-57); its update rule executes on the
device in csrc/optim.hip (clv_adam_wn_step).  `SGDWithWeightnorm` has no call site in the
reference and is not provided.  `data_based_init` (:182-210) is a no-op under the pinned Keras
2.0.0 (it looks for layer attributes `W`/`b`, which Keras-2 layers do not have; SURVEY.md 5.9
B8) and is reproduced as such.
"""
from ..keras_like import OptimizerSpec


class AdamWithWeightnorm(OptimizerSpec):
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-08, decay=0.0):
        super().__init__('adam-wn', lr, beta_1, beta_2, epsilon, decay)


def data_based_init(model, input):
    """No-op, exactly like the reference under Keras 2.0.0 (no layer exposes `W` and `b`)."""
    return None
