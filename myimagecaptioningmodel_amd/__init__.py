"""myimagecaptioningmodel_amd -- MI355X-native training hot path behind the API of
Chgtaxihe/MyImageCaptioningModel's `model/` package and `train.py` step.

Public surface:
    ImageCaptionModel, Executor      reference-shaped facade (model.py)
    CaptionEngine                    the engine: train_step / forward_backward / decode
    config_compat.from_reference_config / default_cfg
Everything numerical runs in libcapmi.so (csrc/, hand-written gfx950 HIP); importing this
package does not need a GPU, constructing a CaptionEngine does.
"""
from .config_compat import default_cfg, from_reference_config          # noqa: F401
from ._lib import CapmiError, LIB_PATH                                  # noqa: F401


def __getattr__(name):
    if name in ('ImageCaptionModel', 'Executor', 'CaptionEngine', 'Var'):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
