"""Factory with the reference's signature (mpd/trainer/train_loaders.py:11-25)."""
from __future__ import annotations


def get_model(model_class=None, checkpoint_path=None, freeze_loaded_model=False, tensor_args=None, **kwargs):
    """getattr(ramp_amd.models, model_class)(**kwargs).to(device); sets ``.submodules = {}`` like
    ``model_loader`` (mpd/utils/decorators.py:88-104)."""
    from . import models
    if checkpoint_path is not None:
        raise NotImplementedError("checkpoint_path: load with model.load_state_dict(torch.load(...)) as the "
                                  "inference scripts do (inference_static.py:107-111)")
    cls = getattr(models, model_class)
    model = cls(**kwargs)
    if tensor_args is not None and 'device' in tensor_args:
        model = model.to(tensor_args['device'])
    model.submodules = {}
    return model
