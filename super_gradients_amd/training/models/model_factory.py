"""models.get() - the entry point recipes use to obtain a network (reference: training/models/model_factory.py:97-256).
Same signature; the architectures come from this package's registry and run on libsgx_hip."""
from typing import Optional, Union

import torch

from ...common.factories import UnknownTypeException
from ...common.registry import ARCHITECTURES
from ..utils.utils import HpmStruct, get_param


def get_architecture(model_name: str, arch_params: HpmStruct):
    if not isinstance(model_name, str):
        raise ValueError("Parameter model_name is expected to be a string.")
    if model_name not in ARCHITECTURES:
        raise UnknownTypeException(message=f'The required model, "{model_name}", was not found in the HIP-path registry', unknown_type=model_name,
                                   choices=[k for k in ARCHITECTURES.keys() if not k.startswith("_")])
    return ARCHITECTURES[model_name], arch_params


def instantiate_model(model_name: str, arch_params: dict, num_classes: int, pretrained_weights: str = None, download_required_code: bool = True):
    arch_params = HpmStruct(**(arch_params or {}))
    cls, arch_params = get_architecture(model_name, arch_params)
    if get_param(arch_params, "num_classes"):
        num_classes = num_classes or arch_params.num_classes
    if num_classes is not None:
        arch_params.override(num_classes=num_classes)
    if pretrained_weights is None and num_classes is None:
        raise ValueError("num_classes or pretrained_weights must be passed to determine net's structure.")
    if pretrained_weights:
        raise NotImplementedError("pretrained weights are downloaded by the reference; this environment has no network. "
                                  "Load a reference checkpoint with checkpoint_path= instead (state_dict keys are identical).")
    net = cls(arch_params=arch_params)
    setattr(net, "_sg_model_name", model_name)
    return net


def get_model_name(model: torch.nn.Module) -> Optional[str]:
    return getattr(model, "_sg_model_name", None)


def get(model_name: str, arch_params: Optional[dict] = None, num_classes: Optional[int] = None, strict_load: Union[str, bool] = "no_key_matching",
        checkpoint_path: Optional[str] = None, pretrained_weights: Optional[str] = None, load_backbone: bool = False,
        download_required_code: bool = True, checkpoint_num_classes: Optional[int] = None, num_input_channels: Optional[int] = None):
    checkpoint_num_classes = checkpoint_num_classes or num_classes
    net = instantiate_model(model_name, arch_params, checkpoint_num_classes or num_classes, pretrained_weights, download_required_code)
    if load_backbone and not checkpoint_path:
        raise ValueError("Please set checkpoint_path when load_backbone=True")
    if checkpoint_path:
        ckpt = torch.load(checkpoint_path, map_location="cpu")
        sd = ckpt.get("ema_net", ckpt.get("net", ckpt)) if isinstance(ckpt, dict) else ckpt
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        net.load_state_dict(sd, strict=strict_load in (True, "on"))
    if checkpoint_num_classes != num_classes:
        net.replace_head(new_num_classes=num_classes)  # transfer learning (model_factory.py:250-251)
    if num_input_channels is not None and num_input_channels != net.get_input_channels():
        raise NotImplementedError("pass in_channels through arch_params instead of num_input_channels")
    return net
