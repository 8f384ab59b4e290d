"""models.get() - the entry point recipes use to obtain a network (reference: training/models/model_factory.py:97-256).
Same signature; the architectures come from this package's registry and run on libsgx_hip."""
from typing import Optional, Union

import warnings

import torch

from ...common.factories import UnknownTypeException
from ...common.registry import ARCHITECTURES
from ..utils.checkpoint_utils import contains_opaque, read_checkpoint
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


def adaptive_load_state_dict(net: torch.nn.Module, state_dict: dict, strict: Union[str, bool]):
    """Reference training/utils/checkpoint_utils.py:79-106.  Every mode except "off" first tries a STRICT load.  When that fails:
    "no_key_matching" pairs the checkpoint's tensors with the model's by position and shape (a checkpoint saved under other layer names) and
    loads that strictly; "key_matching" copies exactly the tensors whose name and shape both match; anything else re-raises with the two
    key lists.  A checkpoint that fits nothing can therefore no longer be "loaded" silently."""
    state_dict = state_dict["net"] if "net" in state_dict else state_dict
    if state_dict and all(k.startswith("module.") for k in state_dict):
        state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
    strict = getattr(strict, "value", strict)  # a StrictLoad member
    mode = strict if isinstance(strict, bool) else {"on": True, "off": False}.get(str(strict), str(strict))
    try:
        net.load_state_dict(state_dict, strict=mode is not False)
        return
    except (RuntimeError, ValueError, KeyError) as ex:
        own = net.state_dict()
        if mode == "no_key_matching":
            ck = [(k, v) for k, v in state_dict.items() if torch.is_tensor(v)]
            if len(ck) != len(own) or any(tuple(v.shape) != tuple(o.shape) for (_, v), o in zip(ck, own.values())):
                raise RuntimeError(f"no_key_matching: the checkpoint's tensors ({len(ck)}) do not pair with the model's ({len(own)}) by position and "
                                   f"shape; first model keys {list(own)[:3]}, first checkpoint keys {[k for k, _ in ck[:3]]}") from ex
            net.load_state_dict({name: v for name, (_, v) in zip(own.keys(), ck)}, strict=True)
        elif mode == "key_matching":
            hit = {k: v for k, v in state_dict.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
            if not hit:
                raise RuntimeError("key_matching: no tensor of the checkpoint matches a model tensor by name and shape") from ex
            net.load_state_dict(hit, strict=False)
        else:
            missing = [k for k in own if k not in state_dict]
            unexpected = [k for k in state_dict if k not in own]
            raise RuntimeError(f"checkpoint does not fit the model: {len(missing)} missing keys (e.g. {missing[:3]}), {len(unexpected)} unexpected "
                               f"(e.g. {unexpected[:3]}); strict_load={strict!r}") from ex


def get_model_name(model: torch.nn.Module) -> Optional[str]:
    return getattr(model, "_sg_model_name", None)


def _maybe_load_preprocessing_params(net, ckpt) -> bool:
    """checkpoint_utils.py:1625-1651: a checkpoint's "processing_params" (class names, image processor, NMS defaults) go to
    model.set_dataset_processing_params; a failure only warns.  The image processor is stored as a `{TypeName: kwargs}` config
    (Processing.to_config) so the file stays loadable with weights_only=True."""
    if not (isinstance(ckpt, dict) and "processing_params" in ckpt and hasattr(net, "set_dataset_processing_params")):
        return False
    if contains_opaque(ckpt["processing_params"]):  # a reference-written file: its image processor is a pickled object, not a config
        warnings.warn("The checkpoint stores its preprocessing pipeline as pickled objects, which are not constructed from a file. Before calling "
                      "predict make sure to call set_dataset_processing_params.")
        return False
    try:
        net.set_dataset_processing_params(**ckpt["processing_params"])
        return True
    except Exception as e:  # noqa: BLE001  (the reference swallows everything here too)
        warnings.warn(f"Could not set preprocessing pipeline from the checkpoint dataset: {e}. Before calling predict make sure to call "
                      "set_dataset_processing_params.")
        return False


def get(model_name: str, arch_params: Optional[dict] = None, num_classes: Optional[int] = None, strict_load: Union[str, bool] = "no_key_matching",
        checkpoint_path: Optional[str] = None, pretrained_weights: Optional[str] = None, load_backbone: bool = False,
        download_required_code: bool = True, checkpoint_num_classes: Optional[int] = None, num_input_channels: Optional[int] = None):
    checkpoint_num_classes = checkpoint_num_classes or num_classes
    net = instantiate_model(model_name, arch_params, checkpoint_num_classes or num_classes, pretrained_weights, download_required_code)
    if load_backbone and not checkpoint_path:
        raise ValueError("Please set checkpoint_path when load_backbone=True")
    if checkpoint_path:
        ckpt = read_checkpoint(checkpoint_path)  # tensors and plain containers only: never runs pickled code (objects come back as placeholders)
        sd = ckpt.get("ema_net", ckpt.get("net", ckpt)) if isinstance(ckpt, dict) else ckpt
        adaptive_load_state_dict(net, sd, strict_load)
        _maybe_load_preprocessing_params(net, ckpt)
    if checkpoint_num_classes != num_classes:
        net.replace_head(new_num_classes=num_classes)  # transfer learning (model_factory.py:250-251)
    if num_input_channels is not None and num_input_channels != net.get_input_channels():
        net.replace_input_channels(in_channels=num_input_channels)  # model_factory.py:253-254
    return net
