"""replace_num_classes_with_random_weights (reference: modules/head_replacement_utils.py:9-49) for the HIP path's layers: a new output
layer with `num_classes` outputs whose weights / bias are drawn from a normal distribution with the mean and std of the layer it replaces."""
import torch

from .layers import ConvLayer, LinearLayer

__all__ = ["replace_num_classes_with_random_weights"]


def replace_num_classes_with_random_weights(module, num_classes: int):
    if isinstance(module, ConvLayer):
        new = type(module)(module.in_channels, num_classes, module.kernel_size, module.stride, module.padding, bias=module.bias is not None)
        torch.nn.init.normal_(new.weight, mean=module.weight.mean().item(), std=module.weight.std(dim=(0, 1, 2, 3)).item())
        if module.bias is not None:
            torch.nn.init.normal_(new.bias, mean=module.bias.mean().item(), std=module.bias.std(dim=0).item())
        return new
    if isinstance(module, LinearLayer):
        new = LinearLayer(module.in_features, num_classes)
        torch.nn.init.normal_(new.weight, mean=module.weight.mean().item(), std=module.weight.std(dim=(0, 1)).item())
        torch.nn.init.normal_(new.bias, mean=module.bias.mean().item(), std=module.bias.std(dim=0).item())
        return new
    raise ValueError(f"Module {module} does not support replacing the number of classes")
