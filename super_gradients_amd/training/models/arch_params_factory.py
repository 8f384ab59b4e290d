"""Default architecture parameters (reference: training/models/arch_params_factory.py:9-27 loads
recipes/arch_params/<name>.yaml through hydra).  Here the YOLO-NAS S/M/L tables are generated from their few
distinguishing numbers; the produced dictionaries have the reference YAMLs' schema ({TypeName: {kwargs}} nesting), so
user `arch_params` overrides written for the reference apply unchanged."""
import copy

# per variant: stage hidden channels, stage concat_intermediates, neck (num_blocks, hidden_channels) x4, head width_mult
_YOLO_NAS = {
    "yolo_nas_s_arch_params": dict(hid=[32, 64, 96, 192], cat=[False, False, False, False], neck=[(2, 64), (2, 48), (2, 64), (2, 64)], head=0.5),
    "yolo_nas_m_arch_params": dict(hid=[64, 128, 256, 384], cat=[True, True, True, False], neck=[(2, 192), (3, 64), (2, 192), (3, 256)], head=0.75),
    "yolo_nas_l_arch_params": dict(hid=[96, 128, 256, 512], cat=[True, True, True, True], neck=[(4, 128), (4, 128), (4, 128), (4, 256)], head=1),
}


def _yolo_nas(spec):
    stage_out, stage_blocks = [96, 192, 384, 768], [2, 3, 5, 2]
    stages = [{"YoloNASStage": dict(out_channels=stage_out[i], num_blocks=stage_blocks[i], activation_type="relu", hidden_channels=spec["hid"][i],
                                    concat_intermediates=spec["cat"][i])} for i in range(4)]
    backbone = {"NStageBackbone": dict(stem={"YoloNASStem": dict(out_channels=48)}, stages=stages,
                                       context_module={"SPP": dict(output_channels=768, activation_type="relu", k=[5, 9, 13])},
                                       out_layers=["stage1", "stage2", "stage3", "context_module"])}
    n = spec["neck"]
    up = lambda oc, nb, hc: {"YoloNASUpStage": dict(out_channels=oc, num_blocks=nb, hidden_channels=hc, width_mult=1, depth_mult=1,  # noqa: E731
                                                    activation_type="relu", reduce_channels=True)}
    down = lambda oc, nb, hc: {"YoloNASDownStage": dict(out_channels=oc, num_blocks=nb, hidden_channels=hc, activation_type="relu",  # noqa: E731
                                                        width_mult=1, depth_mult=1)}
    neck = {"YoloNASPANNeckWithC2": dict(neck1=up(192, *n[0]), neck2=up(96, *n[1]), neck3=down(192, *n[2]), neck4=down(384, *n[3]))}
    heads = {"NDFLHeads": dict(num_classes=80, reg_max=16, heads_list=[
        {"YoloNASDFLHead": dict(inter_channels=c, width_mult=spec["head"], first_conv_group_size=0, stride=s)} for c, s in ((128, 8), (256, 16), (512, 32))])}
    return dict(in_channels=3, backbone=backbone, neck=neck, heads=heads, bn_eps=1e-3, bn_momentum=0.03, inplace_act=True)


# PP-YOLOE (recipes/arch_params/ppyoloe_arch_params.yaml + ppyoloe_{s,m,l,x}_arch_params.yaml): (depth_mult, width_mult)
_PPYOLOE = {"ppyoloe_s_arch_params": (0.33, 0.50), "ppyoloe_m_arch_params": (0.67, 0.75), "ppyoloe_l_arch_params": (1.0, 1.0),
            "ppyoloe_x_arch_params": (1.33, 1.25)}


def _ppyoloe(mults):
    depth, width = mults
    return dict(depth_mult=depth, width_mult=width, num_classes=80,
                backbone=dict(layers=[3, 6, 6, 3], channels=[64, 128, 256, 512, 1024], activation="silu", return_idx=[1, 2, 3], use_large_stem=True,
                              use_alpha=False, pretrained_weights=None),
                neck=dict(in_channels=[256, 512, 1024], out_channels=[768, 384, 192], activation="silu", block_num=3, stage_num=1, spp=True),
                head=dict(in_channels=[768, 384, 192], activation="silu", fpn_strides=[32, 16, 8], grid_cell_scale=5.0, grid_cell_offset=0.5, reg_max=16,
                          eval_size=None))


def get_arch_params(config_name: str, overriding_params: dict = None, recipes_dir_path=None) -> dict:
    from ..utils.utils import recursive_override

    if config_name in _PPYOLOE:
        cfg = _ppyoloe(_PPYOLOE[config_name])
    elif config_name in _YOLO_NAS:
        cfg = _yolo_nas(_YOLO_NAS[config_name])
    else:
        raise ValueError(f"unknown arch params '{config_name}' (available: {sorted(list(_YOLO_NAS) + list(_PPYOLOE))})")
    if overriding_params:
        cfg = copy.deepcopy(cfg)
        recursive_override(cfg, dict(overriding_params))
    return cfg
