"""Reading the reference's experiment yamls (config/train_*.yaml over config/defaults.py) without yacs, and building
the three networks + renderer from them the way NeuconWSystem.__init__ does (lightning_modules/neuconw_system.py:60-146).
Used by scripts/train.py (the PL-free training driver)."""
import ast
import copy
import os

import yaml

# config/defaults.py, restated as plain data (the values the scene yamls do not override)
DEFAULTS = {
    "NEUCONW": {
        "N_SAMPLES": 512, "N_IMPORTANCE": 512, "USE_DISP": False, "PERTURB": 1.0, "NOISE_STD": 1.0, "S_VAL_BASE": 0,
        "BOUNDARY_SAMPLES": 0, "NEAR_FAR_OVERRIDE": False, "VOXEL_SIZE": 0.0, "MIN_TRACK_LENGTH": 0, "SAMPLE_RANGE": 4,
        "SDF_THRESHOLD": 1e-3, "TRAIN_VOXEL_SIZE": 0.01, "UPDATE_FREQ": 2000, "N_VOCAB": 1500, "ENCODE_A": True, "N_A": 48,
        "N_STATIC_HEAD": 1, "ANNEAL_END": 50000, "RENDER_BG": True, "UP_SAMPLE_STEP": 4, "N_OUTSIDE": 32,
        "MESH_MASK_LIST": None, "RAY_MASK_LIST": None, "ENCODE_A_BG": True, "FLOOR_NORMAL": False, "FLOOR_LABELS": ["road"],
        "DEPTH_LOSS": False,
        "SDF_CONFIG": {"d_in": 3, "d_out": 513, "d_hidden": 512, "n_layers": 8, "skip_in": (4,), "multires": 6, "bias": 0.5,
                       "scale": 1, "geometric_init": True, "weight_norm": True, "inside_outside": False},
        "COLOR_CONFIG": {"d_in": 9, "d_feature": 512, "mode": "idr", "d_out": 3, "d_hidden": 256, "n_layers": 4,
                         "head_channels": 128, "static_head_layers": 2, "weight_norm": True, "multires_view": 4},
        "S_CONFIG": {"init_val": 0.03},
        "LOSS": {"coef": 1.0, "igr_weight": 0.1, "mask_weight": 0.1, "depth_weight": 0.1, "floor_weight": 0.01},
    },
    "DATASET": {"ROOT_DIR": None, "DATASET_NAME": None, "SPLIT": "train",
                "PHOTOTOURISM": {"IMG_DOWNSCALE": 1, "USE_CACHE": True, "CACHE_DIR": "cache", "CACHE_TYPE": "npz",
                                 "SEMANTIC_MAP_PATH": "semantic_maps", "WITH_SEMANTICS": True}},
    "TRAINER": {"WORLD_SIZE": 1, "CANONICAL_BS": 2048, "CANONICAL_LR": 1e-3, "SCALING": None, "SAVE_DIR": "checkpoints",
                "VAL_FREQ": 0.125, "SAVE_FREQ": 5000, "OPTIMIZER": "adam", "LR": None, "WEIGHT_DECAY": 0,
                "LR_SCHEDULER": "cosine", "DECAY_STEP": [], "DECAY_GAMMA": 0.1, "SEED": 66},
}


def _decode(v):
    """yacs' value decoding: strings go through literal_eval ("(4,)" -> (4,), "1e-4" -> 1e-4)."""
    if isinstance(v, dict):
        return {k: _decode(x) for k, x in v.items()}
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def load_config(yaml_path=None, overrides=None):
    cfg = copy.deepcopy(DEFAULTS)
    if yaml_path:
        with open(yaml_path, "r") as f:
            _merge(cfg, _decode(yaml.safe_load(f)))
    if overrides:
        _merge(cfg, overrides)
    return cfg


def scale_lr(cfg, world_size, batch_size):
    """train.py:21-25."""
    t = cfg["TRAINER"]
    t["WORLD_SIZE"] = world_size
    t["TRUE_BATCH_SIZE"] = world_size * batch_size
    t["SCALING"] = t["TRUE_BATCH_SIZE"] / t["CANONICAL_BS"]
    t["LR"] = t["CANONICAL_LR"] * t["SCALING"]
    return t["LR"]


def lr_at_epoch(cfg, base_lr, epoch, num_epochs):
    """The learning rate the reference's scheduler (utils/__init__.py:45-61 `get_scheduler`, stepped once per epoch by
    PyTorch-Lightning) holds DURING epoch `epoch` (0-based):
      'none'   -> base_lr (every shipped scene yaml);
      'cosine' -> CosineAnnealingLR(T_max=num_epochs, eta_min=1e-8) in closed form;
      'steplr' -> MultiStepLR(milestones=DECAY_STEP, gamma=DECAY_GAMMA);
      'poly'   -> upstream raises NameError (LambdaLR is never imported, POLY_EXP is not in config/defaults.py)."""
    import math

    t = cfg["TRAINER"]
    kind = t.get("LR_SCHEDULER", "none")
    if kind in (None, "none"):
        return base_lr
    if kind == "cosine":
        eta_min = 1e-8
        return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / max(1, num_epochs))) / 2.0
    if kind == "steplr":
        return base_lr * float(t.get("DECAY_GAMMA", 0.1)) ** sum(1 for m in (t.get("DECAY_STEP") or []) if epoch >= m)
    raise NotImplementedError("TRAINER.LR_SCHEDULER = %r (the reference itself fails on 'poly': LambdaLR is not imported)" % (kind,))


def build_system(cfg, device, prec=None):
    """neuconw_system.py:60-146: embedding, NeuconW, background NeRF and the renderer from the experiment config and the
    scene's config.yaml.  Returns (embedding_a, neuconw, nerf, renderer, scene_config)."""
    import torch

    from . import NeRF, NeuconW, NeuconWRenderer

    n = cfg["NEUCONW"]
    root = cfg["DATASET"]["ROOT_DIR"]
    with open(os.path.join(root, "config.yaml"), "r") as f:
        scene = yaml.load(f, Loader=yaml.FullLoader)
    emb = torch.nn.Embedding(n["N_VOCAB"], n["N_A"])
    neuconw = NeuconW(sdfNet_config=dict(n["SDF_CONFIG"]), colorNet_config=dict(n["COLOR_CONFIG"]),
                      SNet_config=dict(n["S_CONFIG"]), in_channels_a=n["N_A"], encode_a=n["ENCODE_A"])
    nerf = NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                encode_appearance=n["ENCODE_A_BG"], in_channels_a=n["N_A"],
                in_channels_dir=6 * n["COLOR_CONFIG"]["multires_view"] + 3, use_viewdirs=True)
    emb, neuconw, nerf = emb.to(device), neuconw.to(device), nerf.to(device)
    spc = {"voxel_size": scene["voxel_size"], "recontruct_path": root, "min_track_length": scene["min_track_length"]}
    rdr = NeuconWRenderer(nerf=nerf, neuconw=neuconw, embeddings={"a": emb}, n_samples=n["N_SAMPLES"],
                          s_val_base=n["S_VAL_BASE"], n_importance=n["N_IMPORTANCE"], n_outside=n["N_OUTSIDE"],
                          up_sample_steps=n["UP_SAMPLE_STEP"], perturb=1.0, origin=scene["origin"], radius=scene["radius"],
                          render_bg=n["RENDER_BG"], mesh_mask_list=n["MESH_MASK_LIST"], floor_normal=n["FLOOR_NORMAL"],
                          floor_labels=n["FLOOR_LABELS"], depth_loss=n["DEPTH_LOSS"], spc_options=spc,
                          sample_range=n["SAMPLE_RANGE"], boundary_samples=n["BOUNDARY_SAMPLES"],
                          nerf_far_override=n["NEAR_FAR_OVERRIDE"], prec=prec)
    return emb, neuconw, nerf, rdr, scene


def surface_level(voxel_size, bbx):
    """neuconw_system.py:314-335: octree level of the refresh for a voxel size in world coordinates."""
    import numpy as np

    lo, hi = np.array(bbx[0]), np.array(bbx[1])
    return int(np.ceil(np.log2(2 * (np.max(hi - lo) / 2) / voxel_size)))


def neuconw_loss(cfg):
    """losses.py:21-43 (NeuconWLoss) with the experiment's weights and switches -> loss_fn(outputs, rgbs) (one launch each
    way: neuralrecon_w_amd.losses.NeuconWLoss)."""
    from .losses import NeuconWLoss

    w = cfg["NEUCONW"]["LOSS"]
    return NeuconWLoss(coef=w["coef"], igr_weight=w["igr_weight"], mask_weight=w["mask_weight"], depth_weight=w["depth_weight"],
                       floor_weight=w["floor_weight"], config=cfg)
