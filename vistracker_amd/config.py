"""Experiment configuration: the reference's JSON files load unchanged (config/config_loader.py:24-44) and the CLI of the
fit scripts keeps its flags (recon/recon_fit_triplane.py:241-300, recon/recon_fit_trivis_full.py:459-474)."""
from __future__ import annotations

import json
from argparse import ArgumentParser, Namespace
from collections import OrderedDict
from os.path import join


def load_configs(exp_name: str, configs_dir: str = "config") -> Namespace:
    """JSON with ``//`` comments stripped -> Namespace; keys are tested with ``'key' in args`` as in the reference."""
    args = Namespace()
    with open(join(configs_dir, exp_name + ".json")) as f:
        txt = "".join(line.split("//")[0] + "\n" for line in f)
    args.__dict__ = json.loads(txt, object_pairs_hook=OrderedDict)
    if "camera_params" in args and "loadSize" in args:
        assert args.camera_params["crop_size"] == args.loadSize, "please check camera params and crop size!"
    return args


def get_parser() -> ArgumentParser:
    p = ArgumentParser()
    p.add_argument("exp_name", help="experiment name")
    p.add_argument("-s", "--seq_folder", help="path to one BEHAVE sequence")
    p.add_argument("-sn", "--save_name", required=True, help="recon result save name")
    p.add_argument("-o", "--outpath", default=None, help="where to save reconstruction results")
    p.add_argument("-ck", "--checkpoint", default=None)
    p.add_argument("-fv", "--filter_val", type=float, default=0.004)
    p.add_argument("-st", "--sparse_thres", type=float, default=0.03)
    p.add_argument("-t", "--tid", default=1, type=int)
    p.add_argument("-bs", "--batch_size", default=96, type=int, help="optimization batch size")
    p.add_argument("-redo", default=False, action="store_true")
    p.add_argument("-d", "--display", default=False, action="store_true")
    p.add_argument("-fs", "--start", default=0, type=int)
    p.add_argument("-fe", "--end", default=None, type=int)
    p.add_argument("-tt", "--triplane_type", default="smooth", choices=["gt", "mocap", "temporal", "smooth"])
    p.add_argument("-pat", default="t*")
    p.add_argument("-neural_only", default=False, action="store_true")
    p.add_argument("-pred_occ", default=True, action="store_true")
    p.add_argument("-sr", "--smpl_recon_name", required=True)
    p.add_argument("-or", "--obj_recon_name", required=True)
    return p


def merge_configs(args: Namespace, configs: Namespace) -> Namespace:
    for k_cfg, k_arg in (("batch_size", "batch_size"), ("test_kid", "tid"), ("filter_val", "filter_val"), ("sparse_thres", "sparse_thres"),
                         ("seq_folder", "seq_folder"), ("pat", "pat"), ("save_name", "save_name"), ("checkpoint", "checkpoint"),
                         ("outpath", "outpath"), ("redo", "redo"), ("display", "display"), ("start", "start"), ("end", "end"),
                         ("neural_only", "neural_only"), ("pred_occ", "pred_occ"), ("triplane_type", "triplane_type"),
                         ("smpl_recon_name", "smpl_recon_name"), ("obj_recon_name", "obj_recon_name")):
        setattr(configs, k_cfg, getattr(args, k_arg))
    return configs
