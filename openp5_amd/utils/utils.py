"""Flags, seeding, logging, checkpoint paths -- flag names and defaults follow
/root/reference/src/src_t5/utils/utils.py:12-129 so the reference's command lines keep working."""
import logging
import os
import random
import sys

import numpy as np
import torch


def parse_global_args(parser):
    parser.add_argument("--seed", type=int, default=2023, help="Random seed")
    parser.add_argument("--model_dir", type=str, default="../model", help="The model directory")
    parser.add_argument("--checkpoint_dir", type=str, default="../checkpoint", help="The checkpoint directory")
    parser.add_argument("--model_name", type=str, default="model.pt", help="The model name")
    parser.add_argument("--log_dir", type=str, default="../log", help="The log directory")
    parser.add_argument("--distributed", type=int, default=1, help="use distributed data parallel or not.")
    parser.add_argument("--gpu", type=str, default="0,1,2,3", help="gpu ids, if not distributed, only use the first one.")
    parser.add_argument("--master_addr", type=str, default="127.0.0.1", help="Setup MASTER_ADDR for os.environ")
    parser.add_argument("--master_port", type=str, default="12345", help="Setup MASTER_PORT for os.environ")
    parser.add_argument("--logging_level", type=int, default=logging.INFO, help="Logging Level, 0, 10, ..., 50")
    return parser


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def ReadLineFromFile(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path, "r") as fd:
        return [line.rstrip("\n") for line in fd]


def WriteDictToFile(path, write_dict):
    with open(path, "w") as out:
        for key, val in write_dict.items():
            out.write(key + " " + (" ".join(val) if isinstance(val, list) else str(val)) + "\n")


def _folder(args):
    return "SP5" if len(args.datasets.split(",")) > 1 else args.datasets


def log_name(args):
    parts = [args.distributed, args.sample_prompt, args.his_prefix, args.skip_empty_his, args.max_his, args.master_port, _folder(args),
             args.tasks, args.backbone, args.item_indexing, args.lr, args.epochs, args.batch_size, args.sample_num,
             os.path.splitext(os.path.basename(args.prompt_file))[0]]
    return "_".join(str(p) for p in parts)


def setup_logging(args):
    args.log_name = log_name(args)
    folder = os.path.join(args.log_dir, _folder(args))
    os.makedirs(folder, exist_ok=True)
    for handler in logging.root.handlers[:]:
        logging.root.removeHandler(handler)
    logging.basicConfig(filename=os.path.join(folder, args.log_name + ".log"), level=args.logging_level,
                        format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))


def setup_model_path(args):
    if args.model_name == "model.pt":
        model_path = os.path.join(args.model_dir, _folder(args))
        os.makedirs(model_path, exist_ok=True)
        args.model_path = os.path.join(model_path, args.log_name + ".pt")
    else:
        args.model_path = os.path.join(args.checkpoint_dir, args.model_name)


def save_model(model, path):
    torch.save(model.state_dict(), path)


def load_model(model, path, args=None, loc=None):
    state_dict = torch.load(path, map_location=loc or "cpu")
    model.load_state_dict(state_dict, strict=False)
    return model
