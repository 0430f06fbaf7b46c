"""Entry point - drop-in for the reference's main.py: same flags (-t -pm -lf -m -w -b -e -mlr -milr -wd -snt), same global
seeding, same wiring (preprocessor -> Generator/Discriminator -> trainer -> train()).  Run from this directory:

    python main.py -m DCGAN -b 256 -e 1 -mlr 0.0002
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py -m DCGAN -b 256   # data parallel

Note: the CLI default learning rate 0.1 is the reference's (main.py:54); it saturates the losses after one step - pass
-mlr 0.0002 for meaningful training.  `torch.autograd.set_detect_anomaly(True)` of the reference is not enabled: the native
step has no autograd graph to check."""
import argparse
import os
import random
from datetime import datetime

import numpy as np
import torch

from change_randomseed import RANDOMSEED
from enums import ModelEnum
from logger.main_logger import MainLogger


def seed_everything(seed=RANDOMSEED):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_arg_parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-t", "--test", type=int, default=0, help="test mode (unused, kept for compatibility)")
    p.add_argument("-pm", "--model_path", type=str, default="", help="model folder name")
    p.add_argument("-lf", "--log_file", type=int, default=1, help="write a log file: 0=false, 1=true")
    p.add_argument("-m", "--model", type=ModelEnum, choices=list(ModelEnum), default=ModelEnum.DCGAN, help="model to train")
    p.add_argument("-w", "--num_worker", type=int, default=0, help="DataLoader workers")
    p.add_argument("-b", "--batch_size", type=int, default=128, help="training batch size (per GPU)")
    p.add_argument("-e", "--epoch", type=int, default=100, help="epochs")
    p.add_argument("-mlr", "--max_learning_rate", type=float, default=0.1, help="Adam learning rate")
    p.add_argument("-milr", "--min_learning_rate", type=float, default=1e-4, help="unused, kept for compatibility")
    p.add_argument("-wd", "--weight_decay", type=float, default=5e-4, help="unused, kept for compatibility")
    p.add_argument("-snt", "--nesterov", type=int, default=1, help="unused, kept for compatibility")
    return p.parse_args(argv)


def main(args: argparse.Namespace):
    datetime_now = args.model_path if args.model_path != "" else datetime.now().strftime("%Y%m%d_%H%M%S")
    args.save_path = os.path.join(".", "save", str(args.model).lower(), datetime_now)
    os.makedirs(args.save_path, exist_ok=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not torch.distributed.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm

    logger = MainLogger(args)
    logger.debug(f"args: {vars(args)}")
    logger.debug("init data preprocessing")

    if args.model == ModelEnum.DCGAN:
        from model import DCGAN
        from preprocess.dcgan_data_preprocessor import DCGANDataPreprocessor
        from train.dcgan_trainer import DCGANTrainer
        data_pre = DCGANDataPreprocessor(args)
        data_pre.transform_data()
        trainer = DCGANTrainer(args, DCGAN.Generator(), DCGAN.Discriminator(), data_pre)
    else:
        from model import CGAN
        from preprocess.cgan_data_preprocessor import CGANDataPreprocessor
        from train.cgan_trainer import CGANTrainer
        data_pre = CGANDataPreprocessor(args)
        data_pre.transform_data()
        trainer = CGANTrainer(args, CGAN.Generator(), CGAN.Discriminator(), data_pre)
    trainer.train()
    if world > 1:
        torch.distributed.destroy_process_group()


seed_everything()

if __name__ == "__main__":
    main(get_arg_parse())
