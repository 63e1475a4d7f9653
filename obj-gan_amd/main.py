"""Command-line entry of the image generator: same arguments and cfg wiring as the reference
(reference image_generation/main.py:26-140), data-parallel by processes instead of `--gpu 0,1,..`
with nn.DataParallel:

    python main.py --gpu 0 --FLAG --data_dir ../data/coco --BATCH_SIZE 16            # one MI355X
    python -m torch.distributed.run --nproc-per-node 8 main.py --gpu 0,1,2,3,4,5,6,7 --FLAG ...

Under a launcher (WORLD_SIZE in the environment) every rank takes GPU LOCAL_RANK, joins the RCCL
process group and reads its own shard of every epoch (trainDataset.build_loader); `--BATCH_SIZE` is the
per-GPU batch.  Without `--FLAG` main() stops with a pointer to evaluator.condGANEvaluator.sampling(): the
reference's test-set loader and its FID / R-precision bookkeeping are outside the hot path.
"""
from __future__ import print_function

import argparse
import datetime
import os
import pprint
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

dir_path = os.path.abspath(os.path.dirname(os.path.realpath(__file__)))
if dir_path not in sys.path:
    sys.path.append(dir_path)

from miscc.config import cfg       # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Train an image generation network')
    parser.add_argument('--gpu', dest='gpu_ids', type=str, default='-1')
    parser.add_argument('--data_dir', dest='data_dir', type=str, default='../data/coco')
    parser.add_argument('--manualSeed', type=int, help='manual seed')
    parser.add_argument('--output_dir', type=str, default='..')
    parser.add_argument('--MAX_EPOCH', type=int, default=60)
    parser.add_argument('--WORKERS', type=int, default=0)
    parser.add_argument('--NET_G', type=str, default='')
    parser.add_argument('--SAMPLE_VAL', dest='SAMPLE_VAL', action='store_true')
    parser.add_argument('--PRINT_INTERVAL', type=int, default=100)
    parser.add_argument('--DISPLAY_INTERVAL', type=int, default=500)
    parser.add_argument('--DISCRIMINATOR_LR', type=float, default=0.0002)
    parser.add_argument('--GENERATOR_LR', type=float, default=0.0002)
    parser.add_argument('--DAMSM_LAMBDA', type=float, default=100.0)
    parser.add_argument('--TXT_LAMBDA', type=float, default=0.1)
    parser.add_argument('--SHP_LAMBDA', type=float, default=1.0)
    parser.add_argument('--OBJ_LAMBDA', type=float, default=0.1)
    parser.add_argument('--UNCOND_LAMBDA', type=float, default=1.0)
    parser.add_argument('--GLB_R_NUM', type=int, default=7)
    parser.add_argument('--LAYER_D_NUM', type=int, default=4)
    parser.add_argument('--BATCH_SIZE', type=int, default=24)
    parser.add_argument('--BRANCH_NUM', type=int, default=3)
    parser.add_argument('--FLAG', dest='FLAG', action='store_true')
    # beyond the reference surface (all off by default = the reference's host-side item tuple):
    parser.add_argument('--device_hmaps', action='store_true',
                        help='hand over the box masks only and rebuild the layout maps on the device')
    parser.add_argument('--device_imgs', action='store_true',
                        help='hand over the decoded 8-bit images and resize them on the device (Pillow-exact)')
    parser.add_argument('--device_jpeg', action='store_true',
                        help='hand over the JPEG FILES and decode + resize them on the device (Pillow-exact); implies '
                             '--device_imgs')
    parser.add_argument('--device_masks', action='store_true',
                        help='hand over the raw 64 x 64 instance masks and resize them on the device (scipy-exact); '
                             'implies --device_hmaps')
    return parser.parse_args(argv)


def apply_args(args):
    """args -> cfg, exactly the assignments of reference main.py:58-88."""
    if args.data_dir != '':
        cfg.DATA_DIR = args.data_dir
    cfg.TRAIN.NET_G = args.NET_G
    cfg.TRAIN.NET_E = cfg.DATA_DIR + cfg.TRAIN.NET_E
    cfg.TEST.NET_SHP_G = cfg.DATA_DIR + cfg.TEST.NET_SHP_G
    cfg.TRAIN.MAX_EPOCH = args.MAX_EPOCH
    cfg.WORKERS = args.WORKERS
    cfg.TEST.SAMPLE_VAL = args.SAMPLE_VAL
    cfg.TRAIN.PRINT_INTERVAL = args.PRINT_INTERVAL
    cfg.TRAIN.DISPLAY_INTERVAL = args.DISPLAY_INTERVAL
    cfg.TRAIN.DISCRIMINATOR_LR = args.DISCRIMINATOR_LR
    cfg.TRAIN.GENERATOR_LR = args.GENERATOR_LR
    cfg.TRAIN.SMOOTH.DAMSM_LAMBDA = args.DAMSM_LAMBDA
    cfg.TRAIN.SMOOTH.TXT_LAMBDA = args.TXT_LAMBDA
    cfg.TRAIN.SMOOTH.SHP_LAMBDA = args.SHP_LAMBDA
    cfg.TRAIN.SMOOTH.OBJ_LAMBDA = args.OBJ_LAMBDA
    cfg.TRAIN.SMOOTH.UNCOND_LAMBDA = args.UNCOND_LAMBDA
    cfg.GAN.GLB_R_NUM = args.GLB_R_NUM
    cfg.GAN.LAYER_D_NUM = args.LAYER_D_NUM
    cfg.TRAIN.BATCH_SIZE = args.BATCH_SIZE
    cfg.TREE.BRANCH_NUM = args.BRANCH_NUM
    cfg.TRAIN.FLAG = args.FLAG
    if args.gpu_ids != '-1':
        cfg.GPU_IDS = [int(gpu_id) for gpu_id in args.gpu_ids.split(',')]
    else:
        cfg.CUDA = False
    return cfg


def agree_on_seed(seed, world=1, device=None):
    """One seed for all ranks: the value rank 0 holds (a seed drawn independently per process would give every rank
    its own DistributedSampler permutation -- overlapping / missing samples in every epoch)."""
    if world <= 1 or not dist.is_initialized():
        return int(seed)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    return int(t.item())


def seed_everything(args, rank=0, world=1, device=None):
    if not cfg.TRAIN.FLAG:
        args.manualSeed = 100
    elif args.manualSeed is None:
        args.manualSeed = random.randint(1, 10000)
    args.manualSeed = agree_on_seed(args.manualSeed, world, device)
    # data order / caption sampling / permute_seg differ per rank, the weights do not (rank 0's are broadcast)
    random.seed(args.manualSeed + rank)
    np.random.seed(args.manualSeed + rank)
    torch.manual_seed(args.manualSeed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.manualSeed + rank)
    return args.manualSeed


def init_distributed():
    """-> (rank, world, device).  One process per GPU when launched by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not cfg.CUDA or not torch.cuda.is_available():
        raise SystemExit("the MI355X kernels have no CPU path: run with --gpu <ids> on a GPU box")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)
    return rank, world, device


def build_training(args, rank, world, device):
    """-> (dataset, dataloader, trainer) for the training split (reference main.py:113-124)."""
    from trainDataset import TrainDataset, build_loader
    from trainer import condGANTrainer
    now = datetime.datetime.now()
    timestamp = now.strftime('%Y_%m_%d_%H_%M_%S')
    output_dir = '{0}/output_image_generation/{1}_{2}'.format(args.output_dir, cfg.DATASET_NAME, timestamp)
    dataset = TrainDataset(cfg.DATA_DIR, 'train', base_size=cfg.TREE.BASE_SIZE,
                           device_hmaps=getattr(args, "device_hmaps", False),
                           device_imgs=getattr(args, "device_imgs", False),
                           device_masks=getattr(args, "device_masks", False),
                           device_jpeg=getattr(args, "device_jpeg", False))
    assert dataset
    dataloader = build_loader(dataset, cfg.TRAIN.BATCH_SIZE, workers=int(cfg.WORKERS), rank=rank,
                              world=world, seed=args.manualSeed or 0, shuffle=True)
    algo = condGANTrainer(output_dir if rank == 0 else '', dataloader, dataset, device=device)
    return dataset, dataloader, algo


def main(argv=None):
    args = parse_args(argv)
    apply_args(args)
    rank, world, device = init_distributed()
    seed_everything(args, rank, world, device)
    if rank == 0:
        print('Using config:')
        pprint.pprint(cfg)
    start_t = time.time()
    if cfg.TRAIN.FLAG:
        _, _, algo = build_training(args, rank, world, device)
        algo.train()
        split_dir = 'train'
    else:
        raise SystemExit("the evaluation data pipeline (testDataset, FID / R-precision) is outside the hot path: "
                         "use evaluator.condGANEvaluator.sampling() on prepared tensors")
    if rank == 0:
        print('Total time for {0}:'.format(split_dir), time.time() - start_t)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
