#!/usr/bin/env python
"""Train-step throughput of the fp32 training path (fwd + loss + bwd + SGD step) on synthetic
COCO-shaped data; one process per GPU, torch DDP over RCCL when WORLD_SIZE > 1.
    python tools/train_bench.py --batch 16 --size 608 --steps 5
Not the headline metric (bench.py is); documents where the training row stands."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import dist as D


def synthetic_gts(batch, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    boxes, labels = [], []
    for _ in range(batch):
        n = max(1, int(torch.poisson(torch.tensor(12.0), generator=g)))
        c = torch.rand(n, 2, generator=g) * size
        wh = torch.exp(torch.rand(n, 2, generator=g) * (torch.log(torch.tensor(400.0)) - torch.log(torch.tensor(8.0))) +
                       torch.log(torch.tensor(8.0)))
        b = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, size)
        boxes.append(b.to(device))
        labels.append(torch.randint(0, 80, (n,), generator=g).to(device))
    return boxes, labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--model', default='yolov4l')
    a = ap.parse_args()
    rank, local_rank, world = D.env_world()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    D.init('nccl', dev)
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(a.model))
    det.init_weights()
    det.train().to(dev)
    model = det
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(det, device_ids=[local_rank], broadcast_buffers=False)
    opt = torch.optim.SGD(det.parameters(), lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
    img = bench.synthetic_images(a.batch, a.size, 1000 + rank, dev)
    gtb, gtl = synthetic_gts(a.batch, a.size, 2000 + rank, dev)
    metas = [dict() for _ in range(a.batch)]

    def step():
        opt.zero_grad(set_to_none=True)
        losses = model(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl)
        loss, log_vars = det._parse_losses(losses)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(det.parameters(), 35)
        opt.step()
        return log_vars['loss']

    for _ in range(a.warmup):
        l0 = step()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        l1 = step()
    D.barrier()
    el = D.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        fl = 3 * 108.516e9 * (a.size / 608.0) ** 2 if a.model == 'yolov4l' else float('nan')
        print(json.dumps(dict(metric='images/sec (train step) ' + a.model, value=round(a.batch * world * a.steps / el, 2),
                              n_gpus=world, ms_per_step=round(el / a.steps * 1e3, 1), batch_per_gpu=a.batch,
                              dtype='f32', loss_first=round(l0, 3), loss_last=round(l1, 3),
                              approx_conv_tflops=round(fl * a.batch * a.steps / el / 1e12, 1),
                              peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))))
    D.finalize()


if __name__ == '__main__':
    main()
