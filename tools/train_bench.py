#!/usr/bin/env python
"""Train-step throughput of the training path (fwd + fused loss + bwd + gradient exchange + SGD-Nesterov + EMA through
the recipe hooks) on synthetic COCO-shaped data; one process per GPU, ``dist.GradReducer`` over RCCL when WORLD_SIZE > 1
(``--torch-optim`` switches to torch.optim.SGD + torch DDP for comparison).
    python tools/train_bench.py --batch 16 --size 608 --steps 5
Not the headline metric (bench.py is); documents where the training row stands."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import dist as D


synthetic_gts = bench.synthetic_gts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--model', default='yolov4l')
    ap.add_argument('--accumulation', type=int, default=1)
    ap.add_argument('--torch-optim', action='store_true')
    ap.add_argument('--grad-exchange', default=None, choices=['allreduce', 'direct', 'direct_bf16'],
                    help='dist.GradReducer mode (default: YV4_GRAD_EXCHANGE or allreduce)')
    ap.add_argument('--overlap-report', action='store_true',
                    help='record when each gradient bucket becomes exchangeable inside backward (GradReducer timing events) and '
                         'print the share of backward its exchange can overlap; works with one rank (nothing is exchanged)')
    ap.add_argument('--phase-report', action='store_true',
                    help='device time (events) and host enqueue time of forward+loss / log-variable readback / backward+update '
                         'per step: where the device waits for the host')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'],
                    help='activation / conv operand type (master weights, statistics and losses stay fp32)')
    a = ap.parse_args()
    if a.overlap_report:
        os.environ['YV4_REDUCER_AT_WORLD1'] = '1'
    rank, local_rank, world = D.env_world()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    D.init('nccl', dev)
    torch.manual_seed(0)
    det = pkg.build_detector(bench.model_cfg(a.model))
    det.init_weights()
    det.train().to(dev)
    if a.dtype != 'f32':
        pkg.wrap_fp16_model(det, torch.float16 if a.dtype == 'f16' else torch.bfloat16)
    # the recipe of configs/yolov4/yolov4l_coco_mosaic.py:108-147: SGD-Nesterov with one group per
    # parameter, grad clip 35, dynamic loss scale, per-parameter warm-up, EMA of the whole state --
    # all through the flat arenas; --torch-optim switches to torch.optim.SGD + DDP for comparison
    metas = [dict() for _ in range(a.batch)]
    img = bench.synthetic_images(a.batch, a.size, 1000 + rank, dev)
    gtb, gtl = synthetic_gts(a.batch, a.size, 2000 + rank, dev)
    data = dict(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl)
    if a.torch_optim:
        model = det
        if world > 1:
            model = torch.nn.parallel.DistributedDataParallel(det, device_ids=[local_rank], broadcast_buffers=False)
        opt = torch.optim.SGD(det.parameters(), lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            losses = model(**data)
            loss, log_vars = det._parse_losses(losses)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(det.parameters(), 35)
            opt.step()
            return log_vars['loss']
    else:
        from mmdet_yolov4_amd import hooks as H
        from mmdet_yolov4_amd.optim import build_optimizer
        opt = build_optimizer(det, dict(type='SGD', lr=0.01, momentum=0.937, weight_decay=0.0005, nesterov=True,
                                        paramwise_cfg=dict(bias_decay_mult=0., norm_decay_mult=0.)))
        runner = H.Runner(det, opt, max_epochs=1)
        runner.log_buffer = None                                      # no per-step grad-norm readback
        runner.register_hook_from_cfg(dict(type='DetailedLinearWarmUpHook', warmup_iters=10000, priority='NORMAL'))
        runner.register_hook_from_cfg(dict(type='StateEMAHook', momentum=0.9999, interval=1, warm_up=10000,
                                           priority='HIGH'))
        runner.register_hook(H.Fp16GradAccumulateOptimizerHook(accumulation=a.accumulation,
                                                               grad_clip=dict(max_norm=35, norm_type=2),
                                                               loss_scale='dynamic', grad_exchange=a.grad_exchange),
                             'ABOVE_NORMAL')
        runner.data_loader = H.BatchSource([data], a.batch)
        runner.call_hook('before_run')
        runner.call_hook('before_train_epoch')
        last = {}

        phases = []

        def step_phases():
            # the same calls as step() below with train_step opened up (single_stage.train_step = forward_train +
            # _parse_losses; forward_train = extract_feat + bbox_head.forward_train)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            h = [time.perf_counter()]
            ev[0].record()
            runner.call_hook('before_train_iter')
            x = det.extract_feat(data['img'])
            ev[1].record(); h.append(time.perf_counter())
            losses = det.bbox_head.forward_train(x, data['img_metas'], data['gt_bboxes'], data['gt_labels'], None)
            ev[2].record(); h.append(time.perf_counter())
            loss, log_vars = det._parse_losses(losses)
            ev[3].record(); h.append(time.perf_counter())
            loss.register_hook(lambda g: (ev[4].record(), h.append(time.perf_counter()), None)[2])
            runner.outputs = dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))
            runner.call_hook('after_train_iter')
            ev[5].record(); h.append(time.perf_counter())
            runner.iter += 1
            phases.append((ev, h))
            last['loss'] = log_vars['loss']
            return last['loss']

        def step():
            # exactly what the runner does per iteration (mmcv epoch_based_runner.run_iter): hooks around
            # ``model.train_step`` -- including its log-variable exchange (one all-reduce, one D2H copy = one host
            # synchronisation per step, single_stage._parse_losses)
            runner.call_hook('before_train_iter')
            runner.outputs = det.train_step(data, opt)
            runner.call_hook('after_train_iter')
            runner.iter += 1
            last['loss'] = runner.outputs['log_vars']['loss']
            return last['loss']

    if a.phase_report and not a.torch_optim:
        step = step_phases
    for _ in range(a.warmup):
        l0 = step()
    red = None
    if a.overlap_report and not a.torch_optim:
        red = next((h.reducer for h in runner._hooks if getattr(h, 'reducer', None) is not None), None)
        if red is not None:
            red.enable_timing()
    D.barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(a.steps):
        ts = time.perf_counter()
        # step() returns the logged loss: the host waits there for the log variables' device-to-host copy, which is queued
        # BETWEEN the forward and the backward pass -- i.e. the host is released when the device has finished step i's forward,
        # with step i's backward + update still queued.  step_ms is therefore host wall time from one such point to the next: it
        # equals the device's step time in the steady state (53-55 ms), can alternate short / long while the queue depth settles
        # (rounds 4-5 printed 23 / 90 ms pairs summing to two steps), and says nothing a barrier-closed ms_per_step does not.
        l1 = step()
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 1))
    D.barrier()
    el = D.max_over_ranks(time.perf_counter() - t0, dev)
    phase_rep = None
    if a.phase_report and not a.torch_optim:
        torch.cuda.synchronize()
        names = ['backbone+neck+head forward', 'loss forward', 'log_vars (sync)', 'hooks before backward', 'backward+update']
        dev_ms = [0.0] * 5
        host_ms = [0.0] * 5
        tail = phases[-a.steps:]
        for ev, h in tail:
            for i in range(5):
                dev_ms[i] += ev[i].elapsed_time(ev[i + 1]) / len(tail)
                host_ms[i] += (h[i + 1] - h[i]) * 1e3 / len(tail)
        phase_rep = {n: dict(device_ms=round(d, 2), host_ms=round(hh, 2)) for n, d, hh in zip(names, dev_ms, host_ms)}
    if rank == 0:
        per_img = {'yolov4l': 108.516e9 / 608.0 ** 2, 'yolov5l': 108.574e9 / 640.0 ** 2, 'yolov4s': 8.942e9 / 416.0 ** 2,
                   'yolov3': 140.692e9 / 608.0 ** 2}    # 75 convs, Plan.total_flops() (= darknet's 140.69 BFLOPs)
        fl = 3 * per_img[a.model] * a.size ** 2      # fwd + dgrad + wgrad conv FLOPs
        print(json.dumps(dict(metric='images/sec (train step) ' + a.model, value=round(a.batch * world * a.steps / el, 2),
                              n_gpus=world, ms_per_step=round(el / a.steps * 1e3, 1), step_ms=step_ms, batch_per_gpu=a.batch,
                              dtype=a.dtype, loss_first=round(float(l0), 3), loss_last=round(float(l1), 3),
                              optimizer='torch SGD + DDP' if a.torch_optim else 'flat arenas + recipe hooks',
                              backend=D.backend_name(), grad_exchange=D.exchange_name(a.grad_exchange, world),
                              approx_conv_tflops=round(fl * a.batch * world * a.steps / el / 1e12, 1),
                              peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                              **({'overlap': red.timing_report()} if red is not None else {}),
                              **({'phases': phase_rep} if phase_rep is not None else {}))))
    D.finalize()


if __name__ == '__main__':
    main()
