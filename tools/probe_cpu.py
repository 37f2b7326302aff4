import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try:
        print(p, open(p).read().strip())
    except Exception as e:
        print(p, 'n/a')
print('torch threads', torch.get_num_threads())
x = torch.randn(1, 128, 76, 76); w = torch.randn(128, 128, 3, 3)
for nt in (1, 4, 8, 16, 32, 64):
    torch.set_num_threads(nt)
    torch.nn.functional.conv2d(x, w, padding=1)
    t0 = time.perf_counter()
    for _ in range(5):
        torch.nn.functional.conv2d(x, w, padding=1)
    print(nt, 'threads: conv ms', (time.perf_counter() - t0) / 5 * 1e3)
