"""What the numpy-in / numpy-out contract of the reference's call surface costs on this box: copies of Kodak-batch-sized arrays
between pageable / pinned host memory and HBM, and the harness's own numpy passes (reconstructing_eae_kodak.py:170-178, 192)."""
import time

import numpy
import torch


def bench(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t)//2]*1e3


def main():
    dev = torch.device('cuda', 0)
    y = numpy.random.RandomState(0).standard_normal((24, 32, 48, 128)).astype(numpy.float32)      # 18.9 MB
    mean = numpy.random.RandomState(1).standard_normal(128).astype(numpy.float32)
    x = numpy.random.RandomState(2).randint(0, 255, (24, 512, 768), dtype=numpy.uint8)             # 9.4 MB
    yd = torch.empty(y.shape, dtype=torch.float32, device=dev)
    xd = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    yp = torch.empty(y.shape, dtype=torch.float32).pin_memory()
    xp = torch.empty(x.shape, dtype=torch.uint8).pin_memory()
    yt = torch.from_numpy(y)
    xt = torch.from_numpy(x)
    rows = []
    rows.append(('H2D pageable 18.9 MB (copy_)', bench(lambda: yd.copy_(yt))))
    rows.append(('H2D pinned 18.9 MB', bench(lambda: yd.copy_(yp, non_blocking=True))))
    rows.append(('host memcpy pageable -> pinned 18.9 MB', bench(lambda: yp.copy_(yt))))
    rows.append(('D2H pageable 18.9 MB', bench(lambda: yt.copy_(yd))))
    rows.append(('D2H pinned 18.9 MB', bench(lambda: yp.copy_(yd, non_blocking=True))))
    rows.append(('D2H fresh .cpu() 18.9 MB', bench(lambda: yd.cpu())))
    rows.append(('H2D pageable 9.4 MB u8', bench(lambda: xd.copy_(xt))))
    rows.append(('H2D pinned 9.4 MB u8', bench(lambda: xd.copy_(xp, non_blocking=True))))
    rows.append(('D2H pinned 9.4 MB u8', bench(lambda: xp.copy_(xd, non_blocking=True))))
    rows.append(('H2D pageable 0.79 MB (one image latents)', bench(lambda: yd[0].copy_(yt[0]))))
    rows.append(('H2D pageable 0.39 MB u8 (one image)', bench(lambda: xd[0].copy_(xt[0]))))
    rows.append(('numpy.tile(map_mean) 18.9 MB', bench(lambda: numpy.tile(mean, (24, 32, 48, 1)))))
    tiled = numpy.tile(mean, (24, 32, 48, 1))
    rows.append(('numpy y - tiled 18.9 MB', bench(lambda: y - tiled)))
    rows.append(('numpy x.astype(float32) 4 images', bench(lambda: x[:4].astype(numpy.float32))))
    rows.append(('numpy zeros 18.9 MB + fill', bench(lambda: numpy.zeros(y.shape, dtype=numpy.float32).fill(1))))
    rows.append(('pin_memory alloc 18.9 MB', bench(lambda: torch.empty(y.shape, dtype=torch.float32).pin_memory(), n=5)))
    rows.append(('torch.cuda.synchronize alone', bench(lambda: None)))
    s = torch.cuda.Stream()
    e = torch.cuda.Event()

    def tiny():
        torch.zeros(1, device=dev).item()
    rows.append(('tiny kernel + .item()', bench(tiny)))
    for (name, ms) in rows:
        print('{0:50s} {1:9.3f} ms'.format(name, ms))


if __name__ == '__main__':
    main()
