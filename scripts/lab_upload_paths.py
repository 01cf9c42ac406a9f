#!/usr/bin/env python3
"""How fast the weights of a model file (page cache, memory-mapped) reach HBM, weight by weight, by different routes:

  mmap        torch.from_numpy(view of a fresh np.memmap).to("cuda")           what `staging.upload` does with a file's tensor
  populate    madvise(MADV_POPULATE_READ) on the view's pages first            page tables filled in one call instead of by faults
  pread N     os.preadv into a page-locked buffer on N threads, async copy     no mapping of the file at all, two buffers in flight
  memory      the same bytes from an anonymous NumPy array                     the ceiling `staging.py` quotes (56 GB/s)

and how fast results come back into FRESH NumPy arrays (`staging.download`: every result of a model is held until the file is written):

  fresh       np.empty + one blocking copy                                     what `staging.download` does
  huge        the same after madvise(MADV_HUGEPAGE) on the new array           2 MiB first-touch faults where the kernel allows them
  populated   ... and madvise(MADV_POPULATE_WRITE)                             all first-touch faults in one call, before the copy
  pinned      torch.empty(pin_memory=True) per result                          page-locked destination
  populated by N threads                                                       the first-touch faults taken by N threads at once
  reused      one destination for all                                          the ceiling

    python scripts/lab_upload_paths.py [--gib 8] [--chunk-mib 172] [--down-mib 22]
"""
import argparse
import ctypes
import json
import os
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

MADV_POPULATE_READ = 22


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0)
    ap.add_argument("--chunk-mib", type=int, default=172)          # 4096 x 11008 fp32
    ap.add_argument("--down-mib", type=int, default=22)            # the MatMulNBits blob of that weight
    a = ap.parse_args()
    chunk = a.chunk_mib << 20
    count = int(a.gib * (1 << 30)) // chunk
    libc = ctypes.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    page = os.sysconf("SC_PAGE_SIZE")
    torch.zeros(1, device="cuda")
    downloads(a, libc, page)
    with tempfile.TemporaryDirectory(prefix="oq_upload_") as d:
        path = os.path.join(d, "weights.bin")
        block = np.random.default_rng(0).standard_normal(chunk // 4, dtype=np.float32)
        t0 = time.perf_counter()
        with open(path, "wb") as f:
            for _ in range(count):
                f.write(block)
        print(json.dumps({"file_gib": round(count * chunk / 2**30, 2), "chunks": count, "write_s": round(time.perf_counter() - t0, 2)}), flush=True)
        total = count * chunk

        def report(name, seconds, **kw):
            print(json.dumps({"route": name, "seconds": round(seconds, 3), "gb_per_s": round(total / seconds / 1e9, 1), **kw}), flush=True)

        def views():
            m = np.memmap(path, dtype=np.float32, mode="r")
            return m, [m[i * (chunk // 4):(i + 1) * (chunk // 4)] for i in range(count)]

        # memory: the ceiling
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(count):
            torch.from_numpy(block).to("cuda")
        torch.cuda.synchronize()
        report("memory", time.perf_counter() - t0)

        for rep in range(2):
            m, vs = views()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for v in vs:
                torch.from_numpy(v).to("cuda")
            torch.cuda.synchronize()
            report("mmap", time.perf_counter() - t0, rep=rep)
            del m, vs

        m, vs = views()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        advise = 0.0
        for v in vs:
            addr = v.ctypes.data
            lo = addr - addr % page
            t1 = time.perf_counter()
            rc = libc.madvise(lo, addr + v.nbytes - lo, MADV_POPULATE_READ)
            advise += time.perf_counter() - t1
            torch.from_numpy(v).to("cuda")
        torch.cuda.synchronize()
        report("populate", time.perf_counter() - t0, madvise_s=round(advise, 3), rc=rc, errno=ctypes.get_errno())
        del m, vs

        # populate on a helper thread, one weight ahead
        m, vs = views()
        pool = ThreadPoolExecutor(1)

        def populate(v):
            addr = v.ctypes.data
            lo = addr - addr % page
            return libc.madvise(lo, addr + v.nbytes - lo, MADV_POPULATE_READ)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ahead = pool.submit(populate, vs[0])
        for i, v in enumerate(vs):
            ahead.result()
            if i + 1 < count:
                ahead = pool.submit(populate, vs[i + 1])
            torch.from_numpy(v).to("cuda")
        torch.cuda.synchronize()
        report("populate one ahead", time.perf_counter() - t0)
        pool.shutdown()
        del m, vs

        for threads in (1, 4, 8, 16):
            fd = os.open(path, os.O_RDONLY)
            pinned = [torch.empty(chunk, dtype=torch.uint8).pin_memory() for _ in range(2)]
            hosts = [p.numpy() for p in pinned]
            events = [torch.cuda.Event() for _ in range(2)]
            pool = ThreadPoolExecutor(threads)
            piece = -(-chunk // threads)
            piece += -piece % page

            def fill(buf, base):
                def one(j):
                    lo = j * piece
                    hi = min(chunk, lo + piece)
                    got = 0
                    while lo + got < hi:
                        got += os.preadv(fd, [memoryview(buf)[lo + got:hi]], base + lo + got)
                list(pool.map(one, range(threads)))

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            read = 0.0
            used = [False, False]
            for i in range(count):
                b = i % 2
                if used[b]:
                    events[b].synchronize()
                t1 = time.perf_counter()
                fill(hosts[b], i * chunk)
                read += time.perf_counter() - t1
                dev = pinned[b].to("cuda", non_blocking=True)
                events[b].record()
                used[b] = True
            torch.cuda.synchronize()
            report(f"pread {threads}", time.perf_counter() - t0, read_s=round(read, 3))
            pool.shutdown()
            os.close(fd)
            del pinned, hosts, dev


def downloads(a, libc, page):
    MADV_HUGEPAGE, MADV_POPULATE_WRITE = 14, 23
    n = a.down_mib << 20
    count = int(a.gib * (1 << 30) / 8) // n                        # a 4-bit result is an eighth of its fp32 source
    src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
    try:
        thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
    except OSError:
        thp = "?"

    def advise(arr, *advices):
        addr = arr.ctypes.data
        lo = addr + (-addr % page)
        for adv in advices:
            libc.madvise(lo, addr + arr.nbytes - lo - (addr + arr.nbytes - lo) % page, adv)

    def run(name, make):
        kept = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(count):
            out = make()
            torch.from_numpy(out).copy_(src)
            kept.append(out)
        dt = time.perf_counter() - t0
        print(json.dumps({"download": name, "seconds": round(dt, 3), "gb_per_s": round(count * n / dt / 1e9, 1), "results": count, "thp": thp}), flush=True)

    def fresh():
        return np.empty(n, dtype=np.uint8)

    def huge():
        out = np.empty(n, dtype=np.uint8)
        advise(out, MADV_HUGEPAGE)
        return out

    def populated():
        out = np.empty(n, dtype=np.uint8)
        advise(out, MADV_HUGEPAGE, MADV_POPULATE_WRITE)
        return out

    def populated_small():
        out = np.empty(n, dtype=np.uint8)
        advise(out, MADV_POPULATE_WRITE)
        return out

    pools = {k: ThreadPoolExecutor(k) for k in (2, 4, 8)}

    def populated_by(threads):
        def make():
            out = np.empty(n, dtype=np.uint8)
            addr = out.ctypes.data
            lo, hi = addr + (-addr % page), addr + n - (addr + n) % page
            step = -(-(hi - lo) // threads)
            step += -step % page
            list(pools[threads].map(lambda a: libc.madvise(a, min(step, hi - a), MADV_POPULATE_WRITE), range(lo, hi, step)))
            return out
        return make

    one = np.empty(n, dtype=np.uint8)
    for name, make in (("fresh", fresh), ("populated by 2 threads", populated_by(2)), ("populated by 4 threads", populated_by(4)),
                       ("populated by 8 threads", populated_by(8)), ("huge", huge), ("populated", populated), ("populated, 4 KiB pages", populated_small),
                       ("pinned", lambda: torch.empty(n, dtype=torch.uint8, pin_memory=True).numpy()), ("reused", lambda: one), ("fresh", fresh)):
        run(name, make)


if __name__ == "__main__":
    main()
