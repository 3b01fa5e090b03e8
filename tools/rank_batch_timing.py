"""What ONE rank of the 8-GPU schedule multiplies, timed on one GPU (resident operands, HIP events, min / median of `reps`):
the 7-way split's single 32768^3 sub-product against the 47-way split's six 16384^3 sub-products -- one call each, and batched
(m4ri_amd_mul_batch_dev) 2 + 2 + 2, 3 + 3 and 6 at a time.  python tools/rank_batch_timing.py [n] [reps]"""
import os
import sys
import statistics

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import m4ri_amd


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return min(out), statistics.median(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    m4ri_amd.init(0)
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream().cuda_stream

    def mats(dim, count, seed):
        w = dim // 64
        t = torch.empty(count * dim * w, dtype=torch.int64, device="cuda")
        for b in range(count):
            m4ri_amd.fill_dev(t.data_ptr() + 8 * b * dim * w, w, dim, dim, seed + b, st)
        return t

    print(f"# n = {n}: one product on one GPU, then the per-rank pieces of the sharded schedules")
    A, B = mats(n, 1, 1), mats(n, 1, 2)
    C = torch.empty_like(A)
    w = n // 64
    t = timed(lambda: m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, False, 0, st), reps)
    print(f"{n}^3 one product                      min {t[0]:8.3f} ms  median {t[1]:8.3f} ms")
    del A, B, C
    h = n // 2
    A, B = mats(h, 1, 3), mats(h, 1, 4)
    C = torch.empty_like(A)
    t7 = timed(lambda: m4ri_amd.mul_dev(C.data_ptr(), h // 64, A.data_ptr(), h // 64, B.data_ptr(), h // 64, h, h, h, False, 0, st), reps)
    print(f"7-way rank: one {h}^3                   min {t7[0]:8.3f} ms  median {t7[1]:8.3f} ms")
    del A, B, C
    q = n // 4
    wq = q // 64
    A, B = mats(q, 6, 10), mats(q, 6, 20)
    C = torch.empty_like(A)
    bs = q * wq

    def groups(sizes):
        def run():
            b0 = 0
            for g in sizes:
                if g == 1:
                    m4ri_amd.mul_dev(C.data_ptr() + 8 * b0 * bs, wq, A.data_ptr() + 8 * b0 * bs, wq, B.data_ptr() + 8 * b0 * bs, wq, q, q, q, False, 0, st)
                else:
                    m4ri_amd.mul_batch_dev(C.data_ptr() + 8 * b0 * bs, wq, bs, A.data_ptr() + 8 * b0 * bs, wq, bs, B.data_ptr() + 8 * b0 * bs, wq, bs,
                                           q, q, q, g, False, 0, st)
                b0 += g
        return run
    del A, B, C
    A, B = mats(q, 12, 10), mats(q, 12, 40)
    C = torch.empty_like(A)
    for sizes in ([1] * 6, [2, 2, 2], [3, 3], [6], [1] * 5, [3, 2], [5], [1] * 12, [2] * 6, [3] * 4, [4] * 3, [6, 6], [12]):
        t = timed(groups(sizes), reps)
        s = m4ri_amd.get_stats()
        print(f"47-way rank: {sum(sizes)} x {q}^3 as {'+'.join(map(str, sizes)):24s} min {t[0]:8.3f} ms  median {t[1]:8.3f} ms   (last call: L = {int(s.levels)}, "
              f"{int(s.leaf_products)} leaf products in {int(s.leaf_launches)} launch(es); model {1e3 * m4ri_amd.model_seconds_batch(q, q, q, -1, sizes[-1]):.3f} ms)")


def slabs():
    """the row-slab schedule's per-rank product on 4 and on 2 ranks: (n/W) x n x n"""
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    st = torch.cuda.current_stream().cuda_stream
    w = n // 64
    B = torch.empty(n * w, dtype=torch.int64, device="cuda")
    m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 77, st)
    for W in (4, 2):
        rows = n // W
        A = torch.empty(rows * w, dtype=torch.int64, device="cuda")
        m4ri_amd.fill_dev(A.data_ptr(), w, rows, n, 78, st)
        C = torch.empty_like(A)
        t = timed(lambda: m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, rows, n, n, False, 0, st), reps)
        s = m4ri_amd.get_stats()
        print(f"row-slab rank of {W}: {rows} x {n} x {n}        min {t[0]:8.3f} ms  median {t[1]:8.3f} ms   (L = {int(s.levels)}, {int(s.leaf_products)} leaf products)")
        del A, C


if __name__ == "__main__":
    main()
    slabs()

