"""Time pr_depth_order: sample sort against the radix sort, per size.  usage: bench_depth_order.py [sizes...]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [200_000, 1_000_000, 1_250_000]
for n in sizes:
    d = (torch.rand(1, n, device=dev) * 2 + 2)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    nb = lib.pr_depth_order_scratch_bytes(1, n)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    row = {}
    for name, v in (("sample sort", "2"), ("radix sort", None)):
        _lib.set_tuning(POGS_RAST_DEPTH_SORT=v)
        def run():
            assert lib.pr_depth_order(1, n, ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                      ctypes.c_void_p(scratch.data_ptr()), nb, torch.cuda.current_stream().cuda_stream) == 0
        for _ in range(5): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50): run()
        b.record(); torch.cuda.synchronize()
        row[name] = a.elapsed_time(b) / 50 * 1e3
    print(n, {k: "%.1f us" % v for k, v in row.items()})
