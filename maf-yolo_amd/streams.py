"""HIP streams that really run side by side.

ROCm multiplexes the streams of a process onto a few hardware queues (4 by default, GPU_MAX_HW_QUEUES); two streams that land on the
same queue execute their kernels strictly one after the other.  A serving loop that keeps two batches in flight on two streams
(Model.forward(x, slot=k), bench.py) then silently loses the overlap — measured: 1.58 ms per forward on two queues, 2.0 ms when the
streams alias.  `concurrent_streams` creates streams until it has `n` that demonstrably overlap (timed with spin kernels)."""
import time

import torch


def _spin_ms(streams, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def overlap_ratio(streams, cycles=2_000_000, reps=4):
    """time of one spin kernel on every stream at once / time of one spin kernel alone: ~1 if all streams overlap, ~k if k of them share a queue."""
    _spin_ms(streams[:1], cycles)
    one = min(_spin_ms(streams[:1], cycles) for _ in range(reps))
    return min(_spin_ms(streams, cycles) for _ in range(reps)) / one


def concurrent_streams(device, n=2, tries=16, cycles=2_000_000):
    """-> list of n torch.cuda.Stream on `device` whose kernels overlap pairwise (falls back to plain new streams if the probe cannot
    tell, e.g. a device that runs one kernel at a time).  `concurrent_streams.last_ratio` = overlap_ratio of the returned set."""
    device = torch.device(device)
    if not hasattr(torch.cuda, "_sleep"):                          # no spin kernel to time with: plain streams
        concurrent_streams.last_ratio = float("nan")
        return [torch.cuda.Stream(device) for _ in range(n)]
    with torch.cuda.device(device):
        torch.cuda.synchronize()
        picked = [torch.cuda.Stream(device)]
        rejected = []                                              # kept alive: a destroyed stream's queue slot would be handed out again
        for _ in range(tries):
            if len(picked) == n:
                break
            s = torch.cuda.Stream(device)
            if overlap_ratio(picked + [s], cycles) < 1.4:          # all of them together take about as long as one: every pair overlaps
                picked.append(s)
            else:
                rejected.append(s)
        while len(picked) < n:
            picked.append(torch.cuda.Stream(device))
        concurrent_streams._keep = getattr(concurrent_streams, "_keep", []) + rejected
        concurrent_streams.last_ratio = overlap_ratio(picked, cycles) if n > 1 else 1.0
    return picked


def masked_stream(device, n_cus, total_cus=256):
    """A torch stream whose kernels run on `n_cus` of the device's compute units only (csrc/capi.hip maf_stream_create_masked): the LAST n_cus bits of the
    driver's CU numbering, which deals consecutive bits round-robin to the XCDs — n_cus / 8 units of every XCD.  The handle lives as long as the process."""
    import ctypes as C
    from . import lib
    device = torch.device(device)
    words = (C.c_uint32 * (total_cus // 32))()
    for i in range(total_cus - n_cus, total_cus):
        words[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    with torch.cuda.device(device):
        lib.check(lib.load().maf_stream_create_masked(words, len(words), C.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=device)
