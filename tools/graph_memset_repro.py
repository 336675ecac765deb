"""The in-process view of tools/graph_memset_repro.hip: a hipMemsetAsync issued through ctypes on torch's capturing
stream inside torch.cuda.graph -- how the library's own hipMemsetAsync calls were captured in round 2 -- followed by a torch kernel
(add_).  Every torch tensor is a sub-buffer of one of the caching allocator's blocks and every torch process runs on the HIP runtime
torch bundles (torch/lib/libamdhip64.so, 7.0.51831), so every variant fails from the second replay on, whatever runs between the
replays (profiles/r03_graph_memset_repro_torch.txt): the node then writes a wrong VALUE (the element count or address bits), which
is how a "cleared" scratch buffer came to hold garbage.  Prints one line per variant (OK, or the first replay after which the
buffer was not 1)."""
import ctypes
import sys

import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def memset(t, value, stream):
    rc = hip.hipMemsetAsync(t.data_ptr(), value, t.numel() * t.element_size(), stream.cuda_stream)
    if rc != 0:
        raise RuntimeError('hipMemsetAsync -> %d' % rc)


def run(n, eager, replays=50):
    dev = torch.device('cuda')
    a = torch.full((n,), 7, dtype=torch.int32, device=dev)
    b = torch.empty(n, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        memset(a, 0, torch.cuda.current_stream())
        a.add_(1)
    for r in range(replays):
        g.replay()
        if eager == 'hipMemsetAsync(other buffer), current stream':
            memset(b, 0, torch.cuda.current_stream())
        elif eager == 'hipMemsetAsync(other buffer), side stream':
            memset(b, 0, side)
        elif eager == 'tensor.zero_() of another buffer':
            b.zero_()
        elif eager == 'torch.zeros allocation':
            b = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if not bool((a == 1).all()):
            return 'FAIL (replay %d: a[0] = %d)' % (r, int(a[0]))
        if eager == 'hipMemsetAsync(SAME buffer) + kernel, current stream':
            memset(a, 1, torch.cuda.current_stream())
            a.add_(3)
            torch.cuda.synchronize()
    return 'OK'


def main():
    fails = 0
    for n in (1, 1024, 1 << 18, 1 << 24):
        for eager in ('nothing', 'hipMemsetAsync(other buffer), current stream', 'hipMemsetAsync(other buffer), side stream',
                      'hipMemsetAsync(SAME buffer) + kernel, current stream', 'tensor.zero_() of another buffer', 'torch.zeros allocation'):
            res = run(n, eager)
            fails += res != 'OK'
            print('torch.cuda.graph  n=%-9d %-55s : %s' % (n, eager, res))
    print('%d failing variant(s)' % fails)


if __name__ == '__main__':
    sys.exit(main())
