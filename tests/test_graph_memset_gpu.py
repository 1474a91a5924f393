"""What zeroes device words inside a captured HIP graph.

Finding (tools/graph_memset_probe.py, profiles/r03_graph_memset_probe.json; ROCm 7.2, MI355X): a hipMemsetAsync captured into a
graph is right on the graph's FIRST replay; from the second replay on the node may fill the words with stale kernel-argument
bytes — device-address fragments such as 0x7131'6220'0000 — instead of the value (seen for 16-, 32-, 2 560- and 4 096-byte nodes;
which captures are hit varied with the capture history: in the probe's order the 16-byte aligned destinations, here both).
That is the "garbage in the counters" of round 1 (commit e1a0236: the 16-byte counters of the ledger scan and led_count), and it
made round 2's captured persistent rollout read stale barrier words from its second replay on.

The product therefore zeroes with kernels only (fpv_rollout_begin, fpv_ledger_clear, zero_words_kernel): asserted here over
repeated replays at every alignment.  The memset node itself is probed and reported, not asserted — a fixed runtime must not
turn this test red.
"""
import ctypes as C

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _replays(fn_capture, buf, inner, reps=4):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn_capture()
    out = []
    for _ in range(reps):
        buf.fill_(0x5A5A5A5A)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        out.append(inner().cpu())
    return out


def test_kernel_zeroing_in_a_captured_graph_is_right_at_every_replay():
    from fpv_drone_rl_agent_amd import _lib as L
    from fpv_drone_rl_agent_amd.engine import FpvEngine
    lib = L.load()
    eng = FpvEngine(65536, 0)                      # led_count: 1 024 words = 4 096 B, the buffer round 1 zeroed with a memset node
    counters = torch.zeros(8, dtype=torch.int32, device="cuda")
    sums = torch.zeros(4, dtype=torch.float64, device="cuda")
    starts = torch.zeros((2, 65536), dtype=torch.uint8, device="cuda")

    def capture():
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.fpv_rollout_begin(eng._h, C.c_void_p(starts[0].data_ptr()), C.c_void_p(starts[1].data_ptr()),
                                      C.c_void_p(counters.data_ptr()), C.c_void_p(sums.data_ptr()), s), "begin")
        L.check(lib.fpv_debug_atomic_inc(C.c_void_p(eng.led_count.data_ptr()), eng.led_count.numel(), s), "inc")
        L.check(lib.fpv_debug_atomic_inc(C.c_void_p(counters.data_ptr()), 4, s), "inc")

    capture()                                      # warm-up outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        capture()
    for rep in range(4):
        eng.led_count.fill_(0x5A5A5A5A); counters.fill_(0x5A5A5A5A); sums.fill_(7.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert (eng.led_count == 1).all() and (counters[:4] == 1).all() and (counters[4:] == 0x5A5A5A5A).all(), rep
        assert (sums[:2] == 0).all() and (sums[2:] == 7.0).all(), rep
    eng.close()


def test_memset_node_probe_is_reported():
    """hipMemsetAsync + a consumer kernel captured into a graph: first replay right everywhere; later replays are reported (and
    were wrong when this was written)."""
    from fpv_drone_rl_agent_amd import _lib as L
    lib = L.load()
    report = {}
    for offset in (0, 8):
        words = 640                                # 2 560 B: the barrier words of fpv_rollout_steps
        buf = torch.full((64 + words + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        view = buf[64 + offset // 4: 64 + offset // 4 + words]

        def capture():
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            L.check(lib.fpv_debug_memset(C.c_void_p(view.data_ptr()), 0, 4 * words, s), "memset")
            L.check(lib.fpv_debug_atomic_inc(C.c_void_p(view.data_ptr()), words, s), "inc")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            capture()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        outs = _replays(capture, buf, lambda: view, reps=3)
        assert (outs[0] == 1).all()                                 # the first replay is right at either alignment
        report[offset] = [bool((o == 1).all()) for o in outs]
    print("hipMemsetAsync node in a captured graph, replays right? by destination offset from 16-byte alignment:", report)
