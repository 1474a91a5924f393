"""Is a hipMemsetAsync node inside a captured HIP graph executed, in stream order, at every replay?  (round 1 replaced
the captured memsets of the rollout tail by a zeroing kernel because the counters "held garbage"; the persistent launch still
issues a captured hipMemsetAsync for its barrier words.)  Sizes: 16 B (the ledger-scan counters), 32 B, 2 560 B (the barrier
words of fpv_rollout_steps), 4 096 B (led_count of 65 536 drones); destinations at 256-, 16- and 8-byte alignment; each memset is
followed by a kernel that adds 1 to every word (the consumer pattern), and the graph is replayed three times."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpv_drone_rl_agent_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda", 0)
out = []
for nbytes in (16, 32, 2560, 4096):
    for offset in (0, 16, 8):
        words = nbytes // 4
        buf = torch.full((64 + words + 64,), 0x5A5A5A5A, dtype=torch.int32, device=dev)      # guard words on both sides
        view = buf[64 + offset // 4: 64 + offset // 4 + words]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):      # warm-up outside capture
            lib.fpv_debug_memset(C.c_void_p(view.data_ptr()), 0, nbytes, C.c_void_p(side.cuda_stream))
            lib.fpv_debug_atomic_inc(C.c_void_p(view.data_ptr()), words, C.c_void_p(side.cuda_stream))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            L.check(lib.fpv_debug_memset(C.c_void_p(view.data_ptr()), 0, nbytes, s), "memset")
            L.check(lib.fpv_debug_atomic_inc(C.c_void_p(view.data_ptr()), words, s), "inc")
        res = []
        for rep in range(3):
            buf.fill_(0x5A5A5A5A)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            host = buf.cpu()
            inner = host[64 + offset // 4: 64 + offset // 4 + words]
            guards_ok = bool((host[:64 + offset // 4] == 0x5A5A5A5A).all() and (host[64 + offset // 4 + words:] == 0x5A5A5A5A).all())
            res.append({"all_words_equal_1": bool((inner == 1).all()), "guards_untouched": guards_ok,
                        "bad_words": int((inner != 1).sum()),
                        "distinct_values_hex": sorted({hex(int(x) & 0xFFFFFFFF) for x in inner.tolist()})[:4]})
        out.append({"bytes": nbytes, "offset_from_256B_alignment": offset, "replays": res})
print(json.dumps({"probe": "hipMemsetAsync node + consumer kernel in a captured graph, 3 replays", "cases": out,
                  "all_ok": all(r["all_words_equal_1"] and r["guards_untouched"] for c in out for r in c["replays"])}))
