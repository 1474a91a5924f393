"""One call sequence, two libraries: libfpv_env.so (fpv_*, device buffers) and the oracle's twin of the same C ABI
(oracle/fpv_cpu.c: fpv_cpu_*, host buffers).  The driver binds `prefix + name` with the argument types of
fpv_drone_rl_agent_amd._lib.SIGNATURES — the twin's signatures ARE the product's — and differs between the two only in where the
buffers live."""
import ctypes as C

import numpy as np

from fpv_drone_rl_agent_amd import _lib

TWIN = ["fpv_create", "fpv_destroy", "fpv_reset", "fpv_step", "fpv_step_f64", "fpv_dims", "fpv_get_state", "fpv_set_state",
        "fpv_set_tick", "fpv_get_tick", "fpv_ledger_view", "fpv_ledger_clear", "fpv_ledger_pack", "fpv_gae", "fpv_last_error", "fpv_version"]


class AbiDriver:
    def __init__(self, lib, prefix: str, device: bool):
        self.lib, self.prefix, self.device = lib, prefix, device
        for name in TWIN:
            res, args = _lib.SIGNATURES[name]
            f = getattr(lib, prefix + name[len("fpv_"):])
            f.restype, f.argtypes = res, args
        self.h = C.c_void_p()
        if device:
            import torch
            self.torch = torch
            torch.cuda.init()
            # the HIP runtime this process already runs on (torch's): opening another copy by name would start a second runtime
            loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
            self.hip = C.CDLL(loaded[0])

    def call(self, name, *args):
        rc = getattr(self.lib, self.prefix + name)(*args)
        if rc != 0:
            raise RuntimeError(f"{self.prefix}{name}: {getattr(self.lib, self.prefix + 'last_error')().decode()}")

    # ---- buffers: torch device tensors or numpy arrays, one helper each way
    def buf(self, shape, dtype):
        if self.device:
            return self.torch.zeros(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device="cuda")
        return np.zeros(shape, dtype=dtype)

    def put(self, array):
        return self.torch.from_numpy(np.ascontiguousarray(array)).cuda() if self.device else np.ascontiguousarray(array)

    def ptr(self, b):
        if b is None:
            return C.c_void_p(0)
        return C.c_void_p(b.data_ptr()) if self.device else b.ctypes.data_as(C.c_void_p)

    def host(self, b):
        if self.device:
            self.torch.cuda.synchronize()
            return b.cpu().numpy()
        return b

    def read(self, address, count, dtype):
        """count elements at a raw address of the library's memory space -> numpy"""
        out = np.zeros(count, dtype=dtype)
        if count == 0:
            return out
        if self.device:
            self.torch.cuda.synchronize()
            assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(address), C.c_size_t(out.nbytes), 2) == 0
        else:
            C.memmove(out.ctypes.data_as(C.c_void_p), C.c_void_p(address), out.nbytes)
        return out

    # ---- the ABI
    def create(self, n, precision=0, task=0, noise=0.02, seed=0, detector=0, ledger_capacity=0, domain_rand=0, floor_check=1):
        self.n, self.dtype = n, np.float64 if precision == 1 else np.float32
        cfg = _lib.FpvConfig(n, 0, precision, floor_check, task, noise, seed, 0, ledger_capacity, domain_rand, detector)
        self.call("create", C.byref(cfg), C.byref(self.h))
        sw, ow, aw = C.c_int32(), C.c_int32(), C.c_int32()
        self.call("dims", self.h, C.byref(sw), C.byref(ow), C.byref(aw))
        self.S, self.O, self.A = sw.value, ow.value, aw.value
        self.obs, self.reward = self.buf((n, self.O), self.dtype), self.buf((n,), self.dtype)
        self.term, self.trunc = self.buf((n,), np.uint8), self.buf((n,), np.uint8)
        return self

    def destroy(self):
        self.call("destroy", self.h)

    def reset(self, mask=None):
        m = self.put(mask.astype(np.uint8)) if mask is not None else None
        self.call("reset", self.h, self.ptr(m), self.ptr(self.obs), None)
        return self.host(self.obs).copy()

    def step(self, actions):
        a = self.put(actions.astype(self.dtype))
        name = "step_f64" if self.dtype == np.float64 else "step"
        self.call(name, self.h, self.ptr(a), self.ptr(self.obs), self.ptr(self.reward), self.ptr(self.term), self.ptr(self.trunc), None)
        return (self.host(self.obs).copy(), self.host(self.reward).copy(), self.host(self.term).copy(), self.host(self.trunc).copy())

    def get_state(self):
        s = self.buf((self.n, self.S), self.dtype)
        self.call("get_state", self.h, self.ptr(s), None)
        return self.host(s).copy()

    def set_state(self, state):
        s = self.put(state.astype(self.dtype))
        self.call("set_state", self.h, self.ptr(s), None)
        if self.device:
            self.torch.cuda.synchronize()

    def set_tick(self, tick):
        self.call("set_tick", self.h, C.c_uint64(tick), None)

    def get_tick(self):
        t = C.c_uint64()
        self.call("get_tick", self.h, C.byref(t), None)
        return t.value

    def ledger(self):
        """The done ledger, tile after tile in append order: dict of numpy arrays."""
        v = _lib.FpvLedger()
        self.call("ledger_view", self.h, C.byref(v))
        tiles = v.capacity // v.rows_per_tile
        count = self.read(v.count, tiles, np.uint32)
        idx, info, tag = (self.read(p, v.capacity, d) for p, d in ((v.idx, np.int32), (v.info, np.uint32), (v.tag, np.uint32)))
        ret = self.read(v.ep_return, v.capacity, self.dtype)
        obs = self.read(v.obs, v.capacity * self.O, self.dtype).reshape(v.capacity, self.O)
        rows = np.concatenate([np.arange(t * v.rows_per_tile, t * v.rows_per_tile + min(int(c), v.rows_per_tile)) for t, c in enumerate(count)]
                              or [np.zeros(0, np.int64)]).astype(np.int64)
        return {"count": int(count.sum()), "idx": idx[rows], "info": info[rows], "tag": tag[rows], "ep_return": ret[rows], "obs": obs[rows]}

    def ledger_clear(self):
        self.call("ledger_clear", self.h, None)

    def ledger_pack(self, max_rows, obs_rms=None, clip=10.0, eps=1e-8):
        """fpv_ledger_pack: (header[8], rows [packed, 4 + O] as uint32 words)."""
        out = self.buf((8 + max_rows * (4 + self.O),), np.uint32)
        rms = self.put(np.asarray(obs_rms, np.float64)) if obs_rms is not None else None
        self.call("ledger_pack", self.h, self.ptr(rms), C.c_float(clip), C.c_float(eps), self.ptr(out), C.c_int64(max_rows), None)
        w = self.host(out).copy()
        return w[:8], w[8:8 + int(w[1]) * (4 + self.O)].reshape(int(w[1]), 4 + self.O)

    def gae(self, rewards, values, starts, last_values, last_dones, gamma=0.99, lam=0.95):
        T, n = rewards.shape
        r, v = self.put(rewards.astype(np.float32)), self.put(values.astype(np.float32))
        s, lv, ld = self.put(starts.astype(np.uint8)), self.put(last_values.astype(np.float32)), self.put(last_dones.astype(np.uint8))
        adv, ret = self.buf((T, n), np.float32), self.buf((T, n), np.float32)
        stats, ms = self.buf((3,), np.float64), self.buf((2,), np.float64)
        self.call("gae", T, n, self.ptr(r), self.ptr(v), self.ptr(s), self.ptr(lv), self.ptr(ld), gamma, lam, self.ptr(adv), self.ptr(ret),
                  self.ptr(stats), self.ptr(ms), None)
        return self.host(adv).copy(), self.host(ret).copy(), self.host(stats).copy(), self.host(ms).copy()


def twin():
    """Driver over the oracle's fpv_cpu_* set (host buffers)."""
    from oracle import oracle as orc
    orc.build()
    return AbiDriver(C.CDLL(orc._LIB), "fpv_cpu_", device=False)


def product():
    """Driver over libfpv_env.so (device buffers)."""
    import torch          # first: libfpv_env.so then binds to the HIP runtime torch brought, not to a second copy of it
    torch.cuda.init()
    return AbiDriver(_lib.load(), "fpv_", device=True)
