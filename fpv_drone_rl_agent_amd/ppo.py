"""PPO update on top of the on-device rollout (SB3 PPO.train semantics).

north_star replaces the reference's SAC learner (simulation/train_hover.py:47-58) by PPO.  SB3's
defaults are kept where they make sense at this batch scale (clip_range 0.2, vf_coef 0.5,
max_grad_norm 0.5, gamma 0.99, lambda 0.95, Adam); the rollout is n_steps x 65 536 transitions, so
minibatches are large and few (SB3's 2048 x 64 defaults would be a pointless batch here).
The minibatch gradient comes from fpv_ppo_grad (csrc/fpv_ppo.hip: forward, loss derivatives and
backward as hand-written fp32-MFMA kernels reading the rollout buffers through the minibatch's
index vector); `fused_update=False` keeps the same loss on torch autograd as the readable
reference (tests/test_ppo_grad_gpu.py compares the two).  Adam, the gradient clip and the
cross-rank average are single-tensor torch ops on one flat parameter buffer: ONE all-reduce per
minibatch (156 KB - latency-bound on xGMI, so no bucketing and no ring of our own).
"""
from __future__ import annotations

import os
import sys
from collections.abc import MutableMapping

import torch
import torch.distributed as td

from . import dist as D
from . import ops as O
from .rollout import RolloutCollector


class _LazyLog(MutableMapping):
    """model.logger: a mapping whose update-time values may still be on their way from the device.  train_on() leaves the last epoch's
    statistics and the policy's log-std as asynchronous copies into pinned host words (so that the host can queue the next rollout
    while the update still runs) and registers a resolver; any READ of the mapping waits for them and fills the numbers in.  Not a
    dict subclass on purpose: C-level readers of a dict (json.dumps, PyDict_Merge) would walk the storage behind the resolver's back —
    here every access goes through these methods, and json.dumps(model.logger) fails loudly (use dict(model.logger))."""

    def __init__(self):
        self._d: dict[str, float] = {}
        self._pending = None

    def defer(self, resolver) -> None:
        self._resolve()                         # (an earlier update's numbers first: each resolver reads its own pinned words)
        self._pending = resolver

    def _resolve(self) -> None:
        if self._pending is not None:
            resolver, self._pending = self._pending, None
            self._d.update(resolver())

    def record(self, k, v) -> None:
        """A write that does not wait for pending values (PPO.learn's episode statistics)."""
        self._d[k] = v

    def __getitem__(self, k):
        self._resolve()
        return self._d[k]

    def __setitem__(self, k, v):
        self._d[k] = v

    def __delitem__(self, k):
        self._resolve()
        del self._d[k]

    def __iter__(self):
        self._resolve()
        return iter(self._d)

    def __len__(self):
        self._resolve()
        return len(self._d)

    def copy(self):
        return dict(self)

    def __repr__(self):
        self._resolve()
        return repr(self._d)


class PPO:
    def __init__(self, collector: RolloutCollector, learning_rate: float = 3e-4, n_epochs: int = 4,
                 n_minibatches: int = 8, clip_range: float = 0.2, ent_coef: float = 0.0, vf_coef: float = 0.5,
                 max_grad_norm: float = 0.5, normalize_advantage: bool = True, target_kl: float | None = None,
                 use_graph: bool = True, fused_update: bool | None = None, log_std_min: float | None = None,
                 grad_storage: str = "f32", update_path: str | None = None, shuffle: str | None = None):
        self.col = collector
        self.policy = collector.policy
        # capturable Adam: its step counter lives on the device, so one minibatch update (forward, loss, backward,
        # gradient all-reduce, clip, Adam) can be captured once into a HIP graph and replayed
        self.use_graph = bool(use_graph) and next(self.policy.parameters()).is_cuda
        # all parameters / gradients live in one flat buffer: one Adam, one clip, one all-reduce per minibatch
        self._flat, self._flat_grad = self.policy.flatten_parameters()
        self._mb_graph = None
        fp = getattr(collector, "fp", None) if getattr(collector, "fused_policy", False) else None
        if fused_update is None:
            fused_update = fp is not None
        if fused_update and fp is None:
            raise ValueError("fused_update needs a collector with the fused policy kernel (fused_policy=True)")
        self.fused_update = bool(fused_update)
        if grad_storage not in ("f32", "bf16"):
            raise ValueError("grad_storage must be 'f32' or 'bf16'")
        self.grad_storage = grad_storage        # how fpv_ppo_grad passes activations between its kernels (ops_policy.ppo_grad)
        if update_path is None:
            update_path = "one_kernel" if grad_storage == "f32" else "two_kernels"
        if update_path not in ("one_kernel", "two_kernels", "two_kernels_bf16x6", "two_kernels_bf16x3") or (
                grad_storage == "bf16" and update_path != "two_kernels"):
            raise ValueError("update_path must be 'one_kernel', 'two_kernels', 'two_kernels_bf16x6' or 'two_kernels_bf16x3'; grad_storage='bf16' exists on "
                             "'two_kernels' only")
        self.update_path = update_path          # which kernels compute the gradient (ops_policy.ppo_grad's `path`)
        # the epoch's row order (SB3: np.random.permutation per epoch): "keyed" = fpv_shuffle_indices, a keyed bijection of the row
        # indices computed in one launch (the fused update's default); "randperm" = torch.randperm (a sort of random keys: 170 us per
        # epoch at 2 x 10^6 rows, 3.5 % of a training iteration at 65 536 drones)
        if shuffle is None:
            shuffle = "keyed" if self.fused_update else "randperm"
        if shuffle not in ("keyed", "randperm") or (shuffle == "keyed" and not self.fused_update):
            raise ValueError("shuffle must be 'keyed' (fused update only) or 'randperm'")
        self.shuffle = shuffle
        self._shuffle_count = 0
        self._fp = fp
        self._ws = self._adv_n = self._perm = self._epoch_stats = None
        self._stats = torch.zeros(4, dtype=torch.float32, device=collector.device) if self.fused_update else None
        if self.fused_update:                   # clip + Adam as one launch on the flat buffer
            self.opt = O.FusedAdam(self._flat, lr=learning_rate, eps=1e-5, max_grad_norm=max_grad_norm or 0.0)
        else:
            self.opt = torch.optim.Adam([self._flat], lr=learning_rate, eps=1e-5, capturable=self.use_graph)
        self.n_epochs, self.n_minibatches = n_epochs, n_minibatches
        self.clip_range, self.ent_coef, self.vf_coef = clip_range, ent_coef, vf_coef
        self.max_grad_norm, self.normalize_advantage, self.target_kl = max_grad_norm, normalize_advantage, target_kl
        # optional floor under the policy's log-std (not in SB3): with 10^4..10^5 drones the Gaussian narrows to
        # sigma < 0.01 within 10^8 steps, where the importance ratios of a PPO epoch become violently sensitive
        self.log_std_min = log_std_min
        self.num_timesteps = 0
        self.logger: dict[str, float] = _LazyLog()
        self._stats_host = self._stats_event = self._logstd_host = None    # pinned mirrors of the epoch statistics (fused update)
        # target_kl's early stop decided on the device (fpv_ppo_kl_gate / fpv_adam_clip_step_gated): one rank, fused update
        self._device_gate = (self.fused_update and target_kl is not None and D.world_size() == 1
                             and os.environ.get("FPV_PPO_SYNC_LOG", "") in ("", "0")
                             and os.environ.get("FPV_PPO_DEVICE_GATE", "1") != "0")      # (=0: the host decides, one wait per epoch; A/B)
        self._gate = torch.zeros(2, dtype=torch.int32, device=collector.device) if self._device_gate else None
        self._gate_host = torch.zeros(2, dtype=torch.int32).pin_memory() if self._device_gate else None
        self._gen = torch.Generator(device=collector.device)
        self._gen.manual_seed(1234 + D.rank())

    def _allreduce_grads(self) -> None:
        w = D.world_size()
        if w == 1:
            return
        flat = self._flat_grad
        if td.get_backend() == "gloo" and flat.is_cuda:
            host = flat.cpu()
            td.all_reduce(host)
            flat.copy_(host.to(flat.device))
        else:
            td.all_reduce(flat)
        flat /= w

    def _minibatch_update(self, obs, act, old_lp, ret, adv):
        """One clipped-surrogate + value-loss step; returns the detached statistics vector."""
        v, lp, ent = self.policy.evaluate(obs, act)
        ratio = torch.exp(lp - old_lp)
        pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - self.clip_range, 1 + self.clip_range)).mean()
        vl = torch.nn.functional.mse_loss(v, ret)
        loss = pg + self.vf_coef * vl - self.ent_coef * ent.mean()
        self._flat_grad.zero_()
        loss.backward()                      # accumulates into the views of the flat gradient buffer
        self._allreduce_grads()
        torch.nn.utils.clip_grad_norm_([self._flat], self.max_grad_norm)
        self.opt.step()
        if self.log_std_min is not None:
            self.policy.log_std.data.clamp_(min=self.log_std_min)
        with torch.no_grad():
            log_ratio = lp - old_lp
            kl = ((torch.exp(log_ratio) - 1) - log_ratio).mean()
            return torch.stack([pg.detach(), vl.detach(), kl, ((ratio - 1).abs() > self.clip_range).float().mean()])

    def _fused_minibatch(self, idx):
        """The same step with the gradient from fpv_ppo_grad; the rollout arrays are read through idx in the kernel."""
        b = self._fused_src
        self._fp.ppo_grad(b["obs"], b["act"], b["old_lp"], b["adv"], b["ret"], index=idx, clip_range=self.clip_range,
                          vf_coef=self.vf_coef, ent_coef=self.ent_coef, stats=self._stats, workspace=self._ws,
                          storage=self.grad_storage, path=self.update_path)
        self._allreduce_grads()
        if self._device_gate:
            self.opt.step(self._flat_grad, gate=self._gate)
        else:
            self.opt.step(self._flat_grad)
        if self.log_std_min is not None:        # (idempotent: harmless behind a stopped update)
            self.policy.log_std.data.clamp_(min=self.log_std_min)
        return self._stats

    def _fused_epoch(self, perm):
        """Every minibatch of one epoch, in order, on the row order in `perm` — captured as ONE HIP graph: a launch per epoch instead
        of a copy of the indices, a launch and a copy of the statistics per minibatch (at 65 536 drones those were 1 ms of idle and
        copy time per 19.5 ms iteration)."""
        mb = perm.numel() // self.n_minibatches
        for k in range(self.n_minibatches):
            self._stats = self._epoch_stats[k]
            self._fused_minibatch(perm[k * mb:(k + 1) * mb])
        if self._device_gate:                   # does this epoch count, and does it stop the ones queued behind it?
            O.ppo_kl_gate(self._epoch_stats, 1.5 * self.target_kl, self._gate)
        return self._epoch_stats

    def _draw_order(self, total: int, dev, epoch: int) -> torch.Tensor:
        """The epoch's row order into the persistent index buffer (the epoch graph holds its address).  The key of the order is
        (update, epoch) — not a count of the orders drawn, which would depend on whether the epochs behind an early stop were queued."""
        if self.shuffle == "keyed":
            O.shuffle_indices(self._perm, 1234 + D.rank(), self._shuffle_count * self.n_epochs + epoch + 1)
        else:
            torch.randperm(total, device=dev, generator=self._gen, out=self._perm)
        return self._perm

    def _build_graph(self, step, inputs, dev) -> None:
        self._g_in = inputs
        # warm-up on a side stream with the optimiser state saved and restored, then capture
        saved = ({k: v.clone() for k, v in self.policy.state_dict().items()},
                 {"state": {k: {n: (t.clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                            for k, st in self.opt.state_dict()["state"].items()},
                  "param_groups": self.opt.state_dict()["param_groups"]})
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step(*self._g_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self._g_out = step(*self._g_in)
        except Exception as exc:                # e.g. a collective library that refuses stream capture
            if D.world_size() == 1:
                raise
            # the minibatch holds the gradient all-reduce: fall back to eager launches for the rest of the run (every
            # rank issues the same collectives either way, so ranks that captured and ranks that did not stay matched)
            self._graph_failed = True
            g = None
            if D.rank() == 0:
                print(f"fpv PPO: capturing the minibatch (with its all-reduce) into a HIP graph failed ({exc!r}); "
                      "running it eagerly.  FPV_GRAPH_COLLECTIVES=0 selects this path up front.", file=sys.stderr, flush=True)
        self.policy.load_state_dict(saved[0])
        if saved[1]["state"]:
            self.opt.load_state_dict(saved[1])
        else:                                   # first update ever: reset the freshly created Adam moments
            for st in self.opt.state.values():
                for t in st.values():
                    if torch.is_tensor(t):
                        t.zero_()
        self._mb_graph, self._mb_size = g, inputs[0].shape[0]

    def train_on(self, batch: dict) -> None:
        T, N = self.col.T, self.col.N
        obs = batch["obs"].reshape(T * N, -1)
        act = batch["actions"].reshape(T * N, -1)
        old_lp = batch["log_probs"].reshape(-1)
        ret = batch["returns"].reshape(-1)
        adv = batch["advantages"].reshape(-1)
        if self.normalize_advantage and not self.fused_update:
            # global statistics from the rollout's single all-gather (identical on every rank)
            adv = (adv - batch["adv_mean"].float()) / (batch["adv_std"].float() + 1e-8)
        total = T * N
        mb = total // self.n_minibatches
        dev = obs.device
        # HIP-graph replay of the minibatch: off under the gloo rehearsal backend, off when the collective must stay outside
        # any captured graph (FPV_GRAPH_COLLECTIVES=0: the first-RCCL-run escape hatch), off after a failed capture
        graph = self.use_graph and not getattr(self, "_graph_failed", False) and not (
            D.world_size() > 1 and (td.get_backend() == "gloo" or os.environ.get("FPV_GRAPH_COLLECTIVES", "1") == "0"))
        if self.fused_update:
            self._shuffle_count += 1            # (the number of this update: part of the epochs' shuffle keys)
            if self._adv_n is None or self._adv_n.numel() != total:
                self._adv_n = torch.empty(total, dtype=torch.float32, device=dev)
            if self.normalize_advantage:        # the same two roundings as above, straight into the persistent buffer (two passes, not three)
                torch.sub(adv, batch["adv_mean"].float(), out=self._adv_n)
                self._adv_n.div_(batch["adv_std"].float() + 1e-8)
            else:
                self._adv_n.copy_(adv)
            if self._ws is None or self._ws_rows != mb:
                self._ws, self._ws_rows = self._fp.ppo_workspace(mb), mb
            src = {"obs": obs, "act": act, "old_lp": old_lp, "adv": self._adv_n, "ret": ret}
            if graph and self._mb_graph is not None and any(src[k].data_ptr() != self._fused_src[k].data_ptr() for k in src):
                self._mb_graph = None           # the captured launches hold the rollout buffers' addresses
            self._fused_src = src
            if self._perm is None or self._perm.numel() != total:
                self._perm = torch.arange(total, dtype=torch.int64, device=dev)
                self._epoch_stats = torch.zeros((self.n_minibatches, 4), dtype=torch.float32, device=dev)
                self._mb_graph = None
            if graph and (self._mb_graph is None or self._mb_size != total):
                self._build_graph(self._fused_epoch, [self._perm], dev)
        elif graph and (self._mb_graph is None or self._mb_size != mb):
            f32 = torch.float32
            self._build_graph(self._minibatch_update,
                              [torch.zeros((mb, obs.shape[1]), dtype=f32, device=dev), torch.zeros((mb, act.shape[1]), dtype=f32, device=dev),
                               torch.zeros((mb,), dtype=f32, device=dev), torch.zeros((mb,), dtype=f32, device=dev),
                               torch.zeros((mb,), dtype=f32, device=dev)], dev)
        graph = graph and not getattr(self, "_graph_failed", False)
        stats = []
        if self.fused_update and D.world_size() == 1 and os.environ.get("FPV_PPO_SYNC_LOG", "") in ("", "0"):
            # (FPV_PPO_SYNC_LOG=1: the synchronising loop below, for an A/B)
            # One rank, fused update: the host never waits for more than it must.  An epoch is one graph replay; its statistics leave
            # as an asynchronous copy into pinned host words behind it; only target_kl's decision about the NEXT epoch waits for them
            # (nothing depends on the last epoch's), the logged numbers are resolved when somebody reads model.logger, and the host
            # goes on to queue the image repack and the next rollout while the last epoch still runs (tools/train_gaps.py: 0.5 ms of
            # a 18 ms iteration at 65 536 drones was the device waiting for the host after those synchronisations).
            if self._stats_host is None or len(self._stats_host) != self.n_epochs or self._stats_host[0].shape[0] != self.n_minibatches:
                self._stats_host = [torch.zeros((self.n_minibatches, 4), dtype=torch.float32).pin_memory() for _ in range(self.n_epochs)]
                self._stats_event = [torch.cuda.Event() for _ in range(self.n_epochs + 1)]
                self._logstd_host = torch.zeros(self.policy.log_std.shape, dtype=torch.float32).pin_memory()
            self.logger._resolve()               # the previous update's numbers leave the pinned words before they are written again
            if self._device_gate:
                self._gate.zero_()
            ran = 0
            for epoch in range(self.n_epochs):
                self._draw_order(total, dev, epoch)
                if graph:
                    self._mb_graph.replay()
                    st = self._g_out
                else:
                    st = self._fused_epoch(self._perm)
                self._stats_host[epoch].copy_(st, non_blocking=True)
                self._stats_event[epoch].record()
                ran = epoch + 1
                # target_kl: with the device gate every epoch is queued at once — an epoch behind the stop computes its gradients and
                # changes nothing (wasted only in updates that do stop) —; without it the decision about the next epoch waits here
                if self.target_kl is not None and not self._device_gate and epoch + 1 < self.n_epochs:
                    self._stats_event[epoch].synchronize()
                    if float(self._stats_host[epoch][:, 2].mean()) > 1.5 * self.target_kl:
                        break
            self._logstd_host.copy_(self.policy.log_std.detach(), non_blocking=True)
            if self._device_gate:
                self._gate_host.copy_(self._gate, non_blocking=True)
            self._stats_event[self.n_epochs].record()
            hosts, events, logstd, vf = self._stats_host, self._stats_event, self._logstd_host, self.vf_coef
            gate_host, n_last = self._gate_host, self.n_epochs

            def resolve():
                events[n_last].synchronize()
                counted = int(gate_host[1]) if gate_host is not None else ran
                m = torch.cat(hosts[:counted]).mean(0).tolist()
                return {"train/policy_gradient_loss": m[0], "train/value_loss": m[1], "train/approx_kl": m[2],
                        "train/clip_fraction": m[3], "train/loss": m[0] + vf * m[1], "train/std": float(torch.exp(logstd).mean())}

            self.logger.defer(resolve)
            self.col.sync_policy()
            return
        for epoch in range(self.n_epochs):
            if self.fused_update:               # the whole epoch is one graph replay on the order drawn into the persistent buffer
                self._draw_order(total, dev, epoch)
                if graph:
                    self._mb_graph.replay()
                    st = self._g_out.clone()
                else:
                    st = self._fused_epoch(self._perm).clone()
                stats.extend(st.unbind(0))
                if self.target_kl is not None:
                    kl = st.mean(0)[2]
                    if D.world_size() > 1:
                        kl = D.allgather_payload(kl.double().reshape(1)).mean()
                    if float(kl) > 1.5 * self.target_kl:
                        break
                continue
            perm = torch.randperm(total, device=dev, generator=self._gen)
            for k in range(self.n_minibatches):
                idx = perm[k * mb:(k + 1) * mb]
                if graph:
                    for dst, src_t in zip(self._g_in, (obs, act, old_lp, ret, adv)):
                        torch.index_select(src_t, 0, idx, out=dst)
                    self._mb_graph.replay()
                    stats.append(self._g_out.clone())
                else:
                    stats.append(self._minibatch_update(obs[idx], act[idx], old_lp[idx], ret[idx], adv[idx]))
            if self.target_kl is not None:
                kl = torch.stack(stats[-self.n_minibatches:]).mean(0)[2]
                if D.world_size() > 1:          # every rank must take the same decision: it gates the next all-reduces
                    kl = D.allgather_payload(kl.double().reshape(1)).mean()
                if float(kl) > 1.5 * self.target_kl:
                    break
        s = torch.stack(stats).mean(0).tolist()
        self.logger.update({"train/policy_gradient_loss": s[0], "train/value_loss": s[1], "train/approx_kl": s[2],
                            "train/clip_fraction": s[3], "train/loss": s[0] + self.vf_coef * s[1],
                            "train/std": float(torch.exp(self.policy.log_std.detach()).mean())})
        self.col.sync_policy()

    def learn(self, total_timesteps: int, callback=None, log_every: int = 0) -> "PPO":
        """model.learn(total_timesteps, callback) of simulation/train_hover.py:60; timesteps count every drone on
        every rank, as SB3 counts every env."""
        per_iter = self.col.T * self.col.N * D.world_size()
        it = 0
        while self.num_timesteps < total_timesteps:
            batch = self.col.collect()
            self.num_timesteps += per_iter
            self.train_on(batch)
            # (the episode statistics wait for the rollout's end: read behind the update's launches, where that costs nothing, not in
            # front of them, where the device would idle until the host has woken up and queued the update)
            ep = self.col.episodes
            if ep["count"] > 0 and ep["ep_len_mean"] == ep["ep_len_mean"]:
                self.logger.record("rollout/ep_len_mean", ep["ep_len_mean"])
                self.logger.record("rollout/ep_rew_mean", ep["ep_rew_mean"])
            it += 1
            if callback is not None:
                callback(self)
            if log_every and it % log_every == 0 and D.rank() == 0:
                print({"iter": it, "timesteps": self.num_timesteps, **{k: round(v, 4) for k, v in self.logger.items()}},
                      flush=True)
        return self

    def predict(self, obs: torch.Tensor, deterministic: bool = True):
        """model.predict(obs, deterministic=True) of simulation/test_hover.py:17 (obs already normalised)."""
        with torch.no_grad():
            mean, _ = self.policy.mean_value(obs)
            if not deterministic:
                mean = mean + torch.exp(self.policy.log_std) * torch.randn_like(mean)
        return torch.clamp(mean, -1, 1), None
