"""Multi-GPU glue: one process per GPU, env shards are independent.

The only exchange on the rollout path is ONE all-gather per PPO update of a few fp64 words per
rank — {count, sum A, sum A^2} of the advantages, with the VecNormalize moments riding in the
same payload — from which every rank derives the same global mean/std.  The message is < 400 B,
so it is latency-bound: a single all_gather (RCCL `ncclAllGather` on GPUs, gloo on CPU tests),
never a ring of our own.
"""
from __future__ import annotations

import datetime
import os

import torch
import torch.distributed as td


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = world == 1 and os.environ.get("FPV_REHEARSE_RANKS", "") == "1"     # several_ranks(): the N > 1 arrangement with G = 1
    if (world > 1 or rehearse) and not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        timeout = datetime.timedelta(seconds=int(os.environ.get("FPV_DIST_TIMEOUT_S", "300")))
        if backend == "nccl":
            torch.cuda.set_device(local)
            td.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timeout)
        else:
            td.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


def world_size() -> int:
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def several_ranks() -> bool:
    """True when the multi-rank arrangement of the rollout applies: more than one rank — or, for rehearsals on one GPU, an
    initialised process group of ONE rank with FPV_REHEARSE_RANKS=1 in the environment: the collector then takes every step of
    the N > 1 path (side stream, the real collective, advantage statistics from the gathered payload, rank merge of the
    VecNormalize statistics) with G = 1, whose results must equal the one-rank path's (tools/rccl_rehearsal.py)."""
    if world_size() > 1:
        return True
    return td.is_available() and td.is_initialized() and os.environ.get("FPV_REHEARSE_RANKS", "") == "1"


def rank() -> int:
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def allgather_payload(payload: torch.Tensor) -> torch.Tensor:
    """The one collective per update: gather each rank's 1-D fp64 payload.  Returns [world, len]."""
    if payload.dtype != torch.float64 or payload.dim() != 1:
        raise ValueError("payload must be a 1-D float64 tensor")
    w = world_size()
    if w == 1 and not several_ranks():
        return payload.unsqueeze(0)
    if td.get_backend() == "gloo" and payload.is_cuda:
        # rehearsal mode (several ranks on one GPU box over gloo): stage the few words through the host
        host = payload.cpu()
        out = torch.empty((w * host.numel(),), dtype=torch.float64)
        td.all_gather_into_tensor(out, host.contiguous())
        return out.view(w, host.numel()).to(payload.device)
    out = torch.empty((w * payload.numel(),), dtype=torch.float64, device=payload.device)
    td.all_gather_into_tensor(out, payload.contiguous())   # flat output: accepted by RCCL and gloo alike
    return out.view(w, payload.numel())


def merge_advantage_stats(gathered: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """rows = {count, sum, sum_sq} per rank -> global (mean, std) with SB3's unbiased=False std + 1e-8
    applied by the caller.  Summation is in rank order, so every rank computes identical bits."""
    tot = gathered[:, :3].sum(dim=0)
    mean = tot[1] / tot[0]
    var = torch.clamp(tot[2] / tot[0] - mean * mean, min=0.0)
    return mean, torch.sqrt(var)


def merge_moments(gathered_mean: torch.Tensor, gathered_var: torch.Tensor, gathered_count: torch.Tensor):
    """Parallel-variance merge of per-rank RunningMeanStd states ([world, d], [world, d], [world])."""
    mean, var, count = gathered_mean[0].clone(), gathered_var[0].clone(), gathered_count[0].clone()
    for r in range(1, gathered_mean.shape[0]):
        bm, bv, bc = gathered_mean[r], gathered_var[r], gathered_count[r]
        delta = bm - mean
        tot = count + bc
        m2 = var * count + bv * bc + delta * delta * count * bc / tot
        mean = mean + delta * bc / tot
        var, count = m2 / tot, tot
    return mean, var, count


def merge_rank_statistics(base: torch.Tensor, after: torch.Tensor) -> torch.Tensor:
    """Host-side statement of fpv_rank_merge (csrc/fpv_rollout.hip) for one RunningMeanStd: base = {mean[d], var[d], count}
    that every rank started the rollout from, after [world, 2 d + 1] = what each rank holds at its end.  Each rank's batch is
    recovered from (base, after_r) and merged into base in rank order -> the statistics of the union of all ranks' samples."""
    d = (base.numel() - 1) // 2
    m0, v0, c0 = base[:d].double(), base[d:2 * d].double(), base[2 * d].double()
    mean, var, count = m0.clone(), v0.clone(), c0.clone()
    for r in range(after.shape[0]):
        m1, v1, c1 = after[r, :d].double(), after[r, d:2 * d].double(), after[r, 2 * d].double()
        cb = c1 - c0
        if not bool(cb > 0):
            continue
        mb = (m1 * c1 - m0 * c0) / cb
        db = mb - m0
        vb = torch.clamp((v1 * c1 - v0 * c0 - db * db * c0 * cb / c1) / cb, min=0.0)
        delta, tot = mb - mean, count + cb
        m2 = var * count + vb * cb + delta * delta * count * cb / tot
        mean, var, count = mean + delta * cb / tot, m2 / tot, tot
    return torch.cat([mean, var, count.reshape(1)])
