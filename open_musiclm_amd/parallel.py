"""Data-parallel runtime: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / xGMI.

The reference uses HF accelerate -> torch DDP, which all-reduces bucketed gradients on EVERY micro-batch
backward (trainer.py:154-155,439; no no_sync) -- 8 collectives of 366 MB per optimizer step for coarse-small.
Here the backward kernels accumulate into one flat fp32 gradient buffer (optimizer.FusedAdam) and the whole
step does ONE SUM all-reduce of that buffer; the 1/world_size mean is folded into the fused optimizer kernel.
Validation-only collectives (scalar mean, all-gather of predictions) mirror trainer.py:470-473.
"""
from __future__ import annotations

import os
import sys
from typing import Optional

import torch
import torch.distributed as dist


def local_rank_count(world_size: int) -> int:
    """Ranks per node as the launcher states it (rank-independent by construction: every rank reads the same variables)."""
    for k in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"):
        v = os.environ.get(k, "")
        if v.split("(")[0].isdigit():                # (Slurm writes e.g. "8(x2)")
            return int(v.split("(")[0])
    nn = os.environ.get("SLURM_NNODES", os.environ.get("SLURM_JOB_NUM_NODES", ""))
    if nn.isdigit() and int(nn) > 0:
        return -(-world_size // int(nn))
    return world_size


class DataParallel:
    def __init__(self, device: Optional[torch.device] = None, backend: Optional[str] = None):
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.owns_group = False
        self.shared_gpu = False
        if self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            use_cuda = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
            if use_cuda:
                # one GPU per rank; a rank count above the visible GPUs (dry runs of the multi-process path on a 1-GPU box,
                # gloo exchange) shares devices round-robin -- RCCL itself refuses two ranks on one GPU
                torch.cuda.set_device(self.local_rank % max(torch.cuda.device_count(), 1))
            backend = backend or os.environ.get("OMLM_DP_BACKEND")
            # Do the ranks of this node outnumber its GPUs (a dry run: RCCL refuses two ranks on one device)?  The answer must be the SAME on
            # every rank -- a per-rank test (LOCAL_RANK >= devices) sent ranks 0-1 to nccl and ranks 2-3 to gloo on a 4-rank / 2-GPU launch
            # and init_process_group hung -- so it is taken from launch-wide quantities only: the local rank count the launcher
            # exports (torchrun: LOCAL_WORLD_SIZE; Open MPI: OMPI_COMM_WORLD_LOCAL_SIZE; Slurm: SLURM_NTASKS_PER_NODE, or WORLD_SIZE over
            # SLURM_NNODES); without any of them WORLD_SIZE itself counts as the local rank count (a multi-node launch that exports none of
            # these must set LOCAL_WORLD_SIZE or OMLM_DP_BACKEND).
            ndev = torch.cuda.device_count() if use_cuda else 0
            self.shared_gpu = bool(use_cuda and local_rank_count(self.world_size) > ndev)
            if backend is None:
                backend = "nccl" if use_cuda else "gloo"
                if self.shared_gpu:
                    # more local ranks than visible GPUs: a dry run of the multi-process path (RCCL refuses two ranks on one device)
                    backend = "gloo"
                    if self.rank == 0:
                        print(f"open_musiclm_amd.parallel: {self.world_size} ranks on {torch.cuda.device_count()} visible GPU(s) -- "
                              "ranks share devices and exchange through gloo (a dry run of the data-parallel path, not an RCCL measurement)",
                              file=sys.stderr, flush=True)
            # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; the HSA runtime reads it when it initialises) is exported at package import
            # (open_musiclm_amd/__init__.py): too late here, the model is already on the GPU.  Fail loudly if something overrode it.
            if backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") != "0":
                raise RuntimeError("HSA_ENABLE_IPC_MODE_LEGACY must be 0 for RCCL on this platform (dmabuf IPC only); "
                                   "export it in the launch environment")
            dist.init_process_group(backend=backend,
                                    rank=self.rank, world_size=self.world_size)
            self.owns_group = True
        if dist.is_initialized():
            self.world_size, self.rank = dist.get_world_size(), dist.get_rank()

    @property
    def is_distributed(self):
        return self.world_size > 1

    @property
    def is_main(self):
        return self.rank == 0

    @property
    def is_local_main(self):
        return self.local_rank == 0

    def allreduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        """THE gradient exchange: one SUM all-reduce of the flat buffer (mean is applied by the optimizer kernel).

        $OMLM_DP_GRAD_DTYPE=bf16 sends the gradients as bf16 (half the bytes on the xGMI links: 183 MB instead of 366 MB for
        coarse-small; one rounding of each rank's gradient to 8 significand bits and a bf16 ring sum -- replicas stay identical
        because every rank receives the same reduced values, but the step no longer equals the single-process accumulation bit for
        bit, so it is off by default)."""
        if not self.is_distributed:
            return flat
        nb = self.bucket_count(flat)
        if nb > 1 and os.environ.get("OMLM_DP_GRAD_DTYPE", "fp32") != "bf16":
            return self.allreduce_buckets_(flat, nb)
        if os.environ.get("OMLM_DP_GRAD_DTYPE", "fp32") == "bf16":
            buf = getattr(self, "_xbuf", None)
            if buf is None or buf.numel() != flat.numel() or buf.device != flat.device:
                buf = self._xbuf = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            buf.copy_(flat)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            flat.copy_(buf)
            return flat
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    # ---- gradient buckets --------------------------------------------------------------------------------------------------
    # The flat buffer as contiguous buckets of $OMLM_DP_BUCKET_MB (default 0 = one collective), each its own asynchronous SUM all-reduce:
    # element i of the result is the sum over ranks of element i whatever the cut, so the buckets give what the single collective gives
    # (bit for bit on 2 ranks, where a + b is the only order; on more ranks a ring reduces each chunk in its own rank order, so the
    # rounding of an element may differ between two cuts -- like between two NCCL versions -- while every rank still receives identical
    # values).  What the cut buys: a bucket can leave as soon as its gradients are final.  Today that is the end of the backward for
    # all of them (the weight gradients are ONE grouped launch behind the last layer: csrc/gemm.hip, 3.8 machine rounds instead of 30
    # split-K GEMMs), so the buckets queue back to back on the communicator's stream and the host call returns when the last is done --
    # the same wire time as one collective; `bucket_ranges` is what a segmented backward hands over bucket by bucket.
    def bucket_count(self, flat: torch.Tensor) -> int:
        mb = float(os.environ.get("OMLM_DP_BUCKET_MB", "0") or 0)
        if mb <= 0:
            return 1
        return max(1, -(-flat.numel() * flat.element_size() // int(mb * (1 << 20))))

    @staticmethod
    def bucket_ranges(numel: int, nb: int, align: int = 1024):
        """nb contiguous [start, end) element ranges covering [0, numel), cut at multiples of `align` elements."""
        per = -(-numel // nb)
        per = -(-per // align) * align
        return [(s, min(numel, s + per)) for s in range(0, numel, per)]

    def allreduce_buckets_(self, flat: torch.Tensor, nb: int) -> torch.Tensor:
        works = [dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, async_op=True) for s, e in self.bucket_ranges(flat.numel(), nb)]
        for w in works:
            w.wait()            # (nccl: makes the current stream wait for the collective; gloo: blocks the host)
        return flat

    def grad_scale(self) -> float:
        return 1.0 / self.world_size

    def reduce_mean(self, t: torch.Tensor) -> torch.Tensor:
        if self.is_distributed:
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t /= self.world_size
        return t

    def all_gather_cat(self, t: torch.Tensor) -> torch.Tensor:
        if not self.is_distributed:
            return t
        out = [torch.empty_like(t) for _ in range(self.world_size)]
        dist.all_gather(out, t.contiguous())
        return torch.cat(out, dim=0)

    def broadcast_(self, t: torch.Tensor, src: int = 0):
        if self.is_distributed:
            dist.broadcast(t, src=src)
        return t

    def barrier(self):
        if self.is_distributed:
            dist.barrier()

    def print(self, *a, **k):
        if self.is_main:
            print(*a, **k)

    def shutdown(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()
