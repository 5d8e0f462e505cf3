"""Data-parallel exchange of the captioner step: one process per GPU, one RCCL all-reduce
of the flat gradient arena per step over xGMI (backend "nccl" is RCCL on ROCm; "gloo" in the
CPU tests).  Replaces torch.nn.DataParallel's per-step parameter broadcast, output gather to
GPU0 and reduce-add (P/trainer.py:74, SURVEY.md section 2a "Collectives").

Loss normalisation: the reference divides by the mask sum of the WHOLE batch
(P/misc/criterion.py:149).  Each rank therefore scales its rows by 1 / sum_over_ranks(mask sum)
-- a 1-float all-reduce issued before the forward -- and gradients are summed, not averaged.
"""
import torch
import torch.distributed as dist


class GradientExchange(object):
    def __init__(self, group=None):
        self.group = group

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    def global_inv_den(self, den_local, device):
        """1 / (global mask sum) as a device scalar, or None on a single rank (the kernel computes it).

        No host synchronisation: the local sum travels through a ring of pinned host floats and a non-blocking copy
        (a plain `torch.tensor(x, device=...)` is a synchronous pageable copy that would make the host wait for the
        previous step every iteration and forfeit its ~9 steps of run-ahead)."""
        if self.world_size == 1:
            return None
        if torch.is_tensor(den_local) and den_local.is_cuda:
            t = den_local.detach().float().reshape(1).clone()
        elif torch.device(device).type != "cuda":
            t = torch.tensor([float(den_local)], dtype=torch.float32, device=device)
        else:
            if getattr(self, "_pin", None) is None:
                self._pin = torch.zeros(64, dtype=torch.float32).pin_memory()
                self._pin_i = 0
            i = self._pin_i
            self._pin_i = (i + 1) % 64
            self._pin[i] = float(den_local)
            t = self._pin[i:i + 1].to(device, non_blocking=True)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.reciprocal_()

    def allreduce_sum(self, flat):
        """Sum the flat gradient arena over ranks, in place (one collective per step)."""
        if self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def allreduce_sum_overlapped(self, flat, splits, wait_group):
        """Sum `flat` over ranks in len(splits) + 1 collectives.  Piece g = flat[splits[g-1]:splits[g]] starts on a
        communication stream as soon as `wait_group(raw_stream, g)` lets it (uic_topdown_grad_ready_wait: gradient
        group g of the step is final) and so runs beside the rest of the backward pass; the tail flat[splits[-1]:]
        follows on the current stream.  The current stream then waits for the communication stream, so whatever is
        enqueued next (Adam) sees the summed gradients."""
        if self.world_size == 1:
            return flat
        if isinstance(splits, int):
            splits = [splits]
        splits = [int(x) for x in splits]
        ok = flat.is_cuda and splits and all(0 < a < flat.numel() for a in splits) and \
            all(a < b for a, b in zip(splits, splits[1:]))
        if not ok:
            return self.allreduce_sum(flat)
        if getattr(self, "_comm_stream", None) is None or self._comm_stream.device != flat.device:
            self._comm_stream = torch.cuda.Stream(device=flat.device)
        comm = self._comm_stream
        lo = 0
        for g, hi in enumerate(splits):
            wait_group(comm.cuda_stream, g)
            with torch.cuda.stream(comm):
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            lo = hi
        dist.all_reduce(flat[lo:], op=dist.ReduceOp.SUM, group=self.group)
        torch.cuda.current_stream(flat.device).wait_stream(comm)
        return flat

    def allreduce_sum_scalar(self, x):
        if self.world_size > 1:
            x = x.clone()
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x

    def shard_images(self, n_images):
        """Images [lo, hi) of this rank: shard by image so the seq_per_img replicas stay together."""
        w, r = self.world_size, self.rank
        per = (n_images + w - 1) // w
        return min(r * per, n_images), min((r + 1) * per, n_images)
