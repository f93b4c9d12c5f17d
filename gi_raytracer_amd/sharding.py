"""Row-stripe sharding of one frame over the GPUs of a node, and the RCCL gather of the stripes.

The reference parallelises the frame over image rows with `#pragma omp parallel for schedule(dynamic, 10)`
(include/raytracer.h:93): rows are independent given the read-only scene and photon map.  Here the rows are cut into stripes of
STRIPE_H rows dealt round-robin to the ranks (interleaved, because the cost of a row depends strongly on what it sees); every
rank renders its stripes into one compact [local_rows][w][3] buffer; one gather (RCCL over xGMI when the tensors live on the
GPUs, gloo in the CPU tests) brings them to rank 0, which puts the rows back in frame order.  No other collective is needed.
"""
import numpy as np

STRIPE_H = 16


def stripe_rows(h, stripe_h, rank, world):
    """Frame rows rendered by `rank`, in the order they appear in its local buffer (include/gi_hip.h: gi_render_params)."""
    rows = []
    n_stripes = (h + stripe_h - 1) // stripe_h
    for k in range(rank, n_stripes, world):
        rows.extend(range(k * stripe_h, min((k + 1) * stripe_h, h)))
    return np.array(rows, int)


def max_local_rows(h, stripe_h, world):
    return max(len(stripe_rows(h, stripe_h, r, world)) for r in range(world))


class FrameGather:
    """Pre-allocated buffers for gathering the stripes of a w x h frame to rank 0."""

    def __init__(self, torch, dist, w, h, stripe_h, rank, world, device, dtype, always_collective=False):
        """always_collective: issue the gather even when world == 1 (a group of one rank), so that a one-GPU box exercises the RCCL call."""
        self.torch, self.dist, self.rank, self.world, self.h = torch, dist, rank, world, h
        self.collective = world > 1 or (always_collective and dist is not None)
        self.n_collectives = 0
        self.rows = [stripe_rows(h, stripe_h, r, world) for r in range(world)]
        self.pad_rows = max(len(r) for r in self.rows)
        self.local = torch.zeros((self.pad_rows, w, 3), dtype=dtype, device=device)    # render target of this rank (padded)
        self.parts = [torch.zeros_like(self.local) for _ in range(world)] if (rank == 0 and self.collective) else None
        self.frame = torch.zeros((h, w, 3), dtype=dtype, device=device) if rank == 0 else None
        self.index = [torch.as_tensor(r, device=device) for r in self.rows] if rank == 0 else None

    def gather(self):
        """One collective: every rank's padded stripe buffer to rank 0, then rows back into frame order."""
        if not self.collective:
            self.frame[:] = self.local[: self.h]
            return self.frame
        self.n_collectives += 1
        if self.dist.get_backend() == "gloo" and self.local.is_cuda:   # rehearsal on one device: gloo gathers host tensors
            host = self.local.cpu()
            parts = [self.torch.zeros_like(host) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(host, parts, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.parts[r].copy_(parts[r])
        else:
            self.dist.gather(self.local, self.parts, dst=0)
        if self.rank == 0:
            for r in range(self.world):
                self.frame[self.index[r]] = self.parts[r][: len(self.rows[r])]
        return self.frame
