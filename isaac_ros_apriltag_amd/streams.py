"""Multi-GPU front end: independent camera streams sharded across the GPUs of one node.

Frames (and streams) are independent units (reference isaac_ros_apriltag/src/apriltag_node.cpp:613-623
handles one frame at a time with no cross-frame state), so the data path needs no collective: stream s
is owned by rank s % world_size.  The only exchange is one broadcast of the per-stream parameter block
(intrinsics, tag size, decimation, image size) from rank 0 at start-up -- RCCL over xGMI when the
backend is "nccl", gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

PARAM_FIELDS = ("fx", "fy", "cx", "cy", "tag_size", "decimate", "width", "height", "seed")


def assign_streams(num_streams, world_size, rank):
    """Streams owned by `rank` (round-robin)."""
    return [s for s in range(num_streams) if s % world_size == rank]


def make_param_block(num_streams, width=1920, height=1080, decimate=1, tag_size=0.22, base_seed=1234):
    """Rank-0 view of the per-stream parameters (distinct intrinsics per stream,
    fx, fy in [900, 1400], seeds 1234 + 1000*s)."""
    blk = np.zeros((num_streams, len(PARAM_FIELDS)), dtype=np.float64)
    for s in range(num_streams):
        f = 900.0 + 500.0 * ((s * 37) % 8) / 7.0
        blk[s] = (f, f + 10.0 * (s % 3), width / 2.0 + s, height / 2.0 - s, tag_size, decimate, width, height,
                  base_seed + 1000 * s)
    return blk


def broadcast_param_block(block, num_streams, device="cpu", src=0):
    """Every rank returns the block rank `src` holds.  `block` may be None on the other ranks."""
    t = torch.zeros((num_streams, len(PARAM_FIELDS)), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        if dist.get_rank() == src:
            t.copy_(torch.from_numpy(np.asarray(block, dtype=np.float64)))
        dist.broadcast(t, src=src)
    else:
        t.copy_(torch.from_numpy(np.asarray(block, dtype=np.float64)))
    return t.cpu().numpy()


def stream_params(block, s):
    return dict(zip(PARAM_FIELDS, [float(v) for v in block[s]]))
