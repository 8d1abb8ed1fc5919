"""N>1 path on CPU: two gloo ranks broadcast the per-stream parameter block, each rank renders and
detects (CPU oracle here; the HIP path on the GPU box) the streams it owns, and the union covers every
stream exactly once."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from isaac_ros_apriltag_amd import streams


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nstreams = 4
    block = streams.make_param_block(nstreams, 640, 480, decimate=2) if rank == 0 else None
    block = streams.broadcast_param_block(block, nstreams)
    mine = streams.assign_streams(nstreams, world, rank)
    from isaac_ros_apriltag_amd import synth
    from oracle import pyoracle as po
    res = []
    for s in mine:
        sp = streams.stream_params(block, s)
        img, _, _ = synth.scene_c1()
        dets, _ = po.detect(img, params=po.default_params(fx=sp["fx"], fy=sp["fy"], cx=sp["cx"], cy=sp["cy"],
                                                          decimate=int(sp["decimate"]), tag_size=sp["tag_size"]))
        res.append((s, [d["id"] for d in dets], float(dets[0]["t"][2]), sp["fx"]))
    q.put((rank, block.tolist(), res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = streams.make_param_block(4, 640, 480, decimate=2)
    seen = []
    for rank, block, res in out:
        assert np.array_equal(np.array(block), ref)           # every rank holds rank 0's block
        for s, ids, z, fx in res:
            assert s % 2 == rank and ids == [0]
            assert abs(z - fx * 0.22 / 120.0) < 0.02 * z        # depth follows that stream's own intrinsics
            seen.append(s)
    assert sorted(seen) == [0, 1, 2, 3]


def test_assign_streams_partition():
    for world in (1, 2, 4, 8):
        alls = sorted(s for r in range(world) for s in streams.assign_streams(8, world, r))
        assert alls == list(range(8))
