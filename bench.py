#!/usr/bin/env python3
"""bench.py -- AprilTag detections throughput at 1920x1080 (BASELINE.json config 2) on N GPUs.

A "step" is one pass of the whole hot path (threshold -> union-find CC -> clustering -> quad fit ->
decode -> pose) over one batch of B device-resident synthetic 1080p frames per GPU; detections are on
the host when a step ends.  One process per GPU; ranks own independent camera streams (weak scaling,
no data-path collective); rank 0 broadcasts the per-stream parameter block once (RCCL).

Prints ONE JSON line on rank 0.  Besides the driver's contract it carries
  roofline     -- the threshold pass (the kernel BASELINE.json's metric names), HIP-event timed
  cpu_baseline -- the CPU restatement (oracle/) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from isaac_ros_apriltag_amd import streams, synth  # noqa: E402
from isaac_ros_apriltag_amd.detector import AprilTagDetector  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def render_stream(seed, nframes, sigma):
    frames, truths = [], []
    for i in range(nframes):
        img, _, truth = synth.scene_c2(seed=seed + i, sigma=sigma)
        frames.append(img)
        truths.append(truth)
    return np.stack(frames), truths


def cpu_baseline(frames, K, decimate, tag_size, budget_s=15.0):
    """CPU restatement on a bounded sample, frame-parallel over the host cores (ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_util as pu
    from oracle import pyoracle as po
    po.lib()
    prm = pu.oracle_params(K, decimate, tag_size)
    t0 = time.perf_counter()
    ref = [po.detect(frames[0], params=prm)[0]]
    t1 = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    n = int(max(cores, min(len(frames) * 4, budget_s * cores / max(t1, 1e-3))))
    idx = [i % len(frames) for i in range(n)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(lambda i: po.detect(frames[i], params=prm)[0], idx))
    dt = time.perf_counter() - t0
    byframe = {}
    for i, r in zip(idx, res):
        byframe[i] = r
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d x 1080p frames (%d distinct), oracle/apriltag_oracle.c -O2, %d threads frame-parallel; "
                      "single thread %.2f frames/s" % (n, len(frames), cores, 1.0 / t1),
            "single_thread_fps": round(1.0 / t1, 3)}, byframe


def threshold_roofline(frames_dev, width, height, decimate, reps=20):
    """Threshold pass alone over a batch whose in+out footprint exceeds the 256 MiB LLC."""
    nrep = int(np.ceil(160 / frames_dev.shape[0]))
    big = frames_dev.repeat(nrep, 1, 1).contiguous()
    nb = big.shape[0]
    det = AprilTagDetector(width, height, decimate=decimate, max_batch=nb, max_points=4096, hash_slots=256,
                           max_clusters=256, max_quads=64, max_detections=16)
    det.set_profiling(True)
    for _ in range(3):
        det.threshold_only(big)
    ms = []
    for _ in range(reps):
        det.threshold_only(big)
        ms.append(det.stage_ms()["threshold"])
    det.close()
    w, h = 1 + (width - 1) // decimate, 1 + (height - 1) // decimate
    alg_bytes = 2.0 * w * h * nb
    avg_ms = float(np.mean(ms))
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    # HBM traffic per launch comes from separate rocprofv3 --pmc passes of the same launch shape
    # (tools/thr_only.py; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), committed under
    # profiles/; bench.py cannot collect PMC counters itself.
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r01_threshold_pmc.json")
    if decimate == 1 and os.path.exists(pmc):
        rec = json.load(open(pmc))
        if rec.get("frames_per_launch") == nb:
            traffic, traffic_src = rec["traffic_bytes"], "profiles/r01_threshold_pmc.json"
    return {"bound": "hbm", "kernel": "k_threshold<%d>" % decimate, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4), "min_launch_ms": round(float(np.min(ms)), 4),
            "frames_per_launch": nb, "footprint_mib": round((big.numel() * 2) / 2 ** 20, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU (BASELINE.md protocol sweeps 1, 8, 64, 256)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct rendered frames per stream")
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--decimate", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-clean", action="store_true", help="skip the informational sigma=0 measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    W, H = 1920, 1080
    nstreams = world
    block = streams.make_param_block(nstreams, W, H, args.decimate) if rank == 0 else None
    block = streams.broadcast_param_block(block, nstreams, device=dev)   # RCCL broadcast of intrinsics
    mine = streams.assign_streams(nstreams, world, rank)
    sp = streams.stream_params(block, mine[0])
    K = np.array([[sp["fx"], 0, sp["cx"]], [0, sp["fy"], sp["cy"]], [0, 0, 1]])

    frames_np, truths = render_stream(int(sp["seed"]), args.distinct, args.sigma)
    frames_dev = torch.from_numpy(frames_np).to(dev)
    reps = int(np.ceil(args.batch / args.distinct))
    batch = frames_dev.repeat(reps, 1, 1)[:args.batch].contiguous()

    det = AprilTagDetector(W, H, families=("tag36h11",), decimate=args.decimate,
                           intrinsics=(sp["fx"], sp["fy"], sp["cx"], sp["cy"]), tag_size=sp["tag_size"],
                           max_batch=args.batch, device=local_rank)
    prep = det.prepare(batch, max_dets=64)   # marshalling once; a step is exactly one blocking C-ABI call
    for _ in range(args.warmup):
        det.run_prepared(prep)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        det.run_prepared(prep)
    barrier()
    dt = time.perf_counter() - t0
    out = det.unpack(prep)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    flags = det.frame_flags(args.batch)
    det.set_profiling(True)
    det.run_prepared(prep)
    stage_ms = det.stage_ms()
    det.set_profiling(False)
    # secondary workload: the same scenes without background noise (informational, not `value`)
    clean = None
    if args.sigma > 0 and not args.no_clean:
        cf, _ = render_stream(int(sp["seed"]), min(args.distinct, 8), 0.0)
        cbatch = torch.from_numpy(cf).to(dev).repeat(int(np.ceil(args.batch / cf.shape[0])), 1, 1)[:args.batch].contiguous()
        cprep = det.prepare(cbatch, max_dets=64)
        det.run_prepared(cprep)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            det.run_prepared(cprep)
        clean = args.batch * args.steps / (time.perf_counter() - tc)
    det.close()
    # informational: same noisy frames at AprilRobotics' default quad_decimate = 2
    dec2 = None
    if args.decimate == 1 and not args.no_clean:
        d2 = AprilTagDetector(W, H, families=("tag36h11",), decimate=2, intrinsics=(sp["fx"], sp["fy"], sp["cx"], sp["cy"]),
                              tag_size=sp["tag_size"], max_batch=args.batch, device=local_rank)
        p2 = d2.prepare(batch, max_dets=64)
        d2.run_prepared(p2)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            d2.run_prepared(p2)
        dec2 = args.batch * args.steps / (time.perf_counter() - tc)
        d2.close()

    if rank == 0:
        fps = world * args.batch * args.steps / dt
        ndet = [len(o) for o in out]
        rec = {
            "metric": "AprilTag detections fps @1080p tag36h11, 1/2/4/8 GPU; HBM GB/s on threshold pass",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 (threshold/CC/clustering), f64 (quad fit/decode/pose)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 1920x1080 mono8, 10 tag36h11 per frame (ids 0-9), "
                                   "background 150 + noise sigma=%g, %d distinct frames per stream cycled, "
                                   "device-resident" % (args.sigma, args.distinct),
                       "frames_per_step_per_gpu": args.batch, "decimate": args.decimate, "streams": nstreams,
                       "parallelism": "%d independent stream(s), 1 per GPU, RCCL broadcast of intrinsics only" % nstreams},
            "detections_per_frame": float(np.mean(ndet)), "frame_flags_nonzero": int(sum(1 for f in flags if f)),
            "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
        }
        if clean is not None:
            rec["fps_per_gpu_same_scenes_sigma0"] = round(clean, 1)
        if dec2 is not None:
            rec["fps_per_gpu_same_frames_decimate2"] = round(dec2, 1)
        byframe = None
        if not args.no_cpu_baseline:
            rec["cpu_baseline"], byframe = cpu_baseline(frames_np, K, args.decimate, sp["tag_size"])
        if not args.no_roofline:
            rec["roofline"] = threshold_roofline(frames_dev, W, H, args.decimate)
        if byframe is not None:
            # correctness gate in the same run: ids exact, corners bit-identical to the CPU restatement
            ok = True
            for i, odets in byframe.items():
                g = out[i]
                ok &= len(g) == len(odets) and all(a["id"] == b["id"] and np.array_equal(a["p"], b["p"]) for a, b in zip(g, odets))
            rec["parity_gate"] = "pass" if ok else "FAIL"
        print(json.dumps(rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
