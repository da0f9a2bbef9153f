#!/usr/bin/env python3
"""bench.py -- residual+Jacobian evals/s and ESKF-iterations/s of the scan-to-map engine.

A "step" is one full iterated ESKF update of one scan (eskf_lio/src/laserMapping.cpp:820-1102)
with the reference's rematch schedule: BASELINE.json configs[2] -- a 64x1024 = 65,536-point
synthetic scan against a 5,000,000-point map, max_iteration = 5 -- with map and scan already
resident in HBM.  ``value`` = scan points x passes executed / wall time, over all ranks.

N > 1 (launched by torch.distributed.run): the scan grows to N x 65,536 points (N x 64 beams),
each rank holds a contiguous 65,536-point shard and a replica of the map, and the 158-double
normal block is summed with one RCCL all-reduce per iteration ("weak" scaling: per-GPU work is
fixed).  ``--mode replicas`` runs BASELINE configs[4] instead (independent scans, no collective).

Extra objects: ``roofline`` for the dominant kernel (the kNN + plane-fit match kernel; HIP-event
timed inside the engine over the timed region) and ``cpu_baseline`` (the CPU oracle, 1 thread,
rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
BYTES_REMATCH = 88              # 12 B scan point + 5 x 12 B neighbours + 16 B plane
BYTES_REUSE = 28                # 12 B scan point + 16 B cached plane


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4"])
    ap.add_argument("--mode", default="sharded", choices=["sharded", "replicas"])
    ap.add_argument("--max-iter", type=int, default=5)
    ap.add_argument("--cell", type=float, default=0.0)
    ap.add_argument("--cpu-steps", type=int, default=8, help="CPU baseline sample: iterated updates (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU testing)")
    ap.add_argument("--all-on-device0", action="store_true", help="test hook: every rank uses GPU 0")
    ap.add_argument("--beams-mult", type=int, default=1, help="test hook: single rank over a denser scan")
    ap.add_argument("--extrinsic", action="store_true",
                    help="extrinsic_est_en = true: 12-column Jacobian rows, 92-term normal block (SURVEY 8d's extra run)")
    ap.add_argument("--torch-collective", action="store_true",
                    help="sum the block with torch.distributed.all_reduce through the C callback instead of the "
                         "engine's own RCCL communicator")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the sharded/all-reduce code path even with one rank (test hook)")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.all_on_device0:
        local_rank = 0
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    from daliti_amd import Engine, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 or a.force_collective:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cfgd = synth.CONFIGS[a.config]
    sharded = (world > 1 or a.force_collective) and a.mode == "sharded"
    # workload: map replicated; scan = world x (beams x az) points when sharded
    t0 = time.time()
    map_xyz = synth.make_map(cfgd["M"], cfgd["L"], seed=1)
    if sharded:
        scan_all = synth.make_scan(cfgd["beams"] * world, cfgd["az"], cfgd["L"], seed=2)
        per = cfgd["beams"] * cfgd["az"]
        scan = scan_all[rank * per:(rank + 1) * per]
        pos = synth.SENSOR_POS
    else:
        dx = (rank - (world - 1) / 2.0) * 2.0 if world > 1 else 0.0
        pos = synth.SENSOR_POS + np.array([dx, 0.0, 0.0])
        scan = synth.make_scan(cfgd["beams"] * a.beams_mult, cfgd["az"], cfgd["L"], seed=2 + rank, sensor_pos=pos)
    x_true, x_prop, P0 = synth.filter_inputs(pos)
    n_local = len(scan)
    t_gen = time.time() - t0

    eng = Engine(max_iter=a.max_iter, cell_size=a.cell, device=local_rank, feat_threshold=100,
                 extrinsic_est_en=int(a.extrinsic))
    # one explicit (non-default) stream shared by the engine's kernels and torch's collectives: RCCL orders
    # its work against torch's *current* stream, so the all-reduce of a block is only correctly ordered
    # after the kernel that wrote it if both are issued under this stream
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    eng.set_stream(stream.cuda_stream)
    # inputs resident in HBM before the timed region: hand the engine device pointers
    d_map = torch.from_numpy(map_xyz).cuda()
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    torch.cuda.synchronize()
    t0 = time.time()
    eng.map_build_device(d_map.data_ptr(), 3, len(map_xyz))
    torch.cuda.synchronize()
    t_build = time.time() - t0
    eng.scan_set_device(d_scan.data_ptr(), 3, n_local)
    info = eng.map_info()

    blk = torch.zeros(160, dtype=torch.float64, device="cuda")
    builtin_comm = False
    if sharded and a.backend == "nccl" and not a.torch_collective:
        # the engine's own RCCL communicator: the all-reduce is issued from the C++ loop on the engine's
        # stream; torch.distributed only ships the 128-byte unique id
        # every rank first proves it can reach RCCL (a rank that cannot must not leave the others blocked
        # inside ncclCommInitRank); any failure sends all ranks to the torch.distributed callback instead
        def all_ok(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        try:
            probe = Engine.comm_unique_id()
        except Exception as ex:  # noqa: BLE001 - reported, then the fallback path is taken
            probe = None
            sys.stderr.write("rank %d: engine RCCL communicator unavailable (%s)\n" % (rank, ex))
        if all_ok(probe is not None):
            ids = [probe if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                eng.comm_init(ids[0], world, rank)
                ok = True
            except Exception as ex:  # noqa: BLE001
                ok = False
                sys.stderr.write("rank %d: s2m_comm_init failed (%s)\n" % (rank, ex))
            builtin_comm = all_ok(ok)
    if sharded and not builtin_comm:
        from daliti_amd.sharding import allreduce_block

        def reduce_cb():
            allreduce_block(blk)

        def step():
            eng.set_feat_queue([])
            return eng.iterated_update_sharded(x_prop, x_prop, P0, blk.data_ptr(), reduce_cb)
    if not sharded or builtin_comm:
        from daliti_amd.engine import IterLog
        xb, xpb, Pb, logb = np.zeros(36), np.ascontiguousarray(x_prop, np.float64), np.zeros((24, 24)), IterLog()

        def step():
            # lean call: preallocated buffers, no per-step conversions (the degeneracy queue is cleared so
            # every step is the same scan arriving fresh)
            eng.set_feat_queue(())
            xb[:] = xpb
            Pb[:] = P0
            eng.iterated_update_raw(xb, xpb, Pb, logb)
            return logb

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        res = step()
    eng.set_timing(5)  # HIP-event time every 5th rematch pass of the timed region (odd: covers both kinds)
    fence()
    t0 = time.perf_counter()
    passes = iters = rematch = 0
    for _ in range(a.steps):
        res = step()
        if sharded and not builtin_comm:
            iters += res["iters"]
            rematch += res["rematch_passes"]
        else:
            iters += res.iters
            rematch += res.rematch_passes
    fence()
    dt = time.perf_counter() - t0
    if not sharded or builtin_comm:  # final state / log of the last step for the report
        res = dict(x=xb.copy(), P=Pb.copy(), iters=res.iters, effct=np.array(res.effct[:res.iters]))
    tstats = eng.timing_stats()
    eng.set_timing(False)
    passes = iters  # one residual pass per iteration
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    evals_total = float(n_local) * passes * world
    value = evals_total / dt
    pose_err = float(np.abs(res["x"][9:12] - x_true[9:12]).max())

    out = {
        "metric": "residual+Jacobian evals/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 per-point / f64 normal block + ESKF",
        "data": "synthetic (seeded closed-box map, ray-cast scan; SURVEY.md 8d)",
        "config": {
            "workload": "%s: full iterated ESKF (max_iter %d%s), %d-pt scan%s vs %d-pt map" % (
                a.config, a.max_iter, ", extrinsic_est_en" if a.extrinsic else "", n_local,
                " shard" if sharded else "", len(map_xyz)),
            "scan_points_per_gpu": n_local,
            "map_points": len(map_xyz),
            "parallelism": ("single GPU" if (world == 1 and not sharded) else
                            ("scan points sharded x%d, map replicated, RCCL all-reduce of 158 f64 per iteration (%s)"
                             % (world, "engine-owned communicator" if builtin_comm else "torch.distributed callback")
                             if sharded else "replicas x%d (independent scans, no collective)" % world)),
            "cell_size_m": info["cell"],
            "mean_points_per_cell": info["mean_per_cell"],
        },
        "eskf_iters_per_sec": iters * (world if not sharded else 1) / dt,
        "scans_per_sec": a.steps * (world if not sharded else 1) / dt,
        "iters_per_step": iters / a.steps,
        "rematch_passes_per_step": rematch / a.steps,
        "pose_error_vs_truth_m": pose_err,
        "final_pos": [float(v) for v in res["x"][9:12]],
        "map_build_s": t_build,
    }
    # roofline of the dominant kernel: the match (kNN + plane fit) kernel, one launch per rematch pass
    if tstats["match_launches"] > 0:
        ms = tstats["match_ms"] / tstats["match_launches"]
        achieved = n_local * BYTES_REMATCH / (ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic() if (a.config == "C3" and world == 1) else (None, None)
        copy_peak = measured_copy_peak(torch) if (rank == 0 and not a.no_cpu) else None  # --no-cpu: no side legs
        out["roofline"] = {
            "kernel": "match_kernel (exact 5-NN on the brick grid + plane fit)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_eval": BYTES_REMATCH, "evals_per_launch": n_local,
            "avg_launch_ms": ms, "launches": tstats["match_launches"],
            # both fractions, and the box's own copy peak as a second denominator
            "frac_of_measured_traffic": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "peak_measured_copy": copy_peak,
            "frac_of_measured_peak": (achieved / copy_peak) if copy_peak else None,
            "note": "one launch = match_easy + match_hard of one rematch pass (HIP events around both, "
                    "every 5th rematch pass of the timed region sampled)",
        }
    if rank == 0 and world == 1 and not a.no_cpu and a.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(a, map_xyz, scan, x_prop, P0, res)
        out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"]
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        sys.stderr.write("[bench] gen %.1fs, map build %.3fs, cell %.3f m (%.2f pts/cell), %d bricks\n" % (
            t_gen, t_build, info["cell"], info["mean_per_cell"], info["bricks"]))
        # RCCL prints a version banner through C stdio; flush it first so the JSON is the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def measured_copy_peak(torch, nbytes=1 << 30, reps=10):
    """Device-to-device copy of 1 GiB, read + written bytes per second in GB/s (the box's own HBM peak)."""
    try:
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    except Exception:  # noqa: BLE001 - a diagnostic, never fatal
        return None


def pmc_traffic():
    """HBM traffic of one rematch pass of the match kernels from the newest committed PMC summary
    (profiles/*_pmc.json, made by scripts/profile_round.sh + summarize_profile.py: separate
    rocprofv3 --pmc passes of this same command).  Read side doubled as MI355X_MICROARCH.md
    prescribes for gfx950 (FETCH_SIZE tallies 128-B requests at 64 B), so this is an upper bound."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"]
        tot = 0.0
        for name in ("match_easy", "match_hard"):
            tot += (2.0 * k[name]["FETCH_SIZE_avg"] + k[name]["WRITE_SIZE_avg"]) * 1024.0
        return tot, os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError):
        return None, None


def cpu_baseline(a, map_xyz, scan, x_prop, P0, gpu_res):
    """The CPU oracle (port of the reference path, 1 thread like the reference) on the same
    workload; also cross-checks the GPU pose against it."""
    import oracle
    t0 = time.perf_counter()
    tree = oracle.KdTree(map_xyz)
    t_build = time.perf_counter() - t0
    cfg = oracle.default_cfg(max_iter=a.max_iter, nthreads=1, extrinsic_est_en=int(a.extrinsic))
    t0 = time.perf_counter()
    iters = 0
    for _ in range(a.cpu_steps):
        r = oracle.iterated_update(cfg, tree, scan, x_prop, x_prop, P0)
        iters += r["iters"]
    dt = time.perf_counter() - t0
    # generous variant: the same port with OpenMP over scan points on many host cores
    nthr = max(1, min(os.cpu_count() or 1, 64))
    cfg_mt = oracle.default_cfg(max_iter=a.max_iter, nthreads=nthr, extrinsic_est_en=int(a.extrinsic))
    oracle.iterated_update(cfg_mt, tree, scan, x_prop, x_prop, P0)
    t1 = time.perf_counter()
    it_mt = 0
    for _ in range(a.cpu_steps):
        it_mt += oracle.iterated_update(cfg_mt, tree, scan, x_prop, x_prop, P0)["iters"]
    dt_mt = time.perf_counter() - t1
    dpos = float(np.abs(r["x"][9:12] - gpu_res["x"][9:12]).max())
    dR = r["x"][:9].reshape(3, 3).T @ gpu_res["x"][:9].reshape(3, 3)
    drot = float(np.abs(oracle.so3_log(dR)).max())
    return {
        "value": len(scan) * iters / dt, "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": "%d full iterated updates of the same %d-pt scan vs the %d-pt map (k-d tree build %.1f s untimed)"
                  % (a.cpu_steps, len(scan), len(map_xyz), t_build),
        "eskf_iters_per_sec": iters / dt, "ms_per_step": 1e3 * dt / a.cpu_steps,
        "host_cpus": os.cpu_count(),
        "all_cores_variant": {"value": len(scan) * it_mt / dt_mt, "unit": "evals/s", "cores": nthr,
                              "ms_per_step": 1e3 * dt_mt / a.cpu_steps,
                              "note": "OpenMP over scan points; not the reference's configuration "
                                      "(its pragmas are commented out, laserMapping.cpp:827-828)"},
        "gpu_vs_cpu_pose_delta_m": dpos, "gpu_vs_cpu_pose_delta_rad": drot,
        "effct_equal": bool((r["effct"] == gpu_res["effct"]).all()) if len(r["effct"]) == len(gpu_res["effct"]) else False,
    }


if __name__ == "__main__":
    main()
