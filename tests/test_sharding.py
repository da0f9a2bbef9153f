"""N > 1 path on CPU: two gloo ranks each reduce a contiguous shard of the scan (via the oracle),
all-reduce the 160-double block with the product's helper and run the update redundantly; the
result must equal the single-rank run to fp64 round-off, and be bit-identical across ranks."""
import os
import sys

import numpy as np

from conftest import bits
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition():
    from daliti_amd.sharding import shard_range
    for n in (0, 1, 7, 2048, 65536, 131072 + 3):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import oracle
    from daliti_amd import synth
    from daliti_amd.sharding import shard_range, allreduce_block
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    sc = synth.make_small()
    tree = oracle.KdTree(sc["map"])
    lo, hi = shard_range(len(sc["scan"]), rank, world)
    scan = sc["scan"][lo:hi]
    cfg = oracle.default_cfg(max_iter=5)
    x, P = sc["x_prop"].copy(), sc["P"].copy()
    ps = oracle.PassState(len(scan))
    log = []
    rematch_en, rematch_num, K1 = False, 0, None
    for it in range(cfg.max_iter):
        rematch = it == 0 or rematch_en
        oracle.residual_pass(cfg, tree, scan, x, rematch, ps)
        blk = torch.zeros(160, dtype=torch.float64)
        blk[:144] = torch.from_numpy(ps.HtH.ravel())
        blk[144:156] = torch.from_numpy(ps.Htz)
        blk[156] = ps.effct
        blk[157] = ps.total_res
        allreduce_block(blk)
        b = blk.numpy()
        HtH, Htz, effct = b[:144].reshape(12, 12).copy(), b[144:156].copy(), int(b[156])
        x, sol, K1, conv = oracle.eskf_update(cfg, x, sc["x_prop"], P, HtH, Htz)
        log.append(effct)
        rematch_en = False
        if conv or (rematch_num == 0 and it == cfg.max_iter - 2):
            rematch_en, rematch_num = True, rematch_num + 1
        if rematch_num >= 2 or it == cfg.max_iter - 1:
            P = oracle.cov_update(K1, HtH, P)
            break
    q.put((rank, x, P, log))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_allreduce_equals_single_rank():
    sys.path.insert(0, ROOT)
    import oracle
    from daliti_amd import synth
    sc = synth.make_small()
    ref = oracle.iterated_update(oracle.default_cfg(max_iter=5), oracle.KdTree(sc["map"]), sc["scan"],
                                 sc["x_prop"], sc["x_prop"], sc["P"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, x0, P0, log0), (_, x1, P1, log1) = res
    assert (x0.view(np.uint64) == x1.view(np.uint64)).all()       # ranks agree bit-for-bit
    assert (P0.view(np.uint64) == P1.view(np.uint64)).all()
    assert log0 == log1 == list(ref["effct"])                       # integer counts are exact
    assert np.abs(x0 - ref["x"]).max() < 1e-10 and np.abs(P0 - ref["P"]).max() < 1e-13


@pytest.mark.gpu
def test_gpu_shards_sum_to_full_block():
    """Testable on one GPU: run the G shards sequentially and sum the partial blocks in rank order."""
    from daliti_amd import Engine, synth
    from daliti_amd.sharding import shard_range
    sc = synth.make_small()
    x = sc["x_prop"]
    full = Engine()
    full.map_build(sc["map"])
    full.scan_set(sc["scan"])
    ref = full.residual_pass(x, True)
    for world in (2, 8):
        HtH = np.zeros((12, 12)); Htz = np.zeros(12); eff = 0; tot = 0.0
        for r in range(world):
            lo, hi = shard_range(len(sc["scan"]), r, world)
            full.scan_set(sc["scan"][lo:hi])
            o = full.residual_pass(x, True)
            HtH += o["HtH"]; Htz += o["Htz"]; eff += o["effct"]; tot += o["total_res"]
        assert eff == ref["effct"]
        assert np.abs(HtH - ref["HtH"]).max() <= 1e-12 * np.abs(ref["HtH"]).max()
        assert np.abs(Htz - ref["Htz"]).max() <= 1e-12 * max(np.abs(ref["Htz"]).max(), 1)
        assert abs(tot - ref["total_res"]) <= 1e-12 * ref["total_res"]
    full.close()


@pytest.mark.gpu
def test_bench_collective_path_single_rank():
    """bench.py's N > 1 code path (torch NCCL all-reduce through the C callback) with one rank."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300))
    lines = []
    # the engine-owned RCCL communicator, the host shared-memory exchange, then the torch callback
    for extra in (["--collective", "rccl"], ["--collective", "shm"], ["--torch-collective"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "3",
                              "--warmup", "1", "--no-cpu", "--force-collective"] + extra, env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
        assert lines[-1]["n_gpus"] == 1 and lines[-1]["value"] > 0 and lines[-1]["pose_error_vs_truth_m"] < 0.05
    assert "engine-owned communicator" in lines[0]["config"]["parallelism"] and lines[0]["config"]["collective"] == "rccl"
    assert "POSIX shared memory" in lines[1]["config"]["parallelism"] and lines[1]["config"]["collective"] == "shm"
    assert "torch.distributed callback" in lines[2]["config"]["parallelism"]
    assert lines[0]["final_pos"] == lines[1]["final_pos"] == lines[2]["final_pos"]
    # the default (--collective auto): the headline on the shared-memory exchange; afterwards, under the watchdog, detach ->
    # the engine's RCCL communicator -> detach -> re-attach, every form timed
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "3", "--warmup", "1",
                          "--no-cpu", "--force-collective"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    auto = json.loads(out.stdout.strip().splitlines()[-1])
    probe = auto["config"]["collective_probe_ms_per_step"]
    assert set(probe) == {"shm", "rccl", "shm_reattached"} and auto["config"]["collective"] == "shm"
    assert all(v > 0 for v in probe.values())
    assert auto["final_pos"] == lines[0]["final_pos"]
    line = lines[0]
    ref = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "3",
                          "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0, ref.stderr[-2000:]
    rline = json.loads(ref.stdout.strip().splitlines()[-1])
    assert abs(rline["pose_error_vs_truth_m"] - line["pose_error_vs_truth_m"]) < 1e-12


@pytest.mark.gpu
def test_two_process_sharded_bench_equals_single_rank():
    """Two bench.py ranks (torch.distributed.run, gloo on CUDA tensors, both on GPU 0) shard a 32x625
    scan; the registered pose must equal one rank processing the whole scan."""
    import json
    import subprocess
    port = str(29700 + os.getpid() % 200)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--config", "C1", "--scaling", "weak", "--steps", "3", "--warmup", "1", "--no-cpu",
                          "--backend", "gloo", "--all-on-device0", "--torch-collective"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert "torch.distributed callback" in l2["config"]["parallelism"]   # gloo all-reduce of the block through the C callback
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "3",
                          "--warmup", "1", "--no-cpu", "--beams-mult", "2"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert l2["n_gpus"] == 2 and l2["scaling"] == "weak"
    assert l2["config"]["scan_points_per_gpu"] * 2 == l1["config"]["scan_points_per_gpu"]
    assert np.abs(np.array(l2["final_pos"]) - np.array(l1["final_pos"])).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_single_process_multi_handle_update_equals_single_handle(n):
    """s2m_iterated_update_multi: ONE scan in n shard_range pieces on n handles (here all on one GPU, sharing one map),
    the n pinned blocks summed by the host in handle order, one fp64 update -- no collective library.  Same iterations
    and effective counts as the single handle, and -- because the shards are aligned power-of-two pieces and every sum
    on the way is a tree over the point index -- the same bits in state and covariance as the single handle, for
    n = 2 and n = 8 alike; a second run gives the same bits again."""
    from daliti_amd import Engine, synth
    from daliti_amd.sharding import shard_range
    sc = synth.make_small()
    one = Engine(max_iter=5, device_loop=0)    # host-stepped like the multi-handle form: bit-identity compares like with like
    one.map_build(sc["map"])
    one.scan_set(sc["scan"])
    ref = one.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
    engs = []
    for r in range(n):
        lo, hi = shard_range(len(sc["scan"]), r, n)
        e = Engine(max_iter=5)
        e.map_share(one)
        e.scan_set(sc["scan"][lo:hi])
        engs.append(e)
    runs = []
    for _ in range(2):
        for e in engs:
            e.set_feat_queue(())
        x = sc["x_prop"].copy(); P = sc["P"].copy()
        log = Engine.iterated_update_multi(engs, x, np.ascontiguousarray(sc["x_prop"]), P)
        runs.append((x, P, log.iters, list(log.effct[:log.iters]), log.rematch_passes))
    x, P, iters, effct, rematch = runs[0]
    assert iters == ref["iters"] and effct == list(ref["effct"]) and rematch == ref["rematch_passes"]
    assert np.abs(x - ref["x"]).max() < 1e-12 and np.abs(P - ref["P"]).max() < 1e-14
    assert (bits(runs[1][0]) == bits(x)).all() and (bits(runs[1][1]) == bits(P)).all()
    # more than close: the reduce kernel's final sum is a binary tree over the point index and the host combines the
    # n blocks pairwise, so aligned power-of-two pieces (here 1 024 / 256 points of 2 048) reproduce the single-handle
    # normal block -- and with it the state and covariance -- BIT FOR BIT, whatever n
    assert (bits(x) == bits(ref["x"])).all() and (bits(P) == bits(ref["P"])).all()
    # the shards' neighbour lists are the single handle's, piece by piece
    ri, rd = one.get_neighbors()
    for r, e in enumerate(engs):
        lo, hi = shard_range(len(sc["scan"]), r, n)
        i, d = e.get_neighbors()
        assert (i == ri[lo:hi]).all() and (bits(d) == bits(rd[lo:hi])).all()
    with pytest.raises(Exception):
        Engine.iterated_update_multi([engs[0], engs[0]], x, np.ascontiguousarray(sc["x_prop"]), P)
    for e in engs:
        e.close()
    one.close()


@pytest.mark.gpu
def test_single_pass_entry_points_with_communicator_attached():
    """A handle that carries an RCCL communicator still serves s2m_residual_pass / s2m_h_share_model: only the
    sharded loop defers the host hand-off of the block until after its collective."""
    from daliti_amd import Engine, synth
    sc = synth.make_small()
    plain = Engine(max_iter=5, device_loop=0)  # a handle with a communicator is host-stepped
    plain.map_build(sc["map"])
    plain.scan_set(sc["scan"])
    ref = plain.residual_pass(sc["x_prop"], True)
    e = Engine(max_iter=5)
    e.comm_init(Engine.comm_unique_id(), 1, 0)
    e.map_build(sc["map"])
    e.scan_set(sc["scan"])
    got = e.residual_pass(sc["x_prop"], True)
    assert got["effct"] == ref["effct"] and (got["HtH"] == ref["HtH"]).all()
    got2 = e.residual_pass(sc["x_prop"], False)
    assert got2["effct"] == ref["effct"]
    r1 = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])       # the built-in collective path, one rank
    r0 = plain.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
    assert (r1["x"] == r0["x"]).all() and (r1["effct"] == r0["effct"]).all()
    e.close()
    plain.close()


@pytest.mark.gpu
def test_empty_first_scan():
    """s2m_scan_set(n = 0) on a fresh handle (no buffers yet) is a valid empty scan: zero block, EKF stop."""
    from daliti_amd import Engine, synth
    sc = synth.make_small()
    e = Engine(max_iter=5)
    e.map_build(sc["map"])
    e.scan_set(np.zeros((0, 3), np.float32))
    out = e.residual_pass(sc["x_prop"], True)
    assert out["effct"] == 0 and not out["HtH"].any()
    r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
    assert r["ekf_stop"] and r["iters"] == 1 and (r["x"] == sc["x_prop"]).all()
    e.close()


@pytest.mark.gpu
def test_two_process_strong_scaling_bench_equals_single_rank():
    """The default for N > 1 (north star: "a single scan's points shard across GPUs"): ONE scan split by shard_range
    over two ranks (gloo on CUDA tensors, both on GPU 0) registers to the same pose as one rank over the same scan.
    The two ranks are started by bench.py ITSELF: `python bench.py --gpus 2` with no WORLD_SIZE in the environment --
    the driver's single-GPU command form -- must launch them as a child process and relay rank 0's line last."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C1", "--steps", "3",
                          "--warmup", "1", "--no-cpu", "--backend", "gloo", "--all-on-device0"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    assert "launching" in two.stderr and "torch.distributed.run" in two.stderr
    assert two.stdout.rstrip().splitlines()[-1].startswith("{")          # the JSON line is the last line
    l2 = json.loads(two.stdout.rstrip().splitlines()[-1])
    assert "weak_scaling" in l2 and l2["weak_scaling"]["scan_points_total"] == 20000
    # (C1 has 625 azimuths: no sector leg with two ranks; test_two_process_shared_memory_exchange... covers it on C2)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "3",
                          "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert l2["n_gpus"] == 2 and l2["scaling"] == "strong"
    assert l2["config"]["collective"] == "shm"   # gloo cannot carry the engine's RCCL communicator: the host exchange it is
    assert l2["config"]["scan_points_per_gpu"] * 2 == l1["config"]["scan_points_per_gpu"] == 10000
    assert np.abs(np.array(l2["final_pos"]) - np.array(l1["final_pos"])).max() < 1e-10


@pytest.mark.gpu
def test_two_process_default_bench_walks_every_exchange_form():
    """`python bench.py --gpus 2` as the driver runs it (default config C3), two ranks on one GPU over gloo: the headline
    on the shared-memory exchange, then -- after the headline, under the watchdog -- detach, the second exchange form (the
    torch.distributed callback stands in for RCCL under gloo), re-attach; the side legs weak_scaling, sector_sharding and
    replicas_batched (24 scans in flight per rank, summed).  With a watchdog of a millisecond the probe "hangs": rank 0
    still prints the line collected so far and every rank leaves with exit code 0."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu",
           "--backend", "gloo", "--all-on-device0"]
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert two.returncode == 0, two.stderr[-3000:]
    l2 = json.loads(two.stdout.rstrip().splitlines()[-1])
    assert l2["n_gpus"] == 2 and l2["scaling"] == "strong" and l2["config"]["collective"] == "shm"
    probe = l2["config"]["collective_probe_ms_per_step"]
    assert set(probe) == {"shm", "torch_callback", "shm_reattached"} and all(v > 0 for v in probe.values())
    assert l2["weak_scaling"]["scan_points_total"] == 131072 and l2["sector_sharding"]["scan_points_per_gpu"] == 32768
    rb = l2["replicas_batched"]
    assert rb["scans_in_flight_per_gpu"] == 24 and rb["scans_per_sec"] > 1000 and 0 < rb["job_roofline"]["frac"] < 1
    assert rb["job_roofline"]["peak"] == 2 * 8000.0
    hung = subprocess.run(cmd, env=dict(env, S2M_PROBE_TIMEOUT_S="0.001"), capture_output=True, text=True, timeout=1500)
    assert hung.returncode == 0, hung.stderr[-3000:]
    lh = json.loads([ln for ln in hung.stdout.splitlines() if ln.startswith("{")][-1])
    assert lh["rccl_probe"] == "timed out" and lh["ms_per_step"] > 0 and "replicas_batched" in lh


def test_bench_defaults_follow_baseline_configs():
    """CPU-side check of bench.py's argument logic: N > 1 defaults to strong scaling (C4: one 131,072-point scan,
    16,384 points per GPU at N = 8; C3: 8,192), C5 to replicas with the survey's seeds and offsets."""
    from daliti_amd import synth
    from daliti_amd.sharding import shard_range
    c4 = synth.CONFIGS["C4"]
    n = c4["beams"] * c4["az"]
    assert n == 131072 and [shard_range(n, r, 8) for r in (0, 7)] == [(0, 16384), (114688, 131072)]
    assert shard_range(65536, 3, 8) == (24576, 32768)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'a.scaling if a.scaling != "auto" else "strong"' in src and '"replicas" if a.config in ("C5", "C5b")' in src
    c5 = synth.CONFIGS["C5"]
    assert c5["replicas"] == 8 and (c5["M"], c5["beams"] * c5["az"]) == (5_000_000, 65536)
    assert [synth.replica_offset(k) for k in (0, 7)] == [-7.0, 7.0]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with N > 1 and no launcher environment (the driver's single-GPU command form with a
    larger N) must not silently run one rank: it starts N ranks as a CHILD process -- never exec, never after touching
    the GPU -- with a loopback rendezvous; under torch.distributed.run it does not launch again; a world size that
    differs from --gpus is refused.  No GPU needed: only the argument logic runs here."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert "torch" not in bench.__dict__                                   # nothing GPU-side is imported at load time
    argv = ["--gpus", "2", "--steps", "7", "--config", "C1"]
    a = bench.parse(argv)
    assert bench.needs_self_launch(a, {})
    assert not bench.needs_self_launch(a, {"WORLD_SIZE": "2", "RANK": "0"})   # already under a launcher
    assert not bench.needs_self_launch(bench.parse(["--gpus", "1"]), {})
    assert not bench.needs_self_launch(bench.parse(["--gpus", "2", "--collective", "host"]), {})   # one process by design
    cmd = bench.self_launch_argv(a, argv, 29511)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                   "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.join(ROOT, "bench.py")] + argv
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src.replace("never exec", "")
    assert src.index("if needs_self_launch(a, os.environ)") < src.index("import torch\n")
    # a mismatch between --gpus and the world size is an error, not a silent single-rank run
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_bench_host_collective_form_equals_single_handle():
    """`bench.py --collective host --shards 4`: one process, four handles holding the shard_range pieces of ONE C1 scan
    on one GPU, blocks summed on the host -- the registered pose equals the plain single-handle run bit for bit (aligned
    power-of-two shards: 2 500 points each are NOT multiples of 64, so here only to summation order) and the line says
    what ran."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    outs = []
    for extra in (["--collective", "host", "--shards", "4"], []):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "3", "--warmup", "1",
                            "--no-cpu"] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads(r.stdout.rstrip().splitlines()[-1]))
    multi, one = outs
    assert "s2m_iterated_update_multi" in multi["config"]["parallelism"] and multi["scaling"] == "strong"
    assert multi["iters_per_step"] == one["iters_per_step"]
    assert np.abs(np.array(multi["final_pos"]) - np.array(one["final_pos"])).max() < 1e-12


def test_shared_memory_exchange_protocol():
    """The host side of --collective shm on its own (no GPU): forked ranks exchange blocks through the POSIX segment
    with jittered timing; every rank must read every rank's block of the same sequence (two slots per rank by sequence
    parity: a rank cannot overwrite a slot somebody is still reading)."""
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="s2m_shm_"), "shm_exchange_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "c++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                           "-I", os.path.join(ROOT, "daliti_amd", "csrc"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "shm_exchange_test.cpp"), os.path.join(ROOT, "daliti_amd", "csrc", "s2m_comm.cpp"),
                           "-ldl", "-lrt", "-pthread", "-o", exe])
    for ranks, rounds in ((2, 4000), (4, 3000), (7, 1500)):
        out = subprocess.run([exe, str(ranks), str(rounds)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_two_process_shared_memory_exchange_is_bit_identical_to_one_rank():
    """--collective shm: two ranks (both on GPU 0) split the C2 scan into two aligned halves, publish their blocks into
    their own pinned pages, exchange them through POSIX shared memory on the host and sum them pairwise over the rank
    index -- the registered pose must equal the single rank's BIT FOR BIT (tree-shaped sums)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C2", "--steps", "3",
                          "--warmup", "1", "--no-cpu", "--backend", "gloo", "--all-on-device0", "--collective", "shm"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    l2 = json.loads(two.stdout.rstrip().splitlines()[-1])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "3",
                          "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert l2["config"]["collective"] == "shm" and l2["n_gpus"] == 2
    assert l2["config"]["scan_points_per_gpu"] == 32768
    assert l2["final_pos"] == l1["final_pos"]
    assert l2["iters_per_step"] == l1["iters_per_step"]
    # side legs of an N > 1 run: the weak form and the sector-sharded form of the same scan (same pose to rounding)
    assert l2["weak_scaling"]["scan_points_per_gpu"] == 65536
    assert l2["sector_sharding"]["scan_points_per_gpu"] == 32768 and l2["sector_sharding"]["pose_delta_vs_range_sharding_m"] < 1e-10


@pytest.mark.gpu
def test_shared_memory_exchange_survives_ranks_that_bet_differently():
    """Two processes on GPU 0, each with half of the C2 scan, exchange their blocks through POSIX shared memory.  Rank
    0 ALWAYS bets that its far-point list is empty (and loses on the first pass of every scan), rank 1 never bets: every rank
    learns from the exchanged blocks that one of them was void, all of them publish a second time, and the result equals
    one engine over the whole scan bit for bit -- over three scans in a row."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="s2m_shm_")
    helper = os.path.join(ROOT, "tests", "shm_rank_helper.py")
    name = "/s2m_t_%d" % os.getpid()
    procs = [subprocess.Popen([sys.executable, helper, name, "2", str(r), "2" if r == 0 else "0",
                               os.path.join(tmp, "r%d.npy" % r)], stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    one = subprocess.run([sys.executable, helper, name, "1", "0", "1", os.path.join(tmp, "one.npy")], capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a, b, ref = (np.load(os.path.join(tmp, f)) for f in ("r0.npy", "r1.npy", "one.npy"))
    assert np.array_equal(a, b)       # the redundant updates stay in step
    assert np.array_equal(a, ref)     # 32,768-point shards are aligned: bit-identical to the unsplit scan
    # three ranks (always / never / by history), shards that are not aligned: the ranks still agree bit for bit with each
    # other, and with the unsplit scan to rounding
    name3 = name + "_3"
    procs = [subprocess.Popen([sys.executable, helper, name3, "3", str(r), str((2, 0, 1)[r]),
                               os.path.join(tmp, "t%d.npy" % r)], stderr=subprocess.PIPE, text=True) for r in range(3)]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    t = [np.load(os.path.join(tmp, "t%d.npy" % r)) for r in range(3)]
    assert np.array_equal(t[0], t[1]) and np.array_equal(t[0], t[2])
    assert np.abs(t[0][:, :36] - ref[:, :36]).max() < 1e-10 and np.array_equal(t[0][:, 612:621], ref[:, 612:621])
