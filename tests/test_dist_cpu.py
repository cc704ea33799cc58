"""CPU test of the N>1 path (gloo, world_size 2): image sharding + fixed-shape gather of detections."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from demonet_amd.dist import DetectionGatherer, gather_detections, pack_detections, shard_range, unpack_detections


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, D = 3, 5
    g = torch.Generator().manual_seed(100 + rank)
    boxes = torch.rand(B, D, 4, generator=g)
    scores = torch.rand(B, D, generator=g)
    labels = torch.randint(1, 91, (B, D), generator=g)
    counts = torch.tensor([D, 2, 0], dtype=torch.int32) if rank == 0 else torch.tensor([1, D, 3], dtype=torch.int32)
    gp, gc = gather_detections(pack_detections(boxes, scores, labels), counts)
    assert gp.shape == (world * B, D, 6) and gc.tolist() == [D, 2, 0, 1, D, 3]
    mine = unpack_detections(gp[rank * B:(rank + 1) * B], gc[rank * B:(rank + 1) * B])
    for i in range(B):
        c = int(counts[i])
        assert torch.equal(mine[i]["boxes"], boxes[i, :c]) and torch.equal(mine[i]["labels"], labels[i, :c])
        assert torch.equal(mine[i]["scores"], scores[i, :c])
    # windowed gatherer (the form bench.py uses): five steps with a window of two -> two full windows + a flushed partial one
    G = DetectionGatherer(B, D, "cpu", every=2)
    idx = [G.submit(boxes + step, scores, labels, counts) for step in range(5)]
    try:
        G.result(idx[4])                      # the last window is not gathered before flush()
        raise AssertionError("expected RuntimeError")
    except RuntimeError:
        pass
    G.flush()
    for step in (2, 3, 4):                    # steps 0, 1 were overwritten by the third window
        pk, cn = G.result(idx[step])
        assert pk.shape == (world * B, D, 6) and cn.tolist() == [D, 2, 0, 1, D, 3]
        assert torch.equal(pk[rank * B:(rank + 1) * B, :, :4], boxes + step)
        assert torch.equal(pk[rank * B:(rank + 1) * B, :, 5].to(torch.int64), labels)
    # submit after a flush (a gatherer reused across evaluation epochs): tickets restart on a fresh window, the slot the
    # flushed window never filled is not handed out as data
    assert G.n == 6
    idx2 = [G.submit(boxes + 10 + step, scores, labels, counts) for step in range(3)]
    assert idx2 == [6, 7, 8]
    for bad in (5, 8):                        # 5: never submitted; 8: not gathered yet
        try:
            G.result(bad)
            raise AssertionError("expected RuntimeError")
        except RuntimeError:
            pass
    G.flush()
    for step, t in enumerate(idx2):
        pk, cn = G.result(t)
        assert torch.equal(pk[rank * B:(rank + 1) * B, :, :4], boxes + 10 + step) and cn.tolist() == [D, 2, 0, 1, D, 3]
    # the pipelined form (ForwardPipeline: every forward has a payload buffer of its own, the window-closing collective first joins
    # the forwards' streams): src = that buffer, join called once per collective -- per window and per flush
    G2 = DetectionGatherer(B, D, "cpu", every=2)
    joins = []
    bufs = [torch.zeros(B, D + 1, 6) for _ in range(3)]
    tick = []
    for step in range(5):
        b = bufs[step % 3]
        b[:, :D] = pack_detections(boxes + 20 + step, scores, labels)
        b[:, D, 0] = counts.float()
        tick.append(G2.submit(src=b, join=lambda: joins.append(G2.n)))
    assert joins == [2, 4]                    # after the 2nd and 4th submit
    G2.flush(join=lambda: joins.append(-1))
    assert joins == [2, 4, -1]
    for step in (2, 3, 4):
        pk, cn = G2.result(tick[step])
        assert torch.equal(pk[rank * B:(rank + 1) * B, :, :4], boxes + 20 + step) and cn.tolist() == [D, 2, 0, 1, D, 3]
    # bench.py's C4 sharding arithmetic: global batch 256 over 8 ranks -> 32 contiguous images each, every image exactly once
    world8 = [shard_range(256, r, 8) for r in range(8)]
    assert all(hi - lo == 32 for lo, hi in world8) and [lo for lo, _ in world8] == list(range(0, 256, 32))
    lo, hi = shard_range(2 * B, rank, world)          # and end to end here: each rank's shard, gathered, is the global batch in order
    glob = torch.arange(2 * B * D * 6, dtype=torch.float32).view(2 * B, D, 6)
    gp2, _ = gather_detections(glob[lo:hi].contiguous(), counts)
    assert torch.equal(gp2, glob)
    ret[rank] = float(gp.sum())
    dist.destroy_process_group()


def test_gather_detections_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert len(ret) == 2 and abs(ret[0] - ret[1]) < 1e-6          # both ranks hold the same global batch


def test_shard_range():
    assert [shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]       # 32 images per GPU
    assert shard_range(10, 3, 4) == (9, 10) and shard_range(2, 3, 4) == (2, 2)     # ragged / empty shards


def _bench_worker(rank, world, port, ret, extra):
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ret[rank] = bench.main(["--gpus", str(world), "--steps", "37", "--warmup", "5", "--stub-cpu"] + extra)


def test_bench_main_c4_path_runs_at_world2_on_cpu():
    """bench.py's OWN main() at world size 2 over gloo with a stub forward (--stub-cpu): the code the driver launches for the scaling
    curve -- C4 sharding of the global batch 256 (128 per rank here), forwards in flight feeding the windowed gather from their own
    payload buffers, the flush of the last partial window inside the timed region, barriers, MAX over ranks, one JSON line from rank 0
    -- is executed, and what the last step's gather holds on every rank is the global batch in image order (util/misc.py:75-115 and
    engine.py:105 are the reference's counterparts). Also with --batch (weak scaling: the per-rank size fixed)."""
    for extra, per, scaling in (([], 128, "strong"), (["--batch", "8"], 8, "weak")):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ret = mp.Manager().dict()
        mp.spawn(_bench_worker, args=(2, port, ret, extra), nprocs=2, join=True)
        assert len(ret) == 2
        for r in (0, 1):
            d = ret[r]
            assert d["stub_check"] and d["n_gpus"] == 2 and d["per_rank_batch"] == per and d["scaling"] == scaling
            assert d["shard"] == [r * per, (r + 1) * per] and d["global_batch"] == 2 * per
            assert d["gather_windows"] == 3 and d["joins"] == 3          # 6 warm-up + 37 timed steps = two full windows of 16 + the flushed one
        assert ret[0]["ms_per_step"] == ret[1]["ms_per_step"]            # MAX over ranks


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2 --stub-cpu` with NO launcher environment: the process starts two ranks itself (torch.distributed.run,
    one per GPU as README.md:63 / util/misc.py:302-324 do) and relays rank 0's line: n_gpus and the communicator's own world size are 2."""
    import json
    p = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "3", "--stub-cpu"])
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["stub_check"] and d["per_rank_batch"] == 128 and d["scaling"] == "strong"


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--stub-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_bench_gpus_n_fails_loudly_without_n_devices():
    """--gpus 8 on a box with fewer GPUs (none here) must not print a 1-GPU number: non-zero status, no result line."""
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this box has 8 GPUs")
    p = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert p.returncode != 0 and '{"metric"' not in p.stdout and "refusing" in p.stderr
