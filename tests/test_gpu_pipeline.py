"""GPU tests (-m gpu) of demonet_amd.pipeline.ForwardPipeline: several forwards of one plan in flight must return, per batch,
exactly what one forward at a time returns (the oracle parity of that single forward is test_gpu_model.py's subject)."""
import ctypes as C
import os

import pytest
import torch

from demonet_amd import _lib, models, synth
from demonet_amd.dist import pack_detections
from demonet_amd.pipeline import ForwardPipeline

pytestmark = pytest.mark.gpu


def _model(name="ssdlite320_mobilenet_v3_large", ncls=91):
    m = getattr(models, name)(num_classes=ncls)
    models.load_synthetic(m, 0)
    return m.cuda()


def _batches(g, count, n, seed=40):
    W, H = g.size
    return [torch.from_numpy(synth.images(seed + i, n, H, W)).cuda() for i in range(count)]


@pytest.mark.parametrize("n,depth", [(8, 3), (32, 2), (5, 4), (64, 3)])          # (64, 3): BASELINE config C2 as bench.py runs it
def test_pipeline_equals_one_forward_at_a_time(n, depth):
    """Seven different batches through a pipeline `depth` deep, results read as late as the contract allows (just before their
    slot is reused): bit-identical to forward_batch on the same images as single-chain forwards. Runs twice over the same slots,
    so both the first (eager + capture) and the replayed launches of every slot are covered."""
    m = _model()
    batches = _batches(m.graph, 7, n)
    with ForwardPipeline(m, n, depth=depth, packed=True) as pipe:
        assert m.batch_split(n) == 1                      # whole-batch chains while the pipeline is open
        ref = [[t.clone() for t in m.forward_batch(b)] for b in batches]
        torch.cuda.synchronize()
        for _ in range(2):
            got = {}
            base = pipe.n
            grab = lambda j: ([x.clone() for x in pipe.result(base + j)], pipe.packed(base + j).clone())
            for k, b in enumerate(batches):
                assert pipe.submit(b) == base + k
                if k - (depth - 1) >= 0:                   # the oldest forward still alive: the next submit overwrites it
                    got[k - (depth - 1)] = grab(k - (depth - 1))
            for j in range(max(0, len(batches) - depth + 1), len(batches)):
                got[j] = grab(j)
            assert sorted(got) == list(range(len(batches)))
            for k in range(len(batches)):
                out, packed = got[k]
                for a, b in zip(ref[k], out):
                    assert torch.equal(a, b), "batch %d" % k
                D = ref[k][1].shape[1]
                assert torch.equal(packed[:, :D], pack_detections(ref[k][0], ref[k][1], ref[k][2]))
                assert torch.equal(packed[:, D, 0].to(torch.int32), ref[k][3])
            assert int(sum(int(r[3].sum()) for r in ref)) > 0
    # closed: the automatic split is back and the model's own forward still works
    assert m.batch_split(64) == 2
    again = m.forward_batch(batches[0])
    torch.cuda.synchronize()
    if n < 32:
        for a, b in zip(ref[0], again):
            assert torch.equal(a, b)


@pytest.mark.parametrize("name,ncls,n", [("ssd300_vgg16", 91, 2), ("ssd_lite_mobilenet_v2", 21, 33)])
def test_pipeline_other_model_families(name, ncls, n):
    """The dense-conv model (256 x 256-tile kernels with 130+ KB of LDS per workgroup, dense 3x3 heads) and the V2 model at a batch
    that the in-forward split would halve: three forwards in flight equal the single-chain forward of each batch."""
    m = _model(name, ncls)
    batches = _batches(m.graph, 4, n, seed=55)
    with ForwardPipeline(m, n, depth=3) as pipe:
        ref = [[t.clone() for t in m.forward_batch(b)] for b in batches]
        for rnd in range(2):
            ts = [pipe.submit(b) for b in batches[:3]]
            for k, t in enumerate(ts):
                for a, b in zip(ref[k], pipe.result(t)):
                    assert torch.equal(a, b), "%s batch %d" % (name, k)
        t = pipe.submit(batches[3])
        for a, b in zip(ref[3], pipe.result(t)):
            assert torch.equal(a, b)
    assert int(sum(int(r[3].sum()) for r in ref)) > 0


def test_pipeline_persistent_inputs_and_detections_form():
    m = _model()
    n, depth = 4, 2
    batches = _batches(m.graph, depth, n, seed=70)
    ref = [[t.clone() for t in m.forward_batch(b)] for b in batches]
    with ForwardPipeline(m, n, depth=depth) as pipe:
        for rnd in range(3):                              # the same tensor lands on the same slot every time: direct replay
            ts = [pipe.submit(b, persistent_input=True) for b in batches]
            for k, t in enumerate(ts):
                for a, b in zip(ref[k], pipe.result(t)):
                    assert torch.equal(a, b)
        dets = pipe.detections(ts[1])
        assert len(dets) == n and list(dets[0]) == ["boxes", "scores", "labels"]
        for i, d in enumerate(dets):
            c = int(ref[1][3][i])
            assert d["boxes"].shape == (c, 4) and d["labels"].dtype == torch.int64
            assert torch.equal(d["scores"], ref[1][1][i, :c])
        # stream-ordered hand-over without a host wait
        t = pipe.submit(batches[0], persistent_input=True)
        pipe.wait(t)
        s = pipe.slots[t % depth].scores.clone()
        torch.cuda.synchronize()
        assert torch.equal(s, ref[0][1])


def test_pipeline_uint8_input():
    m = _model()
    n = 3
    W, H = m.graph.size
    u8 = [(torch.from_numpy(synth.images(90 + i, n, H, W)).cuda() * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
          for i in range(3)]
    ref = [[t.clone() for t in m.forward_uint8(x)] for x in u8]
    with ForwardPipeline(m, n, depth=2, uint8=True) as pipe:
        for k, x in enumerate(u8):
            t = pipe.submit(x)
            for a, b in zip(ref[k], pipe.result(t)):
                assert torch.equal(a, b)
        with pytest.raises(ValueError):
            pipe.submit(u8[0].float())


def test_evaluate_loop_matches_one_call_per_batch():
    """demonet_amd.engine.evaluate (the reference's engine.evaluate loop, engine.py:84-100, over the pipeline): same-size batches,
    a change of image size (device resize + boxes mapped back), a mixed-size batch (runs through model(images) after a drain) --
    every record equals what model(images) returns for that batch, in the loader's order; the COCO records follow coco_eval.py:76-98."""
    from demonet_amd import engine, evalrec
    m = _model()
    g = torch.Generator().manual_seed(5)
    def imgs(n, h, w, seed):
        return [torch.from_numpy(synth.images(seed + i, 1, h, w)[0]) for i in range(n)]
    loader, next_id = [], [100]
    def add(images):
        t = [{"image_id": torch.tensor(next_id[0] + i)} for i in range(len(images))]
        next_id[0] += len(images)
        loader.append((images, t))
    for b in range(5):
        add(imgs(3, 320, 320, 200 + 10 * b))
    add(imgs(3, 300, 400, 300))
    add(imgs(1, 320, 320, 400) + imgs(1, 500, 400, 401))        # mixed sizes
    add(imgs(3, 320, 320, 500))
    results, stats = engine.evaluate(m, loader, "cuda:0", depth=3)
    assert stats["images"] == 23 and list(results) == list(range(100, 123))
    for images, targets in loader:
        want = m([im.cuda() for im in images])
        for t, w in zip(targets, want):
            got = results[int(t["image_id"])]
            for k in ("boxes", "scores", "labels"):
                assert torch.equal(got[k], w[k].cpu()), (int(t["image_id"]), k)
            assert got["labels"].dtype == torch.int64
    recs = engine.coco_records(results)
    assert len(recs) == sum(len(r["scores"]) for r in results.values()) > 0
    r0 = results[100]
    assert recs[0]["image_id"] == 100 and recs[0]["category_id"] == int(r0["labels"][0]) and recs[0]["score"] == float(r0["scores"][0])
    x1, y1, x2, y2 = r0["boxes"][0].tolist()
    assert recs[0]["bbox"] == [x1, y1, (r0["boxes"][0, 2] - r0["boxes"][0, 0]).item(), (r0["boxes"][0, 3] - r0["boxes"][0, 1]).item()]
    # the padded-array form of evalrec builds the same list
    ids = list(results)[:3]
    D = m.detections_per_img
    boxes = torch.zeros(3, D, 4); scores = torch.zeros(3, D); labels = torch.zeros(3, D, dtype=torch.int64); counts = torch.zeros(3, dtype=torch.int32)
    for i, k in enumerate(ids):
        c = len(results[k]["scores"]); counts[i] = c
        boxes[i, :c], scores[i, :c], labels[i, :c] = results[k]["boxes"], results[k]["scores"], results[k]["labels"]
    assert evalrec.coco_detection_records(boxes, scores, labels, counts, ids) == [r for r in recs if r["image_id"] in ids]


def test_pipeline_errors():
    m = _model()
    n = 2
    b = _batches(m.graph, 1, n)[0]
    with pytest.raises(ValueError):
        ForwardPipeline(m, n, depth=0)
    pipe = ForwardPipeline(m, n, depth=2)
    with pytest.raises(RuntimeError, match="never submitted"):
        pipe.result(0)
    ts = [pipe.submit(b) for _ in range(3)]
    with pytest.raises(RuntimeError, match="overwritten"):
        pipe.result(ts[0])
    pipe.result(ts[1]); pipe.result(ts[2])
    with pytest.raises(ValueError):
        pipe.submit(b[:1])
    with pytest.raises(TypeError):                        # transform.py:130-134
        pipe.submit((b * 255).to(torch.int32))
    with pytest.raises(RuntimeError, match="packed=True"):
        pipe.packed(ts[2])
    with ForwardPipeline(m, n, depth=2) as second:        # two pipelines of one plan: the split is reset when the LAST one closes
        t2 = second.submit(b)
        with pytest.raises(ValueError, match="chains"):
            ForwardPipeline(m, n, depth=2, chains=2)
        t1 = pipe.submit(b)
        assert torch.equal(second.result(t2)[1], pipe.result(t1)[1])
    assert m.batch_split(64) == 1                         # `pipe` is still open
    pipe.result(pipe.submit(b))
    m.invalidate()                                        # the plan the pipeline replays is gone
    with pytest.raises(RuntimeError, match="rebuilt"):
        pipe.submit(b)
    pipe.close()
    with pytest.raises(RuntimeError, match="closed"):
        pipe.submit(b)


def test_set_chains_abi():
    m = _model()
    m.forward_batch(_batches(m.graph, 1, 2)[0])
    L, h = _lib.lib(), C.c_void_p(m._handle)
    assert L.dn_batch_split(h, 64) == 2 and L.dn_batch_split(h, 16) == 1
    ws2 = L.dn_workspace_bytes(h, 64)
    assert L.dn_set_chains(h, 1) == 0
    assert L.dn_batch_split(h, 64) == 1
    assert L.dn_workspace_bytes(h, 64) > ws2 // 2         # sized for the new split (one chain of 64 rows instead of two of 32)
    assert L.dn_set_chains(h, 3) == 0 and L.dn_batch_split(h, 64) == 3 and L.dn_batch_split(h, 2) == 2
    assert L.dn_set_chains(h, 5) != 0 and b"dn_set_chains" in L.dn_last_error()
    assert L.dn_set_chains(h, 0) == 0 and L.dn_batch_split(h, 64) == 2 and L.dn_workspace_bytes(h, 64) == ws2
    m._bufs = {}


@pytest.mark.parametrize("se_small", ["1", "0"])
def test_stress_forwards_in_flight_are_bit_identical_to_serial(se_small, monkeypatch):
    """>= 500 forwards at depths 2 - 4, batch 32 and 64, every result compared bit for bit with the serial forward
    of the same batch. The comparisons are enqueued on the device (no host wait between submits), so the chip stays shared by `depth`
    forwards the whole time -- the regime in which a cross-workgroup hand-over (the squeeze-excitation FCs in the last workgroup of the
    pooling depthwise launch, DN_SE_SMALL=1: depthwise.hip dw_se_tail) or any timing-dependent kernel would show. DN_SE_SMALL=0 runs the
    same load with the FCs folded into the projection (no hand-over) as the control."""
    monkeypatch.setenv("DN_SE_SMALL", se_small)
    m = _model()
    total = 0
    for n, depth, count in [(32, 2, 90), (32, 3, 90), (32, 4, 90), (64, 2, 90), (64, 3, 100), (64, 4, 90)] if se_small == "1" else [(32, 3, 60), (64, 3, 60)]:
        batches = _batches(m.graph, 6, n, seed=90)
        with ForwardPipeline(m, n, depth=depth) as pipe:
            ref = [[t.clone() for t in m.forward_batch(b)] for b in batches]
            torch.cuda.synchronize()
            flags = torch.zeros(count, dtype=torch.int32, device="cuda")
            base = pipe.n

            def check(j):
                pipe.wait(base + j)                           # stream-ordered, no host wait
                s = pipe.slots[(base + j) % depth]
                r = ref[j % len(batches)]
                ne = (s.boxes != r[0]).any() | (s.scores != r[1]).any() | (s.labels != r[2]).any() | (s.counts != r[3]).any()
                flags[j] = ne.to(torch.int32)

            for k in range(count):
                pipe.submit(batches[k % len(batches)])
                if k - (depth - 1) >= 0:
                    check(k - (depth - 1))                    # as late as the contract allows: the next submit overwrites that slot
            for j in range(max(0, count - depth + 1), count):
                check(j)
            torch.cuda.synchronize()
            bad = torch.nonzero(flags).flatten().tolist()
            assert not bad, f"batch {n}, {depth} in flight, DN_SE_SMALL={se_small}: forwards {bad[:10]} of {count} differ from their serial forward"
            total += count
    assert total >= (500 if se_small == "1" else 100)


@pytest.mark.parametrize("name,ncls,kw,n", [("ssdlite320_mobilenet_v3_large", 91, {}, 5), ("ssdlite320_mobilenet_v3_large", 91, {}, 37),
                                           ("ssd_lite_mobilenet_v2", 21, {"image_size": 300}, 9), ("ssd300_vgg16", 21, {}, 2), ("ssd512_vgg16", 21, {}, 1)])
def test_results_do_not_depend_on_stale_lds_or_registers(name, ncls, kw, n, monkeypatch):
    """DN_POISON=1 (round 4, correctness tooling -- dense.hip poison_kernel, plan.hip): a launch that leaves NaN patterns in every LDS byte and every
    vector register (VGPR and AGPR) of the chip runs in front of every launch of the forward (plain launches, no graph). A kernel that reads LDS or a
    register it has not written itself would now see different (NaN) bits than in an undisturbed run: the detections must be equal bit for bit.
    (The round-2 / round-3 one-row depthwise form, whose results depended on what else ran on the chip, was deleted in round 4; this test and the
    in-flight stress test guard the forms that are left.)"""
    size = kw.get("image_size", None)
    m0 = getattr(models, name)(num_classes=ncls, **kw)
    W, H = m0.graph.size
    imgs = torch.from_numpy(synth.images(13, n, H, W)).cuda()
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DN_POISON", flag)
        monkeypatch.setenv("DN_GRAPH", "0")
        m = models.load_synthetic(getattr(models, name)(num_classes=ncls, **kw), 0).cuda()
        res[flag] = [t.clone() for t in m.forward_batch(imgs)]
        heads = [t.clone() for t in m.forward_heads(imgs)]
        res[flag] += heads
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)
    assert int(res["1"][3].sum()) > 0 and torch.isfinite(res["1"][4]).all()


def _world2_worker(rank, world, port, ret):
    """One rank of the C4 data path with REAL kernels: both ranks share cuda:0 (there is one GPU on the test box) and talk over gloo, which
    moves device tensors through the host -- the collective's transport is not what is under test, everything around it is."""
    import torch.distributed as dist
    from demonet_amd.dist import DetectionGatherer, shard_range
    from demonet_amd.pipeline import ForwardPipeline
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G, steps, depth = 12, 5, 3                           # global batch 12 -> 6 images per rank, five steps, window of two
    lo, hi = shard_range(G, rank, world)
    B = hi - lo
    m = models.load_synthetic(models.ssdlite320_mobilenet_v3_large(num_classes=91), 0).to(dev)
    D = m.detections_per_img
    glob = [torch.from_numpy(synth.images(700 + s, G, 320, 320)).to(dev) for s in range(steps)]       # every rank can build the global batch
    pipe = ForwardPipeline(m, B, depth=depth, chains=1, device=dev, packed=True)
    gat = DetectionGatherer(B, D, dev, every=2)
    tickets = []
    for s in range(steps):
        t = pipe.submit(glob[s][lo:hi].contiguous())
        with torch.cuda.stream(pipe.stream_of(t)):
            tickets.append(gat.submit(src=pipe.packed(t), join=pipe.join))
    with torch.cuda.stream(pipe.stream_of(t)):
        gat.flush(join=pipe.join)
    torch.cuda.synchronize(dev)
    pipe.close()
    ok = True
    for s in (2, 3, 4):                                  # (steps 0, 1 were overwritten by the third window)
        pk, cn = gat.result(tickets[s])
        boxes, scores, labels, counts = [x.clone() for x in m.forward_batch(glob[s])]     # the GLOBAL batch in one forward on this rank
        ok = ok and pk.shape == (G, D, 6) and torch.equal(cn, counts)
        ok = ok and torch.equal(pk[..., :4], boxes) and torch.equal(pk[..., 4], scores) and torch.equal(pk[..., 5].long(), labels)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_data_path_with_real_kernels():
    """BASELINE config C4's data path at world size 2 on the one GPU of the test box: each rank shards the global batch (dist.shard_range),
    keeps three forwards of its shard in flight (ForwardPipeline), the merge kernel writes the packed payload, the windowed gatherer
    all-gathers it (util/misc.py:75-115 is the reference's pickled all_gather; engine.py:105 its call site), and what every rank then holds
    for a step equals ONE forward of the global batch: boxes, scores, labels and counts, bit for bit, in image order."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_world2_worker, args=(2, port, ret), nprocs=2, join=True)
    assert len(ret) == 2 and ret[0] and ret[1], dict(ret)


def test_bench_two_ranks_on_one_gpu_runs_the_real_n_rank_path():
    """`python bench.py --gpus 2` with DN_BENCH_SHARE_GPU=1: the process starts two ranks itself (as the driver's N > 1 form would), both on the
    one GPU of the test box, collectives over gloo: config C4's code path -- sharding of the global batch 256 (128 per rank), three forwards
    in flight per rank, the windowed gather of the merge kernel's payload, flush inside the timed region, barriers, MAX over ranks -- with REAL
    forwards. Not a throughput number (the line says so); n_gpus and the communicator's own world size must be 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DN_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "6", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 256
    assert d["value"] > 0 and d["config"]["mean_detections"] > 0 and "NOT a scaling number" in d["config"]["parallelism"]
